/* placeholder: the liblzma-compatible streaming API lands in the next commit */
typedef int xzamd_stream_placeholder_t;
