/*
 * xz_amd_lzma.h -- the slice of liblzma's public C ABI that libxz_amd.so
 * implements as a drop-in: the multi-threaded .xz Stream encoder entry points
 * and the generic lzma_code()/lzma_end() pair that drive it.
 *
 * A client compiled against the real <lzma.h> links against libxz_amd.so
 * unchanged for these symbols (same names, argument meaning, return codes and
 * struct layouts; written from the ABI facts in the src/liblzma/api/lzma/ headers of
 * XZ Utils 5.8.3, symbol versions in src/liblzma/liblzma_generic.map:3-28,103-110).
 * If <lzma.h> was included first, only the prototypes below are used.
 *
 *   lzma_stream_encoder_mt           replaces common/stream_encoder_mt.c:1196
 *   lzma_stream_encoder_mt_memusage  replaces common/stream_encoder_mt.c:1231
 *   lzma_code                        replaces common/common.c:203 (for streams made here)
 *   lzma_end                         replaces common/common.c:379
 *   lzma_get_progress                replaces common/common.c:406
 *
 * Behavioural notes (INTEGRATION.md has the full list):
 *   - lzma_mt.threads caps the number of worker threads = GPUs used (one worker per visible GPU);
 *     lzma_mt.timeout bounds the time a lzma_code call waits for the workers (0 = no limit), a call
 *     that returns because of it returns LZMA_OK like the reference (stream_encoder_mt.c:667-713);
 *     LZMA_FULL_BARRIER returns once the input has been handed over (:803-807).
 *   - filters: {LZMA2} and {up to three BCJ (any of the eight) | delta filters, LZMA2} chains; LZMA_SYNC_FLUSH is unsupported
 *     exactly like the reference MT encoder (stream_encoder_mt.c:1201-1205).
 *   - check: LZMA_CHECK_NONE, LZMA_CHECK_CRC32, LZMA_CHECK_CRC64 (the xz default) and
 *     LZMA_CHECK_SHA256; others return LZMA_UNSUPPORTED_CHECK.
 */
#ifndef XZ_AMD_LZMA_H
#define XZ_AMD_LZMA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef LZMA_H   /* real liblzma headers not in use: provide ABI-identical declarations */

typedef uint64_t lzma_vli;
#define LZMA_VLI_UNKNOWN UINT64_MAX

typedef enum { LZMA_RESERVED_ENUM = 0 } lzma_reserved_enum;

typedef enum {
	LZMA_OK = 0, LZMA_STREAM_END = 1, LZMA_NO_CHECK = 2, LZMA_UNSUPPORTED_CHECK = 3,
	LZMA_GET_CHECK = 4, LZMA_MEM_ERROR = 5, LZMA_MEMLIMIT_ERROR = 6, LZMA_FORMAT_ERROR = 7,
	LZMA_OPTIONS_ERROR = 8, LZMA_DATA_ERROR = 9, LZMA_BUF_ERROR = 10, LZMA_PROG_ERROR = 11,
	LZMA_SEEK_NEEDED = 12
} lzma_ret;

typedef enum {
	LZMA_RUN = 0, LZMA_SYNC_FLUSH = 1, LZMA_FULL_FLUSH = 2, LZMA_FINISH = 3, LZMA_FULL_BARRIER = 4
} lzma_action;

typedef enum {
	LZMA_CHECK_NONE = 0, LZMA_CHECK_CRC32 = 1, LZMA_CHECK_CRC64 = 4, LZMA_CHECK_SHA256 = 10
} lzma_check;

typedef enum { LZMA_MODE_FAST = 1, LZMA_MODE_NORMAL = 2 } lzma_mode;
typedef enum {
	LZMA_MF_HC3 = 0x03, LZMA_MF_HC4 = 0x04, LZMA_MF_BT2 = 0x12, LZMA_MF_BT3 = 0x13, LZMA_MF_BT4 = 0x14
} lzma_match_finder;

#define LZMA_FILTER_LZMA1 UINT64_C(0x4000000000000001)   /* api/lzma/lzma12.h:40 */
#define LZMA_FILTER_LZMA2 UINT64_C(0x21)
#define LZMA_FILTER_X86   UINT64_C(0x04)
#define LZMA_FILTER_POWERPC UINT64_C(0x05)
#define LZMA_FILTER_IA64 UINT64_C(0x06)
#define LZMA_FILTER_ARM UINT64_C(0x07)
#define LZMA_FILTER_ARMTHUMB UINT64_C(0x08)
#define LZMA_FILTER_SPARC UINT64_C(0x09)
#define LZMA_FILTER_ARM64 UINT64_C(0x0A)
#define LZMA_FILTER_RISCV UINT64_C(0x0B)
#define LZMA_FILTER_DELTA UINT64_C(0x03)
/* api/lzma/delta.h:24-90 */
typedef enum { LZMA_DELTA_TYPE_BYTE = 0 } lzma_delta_type;
typedef struct {
	lzma_delta_type type;
	uint32_t dist;
	uint32_t reserved_int1, reserved_int2, reserved_int3, reserved_int4;
	void *reserved_ptr1, *reserved_ptr2;
} lzma_options_delta;
/* api/lzma/bcj.h:81-98 */
typedef struct {
	uint32_t start_offset;
} lzma_options_bcj;
#define LZMA_PRESET_EXTREME UINT32_C(0x80000000)
#define LZMA_PRESET_DEFAULT UINT32_C(6)

typedef struct {
	void *(*alloc)(void *opaque, size_t nmemb, size_t size);
	void (*free)(void *opaque, void *ptr);
	void *opaque;
} lzma_allocator;

typedef struct lzma_internal_s lzma_internal;

typedef struct {
	const uint8_t *next_in;
	size_t avail_in;
	uint64_t total_in;
	uint8_t *next_out;
	size_t avail_out;
	uint64_t total_out;
	const lzma_allocator *allocator;
	lzma_internal *internal;
	void *reserved_ptr1, *reserved_ptr2, *reserved_ptr3, *reserved_ptr4;
	uint64_t seek_pos;
	uint64_t reserved_int2;
	size_t reserved_int3, reserved_int4;
	lzma_reserved_enum reserved_enum1, reserved_enum2;
} lzma_stream;

#define LZMA_STREAM_INIT \
	{ NULL, 0, 0, NULL, 0, 0, NULL, NULL, NULL, NULL, NULL, NULL, 0, 0, 0, 0, \
	  LZMA_RESERVED_ENUM, LZMA_RESERVED_ENUM }

typedef struct {
	lzma_vli id;
	void *options;
} lzma_filter;

typedef struct {
	uint32_t dict_size;
	const uint8_t *preset_dict;
	uint32_t preset_dict_size;
	uint32_t lc, lp, pb;
	lzma_mode mode;
	uint32_t nice_len;
	lzma_match_finder mf;
	uint32_t depth;
	uint32_t ext_flags, ext_size_low, ext_size_high;
	uint32_t reserved_int4, reserved_int5, reserved_int6, reserved_int7, reserved_int8;
	lzma_reserved_enum reserved_enum1, reserved_enum2, reserved_enum3, reserved_enum4;
	void *reserved_ptr1, *reserved_ptr2;
} lzma_options_lzma;

typedef struct {
	uint32_t flags;
	uint32_t threads;
	uint64_t block_size;
	uint32_t timeout;
	uint32_t preset;
	const lzma_filter *filters;
	lzma_check check;
	lzma_reserved_enum reserved_enum1, reserved_enum2, reserved_enum3;
	uint32_t reserved_int1, reserved_int2, reserved_int3, reserved_int4;
	uint64_t memlimit_threading, memlimit_stop;
	uint64_t reserved_int7, reserved_int8;
	void *reserved_ptr1, *reserved_ptr2, *reserved_ptr3, *reserved_ptr4;
} lzma_mt;

#endif /* LZMA_H */

lzma_ret lzma_stream_encoder_mt(lzma_stream *strm, const lzma_mt *options);
uint64_t lzma_stream_encoder_mt_memusage(const lzma_mt *options);
lzma_ret lzma_code(lzma_stream *strm, lzma_action action);
void lzma_end(lzma_stream *strm);
void lzma_get_progress(lzma_stream *strm, uint64_t *progress_in, uint64_t *progress_out);
/* common/filter_encoder.c:270 (block size a chain asks for), hardware_cputhreads.c (here: visible GPUs = the
 * workers lzma_mt.threads can be given), stream_encoder_mt.c:914-950 (new chain between Blocks). */
uint64_t lzma_mt_block_size(const lzma_filter *filters);
uint32_t lzma_cputhreads(void);
lzma_ret lzma_filters_update(lzma_stream *strm, const lzma_filter *filters);

/* One-shot buffer API (common/stream_buffer_encoder.c:43-141, common/easy_buffer_encoder.c:16-27), same
 * return codes (LZMA_BUF_ERROR and *out_pos untouched if the output does not fit).  Like the reference the
 * Stream holds a single Block whatever the input size (stream_buffer_encoder.c:91-101), up to 1 GiB of input;
 * above that, or when the device cannot hold such a Block, the MT layout (one Block per default block_size). */
size_t lzma_stream_buffer_bound(size_t uncompressed_size);
lzma_ret lzma_stream_buffer_encode(lzma_filter *filters, lzma_check check, const lzma_allocator *allocator,
		const uint8_t *in, size_t in_size, uint8_t *out, size_t *out_pos, size_t out_size);
lzma_ret lzma_easy_buffer_encode(uint32_t preset, lzma_check check, const lzma_allocator *allocator,
		const uint8_t *in, size_t in_size, uint8_t *out, size_t *out_pos, size_t out_size);

#ifdef __cplusplus
}
#endif
#endif
