/*
 * compress_mt.c -- minimal liblzma client: stdin -> .xz on stdout through
 * lzma_stream_encoder_mt()/lzma_code().  It compiles unchanged against either
 *   the real liblzma:   gcc compress_mt.c -llzma
 *   libxz_amd (GPU):    gcc -DUSE_XZ_AMD -I../include compress_mt.c -L../xz_amd -lxz_amd
 * which is the whole point of the drop-in boundary (same role as the
 * reference's doc/examples/04_compress_easy_mt.c, written independently).
 *
 * usage: compress_mt [preset 0-9] [block_size_bytes] [io_buffer_bytes] < in > out.xz
 */
#ifdef USE_XZ_AMD
#include "xz_amd_lzma.h"
#else
#include <lzma.h>
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv)
{
	unsigned preset = argc > 1 ? (unsigned)atoi(argv[1]) : 6;
	unsigned long long block_size = argc > 2 ? strtoull(argv[2], NULL, 10) : 0;
	size_t bufsz = argc > 3 ? (size_t)strtoull(argv[3], NULL, 10) : (size_t)1 << 20;

	lzma_stream strm = LZMA_STREAM_INIT;
	lzma_mt mt;
	memset(&mt, 0, sizeof(mt));
	mt.threads = 4;
	mt.preset = preset;
	mt.block_size = block_size;
	mt.check = LZMA_CHECK_CRC64;
	lzma_ret r = lzma_stream_encoder_mt(&strm, &mt);
	if (r != LZMA_OK) {
		fprintf(stderr, "lzma_stream_encoder_mt failed: %d\n", (int)r);
		return 1;
	}
	unsigned char *inbuf = malloc(bufsz), *outbuf = malloc(bufsz);
	lzma_action action = LZMA_RUN;
	strm.next_out = outbuf;
	strm.avail_out = bufsz;
	for (;;) {
		if (strm.avail_in == 0 && action == LZMA_RUN) {
			strm.next_in = inbuf;
			strm.avail_in = fread(inbuf, 1, bufsz, stdin);
			if (feof(stdin))
				action = LZMA_FINISH;
		}
		r = lzma_code(&strm, action);
		if (strm.avail_out == 0 || r == LZMA_STREAM_END) {
			size_t n = bufsz - strm.avail_out;
			if (fwrite(outbuf, 1, n, stdout) != n) { perror("write"); return 1; }
			strm.next_out = outbuf;
			strm.avail_out = bufsz;
		}
		if (r == LZMA_STREAM_END)
			break;
		if (r != LZMA_OK) {
			fprintf(stderr, "lzma_code failed: %d\n", (int)r);
			return 1;
		}
	}
	uint64_t pin = 0, pout = 0;
	lzma_get_progress(&strm, &pin, &pout);
	fprintf(stderr, "in %llu out %llu (progress %llu/%llu)\n", (unsigned long long)strm.total_in,
			(unsigned long long)strm.total_out, (unsigned long long)pin, (unsigned long long)pout);
	lzma_end(&strm);
	free(inbuf); free(outbuf);
	return 0;
}
