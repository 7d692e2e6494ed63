/* xzamd_internal.h -- shared by the plain-C host translation units only (not part of any ABI). */
#ifndef XZAMD_INTERNAL_H
#define XZAMD_INTERNAL_H
#include <stddef.h>
#include <stdint.h>
#include "../../include/xz_amd.h"

uint32_t xzamd_crc32_host_(const uint8_t *p, size_t n);
void *xzamd_ctx_stream_(xzamd_ctx *c);
uint32_t xzamd_ctx_wave_slots_(const xzamd_ctx *c);
int xzamd_ctx_fail_(xzamd_ctx *c, int code, const char *what);
uint64_t xzamd_ctx_progress_in_(xzamd_ctx *c);
void xzamd_ctx_progress_reset_(xzamd_ctx *c);
int xzamd_encode_device_(xzamd_ctx *c, const void *d_in, uint64_t in_size, uint64_t block_size,
		const xzamd_lzma_options *opt, int check, uint32_t flags, void *d_out, uint64_t out_cap, uint64_t *out_size,
		xzamd_block_info *binfo, uint64_t binfo_cap, uint64_t *nblocks_out, void *stream, int *deferred);
int xzamd_encode_finish_(xzamd_ctx *c, uint64_t *out_size);
int xzamd_stored_blocks_host_(const uint8_t *in, uint64_t n, uint64_t block_size, int check,
		uint8_t *out, uint64_t out_cap, uint64_t *out_size, xzamd_block_info *binfo, uint64_t binfo_cap, uint64_t *nblocks);

double xzamd_work_bytes_per_byte_(const xzamd_lzma_options *opt);   /* device work buffers per input byte of a batch */
#endif
