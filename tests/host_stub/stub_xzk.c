/* stub_xzk.c -- TEST INFRASTRUCTURE ONLY: a CPU stand-in for the kernels_api.h layer (lzma_kernels.hip), so that the
 * plain-C host code of the product (xzamd_host.c: batch geometry, span plan bookkeeping, ordered layout, stored-Block
 * fallback, framing; xzamd_stream.c: the lzma_code state machine, worker threads, ordered job queue, timeouts) can
 * run under AddressSanitizer / UndefinedBehaviorSanitizer / ThreadSanitizer on a box without a GPU (SURVEY.md 5:
 * sanitizer builds, race detection).  It is linked ONLY into tests/host_stub's test binary, never into
 * libxz_amd.so.  "Device" memory is host memory, streams are synchronous, the span kernel emits LZMA2 UNCOMPRESSED
 * chunks (lzma2_encoder.c:110-131) -- a valid payload, so the Streams the host code frames decode through the oracle
 * and the reference decoder -- and the Block checks are plain table-driven CRCs. */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include "../../xz_amd/csrc/kernels_api.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

static int g_devices = 1;

int xzk_malloc(void **p, uint64_t bytes) { *p = malloc(bytes ? bytes : 1); return *p ? 0 : 2; }
int xzk_free(void *p) { free(p); return 0; }
int xzk_host_alloc(void **p, uint64_t bytes) { *p = malloc(bytes ? bytes : 1); return *p ? 0 : 2; }
int xzk_host_free(void *p) { free(p); return 0; }
int xzk_h2d(void *d, const void *h, uint64_t bytes, void *st) { (void)st; memcpy(d, h, bytes); return 0; }
int xzk_d2h(void *h, const void *d, uint64_t bytes, void *st) { (void)st; memcpy(h, d, bytes); return 0; }
int xzk_memset(void *d, int v, uint64_t bytes, void *st) { (void)st; memset(d, v, bytes); return 0; }
int xzk_sync(void *st) { (void)st; return 0; }
int xzk_set_device(int dev) { return dev >= 0 && dev < g_devices ? 0 : 101; }
int xzk_get_device(int *dev) { *dev = 0; return 0; }
int xzk_device_count(int *n) { const char *e = getenv("STUB_DEVICES"); g_devices = e ? atoi(e) : 1; *n = g_devices; return 0; }
int xzk_cu_count(int dev, int *cus) { (void)dev; *cus = 4; return 0; }
int xzk_stream_create(void **st) { *st = malloc(1); return *st ? 0 : 2; }
int xzk_stream_create_low(void **st) { *st = malloc(1); return *st ? 0 : 2; }
int xzk_stream_wait_event(void *st, void *ev) { (void)st; (void)ev; return 0; }
int xzk_stream_destroy(void *st) { free(st); return 0; }
int xzk_event_create(void **ev) { *ev = malloc(1); return *ev ? 0 : 2; }
int xzk_event_destroy(void *ev) { free(ev); return 0; }
int xzk_event_record(void *ev, void *st) { (void)ev; (void)st; return 0; }
int xzk_event_query(void *ev) { (void)ev; return 0; }
int xzk_event_elapsed_ms(void *a, void *b, float *ms) { (void)a; (void)b; *ms = 0.0f; return 0; }
const char *xzk_error_string(int e) { (void)e; return "stub error"; }
int xzk_mem_info(uint64_t *free_b, uint64_t *total_b) { *free_b = *total_b = 1ull << 34; return 0; }
int xzk_span_occupancy(int parser, uint32_t nice_len, int *w) { (void)parser; (void)nice_len; *w = 16; return 0; }
int xzk_sort_temp_bytes(uint32_t n, uint32_t end_bit, uint64_t *bytes) { (void)n; (void)end_bit; *bytes = 64; return 0; }
int xzk_sa_temp_bytes(uint32_t n, uint64_t *bytes) { (void)n; *bytes = 64; return 0; }

int xzk_build_chains(const uint8_t *d_in, uint32_t n, uint32_t block_size, uint32_t nblocks,
		uint32_t hash_bytes, uint32_t hash_mask, uint32_t hash_bits, uint32_t sa_depth,
		uint32_t *keys_a, uint32_t *keys_b, uint32_t *vals_a, uint32_t *vals_b,
		void *sort_tmp, uint64_t sort_tmp_bytes,
		uint32_t *rank, uint32_t *sorted_pos, uint32_t *prev2, uint32_t *prev3,
		uint32_t *prev4, uint64_t *rp8, uint64_t *rp16, uint64_t *key64_a, uint64_t *key64_b,
		uint32_t *sa, uint32_t *sa_rank, uint32_t *prev24, uint32_t *prev32, void *stream)
{
	(void)prev24; (void)prev32;
	(void)d_in; (void)n; (void)block_size; (void)nblocks; (void)hash_bytes; (void)hash_mask; (void)hash_bits; (void)sa_depth;
	(void)keys_a; (void)keys_b; (void)vals_a; (void)vals_b; (void)sort_tmp; (void)sort_tmp_bytes; (void)rank; (void)sorted_pos;
	(void)prev2; (void)prev3; (void)prev4; (void)rp8; (void)rp16; (void)key64_a; (void)key64_b; (void)sa; (void)sa_rank; (void)stream;
	return 0;
}

int xzk_find_matches(const xzamd_span_args *a, const uint32_t *sa, const uint32_t *sa_rank, const uint32_t *prev4,
		const uint64_t *rp8, const uint64_t *rp16, const uint32_t *prev24, const uint32_t *prev32,
		uint16_t *mlen, uint32_t *mdist, int part, void *stream)
{
	(void)prev24; (void)prev32;
	(void)a; (void)sa; (void)sa_rank; (void)prev4; (void)rp8; (void)rp16; (void)mlen; (void)mdist; (void)part; (void)stream;
	return 0;
}

/* a plan with the documented shape: spans of 192 KiB (a multiple of the estimate chunk), at least min_len long */
int xzk_span_plan(const xzamd_span_args *a, uint32_t nblocks, uint32_t *est, unsigned long long *totals,
		uint32_t *span_tab, uint32_t *span_cnt, uint32_t cost_min, uint32_t bits_min, uint32_t min_len,
		uint32_t *enc_tab, uint32_t *enc_cnt,
		uint32_t *order_bufs, void *sort_tmp, uint64_t sort_tmp_bytes, uint32_t **order_out, void *stream)
{
	(void)est; (void)bits_min; (void)stream; (void)order_bufs; (void)sort_tmp; (void)sort_tmp_bytes;
	*order_out = NULL;
	uint32_t span = 192u << 10;
	if (span < min_len) span = min_len;
	for (uint32_t b = 0; b < nblocks; ++b) {
		const uint64_t bs = (uint64_t)b * a->block_size;
		const uint64_t be = a->n - bs < a->block_size ? a->n : bs + a->block_size;
		uint32_t k = 0;
		for (uint64_t p = bs; p < be && k < a->max_spb; p += span, ++k) {
			span_tab[2 * ((uint64_t)b * a->max_spb + k)] = (uint32_t)p;
			span_tab[2 * ((uint64_t)b * a->max_spb + k) + 1] = (uint32_t)(be - p < span || k + 1 == a->max_spb ? be : p + span);
		}
		span_cnt[b] = k;
		totals[b] = be - bs;
		if (a->enc_bits) {
			/* two-phase: encode spans of 768 KiB (four spans of the plan above), the last one takes the rest */
			const uint32_t es = 768u << 10;
			uint32_t j = 0;
			for (uint64_t p = bs; p < be && j < a->max_esb; p += es, ++j) {
				enc_tab[2 * ((uint64_t)b * a->max_esb + j)] = (uint32_t)p;
				enc_tab[2 * ((uint64_t)b * a->max_esb + j) + 1] = (uint32_t)(be - p < es || j + 1 == a->max_esb ? be : p + es);
			}
			enc_cnt[b] = j;
		}
	}
	totals[nblocks] = a->n;
	totals[nblocks + 1] = cost_min;
	return 0;
}

int xzk_span_encode(const xzamd_span_args *a, uint32_t nslots, uint32_t waves, uint32_t *counter, void *stream)
{
	(void)waves; (void)counter; (void)stream;
	for (uint32_t wg = 0; wg < nslots; ++wg) {
		const uint32_t s = a->order ? a->order[wg] : wg;
		const uint32_t b = s / a->max_spb, k = s - b * a->max_spb;
		if ((uint64_t)b * a->block_size >= a->n || k >= a->span_cnt[b])
			continue;
		const uint32_t start = a->span_tab[2 * s], end = a->span_tab[2 * s + 1];
		uint8_t *out = a->scratch + ((((uint64_t)start + (start >> 3)) + 15) & ~15ull) + (uint64_t)s * XZAMD_SPAN_SLACK;
		uint32_t o = 0;
		int first = start == b * a->block_size;
		for (uint32_t p = start; p < end; ) {
			const uint32_t c = end - p < 65536 ? end - p : 65536;
			out[o++] = first ? 1 : 2;               /* lzma2_header_uncompressed: 0x01 resets the dictionary */
			out[o++] = (uint8_t)((c - 1) >> 8);
			out[o++] = (uint8_t)(c - 1);
			memcpy(out + o, a->in + p, c);
			o += c;
			p += c;
			first = 0;
		}
		a->span_bytes[s] = o;
	}
	return 0;
}

/* two-phase: the pieces record nothing here, the encode spans are stored chunks of the input */
int xzk_parse_pieces(const xzamd_span_args *a, uint32_t nblocks, int phase, uint32_t waves, uint32_t *counter, void *stream)
{
	(void)a; (void)nblocks; (void)phase; (void)waves; (void)counter; (void)stream;
	return 0;
}

int xzk_model_snapshots(const xzamd_span_args *a, uint32_t nblocks, void *stream)
{
	(void)a; (void)nblocks; (void)stream;
	return 0;
}

int xzk_encode_syms(const xzamd_span_args *a, uint32_t nblocks, void *stream)
{
	(void)stream;
	const uint32_t nslots = nblocks * a->max_esb;
	memset(a->chunks, 0, (size_t)XZAMD_CHUNK_SLOTS(a->n, nslots) * sizeof(xzamd_chunk));
	for (uint32_t s = 0; s < nslots; ++s) {
		const uint32_t b = s / a->max_esb, k = s - b * a->max_esb;
		if ((uint64_t)b * a->block_size >= a->n || k >= a->enc_cnt[b])
			continue;
		const uint32_t start = a->enc_tab[2 * s], end = a->enc_tab[2 * s + 1];
		uint32_t ci = XZAMD_CHUNK_BASE(start, s);
		int first = start == b * a->block_size;
		for (uint32_t p = start; p < end; ++ci) {           /* raw chunks of 48 KiB: more than 32 KiB each, as the table assumes */
			const uint32_t c = end - p < 49152 ? end - p : 49152;
			uint8_t *out = a->scratch + XZAMD_CHUNK_OUT(p, ci);
			out[0] = first ? 1 : 2;                 /* lzma2_header_uncompressed: 0x01 resets the dictionary */
			out[1] = (uint8_t)((c - 1) >> 8);
			out[2] = (uint8_t)(c - 1);
			memcpy(out + 3, a->in + p, c);
			a->chunks[ci].in_start = p; a->chunks[ci].usize = c; a->chunks[ci].csize = 3 + c; a->chunks[ci].flags = XZAMD_CH_RAW;
			p += c;
			first = 0;
		}
	}
	return 0;
}

int xzk_x86_bcj(const uint8_t *d_in, uint8_t *d_out, uint32_t n, uint32_t block_size, uint32_t nblocks, void *stream)
{
	(void)d_in; (void)d_out; (void)n; (void)block_size; (void)nblocks; (void)stream;
	return 1;       /* filters are device work: not part of the host-logic tests */
}
int xzk_prefilter(const uint8_t *d_in, uint8_t *d_out, uint32_t n, uint32_t block_size, uint32_t nblocks, uint32_t kind, uint32_t dist,
		void *stream)
{
	(void)d_in; (void)d_out; (void)n; (void)block_size; (void)nblocks; (void)kind; (void)dist; (void)stream;
	return 1;
}
int xzk_sha256_blocks(const uint8_t *d_in, uint32_t n, uint32_t block_size, uint32_t nblocks, uint8_t *d_out32, void *stream)
{
	(void)d_in; (void)n; (void)block_size; (void)nblocks; (void)d_out32; (void)stream;
	return 1;
}

static uint64_t t64[256];
static uint32_t t32[256];
static pthread_once_t crc_once = PTHREAD_ONCE_INIT;
static void crc_tables(void)
{
	for (uint32_t i = 0; i < 256; ++i) {
		uint64_t r = i; uint32_t q = i;
		for (int k = 0; k < 8; ++k) {
			r = (r >> 1) ^ ((r & 1) ? 0xC96C5795D7870F42ull : 0);
			q = (q >> 1) ^ ((q & 1) ? 0xEDB88320u : 0);
		}
		t64[i] = r; t32[i] = q;
	}
}

int xzk_crc_blocks(const uint8_t *d_in, uint32_t n, uint32_t block_size, uint32_t nblocks,
		uint32_t strip, int crc32, uint64_t *d_strip_crc, uint64_t *d_block_crc, void *stream)
{
	(void)strip; (void)d_strip_crc; (void)stream;
	pthread_once(&crc_once, crc_tables);
	for (uint32_t b = 0; b < nblocks; ++b) {
		const uint64_t bs = (uint64_t)b * block_size;
		if (bs >= n) break;
		const uint64_t be = n - bs < block_size ? n : bs + block_size;
		if (crc32) {
			uint32_t c = 0xFFFFFFFFu;
			for (uint64_t i = bs; i < be; ++i) c = t32[(c ^ d_in[i]) & 0xFF] ^ (c >> 8);
			d_block_crc[b] = (uint32_t)~c;
		} else {
			uint64_t c = ~0ull;
			for (uint64_t i = bs; i < be; ++i) c = t64[(c ^ d_in[i]) & 0xFF] ^ (c >> 8);
			d_block_crc[b] = ~c;
		}
	}
	return 0;
}

int xzk_assemble(const xzamd_copy_seg *d_segs, uint32_t nsegs, const uint8_t *d_scratch,
		const uint8_t *d_lits, const uint8_t *d_in, uint8_t *d_out, void *stream)
{
	(void)stream;
	for (uint32_t i = 0; i < nsegs; ++i) {
		const xzamd_copy_seg *s = &d_segs[i];
		const uint8_t *src = s->kind == 0 ? d_scratch + s->src : s->kind == 1 ? d_lits + s->src : d_in + s->src;
		memcpy(d_out + s->dst, src, s->len);
	}
	return 0;
}
