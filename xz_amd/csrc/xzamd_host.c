/*
 * xzamd_host.c -- plain-C host layer of libxz_amd: splits the input into
 * independent .xz Blocks, drives the HIP kernels over device-resident batches
 * of Blocks and reassembles a standards-conformant .xz Stream.
 *
 * Mirrors the scheduler/container half of the reference MT encoder
 *   src/liblzma/common/stream_encoder_mt.c   (stream_encode_mt :717, worker_encode :219)
 *   src/liblzma/common/block_header_encoder.c, block_buffer_encoder.c,
 *   index_encoder.c, stream_flags_encoder.c, vli_encoder.c
 * with threads replaced by one device batch: all Blocks of a batch are
 * encoded by one grid, the ordered output queue (outqueue.c) becomes a prefix
 * sum over span sizes, and the copy-out becomes one gather kernel.
 */
#include "../../include/xz_amd.h"
#include "kernels_api.h"
#include "xzamd_internal.h"

#include <stdio.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define DEFAULT_SPAN (64u * 1024u)          /* fast parser, dictionaries < 1 MiB (preset 0: 1 MiB Blocks) */
#define DEFAULT_SPAN_FAST_BIG (256u * 1024u) /* fast parser, dictionaries >= 1 MiB (presets 1-3: Blocks of 3 MiB and more): a state
                                             * reset costs ~1.3 KB on text / HTML at these presets -- 64 KiB spans: +1.3 ... +2.3 %
                                             * vs liblzma, 256 KiB: +0.5 ... +0.7 % (round 5, 16 MiB Blocks through the oracle) */
#define DEFAULT_SPAN_OPT (128u * 1024u)     /* optimal parser: fewer state resets, still >> resident waves */
#define DEFAULT_BATCH ((1ull << 31) - (1ull << 20))   /* positions are 31-bit; more spans per launch = shorter tails */
#define CRC_STRIP 4096u

/* ------------------------------------------------------------------ */
/* container primitives (doc/xz-file-format.txt)                        */
/* ------------------------------------------------------------------ */
static uint32_t crc32_tab[256];
static pthread_once_t crc32_once = PTHREAD_ONCE_INIT;

static void crc32_init(void)
{
	for (uint32_t i = 0; i < 256; ++i) {
		uint32_t r = i;
		for (int k = 0; k < 8; ++k)
			r = (r >> 1) ^ (0xEDB88320u & (0u - (r & 1)));
		crc32_tab[i] = r;
	}
}

static uint32_t crc32_buf(const uint8_t *p, size_t n)
{
	pthread_once(&crc32_once, crc32_init);       /* streams may be framed from several threads */
	uint32_t c = 0xFFFFFFFFu;
	while (n--)
		c = crc32_tab[(c ^ *p++) & 0xFF] ^ (c >> 8);
	return ~c;
}

uint32_t xzamd_crc32_host_(const uint8_t *p, size_t n) { return crc32_buf(p, n); }

static void le32(uint8_t *p, uint32_t v)
{
	p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}

static uint32_t vli_len(uint64_t v)
{
	uint32_t n = 0;
	do { ++n; v >>= 7; } while (v);
	return n;
}

static uint32_t vli_put(uint8_t *out, uint64_t v)
{
	uint32_t n = 0;
	for (; v >= 0x80; v >>= 7)
		out[n++] = (uint8_t)(v | 0x80);
	out[n++] = (uint8_t)v;
	return n;
}

static uint32_t check_bytes(int check)
{
	switch (check) {
	case XZAMD_CHECK_NONE: return 0;
	case XZAMD_CHECK_CRC32: return 4;
	case XZAMD_CHECK_CRC64: return 8;
	case XZAMD_CHECK_SHA256: return 32;
	default: return 0xFFFFFFFFu;
	}
}

uint64_t xzamd_frame_header(uint8_t *out, int check)
{
	/* stream_flags_encoder.c:29-52 */
	static const uint8_t magic[6] = { 0xFD, 0x37, 0x7A, 0x58, 0x5A, 0x00 };
	memcpy(out, magic, 6);
	out[6] = 0;
	out[7] = (uint8_t)check;
	le32(out + 8, crc32_buf(out + 6, 2));
	return 12;
}

uint64_t xzamd_frame_index_footer(uint8_t *out, uint64_t cap, int check,
		const uint64_t *unpadded, const uint64_t *uncompressed, uint64_t nblocks)
{
	/* index_encoder.c:43-164, stream_flags_encoder.c:56-85 */
	uint64_t need = 1 + vli_len(nblocks);
	for (uint64_t i = 0; i < nblocks; ++i)
		need += vli_len(unpadded[i]) + vli_len(uncompressed[i]);
	const uint64_t padded = (need + 3) & ~3ull;
	if (cap < padded + 4 + 12)
		return 0;
	uint64_t pos = 0;
	out[pos++] = 0x00;
	pos += vli_put(out + pos, nblocks);
	for (uint64_t i = 0; i < nblocks; ++i) {
		pos += vli_put(out + pos, unpadded[i]);
		pos += vli_put(out + pos, uncompressed[i]);
	}
	while (pos < padded)
		out[pos++] = 0;
	le32(out + pos, crc32_buf(out, pos));
	pos += 4;
	uint8_t *f = out + pos;
	le32(f + 4, (uint32_t)(pos / 4 - 1));
	f[8] = 0;
	f[9] = (uint8_t)check;
	le32(f, crc32_buf(f + 4, 6));
	f[10] = 0x59;
	f[11] = 0x5A;
	return pos + 12;
}

static uint8_t dict_size_byte(uint32_t d)
{
	/* lzma2_encoder.c:376-400 */
	if (d < 4096) d = 4096;
	--d;
	d |= d >> 2; d |= d >> 3; d |= d >> 4; d |= d >> 8; d |= d >> 16;
	if (d == 0xFFFFFFFFu)
		return 40;
	++d;
	uint32_t top = 31;
	while (!(d >> top)) --top;
	return (uint8_t)(2 * top + ((d >> (top - 1)) & 1) - 24);
}

uint64_t xzamd_block_buffer_bound(uint64_t u)
{
	/* block_buffer_encoder.c:20-69 */
	const uint64_t headers = (1 + 1 + 2 * 9 + 3 + 4 + 64 + 3) & ~3ull;
	const uint64_t lz2 = u + ((u + 65535) / 65536) * 3 + 1;
	return headers + ((lz2 + 3) & ~3ull);
}

/* The filters in front of LZMA2 as xzamd_lzma_options.bcj / bcj2 / bcj3 carry them (0 none, XZAMD_BCJ_*,
 * XZAMD_FILTER_DELTA(dist)), in chain order; NULL = the chain {LZMA2}. */
static uint32_t prefilter_list(const xzamd_lzma_options *opt, uint32_t pre[XZAMD_PREFILTERS_MAX])
{
	uint32_t n = 0;
	if (opt != NULL) {
		const uint32_t all[XZAMD_PREFILTERS_MAX] = { opt->bcj, opt->bcj2, opt->bcj3 };
		while (n < XZAMD_PREFILTERS_MAX && all[n] != 0) { pre[n] = all[n]; ++n; }
	}
	return n;
}

static int prefilter_valid(uint32_t pre)
{
	return (pre >= XZAMD_BCJ_X86 && pre <= XZAMD_BCJ_RISCV) || ((pre & 0xFF) == 3 && (pre >> 8) <= 255);
}

static uint32_t prefilter_flags_size(const xzamd_lzma_options *opt)
{
	uint32_t pre[XZAMD_PREFILTERS_MAX], s = 0;
	const uint32_t n = prefilter_list(opt, pre);
	for (uint32_t i = 0; i < n; ++i)
		s += (pre[i] & 0xFF) == 3 ? 3 : 2;
	return s;
}

static uint32_t block_header_size(uint64_t csize, uint64_t usize, const xzamd_lzma_options *opt)
{
	/* block_header_encoder.c:17-70 for the chains {LZMA2} and {up to three of BCJ | delta, LZMA2} */
	uint32_t s = 1 + 1 + 4 + vli_len(csize) + vli_len(usize) + 3 + prefilter_flags_size(opt);
	return (s + 3) & ~3u;
}

static void block_header_put(uint8_t *out, uint32_t hs, uint64_t csize, uint64_t usize, uint8_t dict_byte,
		const xzamd_lzma_options *opt)
{
	/* block_header_encoder.c:73-131 */
	const uint32_t body = hs - 4;
	uint32_t pre[XZAMD_PREFILTERS_MAX];
	const uint32_t npre = prefilter_list(opt, pre);
	memset(out, 0, body);
	out[0] = (uint8_t)(body / 4);
	out[1] = (uint8_t)(0xC0 | npre);   /* both sizes present, number of filters - 1 */
	uint32_t p = 2;
	p += vli_put(out + p, csize);
	p += vli_put(out + p, usize);
	for (uint32_t i = 0; i < npre; ++i) {
		if ((pre[i] & 0xFF) == 3) {      /* delta: id 0x03, one property byte = distance - 1 (delta_encoder.c:99-111) */
			out[p++] = 0x03;
			out[p++] = 0x01;
			out[p++] = (uint8_t)(pre[i] >> 8);
		} else {                         /* filter_flags_encoder.c:30-55: BCJ id, no properties (start offset 0) */
			out[p++] = (uint8_t)(pre[i] & 0xFF);
			out[p++] = 0x00;
		}
	}
	out[p++] = 0x21;
	out[p++] = 0x01;
	out[p++] = dict_byte;
	le32(out + body, crc32_buf(out, body));
}

uint64_t xzamd_stream_buffer_bound(uint64_t in_size, uint64_t block_size)
{
	/* Worst case of the span-parallel layout: every span may add LZMA2 chunk headers of its
	 * own (<= 6 bytes per chunk, at least one chunk per 4 KiB span), on top of the reference's
	 * per-Block header/padding/check and the Index.  in/128 covers 6 bytes per 768 input bytes. */
	if (block_size == 0)
		return 0;
	const uint64_t nb = (in_size + block_size - 1) / block_size;
	uint64_t tot = 12 + 12 + in_size + (in_size >> 7) + nb * (128 + 18) + 4096;
	return (tot + 15) & ~15ull;
}

/* Stored Blocks made on the host (block_buffer_encoder.c:88-162 block_encode_uncompressed: uncompressed LZMA2 chunks
 * of 64 KiB, filter chain reduced to LZMA2), laid out like the device path's XZAMD_F_BLOCKS_ONLY output.  Not an
 * encoder: the error path of the lzma_* front end for a device failure in the middle of a Stream (SURVEY.md section
 * 5: "uncompressed-chunk fallback keeps output valid"), taken only when the client asked for it
 * (XZAMD_STORED_ON_DEVICE_ERROR=1; the default is LZMA_PROG_ERROR).  Checks none / CRC32 / CRC64. */
static uint64_t crc64_tab[256];
static pthread_once_t crc64_once = PTHREAD_ONCE_INIT;
static void crc64_init(void)
{
	for (uint32_t i = 0; i < 256; ++i) {
		uint64_t r = i;
		for (int k = 0; k < 8; ++k) r = (r >> 1) ^ ((r & 1) ? 0xC96C5795D7870F42ull : 0);
		crc64_tab[i] = r;
	}
}

static uint64_t crc64_buf(const uint8_t *p, uint64_t n)
{
	pthread_once(&crc64_once, crc64_init);
	uint64_t c = ~0ull;
	for (uint64_t i = 0; i < n; ++i) c = crc64_tab[(c ^ p[i]) & 0xFF] ^ (c >> 8);
	return ~c;
}

int xzamd_stored_blocks_host_(const uint8_t *in, uint64_t n, uint64_t block_size, int check,
		uint8_t *out, uint64_t out_cap, uint64_t *out_size, xzamd_block_info *binfo, uint64_t binfo_cap, uint64_t *nblocks)
{
	const uint32_t cbytes = check_bytes(check);
	if (block_size == 0 || (check != XZAMD_CHECK_NONE && check != XZAMD_CHECK_CRC32 && check != XZAMD_CHECK_CRC64))
		return XZAMD_OPTIONS_ERROR;
	uint64_t opos = 0, nb = 0;
	for (uint64_t bs = 0; bs < n; bs += block_size, ++nb) {
		const uint64_t usize = n - bs < block_size ? n - bs : block_size;
		const uint64_t csz = usize + ((usize + 65535) / 65536) * 3 + 1;
		const uint32_t hs = block_header_size(csz, usize, NULL);
		const uint64_t pad = (4 - (csz & 3)) & 3;
		if (opos + hs + csz + pad + cbytes > out_cap)
			return XZAMD_BUF_ERROR;
		const uint64_t bstart = opos;
		block_header_put(out + opos, hs, csz, usize, 0x00, NULL);
		opos += hs;
		uint8_t ctl = 0x01;
		for (uint64_t ip = 0; ip < usize; ip += 65536) {
			const uint64_t cs = usize - ip < 65536 ? usize - ip : 65536;
			out[opos++] = ctl;
			out[opos++] = (uint8_t)((cs - 1) >> 8);
			out[opos++] = (uint8_t)(cs - 1);
			memcpy(out + opos, in + bs + ip, cs);
			opos += cs;
			ctl = 0x02;
		}
		out[opos++] = 0x00;
		for (uint64_t i = 0; i < pad; ++i) out[opos++] = 0;
		if (check == XZAMD_CHECK_CRC64) {
			const uint64_t v = crc64_buf(in + bs, usize);
			le32(out + opos, (uint32_t)v);
			le32(out + opos + 4, (uint32_t)(v >> 32));
		} else if (check == XZAMD_CHECK_CRC32) {
			le32(out + opos, crc32_buf(in + bs, usize));
		}
		opos += cbytes;
		if (binfo && nb < binfo_cap) {
			binfo[nb].unpadded_size = hs + csz + cbytes;
			binfo[nb].uncompressed_size = usize;
			binfo[nb].out_offset = bstart;
			binfo[nb].total_size = opos - bstart;
		}
	}
	*out_size = opos;
	if (nblocks) *nblocks = nb;
	return XZAMD_OK;
}

/* ------------------------------------------------------------------ */
/* presets                                                              */
/* ------------------------------------------------------------------ */
/* What a BT2/BT3/BT4 request runs on the device: the suffix-neighbourhood finder over a suffix order deep enough
 * for the request's nice_len, the windowed optimal parser, cost-balanced spans. */
void xzamd_sn_defaults(xzamd_lzma_options *o)
{
	o->gpu_sa_window = XZAMD_SA_WINDOW_MAX;
	o->gpu_parser = 1;
	o->gpu_sa_depth = o->gpu_nice_len <= 32 ? 32 : o->gpu_nice_len <= 64 ? 64 : 256;
	/* nice_len > 128 (the extreme presets) asks for ratio first: twice the output per span on compressible Blocks */
	o->span_cost = XZAMD_SPAN_COST_DEFAULT;
	o->span_bits = (o->gpu_nice_len > 128 ? 2 : 1) * XZAMD_SPAN_BITS_DEFAULT;
	o->enc_span_bits = XZAMD_ENC_SPAN_BITS_DEFAULT;
}

int xzamd_lzma_preset(xzamd_lzma_options *o, uint32_t preset)
{
	/* lzma/lzma_encoder_presets.c:17-63 */
	const uint32_t level = preset & 0x1F;
	const uint32_t flags = preset & ~0x1Fu;
	if (level > 9 || (flags & ~XZAMD_PRESET_EXTREME))
		return 1;
	static const uint8_t dict_log2[10] = { 18, 20, 21, 22, 22, 23, 23, 24, 25, 26 };
	memset(o, 0, sizeof(*o));
	o->dict_size = 1u << dict_log2[level];
	o->lc = 3; o->lp = 0; o->pb = 2;
	if (level <= 3) {
		static const uint8_t depths[4] = { 4, 8, 24, 48 };
		o->mode = XZAMD_MODE_FAST;
		o->mf = level == 0 ? XZAMD_MF_HC3 : XZAMD_MF_HC4;
		o->nice_len = level <= 1 ? 128 : 273;
		o->depth = depths[level];
	} else {
		o->mode = XZAMD_MODE_NORMAL;
		o->mf = XZAMD_MF_BT4;
		o->nice_len = level == 4 ? 16 : (level == 5 ? 32 : 64);
		o->depth = 0;
	}
	if (flags & XZAMD_PRESET_EXTREME) {
		o->mode = XZAMD_MODE_NORMAL;
		o->mf = XZAMD_MF_BT4;
		if (level == 3 || level == 5) { o->nice_len = 192; o->depth = 0; }
		else { o->nice_len = 273; o->depth = 512; }
	}
	/* Device mapping.  Fast-mode HC3/HC4 chains run exactly as requested.
	 * BT4/normal chains (presets 4-9, -e): BT4 relinks its tree at every
	 * insert, i.e. is sequential per Block; the device runs its parallel
	 * successor, the suffix-neighbourhood finder (the recency records of the
	 * 32-byte-prefix suffix order are the nodes BT4's descent visits), and a
	 * windowed form of the optimal parser. */
	if (o->mf == XZAMD_MF_HC3 || o->mf == XZAMD_MF_HC4) {
		o->gpu_mf = o->mf;
		o->gpu_nice_len = o->nice_len;
		o->gpu_depth = o->depth;
	} else {
		/* BT4 + normal mode -> suffix-neighbourhood finder + windowed optimal parser */
		o->gpu_mf = XZAMD_MF_HC4;
		o->gpu_nice_len = o->nice_len;
		o->gpu_depth = 1;
		xzamd_sn_defaults(o);
	}
	o->span_size = XZAMD_SPAN_DEFAULT;
	return 0;
}

uint64_t xzamd_mt_block_size(const xzamd_lzma_options *o)
{
	const uint64_t b = (uint64_t)o->dict_size * 3;
	return b > (1u << 20) ? b : (1u << 20);
}

/* ------------------------------------------------------------------ */
/* context                                                              */
/* ------------------------------------------------------------------ */
typedef struct {
	void *p;
	uint64_t cap;
} dbuf;

/* stage events of one batch (one set per parity of the two-stream pipeline) */
enum { EV_START, EV_CHAINS, EV_FIND, EV_PLAN0, EV_PLAN1, EV_SEED, EV_ITER1, EV_PARSE, EV_FRONT, EV_BACK0, EV_CODE, EV_CRC, EV_ASM, EV_COUNT };

/* What the front end of a batch (match structures, plan, parse -- the caller's stream) hands to its back end (range
 * coder of the two-phase mode, Block checks, sizes, layout, gather -- the second stream when the two are pipelined). */
typedef struct {
	int active;
	uint64_t b0, nb, in_off, n64;
	uint32_t n, nspans, nout, opb;
	int par;                     /* which set of the double-buffered tables / events this batch uses */
	const uint8_t *enc_in;       /* what LZMA2 reads (the filtered copy when a filter runs in front of it) */
	int find_timed, seeds_early;
	uint64_t max_segs, max_lits;
} batch_run;

/* constants of one xzamd_stream_encode_device call */
typedef struct {
	const xzamd_lzma_options *opt;
	const uint8_t *d_in;
	uint8_t *d_out;
	uint64_t block_size, out_cap, bound, binfo_cap;
	uint32_t spb, esb, cbytes, hs_fixed;
	int check, two, adaptive, whole;
	uint8_t dbyte;
	void *st, *stb;
	uint64_t *rec_unp, *rec_unc;
	xzamd_block_info *binfo;
	uint64_t opos;
} job_env;

struct xzamd_ctx {
	int device;
	void *own_stream;
	void *st2;                   /* back-end stream: range coder, checks and gather of batch i run here while the front end
	                                (match structures, plan, parse) of batch i + 1 runs on the caller's stream */
	void *st3;                   /* the seed pieces of a batch run here, underneath the rest of its match finder */
	void *ev_seed[2][3];            /* seed lists ready (caller's stream), seeds begin / done (st3) */
	void *ev_sha;                /* SHA-256 of the batch (second stream) done */
	uint64_t batch_bytes;
	uint32_t wave_slots;         /* span wavefronts resident at once (CUs x occupancy of the standard span kernel) */
	uint32_t cus;
	uint32_t span_waves;         /* != 0: persistent span kernel with this many wavefronts */
	uint64_t alloc_limit;        /* test hook (XZAMD_TEST_ALLOC_LIMIT_MIB, read once at creation): larger allocations fail; 0 = none */
	char err[256];
	char err_msg_buf[200];
	/* device buffers */
	dbuf keys_a, keys_b, vals_a, vals_b, rank, sorted_pos, prev2, prev3, prev4, prev8, prev16, prev24, prev32, key64_a, key64_b, sa, sa_rank, sort_tmp;
	dbuf scratch, span_bytes, strip_crc, block_crc, segs, lits, trace, errw, errw2, litp, mlen, mdist, bcj[2], bcjt;
	dbuf est, totals, span_tab[2], span_cnt[2], mtop, order;       /* span plan (kernels_api.h); the piece table per pipeline parity: the coder's walk reads it */
	dbuf pinfo[2], snap_sr, part_tab;                                        /* per piece: what its parser leaves for the coder's walk; what a piece of iteration 2 starts with */
	dbuf cb_bnd[2], cb_log[2], cb_hdr[2], cb_start[2], cb_carry[2]; /* carried model walk: [0] over iteration 1's records (front end), [1] the coder's (back end) */
	dbuf sym_len[2], sym_dist[2], prior, enc_tab[2], enc_cnt[2];   /* two-phase mode ([2]: one set per pipeline parity) */
	dbuf tok, chunks, h_chunks;                                    /* coder of the two-phase mode: tokens, chunk table */
	/* pinned host buffers */
	dbuf h_span_bytes, h_block_crc, h_segs, h_lits, h_span_tab, h_span_cnt[2], h_enc_tab[2], h_enc_cnt[2], h_err[2];
	void *evp[2][EV_COUNT];
	void *ev_total[2];
	uint32_t trace_cap;
	int trace_on;
	int last_par;                /* pipeline parity of the last batch (debug fetches) */
	/* A call made with `defer` returns once the back end of its last batch has been LAUNCHED (second stream); the batch is
	 * carried into the next call, which finishes it underneath the front end of its own first batch, or into
	 * xzamd_encode_finish_.  pend = that batch and the constants of its call. */
	struct { int active; job_env J; batch_run B; } pend;
	int pend_has_result, pend_rc;
	uint64_t pend_out_size;
	uint64_t par_seq;            /* parity of the double-buffered tables: continues across calls while a batch is carried */
	/* progress of the running xzamd_stream_encode_device call, read by other threads (xzamd_ctx_progress_in_) */
	pthread_mutex_t prog_mu;
	int prog_mu_ok;
	uint64_t prog_done;          /* input bytes of the batches that are through their back end */
	uint64_t prog_cur;           /* input bytes of the batch whose launches are all in flight (0: none) */
	int prog_par;                /* its event set */
	xzamd_stats stats;
};

static int fail(xzamd_ctx *c, int code, const char *what, int hip_err)
{
	if (hip_err)
		snprintf(c->err, sizeof(c->err), "%s: %s (%d)", what, xzk_error_string(hip_err), hip_err);
	else
		snprintf(c->err, sizeof(c->err), "%s", what);
	return code;
}

static int pend_complete(xzamd_ctx *c);

static int dgrow(xzamd_ctx *c, dbuf *b, uint64_t bytes, int host)
{
	if (b->cap >= bytes)
		return 0;
	if (c->pend.active) {
		/* a buffer is about to be replaced while the carried batch of the previous call may still be using it */
		int r = pend_complete(c);
		if (r) return r;
	}
	if (b->p) {
		if (host) xzk_host_free(b->p); else xzk_free(b->p);
		b->p = NULL;
		b->cap = 0;
	}
	bytes = (bytes + 255) & ~255ull;
	if (c->alloc_limit && bytes > c->alloc_limit)   /* test hook: exercises the smaller-batch retry */
		return fail(c, XZAMD_MEM_ERROR, "allocation above XZAMD_TEST_ALLOC_LIMIT_MIB", 0);
	int e = host ? xzk_host_alloc(&b->p, bytes) : xzk_malloc(&b->p, bytes);
	if (e)
		return fail(c, XZAMD_MEM_ERROR, host ? "hipHostMalloc" : "hipMalloc", e);
	b->cap = bytes;
	return 0;
}

int xzamd_ctx_create(xzamd_ctx **out, int device)
{
	*out = NULL;
	int ndev = 0;
	if (xzk_device_count(&ndev) || ndev <= 0)
		return XZAMD_DEVICE_ERROR;    /* no GPU: the product path never falls back to the CPU */
	xzamd_ctx *c = (xzamd_ctx *)calloc(1, sizeof(*c));
	if (!c)
		return XZAMD_MEM_ERROR;
	if (device < 0) {
		if (xzk_get_device(&device)) { free(c); return XZAMD_DEVICE_ERROR; }
	}
	if (device >= ndev || xzk_set_device(device)) { free(c); return XZAMD_DEVICE_ERROR; }
	c->device = device;
	if (pthread_mutex_init(&c->prog_mu, NULL)) { free(c); return XZAMD_MEM_ERROR; }
	c->prog_mu_ok = 1;
	/* from here on every failure goes through xzamd_ctx_destroy (streams and events already made are released) */
	if (xzk_stream_create(&c->own_stream)) { c->own_stream = NULL; xzamd_ctx_destroy(c); return XZAMD_DEVICE_ERROR; }
	/* (XZAMD_ST2_LOW=1, measurement knob: the back end's stream at the device's lowest priority) */
	if ((getenv("XZAMD_ST2_LOW") ? xzk_stream_create_low(&c->st2) : xzk_stream_create(&c->st2))) { c->st2 = NULL; xzamd_ctx_destroy(c); return XZAMD_DEVICE_ERROR; }
	if (xzk_event_create(&c->ev_sha)) { c->ev_sha = NULL; xzamd_ctx_destroy(c); return XZAMD_DEVICE_ERROR; }
	if (xzk_stream_create(&c->st3)) { c->st3 = NULL; xzamd_ctx_destroy(c); return XZAMD_DEVICE_ERROR; }
	for (int i = 0; i < 6; ++i)
		if (xzk_event_create(&c->ev_seed[i / 3][i % 3])) { c->ev_seed[i / 3][i % 3] = NULL; xzamd_ctx_destroy(c); return XZAMD_DEVICE_ERROR; }
	for (int i = 0; i < 2 * EV_COUNT; ++i)
		if (xzk_event_create(&c->evp[i / EV_COUNT][i % EV_COUNT])) { c->evp[i / EV_COUNT][i % EV_COUNT] = NULL; xzamd_ctx_destroy(c); return XZAMD_DEVICE_ERROR; }
	for (int i = 0; i < 2; ++i)
		if (xzk_event_create(&c->ev_total[i])) { c->ev_total[i] = NULL; xzamd_ctx_destroy(c); return XZAMD_DEVICE_ERROR; }
	{
		const char *lim = getenv("XZAMD_TEST_ALLOC_LIMIT_MIB");
		c->alloc_limit = (lim && *lim && atoll(lim) > 0) ? (uint64_t)atoll(lim) << 20 : 0;
	}
	c->batch_bytes = DEFAULT_BATCH;
	{
		int cus = 0;
		if (xzk_cu_count(device, &cus) || cus <= 0) cus = 256;
		int occ = 0;
		if (xzk_span_occupancy(1, 64, &occ) || occ <= 0 || occ > 32) occ = 16;     /* the span kernels are built for 4 waves per SIMD */
		c->cus = (uint32_t)cus;
		c->wave_slots = (uint32_t)cus * (uint32_t)occ;
		/* XZAMD_SPAN_WAVES_PER_CU = k: persistent span kernel with k wavefronts per CU (k < 16 leaves
		 * register space for the low-priority stream's kernels); 0 / unset: one wavefront per span */
		const char *pw = getenv("XZAMD_SPAN_WAVES_PER_CU");
		c->span_waves = (pw && atoi(pw) > 0) ? (uint32_t)cus * (uint32_t)atoi(pw) : 0;
	}
	const char *env = getenv("XZAMD_BATCH_MIB");
	if (env && atoll(env) > 0)
		xzamd_ctx_set_batch_bytes(c, (uint64_t)atoll(env) << 20);
	*out = c;
	return XZAMD_OK;
}

#define CTX_NBUF_MAX 96   /* callers' array size for ctx_device_bufs */
static void ctx_device_bufs(xzamd_ctx *c, dbuf **d, size_t *nd)
{
	dbuf *all[] = { &c->keys_a, &c->keys_b, &c->vals_a, &c->vals_b, &c->rank, &c->sorted_pos,
		&c->prev2, &c->prev3, &c->prev4, &c->prev8, &c->prev16, &c->prev24, &c->prev32, &c->key64_a, &c->key64_b, &c->sa, &c->sa_rank, &c->sort_tmp,
		&c->scratch, &c->litp, &c->mlen, &c->mdist, &c->bcj[0], &c->bcj[1], &c->bcjt, &c->est, &c->mtop, &c->order,
		&c->sym_len[0], &c->sym_len[1], &c->sym_dist[0], &c->sym_dist[1], &c->tok, &c->cb_log[0], &c->cb_log[1],
		/* small ones */
		&c->chunks,
		&c->span_bytes, &c->strip_crc, &c->block_crc, &c->segs, &c->lits, &c->trace, &c->errw, &c->errw2,
		&c->totals, &c->span_tab[0], &c->span_tab[1], &c->span_cnt[0], &c->span_cnt[1], &c->prior, &c->enc_tab[0], &c->enc_tab[1], &c->enc_cnt[0], &c->enc_cnt[1],
		&c->pinfo[0], &c->pinfo[1], &c->snap_sr, &c->part_tab, &c->cb_bnd[0], &c->cb_bnd[1], &c->cb_hdr[0], &c->cb_hdr[1], &c->cb_start[0], &c->cb_start[1],
		&c->cb_carry[0], &c->cb_carry[1] };
	_Static_assert(sizeof(all) / sizeof(all[0]) <= CTX_NBUF_MAX, "ctx_device_bufs: raise CTX_NBUF_MAX");
	*nd = sizeof(all) / sizeof(all[0]);
	memcpy(d, all, sizeof(all));
}
#define CTX_NBIG 35      /* the first CTX_NBIG entries of ctx_device_bufs scale with the batch */

void xzamd_ctx_destroy(xzamd_ctx *c)
{
	if (!c)
		return;
	xzk_set_device(c->device);
	if (c->pend.active) {
		/* a deferred call nobody finished (its owner is tearing down): let the kernels end before their buffers go */
		if (c->st2) xzk_sync(c->st2);
		c->pend.active = 0;
	}
	dbuf *d[CTX_NBUF_MAX];
	size_t nd = 0;
	ctx_device_bufs(c, d, &nd);
	for (size_t i = 0; i < nd; ++i)
		if (d[i]->p) xzk_free(d[i]->p);
	dbuf *h[] = { &c->h_span_bytes, &c->h_block_crc, &c->h_segs, &c->h_lits, &c->h_span_tab, &c->h_span_cnt[0], &c->h_span_cnt[1],
		&c->h_enc_tab[0], &c->h_enc_tab[1], &c->h_enc_cnt[0], &c->h_enc_cnt[1], &c->h_err[0], &c->h_err[1], &c->h_chunks };
	for (size_t i = 0; i < sizeof(h) / sizeof(h[0]); ++i)
		if (h[i]->p) xzk_host_free(h[i]->p);
	for (int i = 0; i < 2 * EV_COUNT; ++i)
		if (c->evp[i / EV_COUNT][i % EV_COUNT]) xzk_event_destroy(c->evp[i / EV_COUNT][i % EV_COUNT]);
	for (int i = 0; i < 2; ++i)
		if (c->ev_total[i]) xzk_event_destroy(c->ev_total[i]);
	if (c->ev_sha) xzk_event_destroy(c->ev_sha);
	for (int i = 0; i < 6; ++i)
		if (c->ev_seed[i / 3][i % 3]) xzk_event_destroy(c->ev_seed[i / 3][i % 3]);
	if (c->st3) xzk_stream_destroy(c->st3);
	if (c->st2) xzk_stream_destroy(c->st2);
	if (c->own_stream) xzk_stream_destroy(c->own_stream);
	if (c->prog_mu_ok) pthread_mutex_destroy(&c->prog_mu);
	free(c);
}

/* lzma_get_progress inside a job (stream_encoder_mt.c:261-267, 1004-1024: the reference's workers publish how far they
 * are inside their Block).  A device batch has no byte position; what it has is stages whose ends are events on the
 * batch's streams.  Input bytes this context has "consumed" of the call it is running = the batches that are complete +
 * the share of the batch in flight that its finished stages stand for (structure build 25 %, finder 15 %, parse 50 %,
 * model pass + range coder 10 %: their shares of a preset-6 job).  Monotonic within a call, 0
 * outside of one; callable from any thread (the events are only queried). */
uint64_t xzamd_ctx_progress_in_(xzamd_ctx *c)
{
	if (!c || !c->prog_mu_ok)
		return 0;
	pthread_mutex_lock(&c->prog_mu);
	uint64_t v = c->prog_done;
	if (c->prog_cur) {
		void **ev = c->evp[c->prog_par];
		uint32_t pct = 0;
		if (xzk_event_query(ev[EV_CHAINS]) == 0) {
			pct = 25;
			if (xzk_event_query(ev[EV_FIND]) == 0) {
				pct = 40;
				if (xzk_event_query(ev[EV_PARSE]) == 0)
					pct = xzk_event_query(ev[EV_CODE]) == 0 ? 100 : 90;
			}
		}
		v += c->prog_cur / 100 * pct;
	}
	pthread_mutex_unlock(&c->prog_mu);
	return v;
}

/* The call's figure stays up until the next call starts or the owner resets it (the lzma_* front end does so in the same
 * critical section in which it books the finished job as a whole). */
void xzamd_ctx_progress_reset_(xzamd_ctx *c)
{
	if (!c || !c->prog_mu_ok)
		return;
	pthread_mutex_lock(&c->prog_mu);
	c->prog_done = c->prog_cur = 0;
	pthread_mutex_unlock(&c->prog_mu);
}

static void progress_set(xzamd_ctx *c, uint64_t done, uint64_t cur, int par)
{
	pthread_mutex_lock(&c->prog_mu);
	c->prog_done = done; c->prog_cur = cur; c->prog_par = par;
	pthread_mutex_unlock(&c->prog_mu);
}

int xzamd_ctx_set_batch_bytes(xzamd_ctx *c, uint64_t bytes)
{
	if (bytes < (1u << 16) || bytes >= (1ull << 31))
		return XZAMD_OPTIONS_ERROR;
	c->batch_bytes = bytes;
	return XZAMD_OK;
}

const char *xzamd_last_error(const xzamd_ctx *c) { return c ? c->err : "no context"; }
/* internal accessors for the other host translation units (xzamd_internal.h) */
void *xzamd_ctx_stream_(xzamd_ctx *c) { return c->own_stream; }
uint32_t xzamd_ctx_wave_slots_(const xzamd_ctx *c) { return c->wave_slots; }
int xzamd_ctx_fail_(xzamd_ctx *c, int code, const char *what) { return fail(c, code, what, 0); }
int xzamd_ctx_device(const xzamd_ctx *c) { return c ? c->device : -1; }
void xzamd_get_stats(const xzamd_ctx *c, xzamd_stats *out) { *out = c->stats; }
const char *xzamd_version(void) { return "xz_amd 0.1 (gfx950)"; }

/* The one place that says which LZMA2 option sets the device path runs (used by the batch entry point and
 * by lzma_stream_encoder_mt, so an unsupported set is refused at init, not mid-stream).  NULL = fine. */
const char *xzamd_options_check(const xzamd_lzma_options *opt)
{
	if (opt->lc > 4 || opt->lp > 4 || opt->lc + opt->lp > 4 || opt->pb > 4)
		return "lc + lp <= 4 and pb <= 4 required (lzma_encoder.c:440-470)";
	if ((opt->gpu_mf != XZAMD_MF_HC3 && opt->gpu_mf != XZAMD_MF_HC4)
			|| opt->gpu_depth < 1 || opt->gpu_depth > 56 || opt->gpu_sa_window > XZAMD_SA_WINDOW_MAX
			|| (opt->gpu_sa_window && (opt->gpu_mf != XZAMD_MF_HC4 || !opt->gpu_parser)) || opt->gpu_parser > 1
			|| opt->gpu_nice_len < (opt->gpu_mf & 0x0F) || opt->gpu_nice_len > 273)
		return "unsupported match finder options for the device path";
	if (opt->dict_size < 4096 || opt->dict_size > (1u << 30))
		return "dict_size must be 4 KiB .. 1 GiB on the device path";
	if ((opt->bcj != 0 && !prefilter_valid(opt->bcj)) || (opt->bcj2 != 0 && (opt->bcj == 0 || !prefilter_valid(opt->bcj2)))
			|| (opt->bcj3 != 0 && (opt->bcj2 == 0 || !prefilter_valid(opt->bcj3))))
		return "filters in front of LZMA2: up to three of x86 / PowerPC / IA-64 / ARM / ARM-Thumb / SPARC / ARM64 / RISC-V BCJ or delta";
	/* pb = 3, 4 (lzma/lzma_common.h:32-37) with the optimal parser: in two-phase mode the parse pieces price with a
	 * pb = 2 view of the positions (their model is the parser's alone) and the coder's continuous model runs the real pb
	 * (k_parse_pieces / k_model_syms).  The single-phase span kernel has ONE model for both: pb <= 2. */
	if (opt->gpu_parser && opt->pb > 2
			&& !(opt->gpu_sa_window && opt->span_cost != 0 && opt->enc_span_bits != 0
				&& (opt->span_size == XZAMD_SPAN_DEFAULT || opt->span_size == XZAMD_SPAN_AUTO)))
		return "pb > 2 with the optimal parser needs the two-phase mode (default spans); the single-phase parser's price tables cover pb <= 2";
	if (opt->gpu_sa_depth != 0 && opt->gpu_sa_depth != 32 && opt->gpu_sa_depth != 64 && opt->gpu_sa_depth != 128
			&& opt->gpu_sa_depth != 256)
		return "gpu_sa_depth: 32, 64, 128 or 256";
	if (opt->part_iters > XZAMD_PART_ITERS_MAX)
		return "part_iters: 0 (default) .. 8";
	return NULL;
}

int xzamd_trace_enable(xzamd_ctx *c, uint32_t cap)
{
	xzk_set_device(c->device);
	int r = dgrow(c, &c->trace, 16ull * cap + 16, 0);
	if (r) return r;
	c->trace_cap = cap;
	c->trace_on = 1;
	if (xzk_memset(c->trace.p, 0, 16, c->own_stream) || xzk_sync(c->own_stream))
		return XZAMD_DEVICE_ERROR;
	return XZAMD_OK;
}

int xzamd_debug_fetch(xzamd_ctx *c, int what, void *out, uint64_t bytes)
{
	if (!c || !out)
		return XZAMD_PROG_ERROR;
	const dbuf *b = what == XZAMD_DEBUG_SA ? &c->sa : what == XZAMD_DEBUG_SA_RANK ? &c->sa_rank
			: what == XZAMD_DEBUG_LISTS ? &c->mdist : what == XZAMD_DEBUG_LIST_LENS ? &c->mlen
			: what == XZAMD_DEBUG_SPAN_TAB ? &c->span_tab[c->last_par] : what == XZAMD_DEBUG_SPAN_CNT ? &c->span_cnt[c->last_par]
			: what == XZAMD_DEBUG_SPAN_EST ? &c->est : what == XZAMD_DEBUG_LITP ? &c->litp
			: what == XZAMD_DEBUG_SYM_LEN ? &c->sym_len[c->last_par] : what == XZAMD_DEBUG_SYM_DIST ? &c->sym_dist[c->last_par]
			: what == XZAMD_DEBUG_ENC_TAB ? &c->enc_tab[c->last_par] : what == XZAMD_DEBUG_ENC_CNT ? &c->enc_cnt[c->last_par]
			: what == XZAMD_DEBUG_PINFO ? &c->pinfo[c->last_par] : what == XZAMD_DEBUG_SNAP_SR ? &c->snap_sr
			: what == XZAMD_DEBUG_PRIOR ? &c->prior : what == XZAMD_DEBUG_CARRY ? &c->cb_carry[1]
			: what == XZAMD_DEBUG_CB_HDR ? &c->cb_hdr[1] : what == XZAMD_DEBUG_CB_START ? &c->cb_start[1] : NULL;
	if (!b || !b->p || b->cap < bytes)
		return XZAMD_PROG_ERROR;
	xzk_set_device(c->device);
	if (xzk_d2h(out, b->p, bytes, c->own_stream) || xzk_sync(c->own_stream))
		return XZAMD_DEVICE_ERROR;
	return XZAMD_OK;
}

int xzamd_trace_read(xzamd_ctx *c, uint32_t *out, uint32_t cap, uint32_t *count)
{
	if (!c->trace.p)
		return XZAMD_PROG_ERROR;
	xzk_set_device(c->device);
	uint32_t cnt = 0;
	if (xzk_d2h(&cnt, c->trace.p, 4, c->own_stream) || xzk_sync(c->own_stream))
		return XZAMD_DEVICE_ERROR;
	*count = cnt;
	uint32_t n = cnt < cap ? cnt : cap;
	if (n > c->trace_cap) n = c->trace_cap;
	if (n && (xzk_d2h(out, (uint8_t *)c->trace.p + 16, 16ull * n, c->own_stream) || xzk_sync(c->own_stream)))
		return XZAMD_DEVICE_ERROR;
	c->trace_on = 0;
	return XZAMD_OK;
}

/* ------------------------------------------------------------------ */
/* batch encode                                                         */
/* ------------------------------------------------------------------ */
static uint32_t hash_mask_for(uint32_t dict_size, uint32_t hash_bytes)
{
	/* lz/lz_encoder.c:306-327 */
	uint32_t hs = dict_size - 1;
	hs |= hs >> 1; hs |= hs >> 2; hs |= hs >> 4; hs |= hs >> 8;
	hs >>= 1;
	hs |= 0xFFFF;
	if (hs > (1u << 24)) {
		if (hash_bytes == 3) hs = (1u << 24) - 1;
		else hs >>= 1;
	}
	return hs;
}

typedef struct {
	uint8_t *lits; uint64_t lits_len, lits_cap;
	xzamd_copy_seg *segs; uint64_t nsegs, segs_cap;
} plan;

static uint64_t plan_lit(plan *p, const uint8_t *bytes, uint64_t n, uint64_t dst)
{
	/* literal pieces are packed back to back; one segment each */
	memcpy(p->lits + p->lits_len, bytes, n);
	xzamd_copy_seg *s = &p->segs[p->nsegs++];
	s->src = p->lits_len; s->dst = dst; s->len = n; s->kind = 1; s->pad_ = 0;
	p->lits_len += n;
	return dst + n;
}

static uint64_t plan_seg(plan *p, uint32_t kind, uint64_t src, uint64_t n, uint64_t dst)
{
	if (n == 0)
		return dst;
	xzamd_copy_seg *s = &p->segs[p->nsegs++];
	s->src = src; s->dst = dst; s->len = n; s->kind = kind; s->pad_ = 0;
	return dst + n;
}

/* Geometry of the batch that starts at Block b0. */
typedef struct {
	uint64_t nb, in_off;
	uint32_t n;
	uint64_t sort_bytes;
} batch_geo;

static int batch_geometry(xzamd_ctx *c, const xzamd_lzma_options *opt, uint64_t b0, uint64_t total_blocks,
		uint64_t max_blocks, uint64_t block_size, uint64_t in_size, uint32_t hbits, batch_geo *g)
{
	g->nb = total_blocks - b0 < max_blocks ? total_blocks - b0 : max_blocks;
	g->in_off = b0 * block_size;
	const uint64_t n64 = in_size - g->in_off < g->nb * block_size ? in_size - g->in_off : g->nb * block_size;
	g->n = (uint32_t)n64;
	g->sort_bytes = 0;
	uint32_t bb = 0;
	while ((1u << bb) < g->nb + 1) ++bb;
	const uint32_t bits[4] = { 10 + bb, 16 + bb, hbits + bb, 32 };   /* 32: the by-position inversions */
	for (int i = 0; i < 4; ++i) {
		uint64_t sbytes = 0;
		int e = xzk_sort_temp_bytes(g->n, bits[i], &sbytes);
		if (e)
			return fail(c, XZAMD_DEVICE_ERROR, "rocprim temp size", e);
		if (sbytes > g->sort_bytes) g->sort_bytes = sbytes;
	}
	if (opt->gpu_sa_window) {
		uint64_t sbytes = 0;
		int e = xzk_sa_temp_bytes(g->n, &sbytes);
		if (e)
			return fail(c, XZAMD_DEVICE_ERROR, "rocprim temp size (suffix order)", e);
		if (sbytes > g->sort_bytes) g->sort_bytes = sbytes;
	}
	return XZAMD_OK;
}

static int launch_chains(xzamd_ctx *c, const xzamd_lzma_options *opt, const uint8_t *enc_in, const batch_geo *g,
		uint64_t block_size, uint32_t hb, uint32_t hmask, uint32_t hbits, void *st)
{
	int e = xzk_build_chains(enc_in, g->n, (uint32_t)block_size, (uint32_t)g->nb, hb, hmask, hbits, opt->gpu_sa_depth,
			(uint32_t *)c->keys_a.p, (uint32_t *)c->keys_b.p, (uint32_t *)c->vals_a.p,
			(uint32_t *)c->vals_b.p, c->sort_tmp.p, g->sort_bytes,
			(uint32_t *)c->rank.p, (uint32_t *)c->sorted_pos.p, (uint32_t *)c->prev2.p,
			(uint32_t *)c->prev3.p,
			opt->gpu_sa_window ? (uint32_t *)c->prev4.p : NULL,
			opt->gpu_sa_window ? (uint64_t *)c->prev8.p : NULL,
			opt->gpu_sa_window ? (uint64_t *)c->prev16.p : NULL,
			opt->gpu_sa_window ? (uint64_t *)c->key64_a.p : NULL,
			opt->gpu_sa_window ? (uint64_t *)c->key64_b.p : NULL,
			opt->gpu_sa_window ? (uint32_t *)c->sa.p : NULL,
			opt->gpu_sa_window ? (uint32_t *)c->sa_rank.p : NULL,
			opt->gpu_sa_window ? (uint32_t *)c->prev24.p : NULL,
			opt->gpu_sa_window ? (uint32_t *)c->prev32.p : NULL, st);
	return e ? fail(c, XZAMD_DEVICE_ERROR, "build_chains", e) : XZAMD_OK;
}

#define HIPCHK(call, what) do { int e_ = (call); if (e_) return fail(c, XZAMD_DEVICE_ERROR, what, e_); } while (0)


/* First half of a batch's back end, enqueued on the back-end stream: model pass + range coder (two-phase), Block checks,
 * D2H of the sizes.  Everything it needs of its batch is in `back_args`, so that the batch loop can enqueue it where it
 * likes: behind the structure build of the NEXT batch, beside its finder (the default for a batch that has one behind it), or
 * right behind the batch's own front end, beside the next batch's sorts (XZAMD_BACK_BESIDE_BUILD=1; the last batch of a call). */
typedef struct {
	int valid;
	xzamd_span_args a;
	uint64_t nb, in_off, block_size;
	uint32_t n, nch, nout;
	int par, two, check, sha_early;
	const uint8_t *d_in;
	void *st, *stb;
	void **ev;
} back_args;

static int back_enqueue(xzamd_ctx *c, back_args *B)
{
	void *stb = B->stb;
	void **ev = B->ev;
	const int par = B->par, two = B->two, check = B->check;
	const uint64_t nb = B->nb, block_size = B->block_size;
	const uint32_t n = B->n;
	int e = 0;
	B->valid = 0;
	if (stb != B->st) e = xzk_stream_wait_event(stb, ev[EV_FRONT]);
	xzk_event_record(ev[EV_BACK0], stb);
	if (!e) e = xzk_memset(c->errw2.p, 0, 512, stb);
	if (!e && two) {
		xzamd_span_args a2 = B->a;
		a2.err = (uint32_t *)c->errw2.p;
		a2.cb_bnd = (uint32_t *)c->cb_bnd[1].p; a2.cb_log = (uint32_t *)c->cb_log[1].p; a2.cb_hdr = (uint32_t *)c->cb_hdr[1].p;
		a2.cb_start = (uint16_t *)c->cb_start[1].p; a2.cb_carry = (uint32_t *)c->cb_carry[1].p;
		e = xzk_encode_syms(&a2, (uint32_t)nb, stb);
	}
	xzk_event_record(ev[EV_CODE], stb);
	if (e) return fail(c, XZAMD_DEVICE_ERROR, "encode_syms launch", e);
	/* Block checks */
	if (check == XZAMD_CHECK_CRC64 || check == XZAMD_CHECK_CRC32) {
		e = xzk_crc_blocks(B->d_in + B->in_off, n, (uint32_t)block_size, (uint32_t)nb, CRC_STRIP,
				check == XZAMD_CHECK_CRC32, (uint64_t *)c->strip_crc.p, (uint64_t *)c->block_crc.p, stb);
		if (e) return fail(c, XZAMD_DEVICE_ERROR, "crc64 launch", e);
		e = xzk_d2h(c->h_block_crc.p, c->block_crc.p, 8ull * nb, stb);
		if (e) return fail(c, XZAMD_DEVICE_ERROR, "d2h crc", e);
	} else if (check == XZAMD_CHECK_SHA256) {
		e = B->sha_early ? xzk_stream_wait_event(stb, c->ev_sha)
				: xzk_sha256_blocks(B->d_in + B->in_off, n, (uint32_t)block_size, (uint32_t)nb, (uint8_t *)c->block_crc.p, stb);
		if (e) return fail(c, XZAMD_DEVICE_ERROR, "sha256 launch", e);
		e = xzk_d2h(c->h_block_crc.p, c->block_crc.p, 32ull * nb, stb);
		if (e) return fail(c, XZAMD_DEVICE_ERROR, "d2h sha256", e);
	}
	xzk_event_record(ev[EV_CRC], stb);
	e = two ? xzk_d2h(c->h_chunks.p, c->chunks.p, (uint64_t)B->nch * sizeof(xzamd_chunk), stb)
			: xzk_d2h(c->h_span_bytes.p, c->span_bytes.p, 4ull * B->nout, stb);
	if (!e) e = xzk_d2h((uint8_t *)c->h_err[par].p + 512, c->errw2.p, 512, stb);
	if (e) return fail(c, XZAMD_DEVICE_ERROR, "d2h sizes", e);
	return XZAMD_OK;
}

/* Second half of a batch's back end: wait for the sizes, lay the Blocks out (the ordered output queue of the
 * reference, outqueue.c) and gather them into the Stream. */
static int back_finish(xzamd_ctx *c, job_env *J, batch_run *B)
{
	const xzamd_lzma_options *opt = J->opt;
	const uint64_t nb = B->nb, block_size = J->block_size, n64 = B->n64;
	const uint32_t spb = J->spb, opb = B->opb, cbytes = J->cbytes;
	const int two = J->two, check = J->check, par = B->par;
	uint8_t small[64];
	B->active = 0;
	{
		int e = xzk_sync(J->stb);
		if (e) return fail(c, XZAMD_DEVICE_ERROR, "span encode / d2h sizes", e);
		const uint32_t *herr = (const uint32_t *)c->h_err[par].p;           /* front end: parser / single-phase span kernel */
		const uint32_t *herr2 = herr + 128;                                /* back end: coder of the two-phase mode */
		if (getenv("XZAMD_TIMING") && herr[8])
			fprintf(stderr, "[timing span0] total %u round1 %u round2 %u encode %u (x256 clk) rounds %u symbols %u\n",
					herr[8], herr[9], herr[10], herr[11], herr[12], herr[13]);
		if (getenv("XZAMD_TIMING")) {
			const uint64_t *t = (const uint64_t *)(herr + 16);
			if (t[8])
				fprintf(stderr, "[timing opt, Mcycles summed over spans] total %llu | derive %llu round %llu bits %llu lit %llu relax %llu "
						"backtrack %llu encode %llu refresh %llu | nodes %llu symbols %llu windows %llu | span max %llu Mcyc | compound nodes %llu price %llu gather %llu\n",
						(unsigned long long)(t[8] >> 20), (unsigned long long)(t[0] >> 20), (unsigned long long)(t[1] >> 20),
						(unsigned long long)(t[2] >> 20), (unsigned long long)(t[3] >> 20), (unsigned long long)(t[4] >> 20),
						(unsigned long long)(t[5] >> 20), (unsigned long long)(t[6] >> 20), (unsigned long long)(t[7] >> 20),
						(unsigned long long)t[9], (unsigned long long)t[10], (unsigned long long)t[11],
						(unsigned long long)(t[12] >> 20), (unsigned long long)t[13], (unsigned long long)(t[14] >> 20), (unsigned long long)(t[15] >> 20));
			const uint64_t *t2 = (const uint64_t *)(herr2 + 48);
			if (t2[0])
				fprintf(stderr, "[timing coder, Mcycles summed over encode spans] total %llu | encode_symbol %llu of which rc_run %llu | "
						"symbols %llu bits %llu | span max %llu Mcyc\n", (unsigned long long)(t2[0] >> 20), (unsigned long long)(t2[1] >> 20),
						(unsigned long long)(t2[2] >> 20), (unsigned long long)t2[3], (unsigned long long)t2[4], (unsigned long long)(t2[5] >> 20));
		}
		const uint32_t *he = herr[0] ? herr : herr2[0] ? herr2 : NULL;
		if (he) {
			snprintf(c->err_msg_buf, sizeof(c->err_msg_buf),
					"span encoder consistency check %u failed: %u %u %u %u %u %u %u",
					he[0], he[1], he[2], he[3], he[4], he[5], he[6], he[7]);
			return fail(c, XZAMD_PROG_ERROR, c->err_msg_buf, 0);
		}
	}

	plan pl;
	pl.lits = (uint8_t *)c->h_lits.p; pl.lits_len = 0; pl.lits_cap = B->max_lits;
	pl.segs = (xzamd_copy_seg *)c->h_segs.p; pl.nsegs = 0; pl.segs_cap = B->max_segs;
	const uint32_t *sb = (const uint32_t *)c->h_span_bytes.p;
	const xzamd_chunk *hch = (const xzamd_chunk *)c->h_chunks.p;
	const uint32_t *hcnt = (const uint32_t *)c->h_span_cnt[par].p;
	/* the slots that hold coded bytes: the spans of the plan, or (two-phase) the encode spans */
	const uint32_t *otab = (const uint32_t *)(two ? c->h_enc_tab[par].p : c->h_span_tab.p);
	const uint32_t *ocnt = (const uint32_t *)(two ? c->h_enc_cnt[par].p : c->h_span_cnt[par].p);
	const uint64_t *bcrc = (const uint64_t *)c->h_block_crc.p;
	uint64_t opos = J->opos;
	for (uint64_t b = 0; b < nb; ++b) {
		const uint64_t boff = b * block_size;                 /* in batch */
		const uint64_t usize = n64 - boff < block_size ? n64 - boff : block_size;
		uint64_t payload = 1;                                 /* end marker */
		const uint32_t nsp = ocnt[b];                         /* coded spans of this Block (slots b * opb ...) */
		if (nsp == 0 || nsp > opb || hcnt[b] == 0 || hcnt[b] > spb)
			return fail(c, XZAMD_PROG_ERROR, "span plan out of range", 0);
		if (two) {
			/* the chunks of the Block's encode spans, in order (k_model_syms / k_rc_chunks) */
			for (uint32_t s = 0; s < nsp; ++s) {
				const uint32_t slot = (uint32_t)(b * opb + s), st0 = otab[2 * slot], en0 = otab[2 * slot + 1];
				const uint32_t cb = XZAMD_CHUNK_BASE(st0, slot), cc = XZAMD_CHUNK_CAP(en0 - st0);
				uint64_t covered = 0;
				for (uint32_t k = 0; k < cc && hch[cb + k].usize != 0; ++k) {
					if (hch[cb + k].csize == 0 || hch[cb + k].in_start != st0 + covered)
						return fail(c, XZAMD_PROG_ERROR, "chunk table inconsistent", 0);
					payload += hch[cb + k].csize;
					covered += hch[cb + k].usize;
				}
				if (covered != (uint64_t)en0 - st0)
					return fail(c, XZAMD_PROG_ERROR, "chunks do not cover their encode span", 0);
			}
		} else {
			for (uint32_t s = 0; s < nsp; ++s)
				payload += sb[b * opb + s];
		}
		const uint64_t pad = (4 - (payload & 3)) & 3;
		const uint64_t bstart = opos;
		uint64_t unp;
		uint8_t tail[48];
		uint32_t tl = 0;
		if (J->hs_fixed + payload + pad + cbytes > J->bound) {
			/* stream_encoder_mt.c:298,316-344 -> block_buffer_encoder.c:88-162 */
			const uint64_t csz = usize + ((usize + 65535) / 65536) * 3 + 1;
			const uint32_t hs = block_header_size(csz, usize, NULL);      /* stored Blocks drop the filters in front of LZMA2 */
			if (opos + hs + csz + 3 + cbytes > J->out_cap) return fail(c, XZAMD_BUF_ERROR, "output buffer too small", 0);
			block_header_put(small, hs, csz, usize, 0x00, NULL);
			opos = plan_lit(&pl, small, hs, opos);
			uint8_t ctl = 0x01;
			for (uint64_t ip = 0; ip < usize; ip += 65536) {
				const uint64_t cs = usize - ip < 65536 ? usize - ip : 65536;
				uint8_t ch[3] = { ctl, (uint8_t)((cs - 1) >> 8), (uint8_t)(cs - 1) };
				ctl = 0x02;
				opos = plan_lit(&pl, ch, 3, opos);
				opos = plan_seg(&pl, 2, boff + ip, cs, opos);
			}
			tail[tl++] = 0x00;
			while ((csz + (tl - 1)) & 3) tail[tl++] = 0;
			unp = hs + csz + cbytes;
			++c->stats.blocks_stored;
		} else {
			if (opos + J->hs_fixed + payload + pad + cbytes > J->out_cap) return fail(c, XZAMD_BUF_ERROR, "output buffer too small", 0);
			block_header_put(small, J->hs_fixed, payload, usize, J->dbyte, opt);
			opos = plan_lit(&pl, small, J->hs_fixed, opos);
			for (uint32_t s = 0; s < nsp; ++s) {
				const uint64_t slot = b * opb + s, start = otab[2 * slot];
				if (two) {
					const uint32_t cb = XZAMD_CHUNK_BASE((uint32_t)start, (uint32_t)slot), cc = XZAMD_CHUNK_CAP(otab[2 * slot + 1] - (uint32_t)start);
					for (uint32_t k = 0; k < cc && hch[cb + k].usize != 0; ++k)
						opos = plan_seg(&pl, 0, XZAMD_CHUNK_OUT(hch[cb + k].in_start, cb + k), hch[cb + k].csize, opos);
				} else
					opos = plan_seg(&pl, 0, ((start + (start >> 3) + 15) & ~15ull) + slot * XZAMD_SPAN_SLACK, sb[slot], opos);
			}
			tail[tl++] = 0x00;
			for (uint64_t i = 0; i < pad; ++i) tail[tl++] = 0;
			unp = J->hs_fixed + payload + cbytes;
		}
		if (check == XZAMD_CHECK_CRC64) {
			const uint64_t v = bcrc[b];
			le32(tail + tl, (uint32_t)v);
			le32(tail + tl + 4, (uint32_t)(v >> 32));
			tl += 8;
		} else if (check == XZAMD_CHECK_CRC32) {
			le32(tail + tl, (uint32_t)bcrc[b]);
			tl += 4;
		} else if (check == XZAMD_CHECK_SHA256) {
			memcpy(tail + tl, (const uint8_t *)c->h_block_crc.p + 32 * b, 32);
			tl += 32;
		}
		opos = plan_lit(&pl, tail, tl, opos);
		const uint64_t gi = B->b0 + b;
		if (J->whole) { J->rec_unp[gi] = unp; J->rec_unc[gi] = usize; }
		if (J->binfo && gi < J->binfo_cap) {
			J->binfo[gi].unpadded_size = unp;
			J->binfo[gi].uncompressed_size = usize;
			J->binfo[gi].out_offset = bstart;
			J->binfo[gi].total_size = opos - bstart;
		}
	}
	J->opos = opos;
	if (pl.nsegs > pl.segs_cap || pl.lits_len > pl.lits_cap) return fail(c, XZAMD_PROG_ERROR, "plan overflow", 0);

	/* gather (the literal pieces and the segment table leave the pinned buffers before the call returns) */
	{
		int e = xzk_h2d(c->segs.p, pl.segs, pl.nsegs * sizeof(xzamd_copy_seg), J->stb);
		if (!e) e = xzk_h2d(c->lits.p, pl.lits, pl.lits_len ? pl.lits_len : 1, J->stb);
		if (!e) e = xzk_assemble((const xzamd_copy_seg *)c->segs.p, (uint32_t)pl.nsegs,
				(const uint8_t *)c->scratch.p, (const uint8_t *)c->lits.p, J->d_in + B->in_off, J->d_out, J->stb);
		xzk_event_record(c->evp[par][EV_ASM], J->stb);
		if (!e) e = xzk_sync(J->stb);
		if (e) return fail(c, XZAMD_DEVICE_ERROR, "assemble", e);
	}
	float ms;
	void **ev = c->evp[par];
	if (!xzk_event_elapsed_ms(ev[EV_START], ev[EV_CHAINS], &ms)) c->stats.ms_chains += ms;
	if (!xzk_event_elapsed_ms(ev[EV_CHAINS], ev[EV_FRONT], &ms)) c->stats.ms_encode += ms;
	if (B->find_timed && !xzk_event_elapsed_ms(ev[EV_CHAINS], ev[EV_FIND], &ms)) c->stats.ms_find += ms;
	if (!xzk_event_elapsed_ms(ev[EV_PLAN0], ev[EV_PLAN1], &ms)) c->stats.ms_plan += ms;
	if (!xzk_event_elapsed_ms(ev[EV_CODE], ev[EV_CRC], &ms)) c->stats.ms_crc += ms;
	if (!xzk_event_elapsed_ms(ev[EV_CRC], ev[EV_ASM], &ms)) c->stats.ms_assemble += ms;
	c->stats.blocks += nb;
	for (uint64_t b = 0; b < nb; ++b) c->stats.spans += hcnt[b];
	if (two) {
		for (uint64_t b = 0; b < nb; ++b) c->stats.enc_spans += ocnt[b];
		if (B->seeds_early) { if (!xzk_event_elapsed_ms(c->ev_seed[par][1], c->ev_seed[par][2], &ms)) c->stats.ms_seed += ms; }
		else if (!xzk_event_elapsed_ms(ev[EV_PLAN1], ev[EV_SEED], &ms)) c->stats.ms_seed += ms;
		if (!xzk_event_elapsed_ms(ev[EV_SEED], ev[EV_PARSE], &ms)) c->stats.ms_parse += ms;
		if (!xzk_event_elapsed_ms(ev[EV_SEED], ev[EV_ITER1], &ms)) c->stats.ms_iter1 += ms;
		if (!xzk_event_elapsed_ms(ev[EV_BACK0], ev[EV_CODE], &ms)) { c->stats.ms_code += ms; c->stats.ms_encode += ms; }
	}
	if (J->adaptive) {
		uint64_t tgt;
		memcpy(&tgt, hcnt + 2 * ((nb + 1) / 2), 8);
		c->stats.span_cost_used = tgt > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)tgt;
	}
	c->stats.batches += 1;
	c->stats.encode_launches += 1;
	return XZAMD_OK;
}

/* back_finish of a batch of the call that is running: its bytes count as done (prog_done is written by this thread only) */
static int back_finish_own(xzamd_ctx *c, job_env *J, batch_run *B)
{
	const uint64_t n64 = B->n64;
	int rc = back_finish(c, J, B);
	if (rc == XZAMD_OK) progress_set(c, c->prog_done + n64, 0, 0);
	return rc;
}

/* Finish the batch carried over from a deferred call and keep that call's result for xzamd_encode_finish_. */
static int pend_complete(xzamd_ctx *c)
{
	if (!c->pend.active)
		return XZAMD_OK;
	c->pend.active = 0;
	int rc = back_finish(c, &c->pend.J, &c->pend.B);
	if (rc != XZAMD_OK) xzk_sync(c->st2);
	c->pend_rc = rc;
	c->pend_out_size = c->pend.J.opos;
	c->pend_has_result = 1;
	return rc;
}

int xzamd_encode_finish_(xzamd_ctx *c, uint64_t *out_size)
{
	if (!c || !out_size)
		return XZAMD_PROG_ERROR;
	/* the oldest deferred call first: its result is already there when a later call has finished it underneath itself
	 * (and may have left a deferred batch of its own) */
	if (!c->pend_has_result && c->pend.active) {
		xzk_set_device(c->device);
		pend_complete(c);
	}
	if (!c->pend_has_result)
		return fail(c, XZAMD_PROG_ERROR, "no deferred call to finish", 0);
	c->pend_has_result = 0;
	*out_size = c->pend_out_size;
	return c->pend_rc;
}

/* Device work buffers per input byte of a batch: what the batch planner budgets with and what
 * lzma_stream_encoder_mt_memusage reports -- one expression for both (round-4 advisor: the two had drifted apart). */
double xzamd_work_bytes_per_byte_(const xzamd_lzma_options *opt)
{
	const int list_packed = opt->gpu_parser && opt->dict_size <= (1u << 23);
	const int two_ = opt->gpu_parser && opt->gpu_sa_window && opt->span_cost != 0 && opt->enc_span_bits != 0
			&& (opt->span_size == XZAMD_SPAN_DEFAULT || opt->span_size == XZAMD_SPAN_AUTO);
	double per_byte = 16.0 + 8.0 + 1.2 + 0.5;                        /* sort buffers, two link arrays, scratch, tables */
	if (opt->gpu_sa_window) per_byte += 4.0 + 16.0 + 8.0 + 16.0 + 8.0;   /* prev4, rp8/16, prev24/32, key64, sa + rank */
	else per_byte += 8.0;                                            /* rank, sorted_pos */
	if (opt->gpu_parser) per_byte += 32.0 + 2.0 + (list_packed ? 0.0 : 16.0);
	if (two_) per_byte += 12.0 + 2.0 * XZAMD_TOK_PER_BYTE + 0.3;     /* recorded parse x 2, tokens, piece models in L2 */
	if (two_) {
		/* the carried model walk: bounds, logged bits and start model per encode-span slot (>= 256 KiB of input), two sets */
		const double mslots = (double)((1846u + (0x300u << (opt->lc + opt->lp)) + 63u) & ~63u);
		per_byte += 2.0 * (4.0 * XZAMD_LOG_WORDS + 6.0) * mslots / (double)XZAMD_ENC_MIN_LEN;
	}
	if (opt->bcj) per_byte += opt->bcj2 ? 3.0 : 2.0;
	return per_byte;
}

int xzamd_stream_encode_device(xzamd_ctx *c,
		const void *d_in_, uint64_t in_size, uint64_t block_size,
		const xzamd_lzma_options *opt, int check, uint32_t flags,
		void *d_out_, uint64_t out_cap, uint64_t *out_size,
		xzamd_block_info *binfo, uint64_t binfo_cap, uint64_t *nblocks_out,
		void *stream)
{
	return xzamd_encode_device_(c, d_in_, in_size, block_size, opt, check, flags, d_out_, out_cap, out_size, binfo, binfo_cap,
			nblocks_out, stream, NULL);
}

/* xzamd_stream_encode_device, and -- with deferred != NULL, XZAMD_F_BLOCKS_ONLY and the two-phase encode -- its pipelined
 * form for callers that run one call after another on the same context (the workers of the lzma_* front end): the call
 * returns when the back end of its LAST batch has been launched (*deferred = 1; *out_size is not set), and that batch is
 * finished by the next call underneath the front end of its own first batch -- the overlap the batches of ONE call have
 * always had -- or by xzamd_encode_finish_, which also hands out the call's out_size and result.  The
 * input, output and binfo buffers of a deferred call stay in use until then. */
int xzamd_encode_device_(xzamd_ctx *c,
		const void *d_in_, uint64_t in_size, uint64_t block_size,
		const xzamd_lzma_options *opt, int check, uint32_t flags,
		void *d_out_, uint64_t out_cap, uint64_t *out_size,
		xzamd_block_info *binfo, uint64_t binfo_cap, uint64_t *nblocks_out,
		void *stream, int *deferred)
{
	if (deferred) *deferred = 0;
	if (!c || !opt || !out_size || (!d_in_ && in_size) || !d_out_)
		return XZAMD_PROG_ERROR;
	if (c->pend_has_result)
		return fail(c, XZAMD_PROG_ERROR, "result of the previous deferred call not collected", 0);
	c->err[0] = 0;
	const uint32_t cbytes = check_bytes(check);
	if ((unsigned)check > 15)
		return fail(c, XZAMD_PROG_ERROR, "check id out of range", 0);
	if (cbytes == 0xFFFFFFFFu)
		return fail(c, XZAMD_UNSUPPORTED_CHECK, "checks: none, CRC32, CRC64, SHA-256", 0);
	{
		const char *why = xzamd_options_check(opt);
		if (why)
			return fail(c, XZAMD_OPTIONS_ERROR, why, 0);
	}
	if (block_size == 0)
		block_size = xzamd_mt_block_size(opt);
	if (block_size >= (1ull << 31))
		return fail(c, XZAMD_OPTIONS_ERROR, "block_size must be < 2 GiB", 0);
	void *st = stream ? stream : c->own_stream;
	const uint8_t *d_in = (const uint8_t *)d_in_;
	uint8_t *d_out = (uint8_t *)d_out_;
	HIPCHK(xzk_set_device(c->device), "hipSetDevice");

	const uint32_t hb = opt->gpu_mf & 0x0F;
	const uint32_t hmask = hash_mask_for(opt->dict_size, hb);
	uint32_t hbits = 0;
	while ((1ull << hbits) <= hmask) ++hbits;
	const uint32_t kbits_max = hbits;   /* widest 32-bit sort key family */
	const uint32_t span0 = opt->gpu_parser ? DEFAULT_SPAN_OPT : opt->dict_size >= (1u << 20) ? DEFAULT_SPAN_FAST_BIG : DEFAULT_SPAN;
	uint32_t span = (opt->span_size == XZAMD_SPAN_DEFAULT || opt->span_size == XZAMD_SPAN_AUTO) ? span0 : opt->span_size;
	if (span > block_size) span = (uint32_t)block_size;
	if (span < 4096)
		return fail(c, XZAMD_OPTIONS_ERROR, "span_size must be >= 4096", 0);

	/* Match lists of the optimal parser: 8 x u32 per position (7 entries length << 23 | distance-1 and a
	 * trailer) when distances fit 23 bits, else 8 x u32 distances + 8 x u16 lengths. */
	const int list_packed = opt->gpu_parser && opt->dict_size <= (1u << 23);
	/* batch = whole Blocks, n < 2^31, (nblocks+1) << hbits < 2^32 */
	uint64_t batch_bytes = c->batch_bytes;
	uint64_t max_blocks = batch_bytes / block_size;
	if (max_blocks == 0) max_blocks = 1;
	const uint64_t key_blocks = (1ull << (32 - kbits_max)) - 2;
	if (max_blocks > key_blocks) max_blocks = key_blocks;
	if (max_blocks * block_size >= (1ull << 31))
		max_blocks = ((1ull << 31) - 1) / block_size;
	if (max_blocks == 0)
		return fail(c, XZAMD_OPTIONS_ERROR, "block_size too large for one device batch", 0);

	const uint64_t total_blocks = (in_size + block_size - 1) / block_size;
	/* A batch must fit the device memory with room to spare: the work buffers take 100 - 165 bytes per input byte
	 * (DESIGN.md section 2), and the runtime allocates kernel scratch (register spills of the parser) lazily at launch --
	 * an allocation that fails there surfaces as an error of some later call, not as a clean out-of-memory here.  So
	 * the batch is capped at 80 % of what is free now plus what this context already holds. */
	{
		const double per_byte = xzamd_work_bytes_per_byte_(opt);
		uint64_t free_b = 0, total_b = 0, held = 0;
		if (xzk_mem_info(&free_b, &total_b) == 0 && total_b != 0) {
			dbuf *d[CTX_NBUF_MAX];
			size_t nd = 0;
			ctx_device_bufs(c, d, &nd);
			for (size_t i = 0; i < nd; ++i) held += d[i]->cap;
			const double budget = 0.80 * (double)(free_b + held);
			uint64_t fit = (uint64_t)(budget / per_byte) / block_size;
			if (fit == 0) fit = 1;
			if (fit < max_blocks) max_blocks = fit;
		}
	}
	/* even batches: a short last launch cannot fill the GPU (one wavefront per span) */
	if (total_blocks > max_blocks) {
		const uint64_t nbatch = (total_blocks + max_blocks - 1) / max_blocks;
		max_blocks = (total_blocks + nbatch - 1) / nbatch;
	}
	if (nblocks_out) *nblocks_out = total_blocks;
	/* Span plan.  Optimal parser over the suffix-neighbourhood finder with no explicit span size: cost-balanced
	 * spans cut on the device from the match lists (xzk_span_plan).  Else spans of `span` bytes, table written here. */
	const int adaptive = opt->gpu_parser && opt->gpu_sa_window && opt->span_cost != 0
			&& (opt->span_size == XZAMD_SPAN_DEFAULT || opt->span_size == XZAMD_SPAN_AUTO);
	const uint32_t spb = adaptive ? (uint32_t)(block_size / XZAMD_SPAN_MIN_LEN + 2) : (uint32_t)((block_size + span - 1) / span);   /* span slots per Block */
	/* Two-phase: the spans of the plan are parse pieces, the symbols they record are coded per encode span (esb slots per Block) */
	const int two = adaptive && opt->enc_span_bits != 0;
	const uint32_t esb = two ? (uint32_t)(block_size / XZAMD_ENC_MIN_LEN + 1) : 0;
	const uint32_t cpb = (uint32_t)((block_size + XZAMD_EST_CHUNK - 1) / XZAMD_EST_CHUNK);
	const int x86 = opt->bcj != 0;          /* any filter in front of LZMA2: the encoder reads a filtered copy */
	/* Two-stream pipeline (two-phase mode): the back end of batch i -- range coder, checks, sizes, gather -- runs on the
	 * second stream while the front end of batch i + 1 -- match structures, plan, parse -- runs on the caller's: the coder
	 * is a few thousand latency-bound wavefronts, the structure build is HBM-bound, they share the GPU well.  The
	 * single-phase kernels read the match structures while they code, so their batches stay serial. */
	const int overlap_ok = two && getenv("XZAMD_NO_OVERLAP") == NULL;
	const int defer = deferred != NULL && overlap_ok && (flags & XZAMD_F_BLOCKS_ONLY) && total_blocks > 0;
	/* a batch carried over from the previous call: it can only be finished underneath this call's front end when this call
	 * runs the same two-stream scheme on the same streams; else it is finished first */
	if (c->pend.active && !(overlap_ok && c->pend.J.st == st))
		(void)pend_complete(c);      /* its result (good or bad) belongs to the deferred call: xzamd_encode_finish_ hands it out;
		                              * nothing of THIS call has failed (round-4 advisor) */
	const int pipelined = overlap_ok && (total_blocks > max_blocks || defer || c->pend.active);

	job_env J;
	memset(&J, 0, sizeof(J));
	J.opt = opt; J.d_in = d_in; J.d_out = d_out; J.block_size = block_size; J.out_cap = out_cap;
	J.bound = xzamd_block_buffer_bound(block_size);
	J.binfo = binfo; J.binfo_cap = binfo_cap;
	J.spb = spb; J.esb = esb; J.cbytes = cbytes; J.hs_fixed = block_header_size(J.bound, block_size, opt);
	J.check = check; J.two = two; J.adaptive = adaptive;
	J.dbyte = dict_size_byte(opt->dict_size);
	J.st = st; J.stb = pipelined ? c->st2 : st;
	J.whole = !(flags & XZAMD_F_BLOCKS_ONLY);

	memset(&c->stats, 0, sizeof(c->stats));
	c->stats.span_size = adaptive ? 0 : span;
	c->stats.wave_slots = c->wave_slots;
	uint8_t small[64];
	if (J.whole) {
		J.rec_unp = (uint64_t *)malloc(sizeof(uint64_t) * (total_blocks + 1) * 2);
		if (!J.rec_unp)
			return fail(c, XZAMD_MEM_ERROR, "malloc", 0);
		J.rec_unc = J.rec_unp + total_blocks + 1;
		if (out_cap < 12) { free(J.rec_unp); return fail(c, XZAMD_BUF_ERROR, "output buffer too small", 0); }
		xzamd_frame_header(small, check);
		int e = xzk_h2d(d_out, small, 12, st);
		if (!e) e = xzk_sync(st);
		if (e) { free(J.rec_unp); return fail(c, XZAMD_DEVICE_ERROR, "h2d header", e); }
		J.opos = 12;
	}

	int rc = XZAMD_OK;
	batch_run prev;
	memset(&prev, 0, sizeof(prev));
	back_args BA;
	memset(&BA, 0, sizeof(BA));
	/* Where the back end of a batch that has another batch behind it starts: behind that batch's structure build, beside its
	 * finder (the default since the end of round 6: the sorts of the build and the coder's walks slow each other down more
	 * than the finder and the walks do -- 4 GiB: 7.14 -> 6.99 s, same Stream); XZAMD_BACK_BESIDE_BUILD=1: right behind its
	 * own front end, beside the next build (rounds 4 - 6) */
	const int back_after_build = getenv("XZAMD_BACK_BESIDE_BUILD") == NULL;
	uint64_t batch_index = 0;
	progress_set(c, 0, 0, 0);
	xzk_event_record(c->ev_total[0], st);
	for (uint64_t b0 = 0; b0 < total_blocks && rc == XZAMD_OK; ) {
		batch_geo g;
		rc = batch_geometry(c, opt, b0, total_blocks, max_blocks, block_size, in_size, hbits, &g);
		if (rc != XZAMD_OK) goto done;
		const uint64_t nb = g.nb, in_off = g.in_off, n64 = g.n;
		const uint32_t n = g.n;
		const uint64_t sort_bytes = g.sort_bytes;
		const uint32_t nspans = (uint32_t)(nb * spb);
		const uint32_t nenc = (uint32_t)(nb * esb);
		const uint32_t nout = two ? nenc : nspans;            /* slots that write coded bytes */
		const uint32_t opb = two ? esb : spb;
		const uint32_t spb_crc = (uint32_t)((block_size + CRC_STRIP - 1) / CRC_STRIP);
		const int par = pipelined ? (int)(c->par_seq++ & 1) : 0;
		void **ev = c->evp[par];

		/* Out of device memory: retry this batch with half the Blocks (retry_smaller releases every per-batch
		 * buffer first). */
#define GROW(buf, bytes, host) do { int r_ = dgrow(c, &c->buf, (bytes), host); \
		if (r_ == XZAMD_MEM_ERROR && nb > 1) { max_blocks = (nb + 1) / 2; goto retry_smaller; } \
		if (r_) { rc = r_; goto done; } } while (0)
		GROW(keys_a, 4ull * n, 0); GROW(keys_b, 4ull * n, 0);
		GROW(vals_a, 4ull * n, 0); GROW(vals_b, 4ull * n, 0);
		GROW(prev2, 4ull * n, 0); GROW(prev3, 4ull * n, 0);
		if (opt->gpu_sa_window) {
			GROW(prev4, 4ull * n, 0); GROW(prev8, 8ull * n, 0); GROW(prev16, 8ull * n, 0);   /* prev8/16: (rank, distance) pairs */
			GROW(prev24, 4ull * n, 0); GROW(prev32, 4ull * n, 0);
			GROW(key64_a, 8ull * n, 0); GROW(key64_b, 8ull * n, 0);
			GROW(sa, 4ull * n, 0); GROW(sa_rank, 4ull * n, 0);
		} else {
			GROW(rank, 4ull * n, 0); GROW(sorted_pos, 4ull * n, 0);
		}
		GROW(sort_tmp, sort_bytes + 256, 0);
		const uint32_t nch = two ? XZAMD_CHUNK_SLOTS(n, nenc) : 0;      /* chunk slots of the two-phase coder */
		const uint32_t mslots = (1846u + (0x300u << (opt->lc + opt->lp)) + 63u) & ~63u;   /* probabilities of the model, padded */
		GROW(scratch, two ? (uint64_t)n + (n >> 3) + 64 + 32ull * nch : (uint64_t)n + (n >> 3) + 32 + (uint64_t)XZAMD_SPAN_SLACK * nout, 0);
		GROW(span_bytes, 4ull * nout, 0);
		GROW(span_tab[par], 8ull * nspans, 0);
		GROW(span_cnt[par], 4ull * nb, 0);
		GROW(h_span_tab, 8ull * nspans, 1);
		GROW(h_span_cnt[par], 4ull * nb + 16, 1);
		if (two) {
			GROW(sym_len[par], 2ull * n + 64, 0);
			GROW(sym_dist[par], 4ull * n + 64, 0);
			GROW(prior, 4ull * XZAMD_PRIOR_WORDS * nspans, 0);
			GROW(enc_tab[par], 8ull * nenc, 0);
			GROW(enc_cnt[par], 4ull * nb, 0);
			GROW(h_enc_tab[par], 8ull * nenc, 1);
			GROW(h_enc_cnt[par], 4ull * nb + 16, 1);
			GROW(tok, 2ull * ((uint64_t)n * XZAMD_TOK_PER_BYTE + 4096ull * nenc + 64), 0);
			GROW(pinfo[par], 4ull * XZAMD_PINFO_WORDS * nspans, 0);
			GROW(snap_sr, 32ull * nspans, 0);
			GROW(part_tab, 4ull * nspans + 16, 0);
			for (int f = 0; f < 2; ++f) {
				GROW(cb_bnd[f], 4ull * mslots * nenc, 0);
				GROW(cb_log[f], 4ull * XZAMD_LOG_WORDS * mslots * nenc, 0);
				GROW(cb_hdr[f], 4ull * nenc + 16, 0);
				GROW(cb_start[f], 2ull * mslots * nenc + 16, 0);
				GROW(cb_carry[f], 4ull * nenc + 16, 0);
			}
			GROW(chunks, (uint64_t)nch * sizeof(xzamd_chunk), 0);
			GROW(h_chunks, (uint64_t)nch * sizeof(xzamd_chunk), 1);
		}
		if (adaptive) {
			GROW(est, 8ull * nb * cpb, 0);
			GROW(totals, 8ull * (nb + 2), 0);
			GROW(order, 16ull * nspans, 0);
		}
		GROW(strip_crc, 8ull * spb_crc * nb, 0);
		GROW(block_crc, 32ull * nb, 0);
		GROW(errw, 512, 0);
		GROW(errw2, 512, 0);
		GROW(h_err[par], 1024, 1);
		GROW(litp, (uint64_t)nspans * (0x300ull << (opt->lc + opt->lp)) * 4ull, 0);
		if (opt->gpu_parser) {
			/* per-position match lists: 8 x u32 (7 entries + trailer), + 8 x u16 lengths when not packed */
			if (!list_packed) GROW(mlen, 16ull * n, 0);
			GROW(mdist, 32ull * n, 0);
			/* + 32: k_span_est reads the summaries eight at a time (16 bytes) and may look past the last position */
			if (opt->gpu_sa_window) GROW(mtop, 2ull * n + 32, 0);
		}
		GROW(h_span_bytes, 4ull * nout, 1);
		GROW(h_block_crc, 32ull * nb, 1);
		/* plan capacity: per Block header + spans + trailer, or the stored form */
		const uint64_t segs_per_block = (two ? (block_size >> 13) + 2ull * esb + 2 : opb) + 2 + 2 * ((block_size + 65535) / 65536) + 2;
		const uint64_t max_segs = nb * segs_per_block + 4;
		const uint64_t max_lits = nb * (64 + 3 * ((block_size + 65535) / 65536) + 32) + 64;
		GROW(segs, max_segs * sizeof(xzamd_copy_seg), 0);
		GROW(lits, max_lits, 0);
		GROW(h_segs, max_segs * sizeof(xzamd_copy_seg), 1);
		GROW(h_lits, max_lits, 1);
		if (x86) GROW(bcj[par], (uint64_t)n + 16, 0);
		if (opt->bcj2) GROW(bcjt, (uint64_t)n + 16, 0);

		batch_run cur;
		memset(&cur, 0, sizeof(cur));
		cur.active = 1; cur.b0 = b0; cur.nb = nb; cur.in_off = in_off; cur.n64 = n64; cur.n = n;
		cur.nspans = nspans; cur.nout = nout; cur.opb = opb; cur.par = par;
		cur.max_segs = max_segs; cur.max_lits = max_lits;
		c->last_par = par;

		/* ================= front end, on the caller's stream ================= */
		/* 0. BCJ pre-pass: LZMA2 sees the filtered copy, the Check and stored Blocks the original */
		const uint8_t *enc_in = d_in + in_off;
		xzk_event_record(ev[EV_START], st);
		int sha_early = 0;
		if (check == XZAMD_CHECK_SHA256 && !pipelined) {
			/* SHA-256 is a serial hash per Block: on the second stream, underneath everything else of the batch */
			int e3 = xzk_stream_wait_event(c->st2, ev[EV_START]);
			if (!e3) e3 = xzk_sha256_blocks(d_in + in_off, n, (uint32_t)block_size, (uint32_t)nb, (uint8_t *)c->block_crc.p, c->st2);
			if (!e3) e3 = xzk_event_record(c->ev_sha, c->st2);
			if (e3) { rc = fail(c, XZAMD_DEVICE_ERROR, "sha256 launch", e3); goto done; }
			sha_early = 1;
		}
		if (x86) {
			/* the filters run one after another, each over the whole batch (they do not alias: in -> X for one,
			 * in -> T -> X for two, in -> X -> T -> X for three; T is only live inside this sequence) */
			uint32_t pre[XZAMD_PREFILTERS_MAX];
			const uint32_t npre = prefilter_list(opt, pre);
			uint8_t *const X = (uint8_t *)c->bcj[par].p, *const T = (uint8_t *)c->bcjt.p;
			const uint8_t *src = d_in + in_off;
			for (uint32_t i = 0; i < npre; ++i) {
				uint8_t *dst = ((npre - 1 - i) & 1) ? T : X;
				int e = pre[i] == XZAMD_BCJ_X86
						? xzk_x86_bcj(src, dst, n, (uint32_t)block_size, (uint32_t)nb, st)
						: xzk_prefilter(src, dst, n, (uint32_t)block_size, (uint32_t)nb, pre[i] & 0xFF, (pre[i] >> 8) + 1, st);
				if (e) { rc = fail(c, XZAMD_DEVICE_ERROR, "filter in front of LZMA2", e); goto done; }
				src = dst;
			}
			enc_in = X;
		}
		cur.enc_in = enc_in;
		/* 1. match-finder structure */
		rc = launch_chains(c, opt, enc_in, &g, block_size, hb, hmask, hbits, st);
		if (rc != XZAMD_OK) goto done;
		xzk_event_record(ev[EV_CHAINS], st);
		if (BA.valid) {
			/* the previous batch's back end starts here, behind this batch's structure build */
			int e_ = BA.stb != st ? xzk_stream_wait_event(BA.stb, ev[EV_CHAINS]) : 0;
			if (e_) { rc = fail(c, XZAMD_DEVICE_ERROR, "stream wait", e_); goto done; }
			rc = back_enqueue(c, &BA);
			if (rc != XZAMD_OK) goto done;
		}
		xzamd_span_args a;
		memset(&a, 0, sizeof(a));
		a.in = enc_in;
		a.rank = (const uint32_t *)c->rank.p;
		a.sorted_pos = (const uint32_t *)c->sorted_pos.p;
		a.prev2 = (const uint32_t *)c->prev2.p;
		a.prev3 = (const uint32_t *)c->prev3.p;
		a.sa_window = opt->gpu_sa_window;
		a.parser = opt->gpu_parser;
		a.scratch = (uint8_t *)c->scratch.p;
		a.span_tab = (const uint32_t *)c->span_tab[par].p;
		a.span_cnt = (const uint32_t *)c->span_cnt[par].p;
		a.max_spb = spb;
		a.span_bytes = (uint32_t *)c->span_bytes.p;
		a.err = (uint32_t *)c->errw.p;
		a.lit = (uint32_t *)c->litp.p;
		if (xzk_memset(c->errw.p, 0, 512, st)) { rc = fail(c, XZAMD_DEVICE_ERROR, "memset", 1); goto done; }
		if (c->trace_on) {
			a.trace_count = (uint32_t *)c->trace.p;
			a.trace = (uint32_t *)((uint8_t *)c->trace.p + 16);
			a.trace_cap = c->trace_cap;
		}
		a.n = n;
		a.block_size = (uint32_t)block_size;
		a.span_size = span;
		a.dict_size = opt->dict_size;
		a.nice_len = opt->gpu_nice_len;
		a.depth = opt->gpu_depth;
		a.hash_bytes = hb;
		a.lc = opt->lc; a.lp = opt->lp; a.pb = opt->pb;
		if (two) {
			a.sym_len = (uint16_t *)c->sym_len[par].p;
			a.sym_dist = (uint32_t *)c->sym_dist[par].p;
			a.prior = (uint32_t *)c->prior.p;
			a.enc_tab = (const uint32_t *)c->enc_tab[par].p;
			a.enc_cnt = (const uint32_t *)c->enc_cnt[par].p;
			a.max_esb = esb;
			a.enc_bits = opt->enc_span_bits;
			{
				/* test knob: a smaller token budget per input byte (never a larger one: the buffer is what it is) */
				const char *tl = getenv("XZAMD_TEST_TOK_PER_BYTE");
				const unsigned long v = tl ? strtoul(tl, NULL, 10) : 0;
				a.tok_limit = v >= 1 && v < XZAMD_TOK_PER_BYTE ? (uint32_t)v : 0;
			}
			{
				/* test knob: fewer logged bits per span and probability (never more: the log is what it is) */
				const char *lc_ = getenv("XZAMD_TEST_LOG_CAP");
				const unsigned long v = lc_ ? strtoul(lc_, NULL, 10) : 0;
				a.log_cap = v >= 1 && v < XZAMD_LOG_CAP ? (uint32_t)v : 0;
			}
			a.tok = (uint16_t *)c->tok.p;
			a.chunks = (xzamd_chunk *)c->chunks.p;
			a.pinfo = (uint32_t *)c->pinfo[par].p;
			a.snap_sr = (uint32_t *)c->snap_sr.p;
			a.part_tab = (uint32_t *)c->part_tab.p;
			a.model_slots_pad = mslots;
			/* (the front end's set of the carried-walk buffers; the back end switches to its own below) */
			a.cb_bnd = (uint32_t *)c->cb_bnd[0].p; a.cb_log = (uint32_t *)c->cb_log[0].p; a.cb_hdr = (uint32_t *)c->cb_hdr[0].p;
			a.cb_start = (uint16_t *)c->cb_start[0].p; a.cb_carry = (uint32_t *)c->cb_carry[0].p;
		}
		{
			int e = 0, seeds_early = 0;
			if (opt->gpu_parser) {
				/* 2a. batch match finder -> lists the parser streams */
				a.mlen = list_packed ? NULL : (uint16_t *)c->mlen.p;
				a.list_packed = (uint32_t)list_packed;
				a.mdist = (uint32_t *)c->mdist.p;
				a.mtop = (uint16_t *)c->mtop.p;
				/* two-phase: the lists of the seed regions first, so that the seed pieces (one wavefront per Block, latency
				 * bound) can be parsed on a third stream underneath the rest of the finder (HBM bound) */
				seeds_early = two && block_size >= XZAMD_SEED_LEN + 1024 && getenv("XZAMD_NO_OVERLAP") == NULL;
				if (seeds_early) {
					e = xzk_find_matches(&a, (const uint32_t *)c->sa.p, (const uint32_t *)c->sa_rank.p, (const uint32_t *)c->prev4.p,
							(const uint64_t *)c->prev8.p, (const uint64_t *)c->prev16.p,
							(const uint32_t *)c->prev24.p, (const uint32_t *)c->prev32.p, (uint16_t *)a.mlen, (uint32_t *)a.mdist, 1, st);
					if (!e) e = xzk_event_record(c->ev_seed[par][0], st);
					if (!e) e = xzk_stream_wait_event(c->st3, c->ev_seed[par][0]);
					if (!e) e = xzk_event_record(c->ev_seed[par][1], c->st3);
					if (!e) e = xzk_parse_pieces(&a, (uint32_t)nb, 0, 0, NULL, c->st3);
					if (!e) e = xzk_event_record(c->ev_seed[par][2], c->st3);
				}
				if (!e) e = xzk_find_matches(&a, (const uint32_t *)c->sa.p, (const uint32_t *)c->sa_rank.p, (const uint32_t *)c->prev4.p,
						(const uint64_t *)c->prev8.p, (const uint64_t *)c->prev16.p,
							(const uint32_t *)c->prev24.p, (const uint32_t *)c->prev32.p, (uint16_t *)a.mlen, (uint32_t *)a.mdist, seeds_early ? 2 : 0, st);
				if (e) { rc = fail(c, XZAMD_DEVICE_ERROR, "find_matches launch", e); goto done; }
				cur.find_timed = 1;
			}
			xzk_event_record(ev[EV_FIND], st);
			/* 2b. span plan */
			uint32_t *const htab = (uint32_t *)c->h_span_tab.p, *const hcnt = (uint32_t *)c->h_span_cnt[par].p;
			xzk_event_record(ev[EV_PLAN0], st);
			if (adaptive) {
				uint32_t *launch_order = NULL;
				{
					int occ = 0;
					c->stats.wave_slots = (xzk_span_occupancy(1, opt->gpu_nice_len, &occ) || occ <= 0 || occ > 32)
							? c->wave_slots : c->cus * (uint32_t)occ;
				}
				e = xzk_span_plan(&a, (uint32_t)nb, (uint32_t *)c->est.p, (unsigned long long *)c->totals.p,
						(uint32_t *)c->span_tab[par].p, (uint32_t *)c->span_cnt[par].p, opt->span_cost, opt->span_bits,
						XZAMD_SPAN_MIN_LEN, two ? (uint32_t *)c->enc_tab[par].p : NULL, two ? (uint32_t *)c->enc_cnt[par].p : NULL,
						(uint32_t *)c->order.p, c->sort_tmp.p, c->sort_tmp.cap, &launch_order, st);
				if (e) { rc = fail(c, XZAMD_DEVICE_ERROR, "span plan launch", e); goto done; }
				a.order = launch_order;
				/* the host lays the Blocks out from the plan: fetched with the span sizes below */
				if (!two) e = xzk_d2h(htab, c->span_tab[par].p, 8ull * nspans, st);
				if (!e) e = xzk_d2h(hcnt, c->span_cnt[par].p, 4ull * nb, st);
				if (!e) e = xzk_d2h(hcnt + 2 * ((nb + 1) / 2), (uint8_t *)c->totals.p + 8ull * (nb + 1), 8, st);   /* target used, behind the counts */
				if (!e && two) e = xzk_d2h(c->h_enc_tab[par].p, c->enc_tab[par].p, 8ull * nenc, st);
				if (!e && two) e = xzk_d2h(c->h_enc_cnt[par].p, c->enc_cnt[par].p, 4ull * nb, st);
				if (e) { rc = fail(c, XZAMD_DEVICE_ERROR, "d2h span plan", e); goto done; }
			} else {
				for (uint64_t b = 0; b < nb; ++b) {
					const uint64_t bs = b * block_size, be = n64 - bs < block_size ? n64 : bs + block_size;
					uint32_t k = 0;
					for (uint64_t p = bs; p < be; p += span, ++k) {
						htab[2 * (b * spb + k)] = (uint32_t)p;
						htab[2 * (b * spb + k) + 1] = (uint32_t)(be - p < span ? be : p + span);
					}
					hcnt[b] = k;
				}
				e = xzk_h2d(c->span_tab[par].p, htab, 8ull * nspans, st);
				if (!e) e = xzk_h2d(c->span_cnt[par].p, hcnt, 4ull * nb, st);
				if (e) { rc = fail(c, XZAMD_DEVICE_ERROR, "h2d span plan", e); goto done; }
			}
			xzk_event_record(ev[EV_PLAN1], st);
			/* 2c. parse (two-phase: seed pieces, then every other piece) or the single-phase span kernel */
			if (two) {
				if (seeds_early) e = xzk_stream_wait_event(st, c->ev_seed[par][2]);
				else e = xzk_parse_pieces(&a, (uint32_t)nb, 0, 0, NULL, st);
				xzk_event_record(ev[EV_SEED], st);
				cur.seeds_early = seeds_early;
				/* partial iterations: the first part of every piece -- from the seed's prior, then from the snapshots -- and the
				 * carried model walk over its records, which leaves every piece the price model the next iteration starts from;
				 * then every piece in full (oracle: parse_block) */
				const uint32_t npart = opt->part_iters ? opt->part_iters : XZAMD_PART_ITERS_DEFAULT;
				for (uint32_t it = 0; it < npart && !e; ++it) {
					a.iter = XZAMD_ITER_PARTIAL | (it ? XZAMD_ITER_SNAP : 0u);
					if (it) e = xzk_memset((uint32_t *)c->errw.p + 60, 0, 4, st);
					if (!e) e = xzk_parse_pieces(&a, (uint32_t)nb, 1, c->span_waves, (uint32_t *)c->errw.p + 60, st);
					if (!e) e = xzk_model_snapshots(&a, (uint32_t)nb, st);
				}
				xzk_event_record(ev[EV_ITER1], st);
				a.iter = XZAMD_ITER_SNAP;
				if (!e) e = xzk_memset((uint32_t *)c->errw.p + 60, 0, 4, st);
				if (!e) e = xzk_parse_pieces(&a, (uint32_t)nb, 1, c->span_waves, (uint32_t *)c->errw.p + 60, st);
				xzk_event_record(ev[EV_PARSE], st);
			} else {
				e = xzk_span_encode(&a, nspans, c->span_waves, (uint32_t *)c->errw.p + 60, st);
				/* single phase: the span kernel parses AND codes; its end is this batch's EV_PARSE for lzma_get_progress (a stage
				 * event that is not recorded for a batch would query as complete: the figure jumped to 90 % at the finder's end) */
				xzk_event_record(ev[EV_PARSE], st);
			}
			if (e) { rc = fail(c, XZAMD_DEVICE_ERROR, "span_encode launch", e); goto done; }
			e = xzk_d2h(c->h_err[par].p, c->errw.p, 512, st);
			if (e) { rc = fail(c, XZAMD_DEVICE_ERROR, "d2h", e); goto done; }
			xzk_event_record(ev[EV_FRONT], st);
		}

		/* ================= back end ================= */
		/* the previous batch's sizes, layout and gather: its coder has had the whole front end above to finish */
		if (pipelined && c->pend.active) {
			/* the last batch of the previous (deferred) call; a failure there is that call's (pend_rc, handed out by
			 * xzamd_encode_finish_), pend_complete has drained the second stream, this call goes on */
			(void)pend_complete(c);
		} else if (pipelined && prev.active) {
			rc = back_finish_own(c, &J, &prev);
			if (rc != XZAMD_OK) goto done;
		}
		{
			BA.valid = 1;
			BA.a = a; BA.nb = nb; BA.in_off = in_off; BA.block_size = block_size; BA.n = n; BA.nch = nch; BA.nout = nout;
			BA.par = par; BA.two = two; BA.check = check; BA.sha_early = sha_early; BA.d_in = d_in; BA.st = st; BA.stb = J.stb;
			BA.ev = ev;
			/* (back_after_build: a batch that has another one behind it leaves its back end to that batch's iteration) */
			if (!(back_after_build && pipelined && b0 + nb < total_blocks)) {
				rc = back_enqueue(c, &BA);
				if (rc != XZAMD_OK) goto done;
			}
		}
		progress_set(c, c->prog_done, n64, par);     /* every stage event of this batch has been recorded */
		if (pipelined) {
			prev = cur;
		} else {
			rc = back_finish_own(c, &J, &cur);
			if (rc != XZAMD_OK) goto done;
		}
		b0 += nb;
		++batch_index;
		continue;
retry_smaller:
		c->err[0] = 0;
		/* The buffers grown so far have full-batch capacity: a retry that kept them would fight for what is left.
		 * Nothing of this batch has been launched; an earlier batch may still be in its back end: finish it, then
		 * release every per-batch device buffer and let the smaller geometry allocate afresh. */
		if (c->pend.active)
			(void)pend_complete(c);          /* (its result is the deferred call's) */
		if (BA.valid) {                       /* (the earlier batch's back end has not been enqueued yet) */
			rc = back_enqueue(c, &BA);
			if (rc != XZAMD_OK) goto done;
		}
		if (prev.active) {
			rc = back_finish_own(c, &J, &prev);
			if (rc != XZAMD_OK) goto done;
		}
		xzk_sync(st);
		xzk_sync(c->st2);
		{
			dbuf *d[CTX_NBUF_MAX];
			size_t nd = 0;
			ctx_device_bufs(c, d, &nd);
			for (size_t i = 0; i < CTX_NBIG && i < nd; ++i)
				if (d[i]->p) { xzk_free(d[i]->p); d[i]->p = NULL; d[i]->cap = 0; }
		}
	}
	if (rc == XZAMD_OK && c->pend.active)
		(void)pend_complete(c);      /* (a call without a batch of its own) */
	if (rc == XZAMD_OK && prev.active) {
		if (defer) {
			/* the back end of the last batch is in flight on the second stream: whoever comes next finishes it */
			c->pend.J = J;
			c->pend.B = prev;
			c->pend.active = 1;
			prev.active = 0;
			*deferred = 1;
		} else {
			rc = back_finish_own(c, &J, &prev);
		}
	}
done:
	if (rc != XZAMD_OK && c->pend.active)
		pend_complete(c);            /* this call failed before it got there: the carried batch still gets finished */
	xzk_sync(st);
	xzk_sync(c->st3);
	if (!(deferred && *deferred))
		xzk_sync(c->st2);          /* a back end abandoned by an error path */
	if (rc == XZAMD_OK && J.whole) {
		const uint64_t isz_cap = 32 + total_blocks * 18 + 16;
		uint8_t *ib = (uint8_t *)malloc(isz_cap);
		if (!ib) rc = fail(c, XZAMD_MEM_ERROR, "malloc", 0);
		else {
			const uint64_t w = xzamd_frame_index_footer(ib, isz_cap, check, J.rec_unp, J.rec_unc, total_blocks);
			if (w == 0 || J.opos + w > out_cap) rc = fail(c, XZAMD_BUF_ERROR, "output buffer too small", 0);
			else {
				int e = xzk_h2d(d_out + J.opos, ib, w, st);
				if (!e) e = xzk_sync(st);
				if (e) rc = fail(c, XZAMD_DEVICE_ERROR, "h2d index", e);
				J.opos += w;
			}
			free(ib);
		}
	}
	xzk_event_record(c->ev_total[1], st);
	xzk_sync(st);
	{
		float ms;
		if (!xzk_event_elapsed_ms(c->ev_total[0], c->ev_total[1], &ms)) c->stats.ms_total = ms;
	}
	free(J.rec_unp);
	c->stats.in_bytes = in_size;
	c->stats.out_bytes = J.opos;
	*out_size = (deferred && *deferred) ? 0 : J.opos;
	return rc;
}
