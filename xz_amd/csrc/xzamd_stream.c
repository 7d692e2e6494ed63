/*
 * xzamd_stream.c -- liblzma-compatible streaming front end of libxz_amd.
 *
 * Exports the reference's own entry points for the multi-threaded .xz Stream
 * encoder so a liblzma client (doc/examples/04_compress_easy_mt.c, xz's
 * src/xz/coder.c:834-837 + :1190-1300) can switch libraries unchanged:
 *
 *   lzma_stream_encoder_mt  <- stream_encoder_mt.c:1196  (init :1028-1168, option checks :956-1000)
 *   lzma_code               <- common/common.c:203-376   (action sequencing, BUF_ERROR rule)
 *   lzma_end                <- common/common.c:379-389
 *   lzma_get_progress       <- common/common.c:406 / stream_encoder_mt.c:1004-1024
 *   lzma_stream_encoder_mt_memusage <- stream_encoder_mt.c:1231
 *
 * stream_encode_mt() (:717-883) copies caller input into per-worker Block
 * buffers and hands them to threads; here the caller's bytes are copied into a
 * pinned host staging buffer holding a batch of Blocks, one batch at a time is
 * uploaded, encoded by xzamd_stream_encode_device(BLOCKS_ONLY) and downloaded
 * into an ordered output buffer (the lzma_outq equivalent), and the Index
 * records are kept on the host until LZMA_FINISH.
 */
#include "../../include/xz_amd.h"
#include "../../include/xz_amd_lzma.h"
#include "kernels_api.h"

#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LZMA_THREADS_MAX 16384
#define XZAMD_MAGIC 0x585A414D44474655ull

enum iseq { ISEQ_RUN, ISEQ_SYNC_FLUSH, ISEQ_FULL_FLUSH, ISEQ_FINISH, ISEQ_FULL_BARRIER, ISEQ_END, ISEQ_ERROR };
enum sseq { SEQ_HEADER, SEQ_BLOCKS, SEQ_TAIL, SEQ_DONE };

struct lzma_internal_s {
	uint64_t magic;
	enum iseq sequence;
	size_t avail_in;
	int allow_buf_error;
	/* encoder */
	xzamd_ctx *ctx;
	xzamd_lzma_options opt;
	uint64_t block_size;
	int check;
	enum sseq sseq;
	/* staging of caller input (pinned) */
	uint8_t *stage; uint64_t stage_len, stage_cap, stage_max;
	void *d_in; uint64_t d_in_cap;
	void *d_out; uint64_t d_out_cap;
	/* ordered output (pinned) */
	uint8_t *outq; uint64_t outq_pos, outq_len, outq_cap;
	/* Index records */
	uint64_t *rec; uint64_t nrec, rec_cap;
	xzamd_block_info *binfo; uint64_t binfo_cap;
	uint64_t progress_in, progress_out;
	const lzma_allocator *allocator;
};

static void *a_alloc(const lzma_allocator *a, size_t n)
{
	/* common/common.c:37-87: all host allocations go through strm->allocator */
	if (n == 0) n = 1;
	if (a && a->alloc) return a->alloc(a->opaque, 1, n);
	return malloc(n);
}

static void a_free(const lzma_allocator *a, void *p)
{
	if (a && a->free) a->free(a->opaque, p);
	else free(p);
}

static lzma_ret map_rc(int rc)
{
	switch (rc) {
	case XZAMD_OK: return LZMA_OK;
	case XZAMD_MEM_ERROR: return LZMA_MEM_ERROR;
	case XZAMD_OPTIONS_ERROR: return LZMA_OPTIONS_ERROR;
	case XZAMD_UNSUPPORTED_CHECK: return LZMA_UNSUPPORTED_CHECK;
	case XZAMD_BUF_ERROR: return LZMA_PROG_ERROR;
	default: return LZMA_PROG_ERROR;   /* device failures: SURVEY.md section 5 */
	}
}

/* One parked set of device/pinned resources: hipMalloc of the work buffers (tens of GiB for a 1 GiB
 * batch) and hipHostMalloc of the staging area cost seconds, so lzma_end() parks them and the next
 * lzma_stream_encoder_mt() of the process picks them up instead of allocating again. */
static struct {
	pthread_mutex_t mu;
	int full;
	xzamd_ctx *ctx;
	uint8_t *stage; uint64_t stage_cap;
	uint8_t *outq; uint64_t outq_cap;
	void *d_in; uint64_t d_in_cap;
	void *d_out; uint64_t d_out_cap;
} g_park = { PTHREAD_MUTEX_INITIALIZER, 0, NULL, NULL, 0, NULL, 0, NULL, 0, NULL, 0 };

static int park_resources(lzma_internal *in)
{
	int parked = 0;
	pthread_mutex_lock(&g_park.mu);
	if (!g_park.full && in->ctx) {
		g_park.ctx = in->ctx;
		g_park.stage = in->stage; g_park.stage_cap = in->stage_cap;
		g_park.outq = in->outq; g_park.outq_cap = in->outq_cap;
		g_park.d_in = in->d_in; g_park.d_in_cap = in->d_in_cap;
		g_park.d_out = in->d_out; g_park.d_out_cap = in->d_out_cap;
		g_park.full = 1;
		parked = 1;
	}
	pthread_mutex_unlock(&g_park.mu);
	return parked;
}

static int unpark_resources(lzma_internal *in)
{
	int got = 0, dev = -1;
	if (xzk_get_device(&dev))
		return 0;
	pthread_mutex_lock(&g_park.mu);
	if (g_park.full && xzamd_ctx_device(g_park.ctx) == dev) {
		in->ctx = g_park.ctx;
		in->stage = g_park.stage; in->stage_cap = g_park.stage_cap;
		in->outq = g_park.outq; in->outq_cap = g_park.outq_cap;
		in->d_in = g_park.d_in; in->d_in_cap = g_park.d_in_cap;
		in->d_out = g_park.d_out; in->d_out_cap = g_park.d_out_cap;
		g_park.full = 0;
		got = 1;
	}
	pthread_mutex_unlock(&g_park.mu);
	return got;
}

static void internal_free(lzma_internal *in)
{
	if (!in) return;
	const lzma_allocator *a = in->allocator;
	if (park_resources(in)) {
		in->ctx = NULL; in->stage = NULL; in->outq = NULL; in->d_in = NULL; in->d_out = NULL;
	}
	if (in->stage) xzk_host_free(in->stage);
	if (in->outq) xzk_host_free(in->outq);
	if (in->d_in) xzk_free(in->d_in);
	if (in->d_out) xzk_free(in->d_out);
	if (in->rec) a_free(a, in->rec);
	if (in->binfo) a_free(a, in->binfo);
	if (in->ctx) xzamd_ctx_destroy(in->ctx);
	in->magic = 0;
	a_free(a, in);
}

/* get_options(): stream_encoder_mt.c:956-1000 */
static lzma_ret parse_options(const lzma_mt *o, xzamd_lzma_options *opt, uint64_t *block_size, int *check)
{
	if (o == NULL)
		return LZMA_PROG_ERROR;
	if (o->flags != 0 || o->threads == 0 || o->threads > LZMA_THREADS_MAX)
		return LZMA_OPTIONS_ERROR;
	if (o->filters != NULL) {
		const lzma_filter *f = o->filters;
		uint32_t bcj = 0;
		if (f[0].id == LZMA_FILTER_X86) {
			/* {x86, LZMA2}: start offset must be 0 (bcj.h:81-98, NULL options = defaults) */
			const lzma_options_bcj *b = (const lzma_options_bcj *)f[0].options;
			if (b != NULL && b->start_offset != 0)
				return LZMA_OPTIONS_ERROR;
			bcj = XZAMD_BCJ_X86;
			++f;
		}
		if (f[0].id != LZMA_FILTER_LZMA2 || f[0].options == NULL || f[1].id != LZMA_VLI_UNKNOWN)
			return LZMA_OPTIONS_ERROR;      /* device path: {LZMA2} and {x86, LZMA2} */
		const lzma_options_lzma *l = (const lzma_options_lzma *)f[0].options;
		if (l->preset_dict != NULL && l->preset_dict_size != 0)
			return LZMA_OPTIONS_ERROR;
		if (l->lc > 4 || l->lp > 4 || l->lc + l->lp > 4 || l->pb > 4
				|| l->nice_len < 2 || l->nice_len > 273
				|| (l->mode != LZMA_MODE_FAST && l->mode != LZMA_MODE_NORMAL)
				|| l->dict_size < 4096 || l->dict_size > (1u << 30) + (1u << 29))
			return LZMA_OPTIONS_ERROR;
		memset(opt, 0, sizeof(*opt));
		opt->dict_size = l->dict_size;
		opt->lc = l->lc; opt->lp = l->lp; opt->pb = l->pb;
		opt->mode = (uint32_t)l->mode;
		opt->nice_len = l->nice_len;
		opt->mf = (uint32_t)l->mf;
		opt->depth = l->depth;
		if (l->mf == LZMA_MF_HC3 || l->mf == LZMA_MF_HC4) {
			opt->gpu_mf = (uint32_t)l->mf;
			opt->gpu_nice_len = l->nice_len < (opt->gpu_mf & 15) ? (opt->gpu_mf & 15) : l->nice_len;
			uint32_t d = l->depth ? l->depth : 4 + opt->gpu_nice_len / 4;
			opt->gpu_depth = d > 56 ? 56 : d;
			opt->gpu_parser = l->mode == LZMA_MODE_NORMAL ? 1 : 0;
		} else if (l->mf == LZMA_MF_BT2 || l->mf == LZMA_MF_BT3 || l->mf == LZMA_MF_BT4) {
			/* binary-tree finders -> suffix-neighbourhood finder + windowed optimal parser (the device has
			 * no fast parser for it: LZMA_MODE_FAST with a BT finder gets the optimal parser too) */
			opt->gpu_mf = XZAMD_MF_HC4;
			opt->gpu_nice_len = l->nice_len < 4 ? 4 : l->nice_len;
			opt->gpu_depth = 1;
			opt->gpu_sa_window = XZAMD_SA_WINDOW_MAX;
			opt->gpu_parser = 1;
		} else {
			return LZMA_OPTIONS_ERROR;
		}
		opt->span_size = XZAMD_SPAN_DEFAULT;
		opt->bcj = bcj;
	} else if (xzamd_lzma_preset(opt, o->preset)) {
		return LZMA_OPTIONS_ERROR;
	}
	{
		const char *au = getenv("XZAMD_SPAN_AUTO");
		if (au && *au == '1')
			opt->span_size = XZAMD_SPAN_AUTO;
	}
	const char *env = getenv("XZAMD_SPAN_KIB");
	if (env && atoi(env) > 0)
		opt->span_size = (uint32_t)atoi(env) << 10;
	*block_size = o->block_size ? o->block_size : xzamd_mt_block_size(opt);
	if (*block_size >= (1ull << 31))
		return LZMA_OPTIONS_ERROR;
	if ((unsigned)o->check > 15)
		return LZMA_PROG_ERROR;
	if (o->check != LZMA_CHECK_NONE && o->check != LZMA_CHECK_CRC32 && o->check != LZMA_CHECK_CRC64)
		return LZMA_UNSUPPORTED_CHECK;
	*check = (int)o->check;
	return LZMA_OK;
}

lzma_ret lzma_stream_encoder_mt(lzma_stream *strm, const lzma_mt *options)
{
	if (strm == NULL)
		return LZMA_PROG_ERROR;
	xzamd_lzma_options opt;
	uint64_t block_size = 0;
	int check = 0;
	lzma_ret r = parse_options(options, &opt, &block_size, &check);
	if (r != LZMA_OK)
		return r;
	/* lzma_next_strm_init (common.h:401-410): re-initialising a stream replaces its coder */
	if (strm->internal != NULL) {
		if (strm->internal->magic == XZAMD_MAGIC)
			internal_free(strm->internal);
		strm->internal = NULL;
	}
	lzma_internal *in = (lzma_internal *)a_alloc(strm->allocator, sizeof(*in));
	if (!in)
		return LZMA_MEM_ERROR;
	memset(in, 0, sizeof(*in));
	in->magic = XZAMD_MAGIC;
	in->allocator = strm->allocator;
	in->opt = opt;
	in->block_size = block_size;
	in->check = check;
	in->sequence = ISEQ_RUN;
	in->sseq = SEQ_HEADER;
	if (!unpark_resources(in)) {
		int rc = xzamd_ctx_create(&in->ctx, -1);
		if (rc) {
			internal_free(in);
			return rc == XZAMD_MEM_ERROR ? LZMA_MEM_ERROR : LZMA_PROG_ERROR;
		}
	}
	/* batch = whole Blocks, at most the context's device batch */
	uint64_t maxb = (1ull << 30) / block_size;
	const char *env = getenv("XZAMD_BATCH_MIB");
	if (env && atoll(env) > 0)
		maxb = ((uint64_t)atoll(env) << 20) / block_size;
	if (maxb == 0) maxb = 1;
	if (maxb * block_size >= (1ull << 31)) maxb = ((1ull << 31) - 1) / block_size;
	in->stage_max = maxb * block_size;
	strm->internal = in;
	strm->total_in = 0;
	strm->total_out = 0;
	return LZMA_OK;
}

uint64_t lzma_stream_encoder_mt_memusage(const lzma_mt *options)
{
	xzamd_lzma_options opt;
	uint64_t bs = 0;
	int check = 0;
	if (parse_options(options, &opt, &bs, &check) != LZMA_OK)
		return UINT64_MAX;
	uint64_t maxb = (1ull << 30) / bs;
	if (maxb == 0) maxb = 1;
	const uint64_t stage = maxb * bs;
	/* host: staging + output queue; device: input + output, 32 B/byte of sort buffers and chain tables
	 * (56 with the suffix-order build), span scratch, and for the optimal parser the match lists
	 * (32 B/byte packed, 48 B/byte for dictionaries above 8 MiB) */
	uint64_t per_byte = 2 + 2 + (opt.gpu_sa_window ? 56 : 32) + 2;
	if (opt.gpu_parser)
		per_byte += opt.dict_size <= (1u << 23) ? 32 : 48;
	return stage * per_byte;
}

static lzma_ret grow_pinned(uint8_t **buf, uint64_t *cap, uint64_t keep, uint64_t want)
{
	if (*cap >= want)
		return LZMA_OK;
	void *p = NULL;
	if (xzk_host_alloc(&p, want))
		return LZMA_MEM_ERROR;
	if (*buf) {
		if (keep) memcpy(p, *buf, keep);
		xzk_host_free(*buf);
	}
	*buf = (uint8_t *)p;
	*cap = want;
	return LZMA_OK;
}

/* Encode everything staged (whole Blocks, last one possibly short) and append the result to the
 * output queue.  worker_encode() x nblocks (stream_encoder_mt.c:219-361) in one device batch. */
static lzma_ret flush_stage(lzma_internal *in)
{
	if (in->stage_len == 0)
		return LZMA_OK;
	const uint64_t n = in->stage_len;
	const uint64_t nb = (n + in->block_size - 1) / in->block_size;
	const uint64_t bound = xzamd_stream_buffer_bound(n, in->block_size);
	if (in->d_in_cap < n) {
		if (in->d_in) xzk_free(in->d_in);
		in->d_in = NULL; in->d_in_cap = 0;
		if (xzk_malloc(&in->d_in, n)) return LZMA_MEM_ERROR;
		in->d_in_cap = n;
	}
	if (in->d_out_cap < bound) {
		if (in->d_out) xzk_free(in->d_out);
		in->d_out = NULL; in->d_out_cap = 0;
		if (xzk_malloc(&in->d_out, bound)) return LZMA_MEM_ERROR;
		in->d_out_cap = bound;
	}
	if (in->binfo_cap < nb) {
		if (in->binfo) a_free(in->allocator, in->binfo);
		in->binfo = (xzamd_block_info *)a_alloc(in->allocator, nb * sizeof(xzamd_block_info));
		if (!in->binfo) { in->binfo_cap = 0; return LZMA_MEM_ERROR; }
		in->binfo_cap = nb;
	}
	if (in->nrec + nb > in->rec_cap) {
		uint64_t nc = in->rec_cap ? in->rec_cap * 2 : 256;
		while (nc < in->nrec + nb) nc *= 2;
		uint64_t *nr = (uint64_t *)a_alloc(in->allocator, nc * 2 * sizeof(uint64_t));
		if (!nr) return LZMA_MEM_ERROR;
		if (in->rec) { memcpy(nr, in->rec, in->nrec * 2 * sizeof(uint64_t)); a_free(in->allocator, in->rec); }
		in->rec = nr;
		in->rec_cap = nc;
	}
	/* pending output must have been drained (we only flush when the queue is empty) */
	lzma_ret r = grow_pinned(&in->outq, &in->outq_cap, 0, bound);
	if (r != LZMA_OK) return r;

	if (xzk_h2d(in->d_in, in->stage, n, NULL) || xzk_sync(NULL))
		return LZMA_PROG_ERROR;
	uint64_t out_size = 0, nblocks = 0;
	int rc = xzamd_stream_encode_device(in->ctx, in->d_in, n, in->block_size, &in->opt, in->check,
			XZAMD_F_BLOCKS_ONLY, in->d_out, in->d_out_cap, &out_size, in->binfo, in->binfo_cap,
			&nblocks, NULL);
	if (rc != XZAMD_OK)
		return map_rc(rc);
	if (xzk_d2h(in->outq, in->d_out, out_size, NULL) || xzk_sync(NULL))
		return LZMA_PROG_ERROR;
	in->outq_pos = 0;
	in->outq_len = out_size;
	for (uint64_t i = 0; i < nblocks; ++i) {
		in->rec[2 * (in->nrec + i)] = in->binfo[i].unpadded_size;
		in->rec[2 * (in->nrec + i) + 1] = in->binfo[i].uncompressed_size;
	}
	in->nrec += nblocks;
	in->progress_in += n;
	in->stage_len = 0;
	return LZMA_OK;
}

static void drain(lzma_internal *in, uint8_t *out, size_t *out_pos, size_t out_size)
{
	uint64_t n = in->outq_len - in->outq_pos;
	if (n > out_size - *out_pos) n = out_size - *out_pos;
	if (n) {
		memcpy(out + *out_pos, in->outq + in->outq_pos, n);
		in->outq_pos += n;
		*out_pos += n;
		in->progress_out += n;
	}
}

/* stream_encode_mt(): stream_encoder_mt.c:717-883 */
static lzma_ret stream_code(lzma_internal *in, const uint8_t *inb, size_t *in_pos, size_t in_size,
		uint8_t *out, size_t *out_pos, size_t out_size, lzma_action action)
{
	for (;;) {
		switch (in->sseq) {
		case SEQ_HEADER: {
			lzma_ret r = grow_pinned(&in->outq, &in->outq_cap, 0, 4096);
			if (r != LZMA_OK) return r;
			in->outq_len = xzamd_frame_header(in->outq, in->check);
			in->outq_pos = 0;
			in->sseq = SEQ_BLOCKS;
			break;
		}
		case SEQ_BLOCKS: {
			drain(in, out, out_pos, out_size);
			if (in->outq_pos < in->outq_len)
				return LZMA_OK;             /* output full */
			/* take input (stream_encode_in: :599-664) */
			while (*in_pos < in_size) {
				if (in->stage_len == in->stage_max)
					break;
				if (in->stage_len == in->stage_cap) {
					/* second growth step goes straight to the full batch: each step is a pinned allocation + copy */
					uint64_t nc = in->stage_cap ? in->stage_max : in->block_size;
					if (nc < (1u << 20)) nc = 1u << 20;
					if (nc > in->stage_max) nc = in->stage_max;
					lzma_ret r = grow_pinned(&in->stage, &in->stage_cap, in->stage_len, nc);
					if (r != LZMA_OK) return r;
				}
				uint64_t room = in->stage_cap - in->stage_len;
				uint64_t take = in_size - *in_pos;
				if (take > room) take = room;
				memcpy(in->stage + in->stage_len, inb + *in_pos, take);
				in->stage_len += take;
				*in_pos += take;
			}
			const int input_done = *in_pos == in_size;
			if (in->stage_len == in->stage_max || (input_done && action != LZMA_RUN && in->stage_len)) {
				lzma_ret r = flush_stage(in);
				if (r != LZMA_OK) return r;
				break;                       /* drain, then continue */
			}
			if (!input_done)
				break;
			if (action == LZMA_RUN)
				return LZMA_OK;             /* :796-801 */
			/* all input handed over and encoded, queue empty */
			if (action == LZMA_FULL_FLUSH || action == LZMA_FULL_BARRIER)
				return LZMA_STREAM_END;     /* :803-823 */
			/* LZMA_FINISH: Index + Stream Footer (:842-883) */
			{
				const uint64_t cap = 64 + in->nrec * 18;
				lzma_ret r = grow_pinned(&in->outq, &in->outq_cap, 0, cap);
				if (r != LZMA_OK) return r;
				uint64_t *unp = (uint64_t *)a_alloc(in->allocator, (in->nrec + 1) * 2 * sizeof(uint64_t));
				if (!unp) return LZMA_MEM_ERROR;
				uint64_t *unc = unp + in->nrec + 1;
				for (uint64_t i = 0; i < in->nrec; ++i) { unp[i] = in->rec[2 * i]; unc[i] = in->rec[2 * i + 1]; }
				in->outq_len = xzamd_frame_index_footer(in->outq, in->outq_cap, in->check, unp, unc, in->nrec);
				in->outq_pos = 0;
				a_free(in->allocator, unp);
				if (in->outq_len == 0) return LZMA_PROG_ERROR;
				in->sseq = SEQ_TAIL;
			}
			break;
		}
		case SEQ_TAIL:
			drain(in, out, out_pos, out_size);
			if (in->outq_pos < in->outq_len)
				return LZMA_OK;
			in->sseq = SEQ_DONE;
			return LZMA_STREAM_END;
		case SEQ_DONE:
			return LZMA_STREAM_END;
		}
	}
}

lzma_ret lzma_code(lzma_stream *strm, lzma_action action)
{
	/* common/common.c:203-376, restated for the one coder this library owns */
	if (strm == NULL || (strm->next_in == NULL && strm->avail_in != 0)
			|| (strm->next_out == NULL && strm->avail_out != 0)
			|| strm->internal == NULL || strm->internal->magic != XZAMD_MAGIC
			|| (unsigned)action > LZMA_FULL_BARRIER || action == LZMA_SYNC_FLUSH)
		return LZMA_PROG_ERROR;
	if (strm->reserved_ptr1 != NULL || strm->reserved_ptr2 != NULL || strm->reserved_ptr3 != NULL
			|| strm->reserved_ptr4 != NULL || strm->reserved_int2 != 0 || strm->reserved_int3 != 0
			|| strm->reserved_int4 != 0 || strm->reserved_enum1 != LZMA_RESERVED_ENUM
			|| strm->reserved_enum2 != LZMA_RESERVED_ENUM)
		return LZMA_OPTIONS_ERROR;
	lzma_internal *in = strm->internal;
	switch (in->sequence) {
	case ISEQ_RUN:
		if (action == LZMA_FULL_FLUSH) in->sequence = ISEQ_FULL_FLUSH;
		else if (action == LZMA_FINISH) in->sequence = ISEQ_FINISH;
		else if (action == LZMA_FULL_BARRIER) in->sequence = ISEQ_FULL_BARRIER;
		break;
	case ISEQ_FULL_FLUSH:
		if (action != LZMA_FULL_FLUSH || in->avail_in != strm->avail_in) return LZMA_PROG_ERROR;
		break;
	case ISEQ_FINISH:
		if (action != LZMA_FINISH || in->avail_in != strm->avail_in) return LZMA_PROG_ERROR;
		break;
	case ISEQ_FULL_BARRIER:
		if (action != LZMA_FULL_BARRIER || in->avail_in != strm->avail_in) return LZMA_PROG_ERROR;
		break;
	case ISEQ_END:
		return LZMA_STREAM_END;
	default:
		return LZMA_PROG_ERROR;
	}
	size_t in_pos = 0, out_pos = 0;
	lzma_ret ret = stream_code(in, strm->next_in, &in_pos, strm->avail_in,
			strm->next_out, &out_pos, strm->avail_out, action);
	if (in_pos) { strm->next_in += in_pos; strm->avail_in -= in_pos; strm->total_in += in_pos; }
	if (out_pos) { strm->next_out += out_pos; strm->avail_out -= out_pos; strm->total_out += out_pos; }
	in->avail_in = strm->avail_in;
	switch (ret) {
	case LZMA_OK:
		if (out_pos == 0 && in_pos == 0) {
			if (in->allow_buf_error) ret = LZMA_BUF_ERROR;
			else in->allow_buf_error = 1;
		} else {
			in->allow_buf_error = 0;
		}
		break;
	case LZMA_STREAM_END:
		if (in->sequence == ISEQ_FULL_FLUSH || in->sequence == ISEQ_FULL_BARRIER)
			in->sequence = ISEQ_RUN;
		else
			in->sequence = ISEQ_END;
		in->allow_buf_error = 0;
		break;
	case LZMA_UNSUPPORTED_CHECK:
		in->allow_buf_error = 0;
		break;
	default:
		in->sequence = ISEQ_ERROR;
		break;
	}
	return ret;
}

void lzma_end(lzma_stream *strm)
{
	if (strm != NULL && strm->internal != NULL) {
		if (strm->internal->magic == XZAMD_MAGIC)
			internal_free(strm->internal);
		strm->internal = NULL;
	}
}

void lzma_get_progress(lzma_stream *strm, uint64_t *progress_in, uint64_t *progress_out)
{
	if (strm && strm->internal && strm->internal->magic == XZAMD_MAGIC) {
		*progress_in = strm->internal->progress_in;
		*progress_out = strm->internal->progress_out;
	} else if (strm) {
		*progress_in = strm->total_in;
		*progress_out = strm->total_out;
	}
}

/* ------------------------------------------------------------------ */
/* One-shot buffer API (common/stream_buffer_encoder.c:43-141,          */
/* common/easy_buffer_encoder.c:16-27): same engine, no caller-visible  */
/* streaming state.  The reference writes ONE Block here; this writes   */
/* the MT layout (a Block per block_size bytes), which decodes the same.*/
/* ------------------------------------------------------------------ */
size_t lzma_stream_buffer_bound(size_t uncompressed_size)
{
	/* stream_buffer_encoder.c:24-40 semantics: 0 = too big */
	const uint64_t bs = 1ull << 20;           /* smallest default Block size (3 x 256 KiB dict -> 1 MiB floor) */
	const uint64_t b = xzamd_stream_buffer_bound(uncompressed_size, bs);
	return b > (uint64_t)SIZE_MAX ? 0 : (size_t)b;
}

static lzma_ret buffer_encode(const lzma_mt *mt, const lzma_allocator *allocator,
		const uint8_t *in, size_t in_size, uint8_t *out, size_t *out_pos_ptr, size_t out_size)
{
	if (out == NULL || out_pos_ptr == NULL || *out_pos_ptr > out_size || (in == NULL && in_size != 0))
		return LZMA_PROG_ERROR;
	lzma_stream strm;
	memset(&strm, 0, sizeof(strm));
	strm.allocator = allocator;
	lzma_ret r = lzma_stream_encoder_mt(&strm, mt);
	if (r != LZMA_OK)
		return r;
	strm.next_in = in;
	strm.avail_in = in_size;
	strm.next_out = out + *out_pos_ptr;
	strm.avail_out = out_size - *out_pos_ptr;
	do {
		r = lzma_code(&strm, LZMA_FINISH);
	} while (r == LZMA_OK && strm.avail_out > 0);
	const size_t produced = (out_size - *out_pos_ptr) - strm.avail_out;
	lzma_end(&strm);
	if (r == LZMA_STREAM_END) {
		*out_pos_ptr += produced;
		return LZMA_OK;
	}
	/* output did not fit: *out_pos is left untouched like the reference does (:128-137) */
	return r == LZMA_OK ? LZMA_BUF_ERROR : r;
}

lzma_ret lzma_stream_buffer_encode(lzma_filter *filters, lzma_check check, const lzma_allocator *allocator,
		const uint8_t *in, size_t in_size, uint8_t *out, size_t *out_pos, size_t out_size)
{
	if (filters == NULL)
		return LZMA_PROG_ERROR;
	lzma_mt mt;
	memset(&mt, 0, sizeof(mt));
	mt.threads = 1;
	mt.filters = filters;
	mt.check = check;
	return buffer_encode(&mt, allocator, in, in_size, out, out_pos, out_size);
}

lzma_ret lzma_easy_buffer_encode(uint32_t preset, lzma_check check, const lzma_allocator *allocator,
		const uint8_t *in, size_t in_size, uint8_t *out, size_t *out_pos, size_t out_size)
{
	lzma_mt mt;
	memset(&mt, 0, sizeof(mt));
	mt.threads = 1;
	mt.preset = preset;
	mt.check = check;
	return buffer_encode(&mt, allocator, in, in_size, out, out_pos, out_size);
}
