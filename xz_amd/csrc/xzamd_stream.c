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
 *   lzma_filters_update     <- stream_encoder_mt.c:914-950 (between Blocks only)
 *   lzma_mt_block_size, lzma_cputhreads <- filter_encoder.c:270, hardware_cputhreads.c
 *
 * stream_encode_mt() (:717-883) copies caller input into per-worker Block
 * buffers and hands them to worker threads (get_thread :540-597,
 * stream_encode_in :599-665), results come back through an ordered output
 * queue (outqueue.c:182-260).  Same shape here with a GPU per worker: the
 * caller's bytes are copied into a pinned staging buffer holding a batch of
 * whole Blocks ("job"); full jobs are dealt in order to one worker thread per
 * visible GPU (own xzamd_ctx each), which uploads, encodes
 * (xzamd_stream_encode_device, BLOCKS_ONLY) and downloads into the job's pinned
 * output buffer; the calling thread drains finished jobs strictly in order,
 * keeps the Index records, and fills the next job meanwhile (staging, H2D,
 * encode and D2H of consecutive batches overlap).  lzma_mt.timeout bounds the
 * time a call may wait for the workers, LZMA_FULL_BARRIER returns once the
 * input is handed over (:803-807).
 */
#include "../../include/xz_amd.h"
#include "../../include/xz_amd_lzma.h"
#include "kernels_api.h"
#include "xzamd_internal.h"

#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LZMA_THREADS_MAX 16384
#define XZAMD_MAGIC 0x585A414D44474655ull
#define MAX_DEVS 16
#define MAX_JOBS (2 * MAX_DEVS + 1)
#define JOB_BYTES (1280ull << 20)     /* a full job (see lzma_stream_encoder_mt) */

#include <errno.h>
#include <time.h>
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <unistd.h>

/* XZAMD_DEBUG_SEGV=1: print a backtrace on SIGSEGV (debug aid for the interposed-client case) */
static void segv_handler(int sig)
{
	void *bt[64];
	int n = backtrace(bt, 64);
	backtrace_symbols_fd(bt, n, 2);
	signal(sig, SIG_DFL);
	raise(sig);
}

enum iseq { ISEQ_RUN, ISEQ_SYNC_FLUSH, ISEQ_FULL_FLUSH, ISEQ_FINISH, ISEQ_FULL_BARRIER, ISEQ_END, ISEQ_ERROR };
enum sseq { SEQ_HEADER, SEQ_BLOCKS, SEQ_TAIL, SEQ_DONE };
enum jstate { J_FREE, J_FILLING, J_QUEUED, J_RUNNING, J_DONE };

/* One batch of whole Blocks on its way through a worker. */
typedef struct {
	enum jstate state;
	uint64_t seq;
	uint8_t *stage; uint64_t stage_len, stage_cap;       /* pinned: caller bytes */
	uint8_t *out; uint64_t out_len, out_pos, out_cap;     /* pinned: encoded Blocks */
	xzamd_block_info *binfo; uint64_t binfo_cap, nblocks;
	xzamd_lzma_options opt;                               /* options in force when the job was queued */
	lzma_ret err;
} job;

/* Device-side input and output of one job. */
typedef struct { void *d_in, *d_out; uint64_t d_in_cap, d_out_cap; } devbufs;

/* One GPU: context + device buffers, driven by one worker thread.  Two sets of job buffers: while the back end of one
 * job (range coder, checks, gather) is still running on the context's second stream, the worker uploads and launches the
 * next one (xzamd_encode_device_ with `deferred`). */
typedef struct {
	int device;
	xzamd_ctx *ctx;
	devbufs io[2];
	int flip;                    /* which of io[] the next job takes */
	pthread_t thr;
	int started;
	const job *running;          /* the job this worker is on (written and read under owner->mu) */
	job *pending;                /* the job before it, launched with its back end still in flight (same lock) */
	int pending_io;
	struct lzma_internal_s *owner;
} devslot;

struct lzma_internal_s {
	uint64_t magic;
	enum iseq sequence;
	size_t avail_in;
	int allow_buf_error;
	/* encoder */
	xzamd_lzma_options opt;
	uint64_t block_size;
	int check;
	uint32_t timeout_ms;
	enum sseq sseq;
	uint64_t stage_max;
	uint64_t test_job_min;       /* XZAMD_TEST_JOB_MIN_MIB << 20 (tests: reach the growing-jobs path on small inputs), 0 = none */
	/* workers */
	devslot dev[MAX_DEVS]; int ndev;
	uint64_t seen_in;             /* input bytes staged since the stream was initialised (sizes the jobs of several GPUs) */
	job jobs[MAX_JOBS]; int njobs;
	int fill;                    /* job being filled by the calling thread, -1 = none */
	int ndone;                   /* jobs in J_DONE (written under mu, read without it by the calling thread: a stale 0 only
	                              * delays the drain by one call) */
	uint64_t next_seq, drain_seq;
	pthread_mutex_t mu;
	pthread_cond_t cv_work, cv_done;
	int shutdown, mu_ok;
	/* small framing pieces (Stream Header, Index + Footer) */
	uint8_t *tailbuf; uint64_t tail_pos, tail_len, tail_cap;
	/* Index records */
	uint64_t *rec; uint64_t nrec, rec_cap;
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

/* Parked resources: hipMalloc of the work buffers (tens of GiB for a 1 GiB batch) and hipHostMalloc of the
 * staging areas cost seconds, so lzma_end() parks them (one set per device, a pool of pinned buffers) and
 * the next lzma_stream_encoder_mt() of the process picks them up.  xzamd_release_parked() gives them back;
 * XZAMD_NO_PARK=1 disables parking. */
static struct {
	pthread_mutex_t mu;
	struct { int full; xzamd_ctx *ctx; devbufs io[2]; } dev[MAX_DEVS];
	struct { uint8_t *p; uint64_t cap; } pinned[2 * MAX_JOBS];
} g_park = { .mu = PTHREAD_MUTEX_INITIALIZER };

static int parking_enabled(void)
{
	const char *e = getenv("XZAMD_NO_PARK");
	return !(e && *e == '1');
}

static void park_pinned(uint8_t *p, uint64_t cap)
{
	if (!p) return;
	if (parking_enabled()) {
		pthread_mutex_lock(&g_park.mu);
		for (int i = 0; i < 2 * MAX_JOBS; ++i)
			if (!g_park.pinned[i].p) {
				g_park.pinned[i].p = p; g_park.pinned[i].cap = cap;
				pthread_mutex_unlock(&g_park.mu);
				return;
			}
		pthread_mutex_unlock(&g_park.mu);
	}
	xzk_host_free(p);
}

static uint8_t *unpark_pinned(uint64_t want, uint64_t *cap)
{
	uint8_t *p = NULL;
	pthread_mutex_lock(&g_park.mu);
	for (int i = 0; i < 2 * MAX_JOBS; ++i)
		if (g_park.pinned[i].p && g_park.pinned[i].cap >= want) {
			p = g_park.pinned[i].p; *cap = g_park.pinned[i].cap;
			g_park.pinned[i].p = NULL; g_park.pinned[i].cap = 0;
			break;
		}
	pthread_mutex_unlock(&g_park.mu);
	return p;
}

void xzamd_release_parked(void)
{
	pthread_mutex_lock(&g_park.mu);
	for (int i = 0; i < MAX_DEVS; ++i)
		if (g_park.dev[i].full) {
			xzk_set_device(xzamd_ctx_device(g_park.dev[i].ctx));
			for (int k = 0; k < 2; ++k) {
				if (g_park.dev[i].io[k].d_in) xzk_free(g_park.dev[i].io[k].d_in);
				if (g_park.dev[i].io[k].d_out) xzk_free(g_park.dev[i].io[k].d_out);
			}
			xzamd_ctx_destroy(g_park.dev[i].ctx);
			memset(&g_park.dev[i], 0, sizeof(g_park.dev[i]));
		}
	for (int i = 0; i < 2 * MAX_JOBS; ++i)
		if (g_park.pinned[i].p) {
			xzk_host_free(g_park.pinned[i].p);
			g_park.pinned[i].p = NULL; g_park.pinned[i].cap = 0;
		}
	pthread_mutex_unlock(&g_park.mu);
}

static void park_dev(devslot *d)
{
	if (!d->ctx) return;
	if (parking_enabled()) {
		pthread_mutex_lock(&g_park.mu);
		for (int i = 0; i < MAX_DEVS; ++i)
			if (!g_park.dev[i].full) {
				g_park.dev[i].full = 1; g_park.dev[i].ctx = d->ctx;
				memcpy(g_park.dev[i].io, d->io, sizeof(d->io));
				pthread_mutex_unlock(&g_park.mu);
				d->ctx = NULL; memset(d->io, 0, sizeof(d->io));
				return;
			}
		pthread_mutex_unlock(&g_park.mu);
	}
	xzk_set_device(d->device);
	for (int k = 0; k < 2; ++k) {
		if (d->io[k].d_in) xzk_free(d->io[k].d_in);
		if (d->io[k].d_out) xzk_free(d->io[k].d_out);
	}
	xzamd_ctx_destroy(d->ctx);
	d->ctx = NULL; memset(d->io, 0, sizeof(d->io));
}

static int unpark_dev(devslot *d, int device)
{
	int got = 0;
	pthread_mutex_lock(&g_park.mu);
	for (int i = 0; i < MAX_DEVS; ++i)
		if (g_park.dev[i].full && xzamd_ctx_device(g_park.dev[i].ctx) == device) {
			d->ctx = g_park.dev[i].ctx;
			memcpy(d->io, g_park.dev[i].io, sizeof(d->io));
			memset(&g_park.dev[i], 0, sizeof(g_park.dev[i]));
			got = 1;
			break;
		}
	pthread_mutex_unlock(&g_park.mu);
	return got;
}

static lzma_ret grow_pinned(uint8_t **buf, uint64_t *cap, uint64_t keep, uint64_t want)
{
	if (*cap >= want)
		return LZMA_OK;
	uint64_t ncap = 0;
	uint8_t *p = unpark_pinned(want, &ncap);
	if (!p) {
		void *q = NULL;
		if (xzk_host_alloc(&q, want))
			return LZMA_MEM_ERROR;
		p = (uint8_t *)q;
		ncap = want;
	}
	if (*buf) {
		if (keep) memcpy(p, *buf, keep);
		park_pinned(*buf, *cap);
	}
	*buf = p;
	*cap = ncap;
	return LZMA_OK;
}

static void vlog(const char *what, const job *j);

/* worker_encode() x nblocks (stream_encoder_mt.c:219-361) for one job on one GPU, in two steps.  job_launch: buffers, upload,
 * device encode -- which may come back DEFERRED: everything of the job is launched, the back end of its last batch still runs
 * (the next job_launch on this context finishes it underneath its own front end).  job_collect: the result of
 * the device encode (the deferred one is fetched with xzamd_encode_finish_), the stored-Block way out, the download. */
static lzma_ret job_launch(lzma_internal *in, devslot *d, job *j, int io, int *rc_out, uint64_t *out_size, int *deferred)
{
	const uint64_t n = j->stage_len;
	const uint64_t nb = (n + in->block_size - 1) / in->block_size;
	const uint64_t bound = xzamd_stream_buffer_bound(n, in->block_size);
	devbufs *b = &d->io[io];
	*deferred = 0;
	*out_size = 0;
	*rc_out = XZAMD_DEVICE_ERROR;
	if (xzk_set_device(d->device))
		return LZMA_PROG_ERROR;
	if (b->d_in_cap < n) {
		if (b->d_in) xzk_free(b->d_in);
		b->d_in = NULL; b->d_in_cap = 0;
		if (xzk_malloc(&b->d_in, n)) return LZMA_MEM_ERROR;
		b->d_in_cap = n;
	}
	if (b->d_out_cap < bound) {
		if (b->d_out) xzk_free(b->d_out);
		b->d_out = NULL; b->d_out_cap = 0;
		if (xzk_malloc(&b->d_out, bound)) return LZMA_MEM_ERROR;
		b->d_out_cap = bound;
	}
	if (j->binfo_cap < nb) {
		free(j->binfo);
		j->binfo = (xzamd_block_info *)malloc(nb * sizeof(xzamd_block_info));
		if (!j->binfo) { j->binfo_cap = 0; return LZMA_MEM_ERROR; }
		j->binfo_cap = nb;
	}
	/* (the pinned output slot is sized in job_collect, when the job's encoded size is known: a pinned allocation costs about a
	 * second per GiB and holds up every other HIP call of the process meanwhile -- round 6: three slots of the full bound were
	 * 3.9 GiB, most of it never written) */
	/* XZAMD_TEST_FAIL_JOB=k (tests): the k-th job behaves as if the device had failed */
	const char *tf = getenv("XZAMD_TEST_FAIL_JOB");
	const int injected = tf && *tf && (uint64_t)atoll(tf) == j->seq;
	/* XZAMD_NO_DEFER=1 (measurement knob): every job is finished before the next one is launched */
	const char *nd = getenv("XZAMD_NO_DEFER");
	/* Copies go through the context's own (front-end) stream: it is idle between two jobs, and unlike the null stream it
	 * is known not to share a hardware queue with the second stream, where the back end of the job before may be running
	 * (measured: a null-stream copy waited ~250 ms for those kernels) */
	void *cs = xzamd_ctx_stream_(d->ctx);
	if (!injected && !(xzk_h2d(b->d_in, j->stage, n, cs) || xzk_sync(cs)) && (vlog("uploaded", j), 1))
		*rc_out = xzamd_encode_device_(d->ctx, b->d_in, n, in->block_size, &j->opt, in->check,
				XZAMD_F_BLOCKS_ONLY, b->d_out, b->d_out_cap, out_size, j->binfo, j->binfo_cap,
				&j->nblocks, NULL, (nd && *nd == '1') ? NULL : deferred);
	return LZMA_OK;
}

static lzma_ret job_collect(lzma_internal *in, devslot *d, job *j, int io, int rc, uint64_t out_size)
{
	const uint64_t n = j->stage_len;
	if (rc == XZAMD_OK && out_size > j->out_cap) {
		/* the pinned slot the encoded Blocks are downloaded into: what this job needs, in steps of 64 MiB */
		const uint64_t bound = xzamd_stream_buffer_bound(n, in->block_size);
		uint64_t want = (out_size + (64ull << 20) - 1) & ~((64ull << 20) - 1);
		if (want > bound) want = bound;
		if (grow_pinned(&j->out, &j->out_cap, 0, want) != LZMA_OK) return LZMA_MEM_ERROR;
	}
	void *cs = xzamd_ctx_stream_(d->ctx);
	if (rc == XZAMD_OK && (xzk_d2h(j->out, d->io[io].d_out, out_size, cs) || xzk_sync(cs)))
		rc = XZAMD_DEVICE_ERROR;
	if (rc == XZAMD_DEVICE_ERROR) {
		/* A device (HIP runtime / kernel launch) failure in the middle of a Stream.  Default: the Stream fails
		 * (LZMA_PROG_ERROR, latched).  With XZAMD_STORED_ON_DEVICE_ERROR=1 the job's Blocks are stored instead (the
		 * reference's own way out when a Block cannot be coded, block_buffer_encoder.c:88-162), so what the client has
		 * written so far stays a valid .xz Stream -- no encoding happens on the host.  The encoder's own consistency
		 * failures (XZAMD_PROG_ERROR) are never papered over this way. */
		const char *sf = getenv("XZAMD_STORED_ON_DEVICE_ERROR");
		uint64_t nblocks = 0;
		if (sf && *sf == '1' && grow_pinned(&j->out, &j->out_cap, 0, xzamd_stream_buffer_bound(n, in->block_size)) == LZMA_OK
				&& xzamd_stored_blocks_host_(j->stage, n, in->block_size, in->check, j->out, j->out_cap, &out_size,
						j->binfo, j->binfo_cap, &nblocks) == XZAMD_OK) {
			if (getenv("XZAMD_VERBOSE"))
				fprintf(stderr, "xz_amd: job of %llu bytes stored after a device error: %s\n", (unsigned long long)n,
						xzamd_last_error(d->ctx));
			j->nblocks = nblocks;
			rc = XZAMD_OK;
		}
	}
	if (rc != XZAMD_OK) {
		if (getenv("XZAMD_VERBOSE"))
			fprintf(stderr, "xz_amd: job %llu failed (%d): %s\n", (unsigned long long)j->seq, rc, xzamd_last_error(d->ctx));
		return map_rc(rc);
	}
	j->out_len = out_size;
	j->out_pos = 0;
	return LZMA_OK;
}

/* XZAMD_VERBOSE=2: a line per step of a worker, with the time since the first one (how the jobs overlap) */
static struct timespec vlog_t0;
static int vlog_on;
static pthread_once_t vlog_once = PTHREAD_ONCE_INIT;
static void vlog_init(void)
{
	const char *e = getenv("XZAMD_VERBOSE");
	vlog_on = e && atoi(e) >= 2;
	clock_gettime(CLOCK_MONOTONIC, &vlog_t0);
}

static void vlog(const char *what, const job *j)
{
	pthread_once(&vlog_once, vlog_init);
	if (!vlog_on) return;
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	fprintf(stderr, "xz_amd: %8.1f ms  job %llu (%llu MiB) %s\n",
			(double)(t.tv_sec - vlog_t0.tv_sec) * 1e3 + (double)(t.tv_nsec - vlog_t0.tv_nsec) * 1e-6,
			(unsigned long long)j->seq, (unsigned long long)(j->stage_len >> 20), what);
}

/* the job is finished: hand it to the draining thread (in->mu held) */
static void job_done_locked(lzma_internal *in, job *j, lzma_ret r)
{
	j->err = r;
	j->state = J_DONE;
	__atomic_add_fetch(&in->ndone, 1, __ATOMIC_RELEASE);
	in->progress_in += j->stage_len;
	pthread_cond_broadcast(&in->cv_done);
}

/* The deferred job of this worker: fetch its result (its device work ends here at the latest), download, report. */
static void finish_pending(lzma_internal *in, devslot *d)
{
	job *p = d->pending;
	uint64_t out_size = 0;
	xzk_set_device(d->device);
	vlog("finish: wait", p);
	const int rc = xzamd_encode_finish_(d->ctx, &out_size);
	vlog("finish: download", p);
	const lzma_ret r = job_collect(in, d, p, d->pending_io, rc, out_size);
	vlog("finish: done", p);
	pthread_mutex_lock(&in->mu);
	d->pending = NULL;
	job_done_locked(in, p, r);
	pthread_mutex_unlock(&in->mu);
}

/* Worker thread: takes the oldest queued job (jobs are dealt strictly in order, stream_encode_in
 * :599-665), runs it on its GPU, reports it done. */
static void *worker_main(void *arg)
{
	devslot *d = (devslot *)arg;
	lzma_internal *in = d->owner;
	pthread_mutex_lock(&in->mu);
	for (;;) {
		int pick = -1;
		for (int i = 0; i < in->njobs; ++i)
			if (in->jobs[i].state == J_QUEUED && (pick < 0 || in->jobs[i].seq < in->jobs[pick].seq))
				pick = i;
		if (pick < 0) {
			if (d->pending) {
				/* nothing to launch on top of the deferred job: finish it now */
				pthread_mutex_unlock(&in->mu);
				finish_pending(in, d);
				pthread_mutex_lock(&in->mu);
				continue;
			}
			if (in->shutdown)
				break;
			pthread_cond_wait(&in->cv_work, &in->mu);
			continue;
		}
		job *j = &in->jobs[pick];
		j->state = J_RUNNING;
		d->running = j;
		const int io = d->flip;
		d->flip ^= 1;
		pthread_mutex_unlock(&in->mu);
		int rc = XZAMD_DEVICE_ERROR, deferred = 0;
		uint64_t out_size = 0;
		vlog("launch", j);
		lzma_ret r = job_launch(in, d, j, io, &rc, &out_size, &deferred);
		vlog(deferred ? "launched, back end in flight" : "encoded", j);
		/* whatever was deferred before has been finished underneath (or in front of) this launch */
		if (d->pending)
			finish_pending(in, d);
		if (r == LZMA_OK && !deferred)
			r = job_collect(in, d, j, io, rc, out_size);
		pthread_mutex_lock(&in->mu);
		d->running = NULL;
		xzamd_ctx_progress_reset_(d->ctx);      /* the job counts as a whole from here on (progress_in, or `pending`) */
		if (r == LZMA_OK && deferred) {
			d->pending = j;
			d->pending_io = io;
		} else {
			job_done_locked(in, j, r);
		}
	}
	pthread_mutex_unlock(&in->mu);
	return NULL;
}

static void internal_free(lzma_internal *in)
{
	if (!in) return;
	const lzma_allocator *a = in->allocator;
	if (in->mu_ok) {
		pthread_mutex_lock(&in->mu);
		in->shutdown = 1;
		/* abandon what is still queued; running jobs finish */
		for (int i = 0; i < in->njobs; ++i)
			if (in->jobs[i].state == J_QUEUED) in->jobs[i].state = J_FREE;
		pthread_cond_broadcast(&in->cv_work);
		pthread_mutex_unlock(&in->mu);
		for (int i = 0; i < in->ndev; ++i)
			if (in->dev[i].started) pthread_join(in->dev[i].thr, NULL);
		pthread_cond_destroy(&in->cv_work);
		pthread_cond_destroy(&in->cv_done);
		pthread_mutex_destroy(&in->mu);
	}
	for (int i = 0; i < in->ndev; ++i)
		park_dev(&in->dev[i]);
	for (int i = 0; i < in->njobs; ++i) {
		park_pinned(in->jobs[i].stage, in->jobs[i].stage_cap);
		park_pinned(in->jobs[i].out, in->jobs[i].out_cap);
		free(in->jobs[i].binfo);
	}
	if (in->tailbuf) a_free(a, in->tailbuf);
	if (in->rec) a_free(a, in->rec);
	in->magic = 0;
	a_free(a, in);
}

/* No library destructor frees the parked set: at process exit the HIP runtime may already be gone (its own
 * teardown runs first), and the process's device memory goes away with it anyway.  A long-running client
 * that is done compressing calls xzamd_release_parked(). */

/* get_options(): stream_encoder_mt.c:956-1000 */
static lzma_ret parse_options(const lzma_mt *o, xzamd_lzma_options *opt, uint64_t *block_size, int *check)
{
	if (o == NULL)
		return LZMA_PROG_ERROR;
	if (o->flags != 0 || o->threads == 0 || o->threads > LZMA_THREADS_MAX)
		return LZMA_OPTIONS_ERROR;
	if (o->filters != NULL) {
		const lzma_filter *f = o->filters;
		uint32_t pre[XZAMD_PREFILTERS_MAX] = { 0, 0, 0 }, npre = 0;
		/* up to three filters in front of LZMA2 (a chain holds at most LZMA_FILTERS_MAX = 4, the last one must be
		 * LZMA2 here; BCJ and delta filters may come in any order: common/filter_common.c:250-334) */
		while (f[0].id != LZMA_FILTER_LZMA2) {
			if (npre == XZAMD_PREFILTERS_MAX)
				return LZMA_OPTIONS_ERROR;
			if (f[0].id >= LZMA_FILTER_X86 && f[0].id <= LZMA_FILTER_RISCV) {
				/* x86, PowerPC, IA-64, ARM, ARM-Thumb, SPARC, ARM64, RISC-V (the filter ids are the values of
				 * xzamd_lzma_options.bcj): start offset must be 0 (bcj.h:81-98, NULL options = defaults) */
				const lzma_options_bcj *b = (const lzma_options_bcj *)f[0].options;
				if (b != NULL && b->start_offset != 0)
					return LZMA_OPTIONS_ERROR;
				pre[npre++] = (uint32_t)f[0].id;
			} else if (f[0].id == LZMA_FILTER_DELTA) {
				/* delta/delta_common.c:38-60 */
				const lzma_options_delta *dl = (const lzma_options_delta *)f[0].options;
				if (dl == NULL || dl->type != LZMA_DELTA_TYPE_BYTE || dl->dist < 1 || dl->dist > 256)
					return LZMA_OPTIONS_ERROR;
				pre[npre++] = XZAMD_FILTER_DELTA(dl->dist);
			} else {
				return LZMA_OPTIONS_ERROR;      /* incl. LZMA_VLI_UNKNOWN: a chain without LZMA2 */
			}
			++f;
		}
		if (f[0].options == NULL || f[1].id != LZMA_VLI_UNKNOWN)
			return LZMA_OPTIONS_ERROR;      /* device path: {LZMA2} and {BCJ | delta ..., LZMA2} */
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
		/* pb = 3, 4 with LZMA_MODE_NORMAL needs the two-phase mode (xzamd_options_check), which comes with the
		 * suffix-neighbourhood finder: a hash-chain finder is mapped like a binary-tree one then */
		const int hc = (l->mf == LZMA_MF_HC3 || l->mf == LZMA_MF_HC4) && !(l->mode == LZMA_MODE_NORMAL && l->pb > 2);
		if (hc) {
			opt->gpu_mf = (uint32_t)l->mf;
			opt->gpu_nice_len = l->nice_len < (opt->gpu_mf & 15) ? (opt->gpu_mf & 15) : l->nice_len;
			uint32_t d = l->depth ? l->depth : 4 + opt->gpu_nice_len / 4;
			opt->gpu_depth = d > 56 ? 56 : d;
			opt->gpu_parser = l->mode == LZMA_MODE_NORMAL ? 1 : 0;
		} else if (l->mf == LZMA_MF_HC3 || l->mf == LZMA_MF_HC4
				|| l->mf == LZMA_MF_BT2 || l->mf == LZMA_MF_BT3 || l->mf == LZMA_MF_BT4) {
			/* binary-tree finders -> suffix-neighbourhood finder + windowed optimal parser (the device has
			 * no fast parser for it: LZMA_MODE_FAST with a BT finder gets the optimal parser too) */
			opt->gpu_mf = XZAMD_MF_HC4;
			opt->gpu_nice_len = l->nice_len < 4 ? 4 : l->nice_len;
			opt->gpu_depth = 1;
			xzamd_sn_defaults(opt);
		} else {
			return LZMA_OPTIONS_ERROR;
		}
		opt->span_size = XZAMD_SPAN_DEFAULT;
		opt->bcj = pre[0]; opt->bcj2 = pre[1]; opt->bcj3 = pre[2];
	} else if (xzamd_lzma_preset(opt, o->preset)) {
		return LZMA_OPTIONS_ERROR;
	}
	if (xzamd_options_check(opt) != NULL)
		return LZMA_OPTIONS_ERROR;      /* refused at init: a preload client then stays on the CPU library */
	{
		const char *au = getenv("XZAMD_SPAN_AUTO");
		if (au && *au == '1')
			opt->span_size = XZAMD_SPAN_AUTO;
	}
	const char *env = getenv("XZAMD_SPAN_KIB");
	if (env && atoi(env) > 0)
		opt->span_size = (uint32_t)atoi(env) << 10;
	if (xzamd_options_check(opt) != NULL)
		return LZMA_OPTIONS_ERROR;      /* the span overrides can leave the two-phase mode (pb > 2 needs it) */
	*block_size = o->block_size ? o->block_size : xzamd_mt_block_size(opt);
	if (*block_size >= (1ull << 31))
		return LZMA_OPTIONS_ERROR;
	if ((unsigned)o->check > 15)
		return LZMA_PROG_ERROR;
	if (o->check != LZMA_CHECK_NONE && o->check != LZMA_CHECK_CRC32 && o->check != LZMA_CHECK_CRC64
			&& o->check != LZMA_CHECK_SHA256)
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
	/* lzma_next_strm_init (common.h:401-410): re-initialising a stream replaces its coder, and an init that
	 * fails ends the stream (lzma_end): a coder of ours must not survive in strm->internal -- an interposer
	 * would hand it to the real liblzma next (preload.c), which would read it as its own lzma_internal */
	if (strm->internal != NULL && strm->internal->magic == XZAMD_MAGIC) {
		internal_free(strm->internal);
		strm->internal = NULL;
	}
	if (r != LZMA_OK)
		return r;
	strm->internal = NULL;
	{
		const char *dbg = getenv("XZAMD_DEBUG_SEGV");
		if (dbg && *dbg == '1') signal(SIGSEGV, segv_handler);
	}
	int ndev_all = 0, cur = 0;
	if (xzk_device_count(&ndev_all) || ndev_all <= 0 || xzk_get_device(&cur))
		return LZMA_PROG_ERROR;      /* no GPU: the product path never falls back to the CPU */
	lzma_internal *in = (lzma_internal *)a_alloc(strm->allocator, sizeof(*in));
	if (!in)
		return LZMA_MEM_ERROR;
	memset(in, 0, sizeof(*in));
	in->magic = XZAMD_MAGIC;
	in->allocator = strm->allocator;
	in->opt = opt;
	in->block_size = block_size;
	in->check = check;
	in->timeout_ms = options->timeout;
	in->sequence = ISEQ_RUN;
	in->sseq = SEQ_HEADER;
	in->fill = -1;
	in->seen_in = 0;
	/* one worker per visible GPU, the current device first; never more workers than lzma_mt.threads
	 * (XZAMD_DEVICES=n caps it further) */
	int ndev = ndev_all;
	if ((uint32_t)ndev > options->threads) ndev = (int)options->threads;
	if (ndev > MAX_DEVS) ndev = MAX_DEVS;
	{
		const char *e = getenv("XZAMD_DEVICES");
		if (e && atoi(e) > 0 && atoi(e) < ndev) ndev = atoi(e);
	}
	/* XZAMD_TEST_WORKERS=n (test knob): n workers, dealt round-robin to the visible GPUs -- two contexts on one
	 * GPU exercise the ordered draining of concurrently finished jobs on a single-GPU box */
	{
		const char *e = getenv("XZAMD_TEST_WORKERS");
		if (e && atoi(e) > 0 && atoi(e) <= MAX_DEVS) ndev = atoi(e);
	}
	for (int i = 0; i < ndev; ++i) {
		devslot *d = &in->dev[i];
		d->device = (cur + i) % ndev_all;
		d->owner = in;
		if (!unpark_dev(d, d->device)) {
			int rc = xzamd_ctx_create(&d->ctx, d->device);
			if (rc) {
				in->ndev = i;
				xzk_set_device(cur);
				internal_free(in);
				return rc == XZAMD_MEM_ERROR ? LZMA_MEM_ERROR : LZMA_PROG_ERROR;
			}
		}
		in->ndev = i + 1;
	}
	xzk_set_device(cur);
	in->njobs = 2 * in->ndev + 1;    /* one job is filled while every GPU works on one and still has the back end of the one before in flight */
	/* One job = one device batch of whole Blocks: 1.25 GiB, i.e. at least two full rounds of parse pieces at preset 6 on an
	 * MI355X (4096 resident wavefronts x 128 KiB), whatever the number of workers -- a job that cannot fill its GPU
	 * wastes it, and every job costs ~100 ms of its own (4 GiB at 24 MiB Blocks: 4 jobs of 43 Blocks, 650 MB/s host to
	 * host; the 5 jobs of 34 that a 1 GiB job size deals: 602 MB/s; 3 jobs of 57: 653 MB/s).  While the end of the input is unknown (LZMA_RUN) full jobs are dealt in order; once the caller has shown
	 * the end (FINISH / FULL_FLUSH / FULL_BARRIER) the rest is dealt evenly over the workers (stream_code, below), as
	 * stream_encoder_mt.c:599-665 deals Blocks to threads. */
	uint64_t batch = JOB_BYTES;
	const char *env = getenv("XZAMD_BATCH_MIB");
	if (env && atoll(env) > 0)
		batch = (uint64_t)atoll(env) << 20;
	uint64_t maxb = batch / block_size;
	if (maxb == 0) maxb = 1;
	if (maxb * block_size >= (1ull << 31)) maxb = ((1ull << 31) - 1) / block_size;
	in->stage_max = maxb * block_size;
	{
		const char *e = getenv("XZAMD_TEST_JOB_MIN_MIB");
		in->test_job_min = e && atoi(e) > 0 ? (uint64_t)atoi(e) << 20 : 0;
	}
	if (pthread_mutex_init(&in->mu, NULL)) { internal_free(in); return LZMA_MEM_ERROR; }
	pthread_cond_init(&in->cv_work, NULL);
	{
		pthread_condattr_t ca;
		pthread_condattr_init(&ca);
		pthread_condattr_setclock(&ca, CLOCK_MONOTONIC);
		pthread_cond_init(&in->cv_done, &ca);
		pthread_condattr_destroy(&ca);
	}
	in->mu_ok = 1;
	for (int i = 0; i < in->ndev; ++i) {
		/* mythread_create (src/common/mythread.h:181-192): workers run with every signal blocked, so the
		 * client's handlers (xz: SIGALRM progress, SIGINT cleanup) only ever run on its own threads */
		sigset_t all, old;
		sigfillset(&all);
		pthread_sigmask(SIG_SETMASK, &all, &old);
		const int pe = pthread_create(&in->dev[i].thr, NULL, worker_main, &in->dev[i]);
		pthread_sigmask(SIG_SETMASK, &old, NULL);
		if (pe) {
			internal_free(in);
			return LZMA_MEM_ERROR;
		}
		in->dev[i].started = 1;
	}
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
	uint64_t maxb = JOB_BYTES / bs;
	if (maxb == 0) maxb = 1;
	const uint64_t stage = maxb * bs;
	/* HOST memory only: `2 x GPUs + 1` pinned job slots of (staging + output bound) each.  Clients compare this figure with a
	 * limit on host RAM (src/xz/coder.c:559-606 lowers the thread count until it fits -- and here threads = GPUs): the device
	 * work buffers (the batch planner caps them at 80 % of what the device has free) are not host memory, and counting them
	 * sent `xz -T0` of 5.4+ down to one worker = one GPU (round-5 advisor).  GPUs = min(options->threads, visible devices), 1
	 * when none can be counted. */
	int ndev = 1;
	if (xzk_device_count(&ndev) || ndev <= 0) ndev = 1;
	if ((uint32_t)ndev > options->threads) ndev = (int)options->threads;
	if (ndev > MAX_DEVS) ndev = MAX_DEVS;
	const uint64_t bound = stage + stage / 64 + 65536;               /* ~ maxb x lzma_block_buffer_bound */
	const double total = (double)(2 * ndev + 1) * (double)(stage + bound);
	return total >= 1.8e19 ? UINT64_MAX - 1 : (uint64_t)total;
}

uint64_t lzma_mt_block_size(const lzma_filter *filters)
{
	/* filter_encoder.c:270-293: the largest block size any filter of the chain asks for; only LZMA2 does
	 * (lzma2_encoder.c:403-413: max(3 x dict_size, 1 MiB), UINT64_MAX for an invalid dictionary size).
	 * UINT64_MAX = error: NULL array, a filter id without an encoder, no filter with a block size
	 * (clients test exactly that value, src/xz/coder.c:482). */
	if (filters == NULL)
		return UINT64_MAX;
	uint64_t max = 0;
	for (size_t i = 0; filters[i].id != LZMA_VLI_UNKNOWN; ++i) {
		if (filters[i].id == LZMA_FILTER_LZMA2) {
			const lzma_options_lzma *l = (const lzma_options_lzma *)filters[i].options;
			if (l == NULL || l->dict_size < 4096 || l->dict_size > (1u << 30) + (1u << 29))
				return UINT64_MAX;
			uint64_t b = (uint64_t)l->dict_size * 3;
			if (b < (1u << 20)) b = 1u << 20;
			if (b > max) max = b;
		} else if (!(filters[i].id >= LZMA_FILTER_X86 && filters[i].id <= LZMA_FILTER_RISCV)
				&& filters[i].id != LZMA_FILTER_DELTA && filters[i].id != LZMA_FILTER_LZMA1) {
			return UINT64_MAX;      /* encoder_find() == NULL */
		}
	}
	return max == 0 ? UINT64_MAX : max;
}

uint32_t lzma_cputhreads(void)
{
	/* hardware_cputhreads.c: "threads" that make sense for lzma_mt.threads = the number of workers this
	 * library can run = visible GPUs (0 when that cannot be determined, like the reference) */
	int n = 0;
	if (xzk_device_count(&n) || n < 0)
		return 0;
	return (uint32_t)n;
}

/* true when every job has been drained */
static int all_drained(const lzma_internal *in)
{
	return in->drain_seq == in->next_seq && (in->fill < 0 || in->jobs[in->fill].stage_len == 0);
}

static void queue_fill_job(lzma_internal *in)
{
	job *j = &in->jobs[in->fill];
	j->opt = in->opt;
	pthread_mutex_lock(&in->mu);
	j->seq = in->next_seq++;
	j->state = J_QUEUED;
	pthread_cond_signal(&in->cv_work);
	pthread_mutex_unlock(&in->mu);
	in->fill = -1;
}

/* Wait until the job to drain next is done or a job slot is free (what = 0), or until job `drain_seq`
 * is done (what = 1).  Returns 0 when the condition holds, 1 on timeout (wait_for_work :667-713). */
static int wait_workers(lzma_internal *in, int what, const struct timespec *deadline)
{
	int rc = 0;
	pthread_mutex_lock(&in->mu);
	for (;;) {
		int ok = 0;
		for (int i = 0; i < in->njobs; ++i) {
			const job *j = &in->jobs[i];
			if (j->state == J_DONE && j->seq == in->drain_seq) ok = 1;
			if (what == 0 && j->state == J_FREE) ok = 1;
		}
		if (ok) break;
		if (deadline) {
			if (pthread_cond_timedwait(&in->cv_done, &in->mu, deadline) == ETIMEDOUT) { rc = 1; break; }
		} else {
			pthread_cond_wait(&in->cv_done, &in->mu);
		}
	}
	pthread_mutex_unlock(&in->mu);
	return rc;
}

/* stream_encode_mt(): stream_encoder_mt.c:717-883.  *timed_out is set when the call returns LZMA_OK only
 * because lzma_mt.timeout expired (common.c:332-335: that return does not count towards LZMA_BUF_ERROR). */
static lzma_ret stream_code(lzma_internal *in, const uint8_t *inb, size_t *in_pos, size_t in_size,
		uint8_t *out, size_t *out_pos, size_t out_size, lzma_action action, int *timed_out)
{
	struct timespec dl, *deadline = NULL;
	if (in->timeout_ms) {
		clock_gettime(CLOCK_MONOTONIC, &dl);
		dl.tv_sec += in->timeout_ms / 1000;
		dl.tv_nsec += (long)(in->timeout_ms % 1000) * 1000000L;
		if (dl.tv_nsec >= 1000000000L) { dl.tv_sec += 1; dl.tv_nsec -= 1000000000L; }
		deadline = &dl;
	}
	for (;;) {
		switch (in->sseq) {
		case SEQ_HEADER: {
			if (in->tail_cap < 64) {
				in->tailbuf = (uint8_t *)a_alloc(in->allocator, 64);
				if (!in->tailbuf) return LZMA_MEM_ERROR;
				in->tail_cap = 64;
			}
			in->tail_len = xzamd_frame_header(in->tailbuf, in->check);
			in->tail_pos = 0;
			in->sseq = SEQ_BLOCKS;
			break;
		}
		case SEQ_BLOCKS: {
			/* 0. Stream Header bytes */
			if (in->tail_pos < in->tail_len) {
				uint64_t k = in->tail_len - in->tail_pos;
				if (k > out_size - *out_pos) k = out_size - *out_pos;
				memcpy(out + *out_pos, in->tailbuf + in->tail_pos, k);
				in->tail_pos += k; *out_pos += k; in->progress_out += k;
				if (in->tail_pos < in->tail_len)
					return LZMA_OK;
			}
			/* 1. drain finished jobs in order (lzma_outq_read, outqueue.c:182-260) */
			for (;;) {
				job *dj = NULL;
				/* nothing finished (the common case of xz's 8 KiB lzma_code(LZMA_RUN) calls): no lock, no scan */
				if (__atomic_load_n(&in->ndone, __ATOMIC_ACQUIRE) == 0)
					break;
				pthread_mutex_lock(&in->mu);
				for (int i = 0; i < in->njobs; ++i)
					if (in->jobs[i].state == J_DONE && in->jobs[i].seq == in->drain_seq)
						dj = &in->jobs[i];
				pthread_mutex_unlock(&in->mu);
				if (!dj)
					break;
				if (dj->err != LZMA_OK)
					return dj->err;
				uint64_t k = dj->out_len - dj->out_pos;
				if (k > out_size - *out_pos) k = out_size - *out_pos;
				if (k) {
					memcpy(out + *out_pos, dj->out + dj->out_pos, k);
					dj->out_pos += k; *out_pos += k; in->progress_out += k;
				}
				if (dj->out_pos < dj->out_len)
					return LZMA_OK;             /* output full */
				/* Index records of this batch (lzma_index_append, :722) */
				if (in->nrec + dj->nblocks > in->rec_cap) {
					uint64_t nc = in->rec_cap ? in->rec_cap * 2 : 256;
					while (nc < in->nrec + dj->nblocks) nc *= 2;
					uint64_t *nr = (uint64_t *)a_alloc(in->allocator, nc * 2 * sizeof(uint64_t));
					if (!nr) return LZMA_MEM_ERROR;
					if (in->rec) { memcpy(nr, in->rec, in->nrec * 2 * sizeof(uint64_t)); a_free(in->allocator, in->rec); }
					in->rec = nr;
					in->rec_cap = nc;
				}
				for (uint64_t i = 0; i < dj->nblocks; ++i) {
					in->rec[2 * (in->nrec + i)] = dj->binfo[i].unpadded_size;
					in->rec[2 * (in->nrec + i) + 1] = dj->binfo[i].uncompressed_size;
				}
				in->nrec += dj->nblocks;
				pthread_mutex_lock(&in->mu);
				dj->state = J_FREE;
				dj->stage_len = 0;
				++in->drain_seq;
				__atomic_sub_fetch(&in->ndone, 1, __ATOMIC_RELEASE);
				pthread_mutex_unlock(&in->mu);
			}
			/* 2. take input (stream_encode_in: :599-664) */
			while (*in_pos < in_size) {
				if (in->fill < 0) {
					pthread_mutex_lock(&in->mu);
					for (int i = 0; i < in->njobs && in->fill < 0; ++i)
						if (in->jobs[i].state == J_FREE) { in->fill = i; in->jobs[i].state = J_FILLING; in->jobs[i].stage_len = 0; }
					pthread_mutex_unlock(&in->mu);
					if (in->fill < 0)
						break;                   /* every job is with a worker or waits to be drained */
				}
				job *j = &in->jobs[in->fill];
				/* Size of this job.  With LZMA_RUN the end of the input is unknown: full jobs.  With FINISH / FULL_FLUSH /
				 * FULL_BARRIER the caller has shown everything that ends this unit (the same avail_in is repeated until it
				 * is done, common.c:253-281): the rest is dealt in EVEN jobs, so that the last one is not a sliver that
				 * cannot fill the GPU (4 GiB at 24 MiB Blocks: 35+34+34+34+34 Blocks instead of 42+42+42+42+3). */
				uint64_t job_max = in->stage_max;
				if (action == LZMA_RUN && in->ndev > 1) {
					/* Several GPUs and the end of the input unknown (how `xz` feeds: 8 KiB at a time, LZMA_RUN until EOF): full
					 * 1.25 GiB jobs dealt in order would keep 4 of 8 workers busy on a 4 GiB file.  A job is therefore no larger
					 * than the input seen so far divided by the GPUs, at least 256 MiB, in whole Blocks -- the jobs grow with the
					 * input (the reference hands a Block to every thread as soon as it is full, stream_encoder_mt.c:599-665). */
					uint64_t cap = in->seen_in / (uint64_t)in->ndev;
					uint64_t floor_b = 256ull << 20;
					if (in->test_job_min) floor_b = in->test_job_min;       /* (XZAMD_TEST_JOB_MIN_MIB, read once at init) */
					if (cap < floor_b) cap = floor_b;
					cap = (cap / in->block_size) * in->block_size;
					if (cap < in->block_size) cap = in->block_size;
					if (cap < job_max) job_max = cap;
				}
				if (action != LZMA_RUN) {
					const uint64_t rem = (in_size - *in_pos) + j->stage_len;
					const uint64_t nbr = (rem + in->block_size - 1) / in->block_size;
					const uint64_t maxb = in->stage_max / in->block_size;
					if (maxb > 0) {
						uint64_t nj = (nbr + maxb - 1) / maxb;
						if (in->ndev > 1) {
							/* several GPUs: a job for every worker, as long as a job keeps at least 128 MiB (what is left of a
							 * 4 GiB input on 8 GPUs: 8 jobs of 22 Blocks instead of 4 of 43) */
							uint64_t minb = (128ull << 20) / in->block_size;
							if (minb == 0) minb = 1;
							uint64_t want = nbr / minb;
							if (want > (uint64_t)in->ndev) want = (uint64_t)in->ndev;
							if (want > nj) nj = want;
						}
						if (nj > 1)
							job_max = ((nbr + nj - 1) / nj) * in->block_size;
					}
				}
				if (j->stage_len >= job_max) {          /* (a job filled under LZMA_RUN beyond what FINISH would deal) */
					queue_fill_job(in);
					continue;
				}
				if (j->stage_len == j->stage_cap) {
					/* second growth step goes straight to what this job may hold (job_max: the full batch with one GPU; with
					 * several, while the jobs still grow with the input, no more than that -- round-5 advisor: 2 x GPUs + 1 slots
					 * of 1.25 GiB pinned each were 21 GiB at 8 GPUs for jobs of 256 ... 512 MiB); each step is a pinned allocation
					 * + copy, and a slot whose job_max has risen since grows again */
					uint64_t nc = j->stage_cap ? job_max : in->block_size;
					if (nc <= j->stage_cap) nc = in->stage_max;
					if (nc < (1u << 20)) nc = 1u << 20;
					if (nc > in->stage_max) nc = in->stage_max;
					lzma_ret r = grow_pinned(&j->stage, &j->stage_cap, j->stage_len, nc);
					if (r != LZMA_OK) return r;
				}
				uint64_t room = (j->stage_cap < job_max ? j->stage_cap : job_max) - j->stage_len;
				uint64_t take = in_size - *in_pos;
				if (take > room) take = room;
				memcpy(j->stage + j->stage_len, inb + *in_pos, take);
				j->stage_len += take;
				*in_pos += take;
				in->seen_in += take;
				if (j->stage_len == job_max)
					queue_fill_job(in);
			}
			const int input_done = *in_pos == in_size;
			if (input_done && action != LZMA_RUN && in->fill >= 0 && in->jobs[in->fill].stage_len)
				queue_fill_job(in);
			if (!input_done) {
				/* no free job: wait for a worker (or for room to drain) */
				int drainable = 0;
				pthread_mutex_lock(&in->mu);
				for (int i = 0; i < in->njobs; ++i)
					if (in->jobs[i].state == J_DONE && in->jobs[i].seq == in->drain_seq) drainable = 1;
				pthread_mutex_unlock(&in->mu);
				if (drainable) {
					if (*out_pos == out_size) return LZMA_OK;
					break;
				}
				if (wait_workers(in, 0, deadline)) { *timed_out = 1; return LZMA_OK; }
				break;
			}
			if (action == LZMA_RUN)
				return LZMA_OK;             /* :796-801 */
			if (action == LZMA_FULL_BARRIER)
				return LZMA_STREAM_END;     /* :803-807: the input is handed over, no waiting */
			if (!all_drained(in)) {
				/* LZMA_FULL_FLUSH / LZMA_FINISH wait for the output queue to empty (:809-823) */
				if (*out_pos == out_size) return LZMA_OK;
				if (wait_workers(in, 1, deadline)) { *timed_out = 1; return LZMA_OK; }
				break;
			}
			if (action == LZMA_FULL_FLUSH)
				return LZMA_STREAM_END;
			/* LZMA_FINISH: Index + Stream Footer (:842-883) */
			{
				const uint64_t cap = 64 + in->nrec * 18;
				if (in->tail_cap < cap) {
					if (in->tailbuf) a_free(in->allocator, in->tailbuf);
					in->tailbuf = (uint8_t *)a_alloc(in->allocator, cap);
					if (!in->tailbuf) { in->tail_cap = 0; return LZMA_MEM_ERROR; }
					in->tail_cap = cap;
				}
				uint64_t *unp = (uint64_t *)a_alloc(in->allocator, (in->nrec + 1) * 2 * sizeof(uint64_t));
				if (!unp) return LZMA_MEM_ERROR;
				uint64_t *unc = unp + in->nrec + 1;
				for (uint64_t i = 0; i < in->nrec; ++i) { unp[i] = in->rec[2 * i]; unc[i] = in->rec[2 * i + 1]; }
				in->tail_len = xzamd_frame_index_footer(in->tailbuf, in->tail_cap, in->check, unp, unc, in->nrec);
				in->tail_pos = 0;
				a_free(in->allocator, unp);
				if (in->tail_len == 0) return LZMA_PROG_ERROR;
				in->sseq = SEQ_TAIL;
			}
			break;
		}
		case SEQ_TAIL: {
			uint64_t k = in->tail_len - in->tail_pos;
			if (k > out_size - *out_pos) k = out_size - *out_pos;
			memcpy(out + *out_pos, in->tailbuf + in->tail_pos, k);
			in->tail_pos += k; *out_pos += k; in->progress_out += k;
			if (in->tail_pos < in->tail_len)
				return LZMA_OK;
			in->sseq = SEQ_DONE;
			return LZMA_STREAM_END;
		}
		case SEQ_DONE:
			return LZMA_STREAM_END;
		}
	}
}

lzma_ret lzma_filters_update(lzma_stream *strm, const lzma_filter *filters)
{
	/* common.c / stream_encoder_mt.c:914-950: the MT encoder takes a new chain only between Blocks: at
	 * the very start or right after LZMA_FULL_FLUSH / LZMA_FULL_BARRIER, and it applies from the next
	 * Block on.  Jobs already dealt keep the options they were queued with. */
	if (strm == NULL || strm->internal == NULL || strm->internal->magic != XZAMD_MAGIC)
		return LZMA_PROG_ERROR;
	lzma_internal *in = strm->internal;
	if (in->sequence != ISEQ_RUN)
		return LZMA_PROG_ERROR;
	if (in->fill >= 0 && in->jobs[in->fill].stage_len % in->block_size != 0)
		return LZMA_PROG_ERROR;      /* in the middle of a Block */
	lzma_mt mt;
	memset(&mt, 0, sizeof(mt));
	mt.threads = 1;
	mt.filters = filters;
	mt.block_size = in->block_size;
	mt.check = (lzma_check)in->check;
	xzamd_lzma_options opt;
	uint64_t bs = 0;
	int check = 0;
	lzma_ret r = parse_options(&mt, &opt, &bs, &check);
	if (r != LZMA_OK)
		return r;
	if (xzamd_block_buffer_bound(in->block_size) == 0)
		return LZMA_OPTIONS_ERROR;
	if (in->fill >= 0 && in->jobs[in->fill].stage_len)
		queue_fill_job(in);          /* whole Blocks staged so far go out with the old chain */
	in->opt = opt;
	return LZMA_OK;
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
	int timed_out = 0;
	lzma_ret ret = stream_code(in, strm->next_in, &in_pos, strm->avail_in,
			strm->next_out, &out_pos, strm->avail_out, action, &timed_out);
	if (in_pos) { strm->next_in += in_pos; strm->avail_in -= in_pos; strm->total_in += in_pos; }
	if (out_pos) { strm->next_out += out_pos; strm->avail_out -= out_pos; strm->total_out += out_pos; }
	in->avail_in = strm->avail_in;
	switch (ret) {
	case LZMA_OK:
		if (timed_out) {
			in->allow_buf_error = 0;        /* common.c:332-335 */
		} else if (out_pos == 0 && in_pos == 0) {
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
		lzma_internal *in = strm->internal;
		/* stream_encoder_mt.c:1004-1024: what is finished plus how far every worker is inside its job -- here the share of
		 * the job's bytes that the finished stages of its device batches stand for (xzamd_ctx_progress_in_) */
		pthread_mutex_lock(&in->mu);
		uint64_t pin = in->progress_in;
		for (int i = 0; i < in->ndev; ++i) {
			const job *j = in->dev[i].running;
			if (in->dev[i].pending != NULL)
				pin += in->dev[i].pending->stage_len;     /* everything but the tail of its back end is done */
			if (j != NULL) {
				const uint64_t part = xzamd_ctx_progress_in_(in->dev[i].ctx);
				pin += part < j->stage_len ? part : j->stage_len;
			}
		}
		pthread_mutex_unlock(&in->mu);
		*progress_in = pin;
		*progress_out = in->progress_out;
	} else if (strm) {
		*progress_in = strm->total_in;
		*progress_out = strm->total_out;
	}
}

/* ------------------------------------------------------------------ */
/* One-shot buffer API (common/stream_buffer_encoder.c:43-141,          */
/* common/easy_buffer_encoder.c:16-27): same engine, no caller-visible  */
/* streaming state.  The reference writes ONE Block whatever the input  */
/* size (:91-101, lzma_block_buffer_encode); so does this up to         */
/* ONE_SHOT_SINGLE_MAX bytes -- the whole input is one device batch of  */
/* one Block, cut into parse pieces / spans like any other Block.       */
/* Above that (or when the device cannot hold such a Block) the MT      */
/* layout is written, a Block per block_size bytes: it decodes the same.*/
/* ------------------------------------------------------------------ */
#define ONE_SHOT_SINGLE_MAX (1ull << 30)
size_t lzma_stream_buffer_bound(size_t uncompressed_size)
{
	/* stream_buffer_encoder.c:24-40 semantics: 0 = too big */
	const uint64_t bs = 1ull << 20;           /* smallest default Block size (3 x 256 KiB dict -> 1 MiB floor) */
	const uint64_t b = xzamd_stream_buffer_bound(uncompressed_size, bs);
	return b > (uint64_t)SIZE_MAX ? 0 : (size_t)b;
}

static lzma_ret buffer_encode_layout(const lzma_mt *mt, const lzma_allocator *allocator,
		const uint8_t *in, size_t in_size, uint8_t *out, size_t *out_pos_ptr, size_t out_size);

static lzma_ret buffer_encode(const lzma_mt *mt, const lzma_allocator *allocator,
		const uint8_t *in, size_t in_size, uint8_t *out, size_t *out_pos_ptr, size_t out_size)
{
	if (out == NULL || out_pos_ptr == NULL || *out_pos_ptr > out_size || (in == NULL && in_size != 0))
		return LZMA_PROG_ERROR;
	if (mt->block_size == 0 && (uint64_t)in_size <= ONE_SHOT_SINGLE_MAX) {
		/* one Block, as the reference writes it: block_size = the input size where that is more than the default */
		xzamd_lzma_options opt;
		uint64_t def_bs = 0;
		int check = 0;
		lzma_ret r = parse_options(mt, &opt, &def_bs, &check);
		if (r != LZMA_OK)
			return r;
		if ((uint64_t)in_size > def_bs) {
			lzma_mt one = *mt;
			one.block_size = (uint64_t)in_size;
			r = buffer_encode_layout(&one, allocator, in, in_size, out, out_pos_ptr, out_size);
			/* a Block the device cannot hold (memory): fall through to the MT layout */
			if (r != LZMA_MEM_ERROR && r != LZMA_OPTIONS_ERROR)
				return r;
		}
	}
	return buffer_encode_layout(mt, allocator, in, in_size, out, out_pos_ptr, out_size);
}

static lzma_ret buffer_encode_layout(const lzma_mt *mt, const lzma_allocator *allocator,
		const uint8_t *in, size_t in_size, uint8_t *out, size_t *out_pos_ptr, size_t out_size)
{
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
