/*
 * ref_shim.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Thin flat-C wrappers around the REAL reference liblzma (built by
 * oracle/Makefile from the sources under /root/reference into
 * oracle/_ref/liblzma_ref.so).  They let the Python tests and bench.py's
 * cpu_baseline leg drive the reference through plain (pointer,size) calls
 * instead of re-declaring lzma_stream in ctypes.
 *
 * Compiled against the reference's own public header (api/lzma.h) where it
 * lies; nothing from the reference is copied into this repository.
 */
#include <lzma.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* lzma_stream_encoder_mt + lzma_code(FINISH) over one in-memory buffer.
 * Mirrors doc/examples/04_compress_easy_mt.c:29-82 but with one big buffer.
 * preset may carry LZMA_PRESET_EXTREME.  block_size 0 = library default.
 * Returns lzma_ret (1 = LZMA_STREAM_END on success). */
int ref_encode_mt(const uint8_t *in, size_t in_size, uint32_t preset,
		uint32_t threads, uint64_t block_size, int check,
		uint8_t *out, size_t out_cap, size_t *out_size)
{
	lzma_stream strm = LZMA_STREAM_INIT;
	lzma_mt mt;
	memset(&mt, 0, sizeof(mt));
	mt.threads = threads;
	mt.block_size = block_size;
	mt.preset = preset;
	mt.check = (lzma_check)check;
	lzma_ret r = lzma_stream_encoder_mt(&strm, &mt);
	if (r != LZMA_OK)
		return (int)r;
	strm.next_in = in;
	strm.avail_in = in_size;
	strm.next_out = out;
	strm.avail_out = out_cap;
	do {
		r = lzma_code(&strm, LZMA_FINISH);
	} while (r == LZMA_OK && strm.avail_out > 0);
	*out_size = out_cap - strm.avail_out;
	lzma_end(&strm);
	return (int)r;
}

/* The reference's one-shot API (common/easy_buffer_encoder.c:16-27 -> stream_buffer_encoder.c:43-141): ONE Block
 * whatever the input size.  Returns lzma_ret (0 = LZMA_OK). */
int ref_easy_buffer_encode(const uint8_t *in, size_t in_size, uint32_t preset, int check,
		uint8_t *out, size_t out_cap, size_t *out_size)
{
	size_t pos = 0;
	lzma_ret r = lzma_easy_buffer_encode(preset, (lzma_check)check, NULL, in, in_size, out, &pos, out_cap);
	*out_size = pos;
	return (int)r;
}

/* Same, but with an explicit LZMA2 option set (filters[] = {LZMA2}). */
int ref_encode_mt_opts(const uint8_t *in, size_t in_size,
		uint32_t dict_size, uint32_t lc, uint32_t lp, uint32_t pb,
		int mode, uint32_t nice_len, int mf, uint32_t depth,
		uint32_t threads, uint64_t block_size, int check,
		uint8_t *out, size_t out_cap, size_t *out_size)
{
	lzma_options_lzma opt;
	memset(&opt, 0, sizeof(opt));
	opt.dict_size = dict_size;
	opt.lc = lc; opt.lp = lp; opt.pb = pb;
	opt.mode = (lzma_mode)mode;
	opt.nice_len = nice_len;
	opt.mf = (lzma_match_finder)mf;
	opt.depth = depth;
	lzma_filter f[2] = { { LZMA_FILTER_LZMA2, &opt }, { LZMA_VLI_UNKNOWN, NULL } };
	lzma_stream strm = LZMA_STREAM_INIT;
	lzma_mt mt;
	memset(&mt, 0, sizeof(mt));
	mt.threads = threads;
	mt.block_size = block_size;
	mt.filters = f;
	mt.check = (lzma_check)check;
	lzma_ret r = lzma_stream_encoder_mt(&strm, &mt);
	if (r != LZMA_OK)
		return (int)r;
	strm.next_in = in;
	strm.avail_in = in_size;
	strm.next_out = out;
	strm.avail_out = out_cap;
	do {
		r = lzma_code(&strm, LZMA_FINISH);
	} while (r == LZMA_OK && strm.avail_out > 0);
	*out_size = out_cap - strm.avail_out;
	lzma_end(&strm);
	return (int)r;
}

/* MT encode with the chain {x86 BCJ, LZMA2(preset)}. */
int ref_encode_mt_x86(const uint8_t *in, size_t in_size, uint32_t preset,
		uint32_t threads, uint64_t block_size, int check,
		uint8_t *out, size_t out_cap, size_t *out_size)
{
	lzma_options_lzma opt;
	if (lzma_lzma_preset(&opt, preset))
		return (int)LZMA_OPTIONS_ERROR;
	lzma_filter f[3] = { { LZMA_FILTER_X86, NULL }, { LZMA_FILTER_LZMA2, &opt }, { LZMA_VLI_UNKNOWN, NULL } };
	lzma_stream strm = LZMA_STREAM_INIT;
	lzma_mt mt;
	memset(&mt, 0, sizeof(mt));
	mt.threads = threads;
	mt.block_size = block_size;
	mt.filters = f;
	mt.check = (lzma_check)check;
	lzma_ret r = lzma_stream_encoder_mt(&strm, &mt);
	if (r != LZMA_OK)
		return (int)r;
	strm.next_in = in;
	strm.avail_in = in_size;
	strm.next_out = out;
	strm.avail_out = out_cap;
	do {
		r = lzma_code(&strm, LZMA_FINISH);
	} while (r == LZMA_OK && strm.avail_out > 0);
	*out_size = out_cap - strm.avail_out;
	lzma_end(&strm);
	return (int)r;
}

/* MT encode with the chain {filter, LZMA2(preset)}: filter_id = LZMA_FILTER_X86 / _ARM64 / _DELTA (dist). */
int ref_encode_mt_chain(const uint8_t *in, size_t in_size, uint32_t preset, uint64_t filter_id, uint32_t delta_dist,
		uint32_t threads, uint64_t block_size, int check,
		uint8_t *out, size_t out_cap, size_t *out_size)
{
	lzma_options_lzma opt;
	if (lzma_lzma_preset(&opt, preset))
		return (int)LZMA_OPTIONS_ERROR;
	lzma_options_delta dl;
	memset(&dl, 0, sizeof(dl));
	dl.type = LZMA_DELTA_TYPE_BYTE;
	dl.dist = delta_dist;
	lzma_filter f[3] = { { filter_id, filter_id == LZMA_FILTER_DELTA ? (void *)&dl : NULL },
			{ LZMA_FILTER_LZMA2, &opt }, { LZMA_VLI_UNKNOWN, NULL } };
	lzma_stream strm = LZMA_STREAM_INIT;
	lzma_mt mt;
	memset(&mt, 0, sizeof(mt));
	mt.threads = threads;
	mt.block_size = block_size;
	mt.filters = f;
	mt.check = (lzma_check)check;
	lzma_ret r = lzma_stream_encoder_mt(&strm, &mt);
	if (r != LZMA_OK)
		return (int)r;
	strm.next_in = in;
	strm.avail_in = in_size;
	strm.next_out = out;
	strm.avail_out = out_cap;
	do {
		r = lzma_code(&strm, LZMA_FINISH);
	} while (r == LZMA_OK && strm.avail_out > 0);
	*out_size = out_cap - strm.avail_out;
	lzma_end(&strm);
	return (int)r;
}

/* MT encode with a chain of `nf` (<= 3) filters in front of LZMA2(preset): ids[i] = LZMA_FILTER_* of the i-th filter,
 * dists[i] = its distance when it is LZMA_FILTER_DELTA (common/filter_common.c:250-334 validates the chain). */
int ref_encode_mt_chain_n(const uint8_t *in, size_t in_size, uint32_t preset, uint32_t nf, const uint64_t *ids,
		const uint32_t *dists, uint32_t threads, uint64_t block_size, int check,
		uint8_t *out, size_t out_cap, size_t *out_size)
{
	lzma_options_lzma opt;
	if (nf > 3 || lzma_lzma_preset(&opt, preset))
		return (int)LZMA_OPTIONS_ERROR;
	lzma_options_delta dl[3];
	lzma_filter f[5];
	memset(dl, 0, sizeof(dl));
	for (uint32_t i = 0; i < nf; ++i) {
		dl[i].type = LZMA_DELTA_TYPE_BYTE;
		dl[i].dist = dists[i];
		f[i].id = ids[i];
		f[i].options = ids[i] == LZMA_FILTER_DELTA ? (void *)&dl[i] : NULL;
	}
	f[nf].id = LZMA_FILTER_LZMA2; f[nf].options = &opt;
	f[nf + 1].id = LZMA_VLI_UNKNOWN; f[nf + 1].options = NULL;
	lzma_stream strm = LZMA_STREAM_INIT;
	lzma_mt mt;
	memset(&mt, 0, sizeof(mt));
	mt.threads = threads;
	mt.block_size = block_size;
	mt.filters = f;
	mt.check = (lzma_check)check;
	lzma_ret r = lzma_stream_encoder_mt(&strm, &mt);
	if (r != LZMA_OK)
		return (int)r;
	strm.next_in = in;
	strm.avail_in = in_size;
	strm.next_out = out;
	strm.avail_out = out_cap;
	do {
		r = lzma_code(&strm, LZMA_FINISH);
	} while (r == LZMA_OK && strm.avail_out > 0);
	*out_size = out_cap - strm.avail_out;
	lzma_end(&strm);
	return (int)r;
}

/* Steady-state throughput of the reference MT encoder: feed `in` through lzma_code(LZMA_RUN) in 4 MiB
 * slices (output discarded) until `seconds` of wall time have passed or the input is used up, then report
 * how many input bytes the workers have actually processed (lzma_get_progress) and the wall time.  With
 * threads = 0 the library's own count (lzma_cputhreads) is used, as `xz -T0` does.  The bench uses it so
 * that the whole input is on offer (every worker has a Block) without waiting minutes for the last one. */
#include <time.h>
static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int ref_encode_mt_timed(const uint8_t *in, size_t in_size, uint32_t preset, int x86,
		uint32_t threads, uint64_t block_size, double seconds,
		uint64_t *processed_in, double *elapsed, uint32_t *threads_used)
{
	lzma_options_lzma opt;
	if (lzma_lzma_preset(&opt, preset))
		return (int)LZMA_OPTIONS_ERROR;
	lzma_filter f[3];
	int nf = 0;
	if (x86) { f[nf].id = LZMA_FILTER_X86; f[nf].options = NULL; ++nf; }
	f[nf].id = LZMA_FILTER_LZMA2; f[nf].options = &opt; ++nf;
	f[nf].id = LZMA_VLI_UNKNOWN; f[nf].options = NULL;
	if (threads == 0)
		threads = lzma_cputhreads();
	if (threads == 0)
		threads = 1;
	lzma_stream strm = LZMA_STREAM_INIT;
	lzma_mt mt;
	memset(&mt, 0, sizeof(mt));
	mt.threads = threads;
	mt.block_size = block_size;
	mt.filters = f;
	mt.check = LZMA_CHECK_CRC64;
	mt.timeout = 100;
	lzma_ret r = lzma_stream_encoder_mt(&strm, &mt);
	if (r != LZMA_OK)
		return (int)r;
	const size_t ocap = 8u << 20;
	uint8_t *obuf = (uint8_t *)malloc(ocap);
	if (!obuf) { lzma_end(&strm); return (int)LZMA_MEM_ERROR; }
	const double t0 = now_s();
	size_t off = 0;
	strm.next_in = in;
	strm.avail_in = 0;
	for (;;) {
		if (strm.avail_in == 0 && off < in_size) {
			size_t k = in_size - off < (4u << 20) ? in_size - off : (4u << 20);
			strm.next_in = in + off;
			strm.avail_in = k;
			off += k;
		}
		strm.next_out = obuf;
		strm.avail_out = ocap;
		const int finishing = off >= in_size && strm.avail_in == 0;
		r = lzma_code(&strm, finishing ? LZMA_FINISH : LZMA_RUN);
		if (r != LZMA_OK)
			break;
		if (now_s() - t0 >= seconds)
			break;
	}
	uint64_t pin = 0, pout = 0;
	lzma_get_progress(&strm, &pin, &pout);
	*elapsed = now_s() - t0;
	*processed_in = pin;
	*threads_used = threads;
	lzma_end(&strm);
	free(obuf);
	return (r == LZMA_OK || r == LZMA_STREAM_END) ? 1 : (int)r;
}

/* The x86 BCJ encoder's output for one buffer (= one Block: fresh filter state, start offset 0):
 * raw-encode with {x86, LZMA2}, raw-decode with {LZMA2} only.  out must hold in_size bytes. */
int ref_x86_filter(const uint8_t *in, size_t in_size, uint8_t *out)
{
	lzma_options_lzma opt;
	if (lzma_lzma_preset(&opt, 0))
		return -1;
	lzma_filter enc[3] = { { LZMA_FILTER_X86, NULL }, { LZMA_FILTER_LZMA2, &opt }, { LZMA_VLI_UNKNOWN, NULL } };
	lzma_filter dec[2] = { { LZMA_FILTER_LZMA2, &opt }, { LZMA_VLI_UNKNOWN, NULL } };
	size_t cap = in_size + in_size / 4 + 65536;
	uint8_t *tmp = malloc(cap);
	if (!tmp)
		return -2;
	size_t tpos = 0;
	lzma_ret r = lzma_raw_buffer_encode(enc, NULL, in, in_size, tmp, &tpos, cap);
	if (r != LZMA_OK) { free(tmp); return (int)r; }
	size_t ipos = 0, opos = 0;
	r = lzma_raw_buffer_decode(dec, NULL, tmp, &ipos, tpos, out, &opos, in_size);
	free(tmp);
	if (r != LZMA_OK || opos != in_size)
		return 100 + (int)r;
	return 0;
}

/* Raw LZMA2 encode of one buffer (what one worker's filter chain produces
 * for one Block: stream_encoder_mt.c:219-298 minus header/padding/check). */
int ref_raw_lzma2_encode(const uint8_t *in, size_t in_size,
		uint32_t dict_size, uint32_t lc, uint32_t lp, uint32_t pb,
		int mode, uint32_t nice_len, int mf, uint32_t depth,
		uint8_t *out, size_t out_cap, size_t *out_size)
{
	lzma_options_lzma opt;
	memset(&opt, 0, sizeof(opt));
	opt.dict_size = dict_size;
	opt.lc = lc; opt.lp = lp; opt.pb = pb;
	opt.mode = (lzma_mode)mode;
	opt.nice_len = nice_len;
	opt.mf = (lzma_match_finder)mf;
	opt.depth = depth;
	lzma_filter f[2] = { { LZMA_FILTER_LZMA2, &opt }, { LZMA_VLI_UNKNOWN, NULL } };
	lzma_stream strm = LZMA_STREAM_INIT;
	lzma_ret r = lzma_raw_encoder(&strm, f);
	if (r != LZMA_OK)
		return (int)r;
	/* Feed in 16 KiB pieces exactly like worker_encode()
	 * (stream_encoder_mt.c:284-296); output is independent of the piece
	 * size, this merely keeps the call pattern identical. */
	size_t pos = 0;
	strm.next_out = out;
	strm.avail_out = out_cap;
	for (;;) {
		size_t n = in_size - pos;
		lzma_action a = LZMA_FINISH;
		if (n > 16384) { n = 16384; a = LZMA_RUN; }
		strm.next_in = in + pos;
		strm.avail_in = n;
		r = lzma_code(&strm, a);
		pos += n - strm.avail_in;
		if (r != LZMA_OK || strm.avail_out == 0)
			break;
	}
	*out_size = out_cap - strm.avail_out;
	lzma_end(&strm);
	return (int)r;
}

/* lzma_stream_decoder over one buffer (the verifier: what `xz -dc` runs). */
int ref_decode(const uint8_t *in, size_t in_size,
		uint8_t *out, size_t out_cap, size_t *out_size)
{
	lzma_stream strm = LZMA_STREAM_INIT;
	lzma_ret r = lzma_stream_decoder(&strm, UINT64_MAX, 0);
	if (r != LZMA_OK)
		return (int)r;
	strm.next_in = in;
	strm.avail_in = in_size;
	strm.next_out = out;
	strm.avail_out = out_cap;
	do {
		r = lzma_code(&strm, LZMA_FINISH);
	} while (r == LZMA_OK && strm.avail_out > 0 );
	*out_size = out_cap - strm.avail_out;
	if (r == LZMA_STREAM_END && strm.avail_in != 0)
		r = LZMA_DATA_ERROR;
	lzma_end(&strm);
	return (int)r;
}

/* The reference's multi-threaded .xz decoder (stream_decoder_mt.c): whole-Stream round trips of multi-GiB bench
 * outputs in seconds instead of a minute.  threads = 0: lzma_cputhreads(). */
int ref_decode_mt(const uint8_t *in, size_t in_size, uint32_t threads,
		uint8_t *out, size_t out_cap, size_t *out_size)
{
	lzma_mt mt;
	memset(&mt, 0, sizeof(mt));
	mt.threads = threads ? threads : lzma_cputhreads();
	if (mt.threads == 0) mt.threads = 1;
	mt.memlimit_threading = UINT64_MAX;
	mt.memlimit_stop = UINT64_MAX;
	lzma_stream strm = LZMA_STREAM_INIT;
	lzma_ret r = lzma_stream_decoder_mt(&strm, &mt);
	if (r != LZMA_OK)
		return (int)r;
	strm.next_in = in;
	strm.avail_in = in_size;
	strm.next_out = out;
	strm.avail_out = out_cap;
	do {
		r = lzma_code(&strm, LZMA_FINISH);
	} while (r == LZMA_OK && strm.avail_out > 0);
	*out_size = out_cap - strm.avail_out;
	if (r == LZMA_STREAM_END && strm.avail_in != 0)
		r = LZMA_DATA_ERROR;
	lzma_end(&strm);
	return (int)r;
}

/* Raw LZMA2 decode (dict_size from the caller). */
int ref_raw_lzma2_decode(const uint8_t *in, size_t in_size, uint32_t dict_size,
		uint8_t *out, size_t out_cap, size_t *out_size)
{
	lzma_options_lzma opt;
	memset(&opt, 0, sizeof(opt));
	opt.dict_size = dict_size;
	lzma_filter f[2] = { { LZMA_FILTER_LZMA2, &opt }, { LZMA_VLI_UNKNOWN, NULL } };
	lzma_stream strm = LZMA_STREAM_INIT;
	lzma_ret r = lzma_raw_decoder(&strm, f);
	if (r != LZMA_OK)
		return (int)r;
	strm.next_in = in;
	strm.avail_in = in_size;
	strm.next_out = out;
	strm.avail_out = out_cap;
	do {
		r = lzma_code(&strm, LZMA_FINISH);
	} while (r == LZMA_OK && strm.avail_out > 0);
	*out_size = out_cap - strm.avail_out;
	if (r == LZMA_STREAM_END && strm.avail_in != 0)
		r = LZMA_DATA_ERROR;
	lzma_end(&strm);
	return (int)r;
}

uint32_t ref_crc32(const uint8_t *buf, size_t n, uint32_t crc) { return lzma_crc32(buf, n, crc); }
uint64_t ref_crc64(const uint8_t *buf, size_t n, uint64_t crc) { return lzma_crc64(buf, n, crc); }

int ref_vli_encode(uint64_t v, uint8_t *out, size_t cap, size_t *n)
{
	*n = 0;
	return (int)lzma_vli_encode(v, NULL, out, n, cap);
}

uint64_t ref_block_buffer_bound(uint64_t n) { return lzma_block_buffer_bound((size_t)n); }
uint64_t ref_mt_block_size_preset(uint32_t preset)
{
	lzma_options_lzma opt;
	if (lzma_lzma_preset(&opt, preset))
		return 0;
	lzma_filter f[2] = { { LZMA_FILTER_LZMA2, &opt }, { LZMA_VLI_UNKNOWN, NULL } };
	return lzma_mt_block_size(f);
}

/* Preset table query (lzma_encoder_presets.c:17-63). out[8] =
 * dict_size, lc, lp, pb, mode, nice_len, mf, depth. */
int ref_preset(uint32_t preset, uint32_t *out)
{
	lzma_options_lzma opt;
	if (lzma_lzma_preset(&opt, preset))
		return 1;
	out[0] = opt.dict_size; out[1] = opt.lc; out[2] = opt.lp; out[3] = opt.pb;
	out[4] = (uint32_t)opt.mode; out[5] = opt.nice_len; out[6] = (uint32_t)opt.mf;
	out[7] = opt.depth;
	return 0;
}

/* Whole-Block uncompressed fallback writer (block_buffer_encoder.c:88-162,
 * public entry :345-354), as used by worker_encode (stream_encoder_mt.c:316-344). */
int ref_block_uncomp_encode(const uint8_t *in, size_t in_size, int check,
		uint8_t *out, size_t out_cap, size_t *out_size,
		uint64_t *unpadded_size)
{
	lzma_block b;
	memset(&b, 0, sizeof(b));
	b.version = 0;
	b.check = (lzma_check)check;
	size_t pos = 0;
	lzma_ret r = lzma_block_uncomp_encode(&b, in, in_size, out, &pos, out_cap);
	*out_size = pos;
	*unpadded_size = lzma_block_unpadded_size(&b);
	return (int)r;
}

const char *ref_version(void) { return lzma_version_string(); }
uint32_t ref_cputhreads(void) { return lzma_cputhreads(); }
