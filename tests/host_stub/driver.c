/* driver.c -- TEST INFRASTRUCTURE ONLY: drives the product's plain-C host layer (xzamd_stream.c + xzamd_host.c over
 * the CPU stand-in stub_xzk.c) through the liblzma entry points the way a client does, under sanitizers.
 * usage: driver OUTDIR   -> writes OUTDIR/caseN.in / caseN.xz; exit code 0 when every call behaved. */
#include "../../include/xz_amd.h"
#include "../../include/xz_amd_lzma.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "driver: %s:%d: %s\n", __FILE__, __LINE__, #c); exit(1); } } while (0)

static void save(const char *dir, const char *name, const uint8_t *p, size_t n)
{
	char path[512];
	snprintf(path, sizeof(path), "%s/%s", dir, name);
	FILE *f = fopen(path, "wb");
	CHECK(f != NULL);
	CHECK(fwrite(p, 1, n, f) == n);
	fclose(f);
}

/* feed `in` in pieces of in_step, drain into pieces of out_step; flush_at: LZMA_FULL_FLUSH once that many bytes are in */
static size_t stream_encode(const lzma_mt *mt, const uint8_t *in, size_t n, size_t in_step, size_t out_step,
		size_t flush_at, lzma_action flush_action, uint8_t *out, size_t out_cap)
{
	lzma_stream s = LZMA_STREAM_INIT;
	CHECK(lzma_stream_encoder_mt(&s, mt) == LZMA_OK);
	size_t ipos = 0, opos = 0;
	uint64_t last_pin = 0;
	int flushed = flush_at == 0;
	for (;;) {
		lzma_action act = LZMA_RUN;
		size_t limit = n;
		if (!flushed && flush_at <= n) limit = flush_at;
		if (s.avail_in == 0 && ipos < limit) {
			size_t k = limit - ipos < in_step ? limit - ipos : in_step;
			s.next_in = in + ipos;
			s.avail_in = k;
			ipos += k;
		}
		if (ipos == limit && !flushed) act = flush_action;
		else if (ipos == n) act = LZMA_FINISH;
		if (s.avail_out == 0) {
			CHECK(opos < out_cap);
			size_t k = out_cap - opos < out_step ? out_cap - opos : out_step;
			s.next_out = out + opos;
			s.avail_out = k;
			opos += k;
		}
		lzma_ret r = lzma_code(&s, act);
		uint64_t pin = 0, pout = 0;
		lzma_get_progress(&s, &pin, &pout);
		CHECK(pin <= n && pin <= s.total_in + s.avail_in);
		CHECK(pin >= last_pin);              /* workers publish progress inside their jobs; it never goes back */
		last_pin = pin;
		if (r == LZMA_STREAM_END) {
			if (act == LZMA_FINISH) break;
			CHECK(act == flush_action && s.avail_in == 0);
			flushed = 1;
			continue;
		}
		CHECK(r == LZMA_OK);
	}
	opos -= s.avail_out;
	CHECK(s.total_in == n && s.total_out == opos);
	/* after STREAM_END every further call says STREAM_END (common.c:283-284) */
	CHECK(lzma_code(&s, LZMA_FINISH) == LZMA_STREAM_END);
	lzma_end(&s);
	return opos;
}

int main(int argc, char **argv)
{
	CHECK(argc == 2);
	const char *dir = argv[1];
	const size_t n1 = (5u << 20) + 12345, n2 = (3u << 20) + 7;
	uint8_t *in = (uint8_t *)malloc(n1);
	uint8_t *out = (uint8_t *)malloc(n1 + (n1 >> 2) + (1u << 20));
	CHECK(in && out);
	xzamd_corpus_lorem(in, n1);

	/* 1. fast-parser preset, many small jobs, three workers, tiny client buffers, timeout */
	setenv("XZAMD_BATCH_MIB", "1", 1);
	setenv("XZAMD_TEST_WORKERS", "3", 1);
	lzma_mt mt;
	memset(&mt, 0, sizeof(mt));
	mt.threads = 4; mt.preset = 1; mt.check = LZMA_CHECK_CRC64; mt.block_size = 256u << 10; mt.timeout = 1;
	size_t w = stream_encode(&mt, in, n1, 8192, 4096, 0, LZMA_RUN, out, n1 + (n1 >> 2) + (1u << 20));
	save(dir, "case1.in", in, n1);
	save(dir, "case1.xz", out, w);

	/* 2. optimal-parser preset (device-side span plan path), FULL_FLUSH in the middle, CRC32, one worker */
	unsetenv("XZAMD_TEST_WORKERS");
	memset(&mt, 0, sizeof(mt));
	mt.threads = 1; mt.preset = 6; mt.check = LZMA_CHECK_CRC32; mt.block_size = 1u << 20;
	w = stream_encode(&mt, in, n2, 100000, 65536, (1u << 20) + 500, LZMA_FULL_FLUSH, out, n1 + (n1 >> 2) + (1u << 20));
	save(dir, "case2.in", in, n2);
	save(dir, "case2.xz", out, w);

	/* 3. FULL_BARRIER, two workers, no Check */
	setenv("XZAMD_TEST_WORKERS", "2", 1);
	memset(&mt, 0, sizeof(mt));
	mt.threads = 2; mt.preset = 0; mt.check = LZMA_CHECK_NONE; mt.block_size = 300000;
	w = stream_encode(&mt, in, n2, 1u << 20, 1u << 20, 700000, LZMA_FULL_BARRIER, out, n1 + (n1 >> 2) + (1u << 20));
	save(dir, "case3.in", in, n2);
	save(dir, "case3.xz", out, w);
	unsetenv("XZAMD_TEST_WORKERS");

	/* 4. option validation and stream life cycle (stream_encoder_mt.c:956-1000, common.c:203-389) */
	{
		lzma_stream s = LZMA_STREAM_INIT;
		CHECK(lzma_stream_encoder_mt(&s, NULL) == LZMA_PROG_ERROR);
		memset(&mt, 0, sizeof(mt));
		mt.threads = 1; mt.preset = 6; mt.check = LZMA_CHECK_CRC64;
		mt.flags = 1;
		CHECK(lzma_stream_encoder_mt(&s, &mt) == LZMA_OPTIONS_ERROR);
		mt.flags = 0; mt.threads = 0;
		CHECK(lzma_stream_encoder_mt(&s, &mt) == LZMA_OPTIONS_ERROR);
		mt.threads = 1; mt.preset = 77;
		CHECK(lzma_stream_encoder_mt(&s, &mt) == LZMA_OPTIONS_ERROR);
		mt.preset = 6; mt.check = (lzma_check)99;
		CHECK(lzma_stream_encoder_mt(&s, &mt) == LZMA_PROG_ERROR);
		mt.check = LZMA_CHECK_CRC64;
		CHECK(lzma_stream_encoder_mt(&s, &mt) == LZMA_OK);
		/* re-init replaces the coder; an init that fails ends the stream (lzma_next_strm_init) */
		CHECK(lzma_stream_encoder_mt(&s, &mt) == LZMA_OK);
		mt.preset = 77;
		CHECK(lzma_stream_encoder_mt(&s, &mt) == LZMA_OPTIONS_ERROR);
		CHECK(s.internal == NULL);
		mt.preset = 6;
		CHECK(lzma_stream_encoder_mt(&s, &mt) == LZMA_OK);
		/* SYNC_FLUSH is not supported by the MT encoder (:1201-1205) */
		s.next_in = in; s.avail_in = 10; s.next_out = out; s.avail_out = 100;
		CHECK(lzma_code(&s, LZMA_SYNC_FLUSH) == LZMA_PROG_ERROR);
		lzma_end(&s);
		lzma_end(&s);           /* idempotent */
		lzma_end(NULL);
		lzma_stream z = LZMA_STREAM_INIT;
		lzma_end(&z);           /* never initialised */
		CHECK(lzma_stream_encoder_mt_memusage(&mt) != UINT64_MAX);
		mt.preset = 77;
		CHECK(lzma_stream_encoder_mt_memusage(&mt) == UINT64_MAX);
		/* filter chains (common/filter_common.c:250-334): up to three BCJ / delta filters in any order in front
		 * of LZMA2; LZMA2 must be there and must be last; five filters are one too many */
		lzma_options_lzma lz;
		memset(&lz, 0, sizeof(lz));      /* preset 1 (lzma_encoder_presets.c:17-63) */
		lz.dict_size = 1u << 20; lz.lc = 3; lz.lp = 0; lz.pb = 2;
		lz.mode = LZMA_MODE_FAST; lz.nice_len = 128; lz.mf = LZMA_MF_HC4; lz.depth = 8;
		lzma_options_delta dl;
		memset(&dl, 0, sizeof(dl));
		dl.type = LZMA_DELTA_TYPE_BYTE; dl.dist = 4;
		lzma_options_bcj off;
		memset(&off, 0, sizeof(off));
		off.start_offset = 16;
		const lzma_filter END = { LZMA_VLI_UNKNOWN, NULL }, L2 = { LZMA_FILTER_LZMA2, &lz }, X86 = { LZMA_FILTER_X86, NULL },
				DL = { LZMA_FILTER_DELTA, &dl }, A64 = { LZMA_FILTER_ARM64, NULL }, PPC = { LZMA_FILTER_POWERPC, NULL },
				XOFF = { LZMA_FILTER_X86, &off }, DNULL = { LZMA_FILTER_DELTA, NULL };
		const struct { lzma_filter f[6]; lzma_ret want; } chains[] = {
			{ { L2, END }, LZMA_OK },
			{ { DL, L2, END }, LZMA_OK },
			{ { X86, DL, L2, END }, LZMA_OK },
			{ { DL, X86, A64, L2, END }, LZMA_OK },
			{ { A64, A64, A64, L2, END }, LZMA_OK },
			{ { DL, X86, A64, PPC, L2, END }, LZMA_OPTIONS_ERROR },
			{ { X86, END }, LZMA_OPTIONS_ERROR },
			{ { END }, LZMA_OPTIONS_ERROR },
			{ { L2, X86, END }, LZMA_OPTIONS_ERROR },
			{ { L2, L2, END }, LZMA_OPTIONS_ERROR },
			{ { XOFF, L2, END }, LZMA_OPTIONS_ERROR },
			{ { X86, DNULL, L2, END }, LZMA_OPTIONS_ERROR },
		};
		mt.preset = 6;
		for (size_t i = 0; i < sizeof(chains) / sizeof(chains[0]); ++i) {
			lzma_stream c = LZMA_STREAM_INIT;
			mt.filters = chains[i].f;
			CHECK(lzma_stream_encoder_mt(&c, &mt) == chains[i].want);
			lzma_end(&c);
		}
		/* LZMA2 option sets (xzamd_options_check, lzma/lzma_common.h:32-37): pb = 3, 4 runs -- with LZMA_MODE_NORMAL in
		 * two-phase mode, a hash-chain finder mapped like a binary-tree one; a span override that leaves the two-phase mode
		 * (XZAMD_SPAN_KIB) is refused with pb > 2; a preset dictionary and a dictionary above 1 GiB are refused */
		{
			lzma_options_lzma q = lz;
			const lzma_filter chain[2] = { { LZMA_FILTER_LZMA2, &q }, { LZMA_VLI_UNKNOWN, NULL } };
			mt.filters = chain;
			const struct { uint32_t pb; lzma_mode mode; lzma_match_finder mf; lzma_ret want; } sets[] = {
				{ 4, LZMA_MODE_FAST, LZMA_MF_HC4, LZMA_OK }, { 3, LZMA_MODE_NORMAL, LZMA_MF_BT4, LZMA_OK },
				{ 4, LZMA_MODE_NORMAL, LZMA_MF_HC4, LZMA_OK }, { 4, LZMA_MODE_NORMAL, LZMA_MF_BT2, LZMA_OK },
				{ 5, LZMA_MODE_NORMAL, LZMA_MF_BT4, LZMA_OPTIONS_ERROR }, { 2, LZMA_MODE_NORMAL, LZMA_MF_HC3, LZMA_OK },
			};
			for (size_t i = 0; i < sizeof(sets) / sizeof(sets[0]); ++i) {
				lzma_stream c = LZMA_STREAM_INIT;
				q.pb = sets[i].pb; q.mode = sets[i].mode; q.mf = sets[i].mf;
				CHECK(lzma_stream_encoder_mt(&c, &mt) == sets[i].want);
				lzma_end(&c);
			}
			q.pb = 4; q.mode = LZMA_MODE_NORMAL; q.mf = LZMA_MF_BT4;
			setenv("XZAMD_SPAN_KIB", "128", 1);
			{ lzma_stream c = LZMA_STREAM_INIT; CHECK(lzma_stream_encoder_mt(&c, &mt) == LZMA_OPTIONS_ERROR); lzma_end(&c); }
			q.pb = 2;
			{ lzma_stream c = LZMA_STREAM_INIT; CHECK(lzma_stream_encoder_mt(&c, &mt) == LZMA_OK); lzma_end(&c); }
			unsetenv("XZAMD_SPAN_KIB");
			static const uint8_t pd[16] = { 1, 2, 3 };
			q.preset_dict = pd; q.preset_dict_size = sizeof(pd);
			{ lzma_stream c = LZMA_STREAM_INIT; CHECK(lzma_stream_encoder_mt(&c, &mt) == LZMA_OPTIONS_ERROR); lzma_end(&c); }
			q.preset_dict = NULL; q.preset_dict_size = 0;
			q.dict_size = (1u << 30) + (1u << 28);
			{ lzma_stream c = LZMA_STREAM_INIT; CHECK(lzma_stream_encoder_mt(&c, &mt) == LZMA_OPTIONS_ERROR); lzma_end(&c); }
		}
		mt.filters = NULL;
	}

	/* 5. one-shot buffer API, empty input */
	{
		size_t pos = 0;
		CHECK(lzma_easy_buffer_encode(1, LZMA_CHECK_CRC64, NULL, in, 600000, out, &pos, n1) == LZMA_OK);
		save(dir, "case5.in", in, 600000);
		save(dir, "case5.xz", out, pos);
		pos = 0;
		CHECK(lzma_easy_buffer_encode(6, LZMA_CHECK_CRC64, NULL, in, 0, out, &pos, n1) == LZMA_OK);
		save(dir, "case6.in", in, 0);
		save(dir, "case6.xz", out, pos);
		pos = 5;
		CHECK(lzma_easy_buffer_encode(6, LZMA_CHECK_CRC64, NULL, in, 600000, out, &pos, 100) == LZMA_BUF_ERROR && pos == 5);
	}
	/* 7. a device failure in the middle of a Stream (XZAMD_TEST_FAIL_JOB: the job with that sequence number fails):
	 * by default the Stream fails and the error latches (common.c:368-372); with XZAMD_STORED_ON_DEVICE_ERROR=1 the
	 * job's Blocks are stored and the Stream stays valid */
	{
		setenv("XZAMD_BATCH_MIB", "1", 1);
		setenv("XZAMD_TEST_FAIL_JOB", "1", 1);
		memset(&mt, 0, sizeof(mt));
		mt.threads = 1; mt.preset = 1; mt.check = LZMA_CHECK_CRC64; mt.block_size = 256u << 10;
		lzma_stream s = LZMA_STREAM_INIT;
		CHECK(lzma_stream_encoder_mt(&s, &mt) == LZMA_OK);
		s.next_in = in; s.avail_in = n2; s.next_out = out; s.avail_out = n1;
		lzma_ret r;
		do r = lzma_code(&s, LZMA_FINISH); while (r == LZMA_OK);
		CHECK(r == LZMA_PROG_ERROR);
		CHECK(lzma_code(&s, LZMA_FINISH) == LZMA_PROG_ERROR);
		lzma_end(&s);
		setenv("XZAMD_STORED_ON_DEVICE_ERROR", "1", 1);
		w = stream_encode(&mt, in, n2, 1u << 20, 1u << 20, 0, LZMA_RUN, out, n1 + (n1 >> 2) + (1u << 20));
		save(dir, "case7.in", in, n2);
		save(dir, "case7.xz", out, w);
		unsetenv("XZAMD_STORED_ON_DEVICE_ERROR");
		unsetenv("XZAMD_TEST_FAIL_JOB");
	}
	/* 8. consecutive jobs of the two-phase encode are pipelined by the worker (the call of a job returns with the back end
	 * of its last batch in flight, the next job finishes it): same bytes as with XZAMD_NO_DEFER=1, three workers on top
	 * of each other, and a device failure of a job in the middle -- while the job before it is still deferred -- either
	 * fails the Stream or, with the knob, stores that job's Blocks only */
	{
		setenv("XZAMD_BATCH_MIB", "1", 1);
		memset(&mt, 0, sizeof(mt));
		mt.threads = 1; mt.preset = 6; mt.check = LZMA_CHECK_CRC64; mt.block_size = 256u << 10;
		const size_t cap = n1 + (n1 >> 2) + (1u << 20);
		uint8_t *alt = (uint8_t *)malloc(cap);
		CHECK(alt != NULL);
		w = stream_encode(&mt, in, n1, 300000, 65536, 0, LZMA_RUN, out, cap);
		save(dir, "case8.in", in, n1);
		save(dir, "case8.xz", out, w);
		setenv("XZAMD_NO_DEFER", "1", 1);
		size_t w2 = stream_encode(&mt, in, n1, 300000, 65536, 0, LZMA_RUN, alt, cap);      /* (same job split: the stub's bytes depend on it) */
		unsetenv("XZAMD_NO_DEFER");
		CHECK(w2 == w && memcmp(out, alt, w) == 0);
		setenv("XZAMD_TEST_WORKERS", "3", 1);
		mt.threads = 3; mt.timeout = 1;
		w2 = stream_encode(&mt, in, n1, 300000, 4096, 0, LZMA_RUN, alt, cap);
		unsetenv("XZAMD_TEST_WORKERS");
		CHECK(w2 == w && memcmp(out, alt, w) == 0);
		mt.threads = 1; mt.timeout = 0;
		setenv("XZAMD_TEST_FAIL_JOB", "2", 1);
		lzma_stream s = LZMA_STREAM_INIT;
		CHECK(lzma_stream_encoder_mt(&s, &mt) == LZMA_OK);
		s.next_in = in; s.avail_in = n1; s.next_out = alt; s.avail_out = cap;
		lzma_ret r;
		do r = lzma_code(&s, LZMA_FINISH); while (r == LZMA_OK);
		CHECK(r == LZMA_PROG_ERROR);
		lzma_end(&s);
		setenv("XZAMD_STORED_ON_DEVICE_ERROR", "1", 1);
		w2 = stream_encode(&mt, in, n1, 1u << 20, 1u << 20, 0, LZMA_RUN, alt, cap);
		save(dir, "case9.in", in, n1);
		save(dir, "case9.xz", alt, w2);
		unsetenv("XZAMD_STORED_ON_DEVICE_ERROR");
		unsetenv("XZAMD_TEST_FAIL_JOB");
		free(alt);
	}
	/* 10. the one-shot API writes ONE Block whatever the input size, like the reference (stream_buffer_encoder.c:91-101):
	 * 5 MiB at preset 1, whose default Block size is 3 MiB.  11: when the device cannot hold such a Block (here: an
	 * allocation limit) the MT layout is written instead */
	{
		unsetenv("XZAMD_BATCH_MIB");
		size_t pos = 0;
		CHECK(lzma_easy_buffer_encode(1, LZMA_CHECK_CRC64, NULL, in, n1, out, &pos, n1 + (n1 >> 2) + (1u << 20)) == LZMA_OK);
		save(dir, "case10.in", in, n1);
		save(dir, "case10.xz", out, pos);
		xzamd_release_parked();
		setenv("XZAMD_TEST_ALLOC_LIMIT_MIB", getenv("DRIVER_ALLOC_LIMIT_MIB") ? getenv("DRIVER_ALLOC_LIMIT_MIB") : "20", 1);
		pos = 0;
		CHECK(lzma_easy_buffer_encode(1, LZMA_CHECK_CRC64, NULL, in, n1, out, &pos, n1 + (n1 >> 2) + (1u << 20)) == LZMA_OK);
		save(dir, "case11.in", in, n1);
		save(dir, "case11.xz", out, pos);
		unsetenv("XZAMD_TEST_ALLOC_LIMIT_MIB");
	}
	xzamd_release_parked();
	free(in);
	free(out);
	puts("driver: ok");
	return 0;
}
