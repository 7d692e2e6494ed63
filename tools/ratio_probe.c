/* ratio_probe.c -- measurement tool (not product code): compressed size of the product's algorithm (through the
 * oracle restatement, compiled in with whatever -D overrides are under test) next to the real liblzma
 * (oracle/_ref/libref_shim.so) on a corpus file, Block by Block.
 *   cc -O2 -pthread -o /tmp/ratio_probe tools/ratio_probe.c oracle/lzma_fast_enc.c oracle/xz_container.c oracle/lzma2_dec.c -ldl
 *   /tmp/ratio_probe FILE PRESET SPAN W [BLOCK_MIB] [noref|ref] [SA_DEPTH] [SPAN_COST] [SPAN_BITS]
 * PRESET: 6, 9e ...; SPAN 0 = whole Block; prints the total sizes and the delta. */
#define _GNU_SOURCE
#include "../oracle/oracle.h"
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int (*ref_raw_fn)(const uint8_t *, size_t, uint32_t, uint32_t, uint32_t, uint32_t, int, uint32_t, int, uint32_t,
		uint8_t *, size_t, size_t *);
typedef int (*ref_raw_dec_fn)(const uint8_t *, size_t, uint32_t, uint8_t *, size_t, size_t *);

static const uint8_t *g_in;
static uint64_t g_n, g_block;
static orc_enc_params g_prm;
static uint32_t g_nice, g_depth;
static ref_raw_fn g_ref;
static ref_raw_dec_fn g_refdec;
static uint64_t *g_ours, *g_refsz;
static int g_do_ref, g_verify;
static uint64_t g_nb;
static volatile uint64_t g_next;

static void *work(void *arg)
{
	(void)arg;
	for (;;) {
		const uint64_t job = __sync_fetch_and_add(&g_next, 1);
		if (job >= 2 * g_nb) break;
		const uint64_t b = job >> 1;
		const uint64_t off = b * g_block, len = g_n - off < g_block ? g_n - off : g_block;
		const uint64_t cap = len + len / 8 + 65536;
		uint8_t *out = (uint8_t *)malloc(cap);
		if (job & 1) {
			if (g_do_ref) {
				size_t sz = 0;
				int r = g_ref(g_in + off, len, g_prm.dict_size, 3, 0, 2, 2, g_nice, 0x14, g_depth, out, cap, &sz);
				if (r != 1) { fprintf(stderr, "ref encode failed %d\n", r); exit(2); }
				g_refsz[b] = sz;
			}
		} else {
			uint64_t sz = 0;
			int r = orc_lzma2_encode_block(g_in + off, (uint32_t)len, &g_prm, out, cap, &sz, NULL);
			if (r) { fprintf(stderr, "oracle encode failed %d\n", r); exit(2); }
			g_ours[b] = sz;
			if (g_verify) {
				uint8_t *dec = (uint8_t *)malloc(len + 16);
				size_t dn = 0;
				r = g_refdec(out, sz, g_prm.dict_size, dec, len + 16, &dn);
				if (r != 1 || dn != len || memcmp(dec, g_in + off, len)) { fprintf(stderr, "ROUND TRIP FAILED block %llu (r=%d)\n", (unsigned long long)b, r); exit(3); }
				free(dec);
			}
		}
		free(out);
	}
	return NULL;
}

int main(int argc, char **argv)
{
	if (argc < 5) { fprintf(stderr, "usage: %s FILE PRESET SPAN W [BLOCK_MIB] [noref]\n", argv[0]); return 1; }
	FILE *f = fopen(argv[1], "rb");
	if (!f) { perror(argv[1]); return 1; }
	fseek(f, 0, SEEK_END); g_n = (uint64_t)ftell(f); fseek(f, 0, SEEK_SET);
	uint8_t *in = (uint8_t *)malloc(g_n);
	if (fread(in, 1, g_n, f) != g_n) return 1;
	fclose(f);
	g_in = in;
	uint32_t preset = (uint32_t)strtoul(argv[2], NULL, 10);
	if (strchr(argv[2], 'e')) preset |= 0x80000000u;
	uint32_t normal = 0;
	if (orc_preset(preset, &g_prm, &normal)) return 1;
	g_nice = g_prm.nice_len; g_depth = g_prm.depth;
	g_prm.mf = 4; g_prm.depth = 1; g_prm.parser = 1;
	g_prm.span_size = (uint32_t)strtoul(argv[3], NULL, 10);
	g_prm.sa_window = (uint32_t)strtoul(argv[4], NULL, 10);
	g_block = (uint64_t)g_prm.dict_size * 3;
	if (argc > 5 && atof(argv[5]) > 0) g_block = (uint64_t)(atof(argv[5]) * 1048576.0);
	g_do_ref = !(argc > 6 && !strcmp(argv[6], "noref"));
	if (argc > 7) g_prm.sa_depth = (uint32_t)strtoul(argv[7], NULL, 10);
	if (argc > 8) g_prm.span_cost = (uint32_t)strtoul(argv[8], NULL, 10);
	if (argc > 9) g_prm.span_bits = (uint32_t)strtoul(argv[9], NULL, 10);
	g_verify = getenv("PROBE_VERIFY") != NULL;
	if (getenv("PROBE_ENC_BITS")) g_prm.enc_bits = (uint32_t)strtoul(getenv("PROBE_ENC_BITS"), NULL, 10);
	void *h = dlopen("/root/repo/oracle/_ref/libref_shim.so", RTLD_NOW);
	if (!h) { fprintf(stderr, "%s\n", dlerror()); return 1; }
	g_ref = (ref_raw_fn)dlsym(h, "ref_raw_lzma2_encode");
	g_refdec = (ref_raw_dec_fn)dlsym(h, "ref_raw_lzma2_decode");
	g_nb = (g_n + g_block - 1) / g_block;
	g_ours = (uint64_t *)calloc(g_nb, 8); g_refsz = (uint64_t *)calloc(g_nb, 8);
	int nt = getenv("PROBE_THREADS") ? atoi(getenv("PROBE_THREADS")) : 8;
	pthread_t th[64];
	for (int t = 0; t < nt; ++t) pthread_create(&th[t], NULL, work, NULL);
	for (int t = 0; t < nt; ++t) pthread_join(th[t], NULL);
	uint64_t so = 0, sr = 0;
	for (uint64_t b = 0; b < g_nb; ++b) { so += g_ours[b]; sr += g_refsz[b]; }
	printf("%s preset %s span %u W %u depth %u cost %u bits %u enc_bits %u block %llu: ours %llu ref %llu", argv[1], argv[2], g_prm.span_size, g_prm.sa_window,
			g_prm.sa_depth, g_prm.span_cost, g_prm.span_bits, g_prm.enc_bits, (unsigned long long)g_block, (unsigned long long)so, (unsigned long long)sr);
	if (sr) printf("  delta %+.2f%%  (ratio ours %.4f ref %.4f)", 100.0 * ((double)so / (double)sr - 1.0), (double)so / g_n, (double)sr / g_n);
	printf("\n");
	return 0;
}
