/*
 * lzma_fast_enc.c -- TEST INFRASTRUCTURE ONLY (oracle).
 *
 * CPU restatement of the LZMA2 encode path for ONE Block -- the code a worker
 * thread of lzma_stream_encoder_mt runs:
 *
 *   lz/lz_encoder_mf.c:22-79     lzma_mf_find (longest-match extension)
 *   lz/lz_encoder_mf.c:190-201   header macro (len_limit / pending rule)
 *   lz/lz_encoder_mf.c:250-287   hc_find_func (chain walk)
 *   lz/lz_encoder_mf.c:305-361   HC3 find/skip
 *   lz/lz_encoder_mf.c:366-441   HC4 find/skip
 *   lz/lz_encoder_hash.h:55-75   hash_3_calc / hash_4_calc
 *   lz/lz_encoder.c:306-345      hash_mask / table sizing, :359-365 depth default
 *   lzma/lzma_encoder_optimum_fast.c:20-169  greedy/lazy parser
 *   lzma/lzma_encoder.c:23-263   literal/match/rep symbol coding
 *   lzma/lzma_encoder.c:313-436  per-chunk loop and its two cut-off rules
 *   rangecoder/range_encoder.h:136-263       rc_shift_low / rc_encode
 *   rangecoder/price.h, price_tablegen.c     bit prices (optimal parser)
 *   lzma/lzma2_encoder.c:54-259  chunk headers, uncompressed-chunk fallback
 *
 * Layout mirrors the HIP kernels: the whole Block is resident, positions are
 * absolute offsets, and -- because the reference inserts EVERY position into
 * its hash tables (find and skip both do) -- the chain links are a pure
 * function of the data and are precomputed once (prev2/prev3/son arrays)
 * instead of being maintained while parsing.  Bits are coded directly instead
 * of being queued.  Every decision on the "exact" path (mf = 3/4, parser 0,
 * span_size 0) is the reference's, so the output is byte-identical to
 * liblzma's raw LZMA2 encoder (pinned in tests/test_oracle_encoder.py).
 *
 * OUR definitions (not the reference's; the HIP path must reproduce them
 * bit-exactly, tests/test_gpu_parity.py):
 *   span_size != 0   independent state-reset spans (control 0xC0), matches may
 *                    not cross a span end, chains stay Block-global;
 *   sa_window != 0   "suffix-neighbourhood" match finder, the parallel successor
 *                    of BT4 (lz_encoder_mf.c:450-743 relinks its tree at every
 *                    insert, i.e. is sequential per Block).  BT4's tree is the
 *                    Cartesian tree of (suffix order, insertion time), so the
 *                    nodes its descent visits from position p are exactly the
 *                    "recency records" met when walking away from p in suffix
 *                    order: earlier positions, each more recent than every one
 *                    nearer in suffix order.  We therefore sort all positions of
 *                    the Block by their first 32 bytes (8-byte chunks, two
 *                    rank-doubling rounds, ties by position) and take, for each
 *                    position, the records among the sa_window (<= 5) slots to the
 *                    left and to the right of its own slot, plus the nearest
 *                    previous position with equal hash2 / hash4 and with
 *                    the same 8 / 16 bytes (by-products of the sort rounds).  All candidates
 *                    are merged by the Pareto rule "longer than everything
 *                    closer"; the LIST_K longest survive.
 *   parser == 1      price-based forward dynamic program over a bounded window
 *                    (the role of lzma_encoder_optimum_normal.c:803-858, not its
 *                    code): exact current-probability prices, edges literal /
 *                    short rep / rep0-3 x all lengths / match x all lengths and
 *                    the reference's three compound edges "literal + rep0",
 *                    "rep + literal + rep0", "match + literal + rep0"
 *                    (:562-597, :635-687, :728-790).  A window that is cut by
 *                    the node limit (not by convergence) only commits the
 *                    symbols that end WTAIL nodes before the cut.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#ifdef ORC_SA_STATS
#include <stdio.h>
#endif

#define MATCH_LEN_MAX 273u
#define LIT 0xFFFFFFFFu
#define NO_DELTA 0xFFFFFFFFu
#ifndef WMAX_STD
#define WMAX_STD 384u             /* optimal-parser window (nodes 0..WMAX); sized so the GPU's node arrays +
                                   * price tables fit 10 KiB of LDS per wavefront (round 5: the model moved to L2) */
#endif
#ifndef WTAIL
#define WTAIL 32u
#endif
#define WMAX_LONG 384u            /* nice_len > 128 (the extreme presets): the same window since round 5 */
#define WMAX_CAP WMAX_LONG
#define WTAIL_                 /* symbols ending less than WTAIL nodes before a forced window cut are re-parsed */
#ifndef LIST_K
#define LIST_K 7u
#endif
#define LIST_K_                 /* matches kept per position (the LIST_K longest) */
#ifndef SA_WMAX
#define SA_WMAX 5u
#endif
#define SA_WMAX_                  /* widest suffix-order window per side: the GPU gives every position 16 lanes
                                   * (5 + 5 neighbours, hash2/3/4, prev8, prev16) and four positions to a wavefront */
#define LEN2_MAX 127u             /* cap of the rep0 run of a compound edge */
#define PRICE_INF (1u << 30)

/* ------------------------------------------------------------------ */
/* Probability model layout (flat u16 array; HIP kernels share it)      */
/* lzma/lzma_encoder_private.h:121-135, lzma/lzma_common.h              */
/* ------------------------------------------------------------------ */
enum {
	P_IS_MATCH = 0,                       /* [12][16] */
	P_IS_REP = P_IS_MATCH + 12 * 16,      /* [12] */
	P_IS_REP0 = P_IS_REP + 12,
	P_IS_REP1 = P_IS_REP0 + 12,
	P_IS_REP2 = P_IS_REP1 + 12,
	P_IS_REP0_LONG = P_IS_REP2 + 12,      /* [12][16] */
	P_DIST_SLOT = P_IS_REP0_LONG + 12 * 16, /* [4][64] */
	P_DIST_SPECIAL = P_DIST_SLOT + 4 * 64,  /* [114] */
	P_DIST_ALIGN = P_DIST_SPECIAL + 114,    /* [16] */
	P_MATCH_LEN = P_DIST_ALIGN + 16,        /* len coder: 2 + 16*8 + 16*8 + 256 */
	LEN_CHOICE = 0, LEN_CHOICE2 = 1, LEN_LOW = 2, LEN_MID = 2 + 16 * 8, LEN_HIGH = 2 + 32 * 8,
	LEN_CODER_SIZE = 2 + 32 * 8 + 256,
	P_REP_LEN = P_MATCH_LEN + LEN_CODER_SIZE,
	P_LITERAL = P_REP_LEN + LEN_CODER_SIZE,
	P_TOTAL_MAX = P_LITERAL + (0x300 << 4)
};

typedef struct {
	uint32_t price;
	uint32_t back;      /* edge arriving here: LIT / rep index / dist + 4 */
	uint32_t len;       /* its length (0: the compound edge is just "literal + rep0") */
	uint32_t len2;      /* != 0: compound edge, followed by one literal and a rep0 match of len2 */
	uint32_t state;     /* valid once the node has been visited */
	uint32_t reps[4];
} node;

typedef struct {
	const uint8_t *in;
	uint32_t n;
	orc_enc_params prm;
	uint32_t depth, hash_mask, cyclic_size;
	/* parse-independent chain links (0 = none) */
	uint32_t *prev2, *prev3;    /* delta to the previous position with equal hash2 / hash3 */
	uint32_t *son;              /* main chain: previous position + 1 */
	uint32_t *prev4;            /* delta to the previous position with equal hash4 (sa_window != 0) */
	uint32_t *prev8, *prev16;   /* delta to the previous position with the same 8 / 16 bytes: by-products of the
	                             * suffix-order build (left neighbour inside the group of equal keys) */
	uint32_t *prev24, *prev32;  /* the same for 24 bytes (one extra sort of the 16-byte groups by the 8 bytes behind
	                             * them) and 32 bytes (by-product of round h = 16) */
	uint32_t *sa, *sa_rank;     /* suffix-neighbourhood finder: slot -> position, position -> slot */
	uint32_t span_end;          /* exclusive end for avail computations */
	/* result of the last find */
	uint32_t m_len[MATCH_LEN_MAX + 8], m_dist[MATCH_LEN_MAX + 8];
	uint32_t m_count, m_longest;
	uint32_t m_len2[2];         /* [0]: longest entry, [1]: second longest: rep0 run after the byte following the match */
	uint32_t rep_len[4];
	/* lzma state */
	uint16_t probs[P_TOTAL_MAX];
	uint32_t state, reps[4];
	/* range coder */
	uint64_t low;
	uint64_t cache_size;
	uint32_t range;
	uint8_t cache;
	uint8_t *cbuf;              /* chunk payload buffer */
	uint32_t cpos;
	/* optimal parser */
	node *nodes;
	uint32_t q_back[WMAX_CAP + 2], q_len[WMAX_CAP + 2], q_head, q_count;
	uint32_t wmax;              /* window of this option set: WMAX_STD, or WMAX_LONG when nice_len > 128 */
	uint8_t price_tab[128];
	/* cached price tables (role of length_update_prices / fill_dist_prices /
	 * fill_align_prices), refreshed at window starts by symbol counters */
	uint16_t lp[2][4][272];     /* [match,rep][pos_state][len-2] */
	uint16_t dsp[4][64];        /* dist slot price (+ direct bits for slot >= 14) */
	uint16_t dp[4][128];        /* full price of distances < 128 */
	uint16_t ap[16];            /* align bits */
	uint32_t cnt_len, cnt_match, cnt_align;
	int tables_valid;
	orc_trace *trace;
	/* two-phase mode: the recorded parse (valid at symbol starts) */
	uint16_t *sy_len;           /* 0 = literal, else the length of the match / rep (1 = short rep) */
	uint32_t *sy_dist;          /* zero-based distance; literal: byte | previous byte << 8 | match byte << 16 |
	                             * (parser state >= 7) << 24 -- what the device's coder reads instead of the input */
	int rc_off;                 /* phase 1: adapt the model, code nothing */
	uint32_t lit_rec;           /* phase 2: != 0: the record of the literal being coded | 1 << 31 */
	int est_on;                 /* phase 2: sum the prices of the coded decisions (chunk rule of the two-phase coder) */
	uint64_t est;               /* in 1/16 bit, from the probabilities before their update */
	uint64_t ntok;              /* phase 2: binary decisions (tokens of the device's model pass) of the encode span so far */
	/* carried encode spans (round 6): the bounds every probability of a span's walk stays between whatever the span's
	 * start model is -- what the device's k_model_bounds computes per span in parallel (see carry_* below) */
	int bnd_on;
	uint16_t blo[P_TOTAL_MAX], bhi[P_TOTAL_MAX], bcnt[P_TOTAL_MAX];
#ifdef ORC_XPREV
	uint32_t *xprev[8];
	int xprev_n;
#endif
} enc;

static const uint32_t *crc_table0(void)
{
	/* lz_encoder_hash.h:31-39: the hash table is CRC32's table[0]. */
	static uint32_t t[256];
	static int ready;
	if (!ready) {
		for (uint32_t b = 0; b < 256; ++b) {
			uint32_t r = b;
			for (int k = 0; k < 8; ++k)
				r = (r >> 1) ^ ((r & 1) ? 0xEDB88320u : 0);
			t[b] = r;
		}
		ready = 1;
	}
	return t;
}

static uint32_t cmplen(const uint8_t *a, const uint8_t *b, uint32_t len, uint32_t limit)
{
	/* common/memcmplen.h:47: first `len` bytes are known equal */
	while (len < limit && a[len] == b[len])
		++len;
	return len;
}

/* ---- chain links: what the reference's hash/son arrays contain when each
 * position is reached (lz_encoder_mf.c:366-441: find and skip both insert) ---- */
static int build_links(enc *e)
{
	const uint32_t *T = crc_table0();
	const uint32_t hb = e->prm.mf;
	const uint32_t n = e->n;
	uint32_t *head2 = (uint32_t *)calloc(1024, 4);
	uint32_t *head3 = (uint32_t *)calloc(65536, 4);
	uint32_t *headm = (uint32_t *)calloc((size_t)e->hash_mask + 1, 4);
	if (!head2 || !head3 || !headm)
		return -1;
	for (uint32_t p = 0; p < n; ++p) {
		const uint8_t *cur = e->in + p;
		if (n - p < hb)
			continue;       /* "pending": never inserted (lz_encoder_mf.c:190-201) */
		const uint32_t temp = T[cur[0]] ^ cur[1];
		const uint32_t h2 = temp & 0x3FF;
		e->prev2[p] = head2[h2] ? p + 1 - head2[h2] : 0;
		head2[h2] = p + 1;
		uint32_t h;
		if (hb == 3) {
			h = (temp ^ ((uint32_t)cur[2] << 8)) & e->hash_mask;
		} else {
			const uint32_t h3 = (temp ^ ((uint32_t)cur[2] << 8)) & 0xFFFF;
			e->prev3[p] = head3[h3] ? p + 1 - head3[h3] : 0;
			head3[h3] = p + 1;
			h = (temp ^ ((uint32_t)cur[2] << 8) ^ (T[cur[3]] << 5)) & e->hash_mask;
		}
		e->son[p] = headm[h];
		if (e->prev4)
			e->prev4[p] = headm[h] ? p + 1 - headm[h] : 0;
		headm[h] = p + 1;
	}
	free(head2); free(head3); free(headm);
	return 0;
}

/* ---- suffix order of the Block by its first 32 bytes (OUR definition) --------
 * Order: compare four 8-byte chunks (big-endian, bytes past the Block end read as
 * zero); a chunk that STARTS at or past the Block end sorts before every real
 * chunk; remaining ties by position.  Built the way the GPU builds it: a stable
 * LSD radix sort on the first chunk, then two rank-doubling rounds on the key
 * (rank(p), rank(p + h)), h = 8, 16, where rank = 1 + first slot of the group of
 * equal keys and 0 stands for "past the end". */
static void radix_sort_u64(uint64_t *key, uint32_t *val, uint64_t *key2, uint32_t *val2, uint32_t n)
{
	uint32_t *cnt = (uint32_t *)malloc(65537 * sizeof(uint32_t));
	for (int pass = 0; pass < 4; ++pass) {
		const int sh = pass * 16;
		memset(cnt, 0, 65537 * sizeof(uint32_t));
		for (uint32_t i = 0; i < n; ++i)
			++cnt[((key[i] >> sh) & 0xFFFF) + 1];
		for (uint32_t d = 0; d < 65536; ++d)
			cnt[d + 1] += cnt[d];
		for (uint32_t i = 0; i < n; ++i) {
			const uint32_t o = cnt[(key[i] >> sh) & 0xFFFF]++;
			key2[o] = key[i];
			val2[o] = val[i];
		}
		uint64_t *tk = key; key = key2; key2 = tk;
		uint32_t *tv = val; val = val2; val2 = tv;
	}
	free(cnt);      /* four passes: the result is back in (key, val) */
}

static int build_sa(enc *e)
{
	const uint32_t n = e->n;
	uint64_t *key = (uint64_t *)malloc((size_t)(n + 1) * 8), *key2 = (uint64_t *)malloc((size_t)(n + 1) * 8);
	uint32_t *val2 = (uint32_t *)malloc((size_t)(n + 1) * 4);
	uint32_t *rk8 = (uint32_t *)malloc((size_t)(n + 1) * 4);
	uint32_t *sa = e->sa, *rk = e->sa_rank;
	if (!key || !key2 || !val2 || !rk8) { free(key); free(key2); free(val2); free(rk8); return -1; }
	for (uint32_t p = 0; p < n; ++p) {
		uint64_t v = 0;
		for (uint32_t i = 0; i < 8; ++i)
			v = (v << 8) | (p + i < n ? e->in[p + i] : 0);
		key[p] = v;
		sa[p] = p;
	}
	for (uint32_t h = 8; ; h *= 2) {
		radix_sort_u64(key, sa, key2, val2, n);
		/* rank = 1 + first slot of the group; inside a group positions ascend, so the left neighbour of a
		 * group member is the nearest earlier position with the same 8 (round 0) / 16 (round 1) bytes */
		uint32_t g = 0;
		uint32_t *prevx = h == 8 ? e->prev8 : (h == 16 ? e->prev16 : (h == 32 ? e->prev32 : NULL));
		for (uint32_t i = 0; i < n; ++i) {
			if (i && key[i] != key[i - 1]) g = i;
			rk[sa[i]] = g + 1;
			if (prevx)
				prevx[sa[i]] = g != i ? sa[i] - sa[i - 1] : 0;
		}
#ifdef ORC_SA_STATS
		{
			uint32_t unres = 0;
			for (uint32_t i = 0; i < n; ++i)
				if ((i && key[i] == key[i - 1]) || (i + 1 < n && key[i] == key[i + 1])) ++unres;
			fprintf(stderr, "SA after %u bytes: %.2f%% of positions in groups of >= 2\n", h, 100.0 * unres / (n ? n : 1));
		}
#endif
		if (h == 8)
			memcpy(rk8, rk, (size_t)n * 4);
		if (h == 16) {
			/* prev24: the positions ordered by (16-byte group, rank after 8 bytes of p + 16); stable from the
			 * current order, so equal keys keep ascending positions */
			uint64_t *k24 = (uint64_t *)malloc((size_t)(n + 1) * 8);
			uint32_t *v24 = (uint32_t *)malloc((size_t)(n + 1) * 4);
			if (!k24 || !v24) { free(k24); free(v24); free(key); free(key2); free(val2); free(rk8); return -1; }
			for (uint32_t i = 0; i < n; ++i) {
				const uint32_t p = sa[i];
				k24[i] = ((uint64_t)rk[p] << 32) | (p + 16 < n ? rk8[p + 16] : 0);
				v24[i] = p;
			}
			radix_sort_u64(k24, v24, key2, val2, n);
			for (uint32_t i = 0; i < n; ++i)
				e->prev24[v24[i]] = (i && k24[i] == k24[i - 1]) ? v24[i] - v24[i - 1] : 0;
			free(k24); free(v24);
		}
		if (h >= (e->prm.sa_depth ? e->prm.sa_depth : 32u))
			break;
		for (uint32_t i = 0; i < n; ++i) {
			const uint32_t p = sa[i];
			key[i] = ((uint64_t)rk[p] << 32) | (p + h < n ? rk[p + h] : 0);
		}
	}
	for (uint32_t i = 0; i < n; ++i)
		rk[sa[i]] = i;              /* position -> slot */
	free(key); free(key2); free(val2); free(rk8);
	return 0;
}

/* ---- exact HC3/HC4: what lzma_mf_find() reports at position p ------------- */
static void find_exact(enc *e, uint32_t p)
{
	const uint8_t *cur = e->in + p;
	const uint32_t hb = e->prm.mf;
	const uint32_t nice = e->prm.nice_len;
	const uint32_t avail = e->span_end - p;
	uint32_t count = 0;
	e->m_count = 0;
	e->m_longest = 0;

	/* header(): lz_encoder_mf.c:190-201 */
	uint32_t len_limit = avail;
	if (nice <= len_limit)
		len_limit = nice;
	else if (len_limit < hb)
		return;

	uint32_t len_best;
	int done = 0;
	/* An empty slot gives delta = pos - 0 >= cyclic_size in the reference
	 * (positions start at cyclic_size, lz_encoder.c:395). */
	uint32_t delta2 = e->prev2[p] ? e->prev2[p] : NO_DELTA;
	if (hb == 3) {
		/* lz_encoder_mf.c:305-335 */
		len_best = 2;
		if (delta2 < e->cyclic_size && cur[-(int64_t)delta2] == cur[0]) {
			len_best = cmplen(cur - delta2, cur, len_best, len_limit);
			e->m_len[0] = len_best;
			e->m_dist[0] = delta2 - 1;
			count = 1;
			if (len_best == len_limit)
				done = 1;
		}
	} else {
		/* lz_encoder_mf.c:366-413 */
		const uint32_t delta3 = e->prev3[p] ? e->prev3[p] : NO_DELTA;
		len_best = 1;
		if (delta2 < e->cyclic_size && cur[-(int64_t)delta2] == cur[0]) {
			len_best = 2;
			e->m_len[0] = 2;
			e->m_dist[0] = delta2 - 1;
			count = 1;
		}
		if (delta2 != delta3 && delta3 < e->cyclic_size && cur[-(int64_t)delta3] == cur[0]) {
			len_best = 3;
			e->m_dist[count++] = delta3 - 1;
			delta2 = delta3;
		}
		if (count != 0) {
			len_best = cmplen(cur - delta2, cur, len_best, len_limit);
			e->m_len[count - 1] = len_best;
			if (len_best == len_limit)
				done = 1;
		}
		if (len_best < 3)
			len_best = 3;
	}

	if (!done) {
		/* hc_find_func: lz_encoder_mf.c:250-287 */
		uint32_t depth = e->depth;
		uint32_t cur_match = e->son[p];
		for (;;) {
			if (cur_match == 0)
				break;
			const uint32_t delta = p + 1 - cur_match;
			if (depth-- == 0 || delta >= e->cyclic_size)
				break;
			const uint8_t *pb = cur - delta;
			cur_match = e->son[p - delta];
			if (pb[len_best] == cur[len_best] && pb[0] == cur[0]) {
				uint32_t len = cmplen(pb, cur, 1, len_limit);
				if (len_best < len) {
					len_best = len;
					e->m_len[count] = len;
					e->m_dist[count] = delta - 1;
					++count;
					if (len == len_limit)
						break;
				}
			}
		}
	}

	/* lzma_mf_find wrapper: lz_encoder_mf.c:22-79 */
	e->m_count = count;
	if (count > 0) {
		uint32_t lb = e->m_len[count - 1];
		if (lb == nice) {
			uint32_t limit = avail < MATCH_LEN_MAX ? avail : MATCH_LEN_MAX;
			lb = cmplen(cur, cur - e->m_dist[count - 1] - 1, lb, limit);
		}
		e->m_longest = lb;
	}
}

/* ---- suffix-neighbourhood finder (OUR definition, see file header) -------- */
static void find_sn(enc *e, uint32_t p)
{
	const uint8_t *cur = e->in + p;
	const uint32_t nice = e->prm.nice_len;
	const uint32_t avail = e->n - p;          /* the lists are span independent: the parser clamps them (do_round) */
	const uint32_t W = e->prm.sa_window;
	e->m_count = 0;
	e->m_longest = 0;
	e->m_len2[0] = e->m_len2[1] = 0;
	uint32_t len_limit = avail;
	if (nice <= len_limit)
		len_limit = nice;
	else if (len_limit < 4)
		return;

	/* candidates: nearest equal hash2 / hash4 / 8 bytes / 16 bytes, then the recency records of both sides.
	 * (No hash3 head: next to the others it was measured to be worth 0.0 % on text and 0.1 % on executables,
	 * and on the GPU it costs a sort and an inversion of the whole batch.) */
	uint32_t cd[64], cl[64], nc = 0;
	const uint32_t d2 = e->prev2[p], d4 = e->prev4[p], d8 = e->prev8[p], d16 = e->prev16[p];
	const uint32_t d24 = e->prev24[p], d32 = e->prev32[p];
	if (d2 && d2 < e->cyclic_size) {
		uint32_t L = cmplen(cur - d2, cur, 0, len_limit);
		if (L >= 2) { cd[nc] = d2; cl[nc] = L; ++nc; }
	}
	if (d4 && d4 < e->cyclic_size) {
		uint32_t L = cmplen(cur - d4, cur, 0, len_limit);
		if (L >= 4) { cd[nc] = d4; cl[nc] = L; ++nc; }
	}
	if (d8 && d8 < e->cyclic_size) {
		uint32_t L = cmplen(cur - d8, cur, 0, len_limit);
		if (L >= 4) { cd[nc] = d8; cl[nc] = L; ++nc; }
	}
	if (d16 && d16 < e->cyclic_size) {
		uint32_t L = cmplen(cur - d16, cur, 0, len_limit);
		if (L >= 4) { cd[nc] = d16; cl[nc] = L; ++nc; }
	}
	if (d24 && d24 < e->cyclic_size) {
		uint32_t L = cmplen(cur - d24, cur, 0, len_limit);
		if (L >= 4) { cd[nc] = d24; cl[nc] = L; ++nc; }
	}
	if (d32 && d32 < e->cyclic_size) {
		uint32_t L = cmplen(cur - d32, cur, 0, len_limit);
		if (L >= 4) { cd[nc] = d32; cl[nc] = L; ++nc; }
	}
#ifdef ORC_CHAIN4
	{       /* experiment: further steps along the hash4 / 8-byte / 16-byte chains */
		uint32_t q = p;
		for (int st = 0; st < ORC_CHAIN4 + 1; ++st) {
			const uint32_t d = e->prev4[q];
			if (!d) break;
			q -= d;
			if (p - q >= e->cyclic_size) break;
			if (st == 0) continue;
			uint32_t L = cmplen(cur - (p - q), cur, 0, len_limit);
			if (L >= 4 && nc < 60) { cd[nc] = p - q; cl[nc] = L; ++nc; }
		}
#ifdef ORC_CHAIN8
		q = p;
		for (int st = 0; st < ORC_CHAIN8 + 1; ++st) {
			const uint32_t d = e->prev8[q];
			if (!d) break;
			q -= d;
			if (p - q >= e->cyclic_size) break;
			if (st == 0) continue;
			uint32_t L = cmplen(cur - (p - q), cur, 0, len_limit);
			if (L >= 4 && nc < 60) { cd[nc] = p - q; cl[nc] = L; ++nc; }
		}
#endif
	}
#endif
#ifdef ORC_XPREV
	for (int xi = 0; xi < e->xprev_n; ++xi) {        /* experiment: nearest earlier position with the same K bytes */
		const uint32_t dx = e->xprev[xi][p];
		if (dx && dx < e->cyclic_size) {
			uint32_t L = cmplen(cur - dx, cur, 0, len_limit);
			if (L >= 2) { cd[nc] = dx; cl[nc] = L; ++nc; }
		}
	}
#endif
	const uint32_t r = e->sa_rank[p];
	for (int side = 0; side < 2; ++side) {
		uint32_t recent = 0;        /* most recent eligible position seen so far on this side, + 1 */
		for (uint32_t k = 1; k <= W; ++k) {
			if (side == 0 ? r < k : r + k >= e->n)
				break;
			const uint32_t q = e->sa[side == 0 ? r - k : r + k];
			if (q < p && p - q < e->cyclic_size && q + 1 > recent) {
				recent = q + 1;
				uint32_t L = cmplen(cur - (p - q), cur, 0, len_limit);
				if (L >= 4) { cd[nc] = p - q; cl[nc] = L; ++nc; }
			}
		}
	}
	/* Pareto set: (L, delta) survives iff no candidate is closer and at least as long; duplicates merge */
	uint32_t count = 0;
	for (uint32_t i = 0; i < nc; ++i) {
		int keep = 1;
		for (uint32_t j = 0; j < nc && keep; ++j)
			if ((cd[j] < cd[i] && cl[j] >= cl[i]) || (cd[j] == cd[i] && j < i))
				keep = 0;
		if (!keep)
			continue;
		uint32_t k = count++;       /* insert sorted by delta (== sorted by length) */
		while (k > 0 && e->m_dist[k - 1] > cd[i] - 1) {
			e->m_dist[k] = e->m_dist[k - 1];
			e->m_len[k] = e->m_len[k - 1];
			--k;
		}
		e->m_dist[k] = cd[i] - 1;
		e->m_len[k] = cl[i];
	}
	if (count > LIST_K) {
		/* per-position lists hold the LIST_K longest (the GPU's k_find writes them for the whole batch) */
		const uint32_t drop = count - LIST_K;
		memmove(e->m_len, e->m_len + drop, LIST_K * sizeof(e->m_len[0]));
		memmove(e->m_dist, e->m_dist + drop, LIST_K * sizeof(e->m_dist[0]));
		count = LIST_K;
	}
	e->m_count = count;
	if (count > 0) {
		uint32_t lb = e->m_len[count - 1];
		if (lb == nice) {
			uint32_t limit = avail < MATCH_LEN_MAX ? avail : MATCH_LEN_MAX;
			lb = cmplen(cur, cur - e->m_dist[count - 1] - 1, lb, limit);
		}
		e->m_longest = lb;
		/* rep0 run behind the byte that follows the match, for the two longest entries
		 * (the "match + literal + rep0" edge, lzma_encoder_optimum_normal.c:728-790) */
		for (uint32_t t = 0; t < 2 && t < count; ++t) {
			const uint32_t L = t == 0 ? lb : e->m_len[count - 2];
			const uint32_t dist = e->m_dist[count - 1 - t];
			uint32_t l2 = 0;
			if (L + 1 < avail) {
				uint32_t lim = L + 1 + LEN2_MAX;
				if (lim > avail) lim = avail;
				l2 = cmplen(cur, cur - dist - 1, L + 1, lim) - (L + 1);
			}
			e->m_len2[t] = l2;
		}
	}
}

/* One "round" at position x: matches + the four rep-match lengths with reps r[]. */
static void do_round(enc *e, uint32_t x, const uint32_t r[4])
{
	if (e->prm.sa_window) {
		find_sn(e, x);
		/* The record of a position is made for the whole Block (k_find_sn runs before the spans are cut); a
		 * span may not reference bytes behind its end, so the parser clamps what it reads: lengths to the
		 * bytes left in the span, the rep0 run of the compound edge to what follows match + literal. */
		const uint32_t left = e->span_end - x;
		if (left < MATCH_LEN_MAX + 1 + LEN2_MAX) {
			for (uint32_t t = 0; t < 2 && t < e->m_count; ++t) {
				const uint32_t L = t == 0 ? e->m_longest : e->m_len[e->m_count - 2];
				uint32_t l2 = e->m_len2[t];
				if (L + 1 >= left) l2 = 0;
				else if (l2 > left - L - 1) l2 = left - L - 1;
				e->m_len2[t] = l2;
			}
			for (uint32_t k = 0; k < e->m_count; ++k)
				if (e->m_len[k] > left) e->m_len[k] = left;
			if (e->m_longest > left) e->m_longest = left;
		}
	} else {
		find_exact(e, x);
		e->m_len2[0] = e->m_len2[1] = 0;
		if (e->prm.parser && e->m_count > LIST_K) {
			const uint32_t drop = e->m_count - LIST_K;
			memmove(e->m_len, e->m_len + drop, LIST_K * sizeof(e->m_len[0]));
			memmove(e->m_dist, e->m_dist + drop, LIST_K * sizeof(e->m_dist[0]));
			e->m_count = LIST_K;
		}
	}
	const uint32_t rem = e->span_end - x;
	const uint32_t buf_avail = rem < MATCH_LEN_MAX ? rem : MATCH_LEN_MAX;
	for (uint32_t i = 0; i < 4; ++i)
		e->rep_len[i] = cmplen(e->in + x, e->in + x - r[i] - 1, 0, buf_avail);
}

/* ---- range coder: rangecoder/range_encoder.h:136-263 --------------------- */
static void rc_reset(enc *e)
{
	e->low = 0;
	e->cache_size = 1;
	e->range = 0xFFFFFFFFu;
	e->cache = 0;
}

static void rc_shift_low(enc *e)
{
	if ((uint32_t)e->low < 0xFF000000u || (uint32_t)(e->low >> 32) != 0) {
		do {
			e->cbuf[e->cpos++] = (uint8_t)(e->cache + (uint8_t)(e->low >> 32));
			e->cache = 0xFF;
		} while (--e->cache_size != 0);
		e->cache = (uint8_t)(e->low >> 24);
	}
	++e->cache_size;
	e->low = (e->low & 0x00FFFFFF) << 8;
}

/* Bounds of a probability over one encode span, whatever value it has at the span start: lo starts at 31, hi at 2017 (the
 * extremes an adapted probability can reach), both take every update of the walk; the update rule is monotone, so the true
 * value stays between them, and once they meet the value is KNOWN without knowing the start.  Until then the bits of the
 * slot are counted (the device logs them: the chain kernel replays them on the true start value), at most ORC_LOG_CAP. */
#define ORC_LOG_CAP 1023u
static uint32_t orc_log_cap;      /* tests: fewer logged bits (orc_set_log_cap; the device: XZAMD_TEST_LOG_CAP) */
void orc_set_log_cap(uint32_t v) { orc_log_cap = v >= 1 && v < ORC_LOG_CAP ? v : 0; }
#define LOG_CAP_NOW (orc_log_cap ? orc_log_cap : ORC_LOG_CAP)
static inline void bnd_update(enc *e, uint32_t idx, uint32_t bit)
{
	uint32_t lo = e->blo[idx], hi = e->bhi[idx];
	if (lo != hi && e->bcnt[idx] < LOG_CAP_NOW) ++e->bcnt[idx];
	e->blo[idx] = (uint16_t)(bit ? lo - (lo >> 5) : lo + ((2048 - lo) >> 5));
	e->bhi[idx] = (uint16_t)(bit ? hi - (hi >> 5) : hi + ((2048 - hi) >> 5));
}

static inline void rc_bit(enc *e, uint16_t *prob, uint32_t bit)
{
	if (e->bnd_on)
		bnd_update(e, (uint32_t)(prob - e->probs), bit);
	if (e->rc_off) {            /* phase 1 of the two-phase mode: the price model adapts, nothing is coded; the prices of the
		                         * decisions (probabilities before their update) are summed: the price of a parse piece */
		uint32_t p = *prob;
		e->est += e->price_tab[(p ^ ((0u - bit) & 0x7FFu)) >> 4];
		*prob = (uint16_t)(bit ? p - (p >> 5) : p + ((2048 - p) >> 5));
		return;
	}
	if (e->range < (1u << 24)) {
		rc_shift_low(e);
		e->range <<= 8;
	}
	uint32_t p = *prob;
	if (e->est_on) {
		e->est += e->price_tab[(p ^ ((0u - bit) & 0x7FFu)) >> 4];
		++e->ntok;
	}
	const uint32_t bound = (e->range >> 11) * p;
	if (!bit) {
		e->range = bound;
		p += (2048 - p) >> 5;
	} else {
		e->low += bound;
		e->range -= bound;
		p -= p >> 5;
	}
	*prob = (uint16_t)p;
}

static void rc_tree(enc *e, uint16_t *probs, uint32_t nbits, uint32_t sym)
{
	uint32_t m = 1;
	do {
		const uint32_t b = (sym >> --nbits) & 1;
		rc_bit(e, &probs[m], b);
		m = (m << 1) + b;
	} while (nbits);
}

static void rc_tree_rev(enc *e, uint16_t *probs, uint32_t nbits, uint32_t sym)
{
	uint32_t m = 1;
	do {
		const uint32_t b = sym & 1;
		sym >>= 1;
		rc_bit(e, &probs[m], b);
		m = (m << 1) + b;
	} while (--nbits);
}

static void rc_direct(enc *e, uint32_t value, uint32_t nbits)
{
	if (e->rc_off) { e->est += 16u * nbits; return; }
	if (e->est_on) { e->est += 16u * nbits; e->ntok += nbits; }
	do {
		if (e->range < (1u << 24)) {
			rc_shift_low(e);
			e->range <<= 8;
		}
		e->range >>= 1;
		if ((value >> --nbits) & 1)
			e->low += e->range;
	} while (nbits);
}

static void rc_flush(enc *e)
{
	/* The queue loop normalizes before the first RC_FLUSH symbol
	 * (range_encoder.h:196-203), then shifts five times (:236-246). */
	if (e->range < (1u << 24)) {
		rc_shift_low(e);
		e->range <<= 8;
	}
	for (int i = 0; i < 5; ++i)
		rc_shift_low(e);
	rc_reset(e);
}

/* ---- LZMA symbol coding: lzma/lzma_encoder.c:23-263 ----------------------- */
static void lzma_state_reset(enc *e)
{
	/* lzma_lzma_encoder_reset, lzma_encoder.c:529-598 */
	for (uint32_t i = 0; i < P_TOTAL_MAX; ++i)
		e->probs[i] = 1024;
	e->state = 0;
	e->reps[0] = e->reps[1] = e->reps[2] = e->reps[3] = 0;
	rc_reset(e);
	e->tables_valid = 0;
}

static uint32_t dist_slot_of(uint32_t d)
{
	/* lzma/fastpos.h:78-86 (bsr form) */
	if (d <= 4)
		return d;
	uint32_t i = 31;
	while (!(d >> i)) --i;
	return (i + i) + ((d >> (i - 1)) & 1);
}

static void enc_length(enc *e, uint32_t base, uint32_t ps, uint32_t len)
{
	uint16_t *lc = e->probs + base;
	len -= 2;
	if (len < 8) {
		rc_bit(e, &lc[LEN_CHOICE], 0);
		rc_tree(e, lc + LEN_LOW + ps * 8, 3, len);
	} else {
		rc_bit(e, &lc[LEN_CHOICE], 1);
		len -= 8;
		if (len < 8) {
			rc_bit(e, &lc[LEN_CHOICE2], 0);
			rc_tree(e, lc + LEN_MID + ps * 8, 3, len);
		} else {
			rc_bit(e, &lc[LEN_CHOICE2], 1);
			rc_tree(e, lc + LEN_HIGH, 8, len - 8);
		}
	}
}

static uint32_t literal_sub(const enc *e, uint32_t pos)
{
	const uint32_t prev = pos ? e->in[pos - 1] : 0;
	const uint32_t lc = e->prm.lc;
	const uint32_t mask = (0x100u << e->prm.lp) - (0x100u >> lc);
	return P_LITERAL + 3u * ((((pos << 8) + prev) & mask) << lc);
}

static void enc_literal(enc *e, uint32_t pos)
{
	const uint8_t cur = e->in[pos];
	uint16_t *sub = e->probs + literal_sub(e, pos);
	if (e->lit_rec) {
		/* two-phase coder: the literal is coded from its RECORD, as the device does (k_encode_syms never reads the
		 * input for a literal): byte, previous byte (context), and the match byte when the record carries one */
		const uint32_t r = e->lit_rec;
		const uint32_t lc = e->prm.lc, mask = (0x100u << e->prm.lp) - (0x100u >> lc);
		const uint8_t c2 = (uint8_t)r;
		sub = e->probs + P_LITERAL + 3u * ((((pos << 8) + ((r >> 8) & 0xFF)) & mask) << lc);
		if (e->state < 7) {
			e->state = e->state <= 3 ? 0 : e->state - 3;
			rc_tree(e, sub, 8, c2);
		} else {
			e->state = e->state <= 9 ? e->state - 3 : e->state - 6;
			uint32_t mb = (r >> 24) & 1 ? (r >> 16) & 0xFF : e->in[pos - e->reps[0] - 1];
			uint32_t off = 0x100, sym = 0x100u + c2;
			do {
				mb <<= 1;
				const uint32_t mbit = mb & off;
				const uint32_t idx = off + mbit + (sym >> 8);
				const uint32_t b = (sym >> 7) & 1;
				rc_bit(e, &sub[idx], b);
				sym <<= 1;
				off &= ~(mb ^ sym);
			} while (sym < 0x10000);
		}
		return;
	}
	if (e->state < 7) {
		e->state = e->state <= 3 ? 0 : e->state - 3;
		rc_tree(e, sub, 8, cur);
	} else {
		e->state = e->state <= 9 ? e->state - 3 : e->state - 6;
		uint32_t mb = e->in[pos - e->reps[0] - 1];
		uint32_t off = 0x100, sym = 0x100u + cur;
		do {
			mb <<= 1;
			const uint32_t mbit = mb & off;
			const uint32_t idx = off + mbit + (sym >> 8);
			const uint32_t b = (sym >> 7) & 1;
			rc_bit(e, &sub[idx], b);
			sym <<= 1;
			off &= ~(mb ^ sym);
		} while (sym < 0x10000);
	}
}

static void trace_sym(enc *e, uint32_t pos, uint32_t back, uint32_t len)
{
	orc_trace *t = e->trace;
	if (!t) return;
	if (t->sym && t->sym_count < t->sym_cap) {
		t->sym[t->sym_count].pos = pos;
		t->sym[t->sym_count].back = back;
		t->sym[t->sym_count].len = len;
	}
	++t->sym_count;
}

static void enc_symbol(enc *e, uint32_t pos, uint32_t back, uint32_t len)
{
	const uint32_t ps = pos & ((1u << e->prm.pb) - 1);
	uint16_t *P = e->probs;
	trace_sym(e, pos, back, len);
	if (back == LIT) {
		rc_bit(e, &P[P_IS_MATCH + e->state * 16 + ps], 0);
		enc_literal(e, pos);
		return;
	}
	rc_bit(e, &P[P_IS_MATCH + e->state * 16 + ps], 1);
	if (back < 4) {
		rc_bit(e, &P[P_IS_REP + e->state], 1);
		if (back == 0) {
			rc_bit(e, &P[P_IS_REP0 + e->state], 0);
			rc_bit(e, &P[P_IS_REP0_LONG + e->state * 16 + ps], len != 1);
		} else {
			const uint32_t dist = e->reps[back];
			rc_bit(e, &P[P_IS_REP0 + e->state], 1);
			if (back == 1) {
				rc_bit(e, &P[P_IS_REP1 + e->state], 0);
			} else {
				rc_bit(e, &P[P_IS_REP1 + e->state], 1);
				rc_bit(e, &P[P_IS_REP2 + e->state], back - 2);
				if (back == 3)
					e->reps[3] = e->reps[2];
				e->reps[2] = e->reps[1];
			}
			e->reps[1] = e->reps[0];
			e->reps[0] = dist;
		}
		if (len == 1) {
			e->state = e->state < 7 ? 9 : 11;
		} else {
			enc_length(e, P_REP_LEN, ps, len);
			e->state = e->state < 7 ? 8 : 11;
			++e->cnt_len;
		}
		return;
	}
	rc_bit(e, &P[P_IS_REP + e->state], 0);
	const uint32_t dist = back - 4;
	e->state = e->state < 7 ? 7 : 10;
	enc_length(e, P_MATCH_LEN, ps, len);
	++e->cnt_len;
	++e->cnt_match;
	const uint32_t slot = dist_slot_of(dist);
	const uint32_t ds = len < 6 ? len - 2 : 3;
	rc_tree(e, P + P_DIST_SLOT + ds * 64, 6, slot);
	if (slot >= 4) {
		const uint32_t fb = (slot >> 1) - 1;
		const uint32_t base = (2 | (slot & 1)) << fb;
		const uint32_t red = dist - base;
		if (slot < 14) {
			rc_tree_rev(e, P + P_DIST_SPECIAL + base - slot - 1, fb, red);
		} else {
			rc_direct(e, red >> 4, fb - 4);
			rc_tree_rev(e, P + P_DIST_ALIGN, 4, red & 15);
			++e->cnt_align;
		}
	}
	e->reps[3] = e->reps[2];
	e->reps[2] = e->reps[1];
	e->reps[1] = e->reps[0];
	e->reps[0] = dist;
}

/* ---- fast parser: lzma/lzma_encoder_optimum_fast.c:20-169 -------------------
 * `cached` = the lookahead round for position `pos` has already been done
 * (coder->matches / longest_match_length of the reference, plus the rep lengths:
 * a literal does not change reps, so they are still valid).  Returns the number
 * of positions the round cache is ahead after the decision (0 or 1). */
#define change_pair(small, big) (((big) >> 7) > (small))

static int optimum_fast(enc *e, uint32_t pos, int cached, uint32_t *back_res, uint32_t *len_res)
{
	const uint32_t nice = e->prm.nice_len;
	if (!cached)
		do_round(e, pos, e->reps);
	uint32_t len_main = e->m_longest;
	uint32_t count = e->m_count;

	const uint32_t rem = e->span_end - pos;
	const uint32_t buf_avail = rem < MATCH_LEN_MAX ? rem : MATCH_LEN_MAX;
	*back_res = LIT; *len_res = 1;
	if (buf_avail < 2)
		return 0;

	uint32_t rep_len = 0, rep_index = 0;
	for (uint32_t i = 0; i < 4; ++i) {
		const uint32_t len = e->rep_len[i];
		if (len < 2)        /* not_equal_16 */
			continue;
		if (len >= nice) {
			*back_res = i; *len_res = len;
			return 0;
		}
		if (len > rep_len) {
			rep_index = i;
			rep_len = len;
		}
	}

	if (len_main >= nice) {
		*back_res = e->m_dist[count - 1] + 4;
		*len_res = len_main;
		return 0;
	}

	uint32_t back_main = 0;
	if (len_main >= 2) {
		back_main = e->m_dist[count - 1];
		while (count > 1 && len_main == e->m_len[count - 2] + 1) {
			if (!change_pair(e->m_dist[count - 2], back_main))
				break;
			--count;
			len_main = e->m_len[count - 1];
			back_main = e->m_dist[count - 1];
		}
		if (len_main == 2 && back_main >= 0x80)
			len_main = 1;
	}

	if (rep_len >= 2) {
		if (rep_len + 1 >= len_main
				|| (rep_len + 2 >= len_main && back_main > (1u << 9))
				|| (rep_len + 3 >= len_main && back_main > (1u << 15))) {
			*back_res = rep_index; *len_res = rep_len;
			return 0;
		}
	}

	if (len_main < 2 || buf_avail <= 2)
		return 0;

	/* lookahead: matches (and rep lengths) of the next byte */
	do_round(e, pos + 1, e->reps);
	if (e->m_longest >= 2) {
		const uint32_t nl = e->m_longest;
		const uint32_t new_dist = e->m_dist[e->m_count - 1];
		if ((nl >= len_main && new_dist < back_main)
				|| (nl == len_main + 1 && !change_pair(back_main, new_dist))
				|| (nl > len_main + 1)
				|| (nl + 1 >= len_main && len_main >= 3 && change_pair(new_dist, back_main)))
			return 1;
	}
	const uint32_t limit = len_main - 1 > 2 ? len_main - 1 : 2;
	for (uint32_t i = 0; i < 4; ++i)
		if (e->rep_len[i] >= limit)     /* memcmp(buf+1, buf+1-rep-1, limit) == 0 */
			return 1;

	*back_res = back_main + 4;
	*len_res = len_main;
	return 0;
}

/* ---- bit prices: rangecoder/price.h:28-92, table from price_tablegen.c:31-58 -- */
static void price_table_init(enc *e)
{
	for (uint32_t i = 8; i < 2048; i += 16) {
		uint32_t w = i, bit_count = 0;
		for (uint32_t j = 0; j < 4; ++j) {
			w *= w;
			bit_count <<= 1;
			while (w >= (1u << 16)) {
				w >>= 1;
				++bit_count;
			}
		}
		e->price_tab[i >> 4] = (uint8_t)((11 << 4) - 15 - bit_count);
	}
}

static inline uint32_t pr_bit(const enc *e, uint32_t idx, uint32_t bit)
{
	return e->price_tab[(e->probs[idx] ^ ((0u - bit) & 0x7FF)) >> 4];
}

static uint32_t pr_tree(const enc *e, uint32_t base, uint32_t nbits, uint32_t sym)
{
	uint32_t price = 0;
	sym += 1u << nbits;
	do {
		const uint32_t bit = sym & 1;
		sym >>= 1;
		price += pr_bit(e, base + sym, bit);
	} while (sym != 1);
	return price;
}

static uint32_t pr_tree_rev(const enc *e, uint32_t base, uint32_t nbits, uint32_t sym)
{
	uint32_t price = 0, m = 1;
	do {
		const uint32_t bit = sym & 1;
		sym >>= 1;
		price += pr_bit(e, base + m, bit);
		m = (m << 1) + bit;
	} while (--nbits);
	return price;
}

static uint32_t pr_literal(const enc *e, uint32_t pos, uint32_t state, uint32_t rep0)
{
	/* get_literal_price: lzma_encoder_optimum_normal.c:21-53 */
	const uint32_t sub = literal_sub(e, pos);
	uint32_t sym = e->in[pos];
	if (state < 7)
		return pr_tree(e, sub, 8, sym);
	uint32_t price = 0, mb = e->in[pos - rep0 - 1], off = 0x100;
	sym += 0x100;
	do {
		mb <<= 1;
		const uint32_t mbit = mb & off;
		const uint32_t idx = off + mbit + (sym >> 8);
		price += pr_bit(e, sub + idx, (sym >> 7) & 1);
		sym <<= 1;
		off &= ~(mb ^ sym);
	} while (sym < 0x10000);
	return price;
}

static uint32_t pr_len(const enc *e, uint32_t base, uint32_t ps, uint32_t len)
{
	/* length_update_prices: lzma_encoder.c:77-102, evaluated on demand */
	len -= 2;
	if (len < 8)
		return pr_bit(e, base + LEN_CHOICE, 0) + pr_tree(e, base + LEN_LOW + ps * 8, 3, len);
	len -= 8;
	if (len < 8)
		return pr_bit(e, base + LEN_CHOICE, 1) + pr_bit(e, base + LEN_CHOICE2, 0)
				+ pr_tree(e, base + LEN_MID + ps * 8, 3, len);
	return pr_bit(e, base + LEN_CHOICE, 1) + pr_bit(e, base + LEN_CHOICE2, 1)
			+ pr_tree(e, base + LEN_HIGH, 8, len - 8);
}

static uint32_t pr_dist(const enc *e, uint32_t dist, uint32_t dist_state)
{
	/* fill_dist_prices / fill_align_prices: lzma_encoder_optimum_normal.c:132-195 */
	const uint32_t slot = dist_slot_of(dist);
	uint32_t price = pr_tree(e, P_DIST_SLOT + dist_state * 64, 6, slot);
	if (slot >= 4) {
		const uint32_t fb = (slot >> 1) - 1;
		const uint32_t base = (2 | (slot & 1)) << fb;
		const uint32_t red = dist - base;
		if (slot < 14)
			price += pr_tree_rev(e, P_DIST_SPECIAL + base - slot - 1, fb, red);
		else
			price += ((fb - 4) << 4) + pr_tree_rev(e, P_DIST_ALIGN, 4, red & 15);
	}
	return price;
}

/* Refresh policy (ours; the reference uses per-table countdowns, lzma_encoder.c:129-133,
 * lzma_encoder_optimum_normal.c:819-826): at a window start, length tables after 64 coded
 * lengths, distance tables after 128 matches, align table after 16 align-coded matches. */
static void refresh_tables(enc *e)
{
	const uint32_t nps = 1u << e->prm.pb;
	if (!e->tables_valid || e->cnt_len >= 64) {
		for (uint32_t c = 0; c < 2; ++c)
			for (uint32_t ps = 0; ps < nps && ps < 4; ++ps)
				for (uint32_t l = 2; l <= MATCH_LEN_MAX; ++l)
					e->lp[c][ps][l - 2] = (uint16_t)pr_len(e, c ? P_REP_LEN : P_MATCH_LEN, ps, l);
		e->cnt_len = 0;
	}
	if (!e->tables_valid || e->cnt_match >= 128) {
		for (uint32_t ds = 0; ds < 4; ++ds) {
			for (uint32_t slot = 0; slot < 64; ++slot) {
				uint32_t pr = pr_tree(e, P_DIST_SLOT + ds * 64, 6, slot);
				if (slot >= 14)
					pr += (((slot >> 1) - 1) - 4) << 4;
				e->dsp[ds][slot] = (uint16_t)pr;
			}
			for (uint32_t d = 0; d < 128; ++d)
				e->dp[ds][d] = (uint16_t)pr_dist(e, d, ds);
		}
		e->cnt_match = 0;
	}
	if (!e->tables_valid || e->cnt_align >= 16) {
		for (uint32_t i = 0; i < 16; ++i)
			e->ap[i] = (uint16_t)pr_tree_rev(e, P_DIST_ALIGN, 4, i);
		e->cnt_align = 0;
	}
	e->tables_valid = 1;
}

static inline uint32_t tab_dist(const enc *e, uint32_t dist, uint32_t ds)
{
	if (dist < 128)
		return e->dp[ds][dist];
	return (uint32_t)e->dsp[ds][dist_slot_of(dist)] + e->ap[dist & 15];
}

static inline uint32_t state_after(uint32_t s, uint32_t back, uint32_t len)
{
	if (back == LIT)
		return s <= 3 ? 0 : (s <= 9 ? s - 3 : s - 6);
	if (back < 4)
		return len == 1 ? (s < 7 ? 9 : 11) : (s < 7 ? 8 : 11);
	return s < 7 ? 7 : 10;
}

static void reps_after(const uint32_t r[4], uint32_t back, uint32_t out[4])
{
	if (back == LIT || back == 0) {
		out[0] = r[0]; out[1] = r[1]; out[2] = r[2]; out[3] = r[3];
	} else if (back < 4) {
		out[0] = r[back];
		uint32_t k = 1;
		for (uint32_t i = 0; i < 4; ++i)
			if (i != back)
				out[k++] = r[i];
		/* rep_match(): the chosen rep moves to the front, the others keep their order */
	} else {
		out[0] = back - 4; out[1] = r[0]; out[2] = r[1]; out[3] = r[2];
	}
}

static inline void relax(node *nd, uint32_t price, uint32_t back, uint32_t len, uint32_t len2)
{
	if (price < nd->price) {
		nd->price = price;
		nd->back = back;
		nd->len = len;
		nd->len2 = len2;
	}
}

/* Number of equal bytes of in[x..] and in[x - dist - 1 ..] behind the first mismatch, as the GPU sees
 * it: one 64-byte row starting at x (offsets >= buf_avail count as mismatches).  `first` = offset of
 * the first mismatch, must be a real one inside the row.  Returns 0 when there is no such run. */
static uint32_t row_run_after(const uint8_t *cur, uint32_t dist, uint32_t first, uint32_t buf_avail)
{
	const uint32_t row = buf_avail < 64 ? buf_avail : 64;
	if (first >= row)
		return 0;
	uint32_t o = first + 1;
	while (o < row && cur[o] == cur[(int64_t)o - dist - 1])
		++o;
	return o - (first + 1);
}

/* ---- optimal parser (OUR definition).  Parses one window starting at `pos`
 * with the coder's current state/reps/probabilities, stores the chosen symbols
 * in the queue.  `cached` as in optimum_fast.  Returns 1 if the round for the
 * position right after the window has been done (nice-length cut). */
#ifdef ORC_STATS
#include <stdio.h>
static uint64_t st_nodes, st_syms;
#define ST_NODE() (++st_nodes)
#define ST_SYM() (++st_syms)
#else
#define ST_NODE() ((void)0)
#define ST_SYM() ((void)0)
#endif
static int optimum_window(enc *e, uint32_t pos, int cached)
{
	const uint32_t nice = e->prm.nice_len;
	const uint32_t pbm = (1u << e->prm.pb) - 1;
	node *nd = e->nodes;
	refresh_tables(e);
	nd[0].price = 0;
	nd[0].state = e->state;
	memcpy(nd[0].reps, e->reps, sizeof(nd[0].reps));
	uint32_t n_end = 0;
	int next_cached = 0, forced = 0;
	uint32_t j = 0;
	for (;;) {
		const uint32_t x = pos + j;
		ST_NODE();
		if (j > 0) {
			/* the path into node j is final: derive its coder state */
			const uint32_t tot = nd[j].len + (nd[j].len2 ? 1 + nd[j].len2 : 0);
			const node *pv = &nd[j - tot];
			uint32_t st = pv->state;
			if (nd[j].len) {
				st = state_after(st, nd[j].back, nd[j].len);
				reps_after(pv->reps, nd[j].back, nd[j].reps);
			} else {
				memcpy(nd[j].reps, pv->reps, sizeof(nd[j].reps));
			}
			if (nd[j].len2)
				st = state_after(state_after(st, LIT, 1), 0, 2);    /* literal, then a long rep0: always 8 */
			nd[j].state = st;
		}
		if (!(j == 0 && cached))
			do_round(e, x, nd[j].reps);
		const uint32_t rem = e->span_end - x;
		const uint32_t buf_avail = rem < MATCH_LEN_MAX ? rem : MATCH_LEN_MAX;
		uint32_t longest = e->m_longest;
		if (j > 0 && longest >= nice) {
			/* lzma_encoder_optimum_normal.c:845-849: cut the window here; the long match is
			 * taken at the start of the next window */
			next_cached = 1;
			break;
		}
		uint32_t rl[4], rmax = 0;
		for (uint32_t i = 0; i < 4; ++i) {
			rl[i] = e->rep_len[i] >= 2 ? e->rep_len[i] : 0;
			if (rl[i] > rmax) rmax = rl[i];
		}
		if (j == 0) {
			/* helper1 shortcuts (:271-439): a nice-length rep or match is taken at once */
			for (uint32_t i = 0; i < 4; ++i)
				if (rl[i] >= nice) {
					e->q_back[0] = i; e->q_len[0] = rl[i]; e->q_count = 1; e->q_head = 0;
					return 0;
				}
			if (longest >= nice) {
				e->q_back[0] = e->m_dist[e->m_count - 1] + 4; e->q_len[0] = longest;
				e->q_count = 1; e->q_head = 0;
				return 0;
			}
		}
		const uint32_t room = e->wmax - j;              /* targets must stay inside the window */
		if (longest > room) longest = room;
		for (uint32_t i = 0; i < 4; ++i)
			if (rl[i] > room) rl[i] = room;
		if (rmax > room) rmax = room;
		const uint32_t reach = longest > rmax ? longest : rmax;
		while (n_end < j + reach || n_end < j + 1)
			nd[++n_end].price = PRICE_INF;
		if (buf_avail == 0)
			break;      /* cannot happen: j < n_end <= span */

		const uint8_t *cur = e->in + x;
		const uint32_t s = nd[j].state, ps = x & pbm, P = nd[j].price;
		const uint32_t pm1 = P + pr_bit(e, P_IS_MATCH + s * 16 + ps, 1);
		/* literal */
		const uint32_t plit = P + pr_bit(e, P_IS_MATCH + s * 16 + ps, 0) + pr_literal(e, x, s, nd[j].reps[0]);
		relax(&nd[j + 1], plit, LIT, 1, 0);
		/* short rep */
		const uint32_t prep = pm1 + pr_bit(e, P_IS_REP + s, 1);
		if (e->rep_len[0] >= 1)
			relax(&nd[j + 1], prep + pr_bit(e, P_IS_REP0 + s, 0) + pr_bit(e, P_IS_REP0_LONG + s * 16 + ps, 0), 0, 1, 0);
		/* reps, every length */
		uint32_t pure[4];
		pure[0] = pr_bit(e, P_IS_REP0 + s, 0) + pr_bit(e, P_IS_REP0_LONG + s * 16 + ps, 1);
		pure[1] = pr_bit(e, P_IS_REP0 + s, 1) + pr_bit(e, P_IS_REP1 + s, 0);
		pure[2] = pr_bit(e, P_IS_REP0 + s, 1) + pr_bit(e, P_IS_REP1 + s, 1) + pr_bit(e, P_IS_REP2 + s, 0);
		pure[3] = pr_bit(e, P_IS_REP0 + s, 1) + pr_bit(e, P_IS_REP1 + s, 1) + pr_bit(e, P_IS_REP2 + s, 1);
		for (uint32_t i = 0; i < 4; ++i)
			for (uint32_t l = 2; l <= rl[i]; ++l)
				relax(&nd[j + l], prep + pure[i] + e->lp[1][ps & 3][l - 2], i, l, 0);
		/* matches, every length: the closest candidate that is long enough */
		const uint32_t pmatch = pm1 + pr_bit(e, P_IS_REP + s, 0);
		if (longest >= 2) {
			uint32_t k = 0;
			for (uint32_t l = 2; l <= longest; ++l) {
				while (k + 1 < e->m_count && e->m_len[k] < l)
					++k;
				const uint32_t dist = e->m_dist[k];
				relax(&nd[j + l], pmatch + e->lp[0][ps & 3][l - 2] + tab_dist(e, dist, l < 6 ? l - 2 : 3),
						dist + 4, l, 0);
			}
		}
		/* compound edges: X + literal + rep0 (lzma_encoder_optimum_normal.c:562-597, 635-687,
		 * 728-790).  After "literal, long rep0" the state is 8 whatever X was.  rep0 price after a
		 * literal coded in state sl at position-state psn: */
#define REP0_AFTER_LIT(sl, psn, l2) (pr_bit(e, P_IS_MATCH + (sl) * 16 + (psn), 1) + pr_bit(e, P_IS_REP + (sl), 1) \
		+ pr_bit(e, P_IS_REP0 + (sl), 0) + pr_bit(e, P_IS_REP0_LONG + (sl) * 16 + (psn), 1) + e->lp[1][(psn) & 3][(l2) - 2])
		/* (a) literal + rep0: the rep0 byte differs here and a run of >= 2 follows inside the row */
		if (e->rep_len[0] == 0) {
			const uint32_t l2 = row_run_after(cur, nd[j].reps[0], 0, buf_avail);
			if (l2 >= 2 && j + 1 + l2 <= e->wmax) {
				const uint32_t s2 = state_after(s, LIT, 1), ps2 = (x + 1) & pbm;
				while (n_end < j + 1 + l2) nd[++n_end].price = PRICE_INF;
				relax(&nd[j + 1 + l2], plit + REP0_AFTER_LIT(s2, ps2, l2), LIT, 0, l2);
			}
		}
		/* (b) rep_i at its full length + literal + rep0 (row-limited: the rep ends inside the row) */
		for (uint32_t i = 0; i < 4; ++i) {
			const uint32_t L1 = e->rep_len[i];
			if (L1 < 2 || L1 > room || L1 >= buf_avail)
				continue;
			const uint32_t l2 = row_run_after(cur, nd[j].reps[i], L1, buf_avail);
			if (l2 < 2 || j + L1 + 1 + l2 > e->wmax)
				continue;
			const uint32_t sr = state_after(s, i, L1), psl = (x + L1) & pbm;
			uint32_t pr = prep + pure[i] + e->lp[1][ps & 3][L1 - 2] + pr_bit(e, P_IS_MATCH + sr * 16 + psl, 0)
				+ pr_literal(e, x + L1, sr, nd[j].reps[i]);
			const uint32_t sl = state_after(sr, LIT, 1), psn = (x + L1 + 1) & pbm;
			pr += REP0_AFTER_LIT(sl, psn, l2);
			while (n_end < j + L1 + 1 + l2) nd[++n_end].price = PRICE_INF;
			relax(&nd[j + L1 + 1 + l2], pr, i, L1, l2);
		}
		/* (c) one of the two longest matches at its full length + literal + rep0; the run lengths come
		 * with the match list (find_sn) */
		for (uint32_t t = 0; t < 2 && t < e->m_count; ++t) {
			const uint32_t k = e->m_count - 1 - t;
			const uint32_t L1 = t == 0 ? e->m_longest : e->m_len[k];
			const uint32_t dist = e->m_dist[k], l2 = e->m_len2[t];
			if (L1 < 2 || L1 > room || l2 < 2 || L1 > 61 || j + L1 + 1 + l2 > e->wmax)
				continue;
			const uint32_t sm = state_after(s, dist + 4, L1), psl = (x + L1) & pbm;
			uint32_t pr = pmatch + e->lp[0][ps & 3][L1 - 2] + tab_dist(e, dist, L1 < 6 ? L1 - 2 : 3)
				+ pr_bit(e, P_IS_MATCH + sm * 16 + psl, 0) + pr_literal(e, x + L1, sm, dist);
			const uint32_t sl = state_after(sm, LIT, 1), psn = (x + L1 + 1) & pbm;
			pr += REP0_AFTER_LIT(sl, psn, l2);
			while (n_end < j + L1 + 1 + l2) nd[++n_end].price = PRICE_INF;
			relax(&nd[j + L1 + 1 + l2], pr, dist + 4, L1, l2);
		}
#undef REP0_AFTER_LIT
		++j;
		if (j == n_end) {
			forced = j >= e->wmax;     /* cut by the node limit, not by convergence */
			break;
		}
	}
	/* backtrack from node j */
	uint32_t cnt = 0, t = j;
	while (t > 0) {
		cnt += nd[t].len2 ? (nd[t].len ? 3 : 2) : 1;
		t -= nd[t].len + (nd[t].len2 ? 1 + nd[t].len2 : 0);
	}
	e->q_count = cnt;
	e->q_head = 0;
	t = j;
	for (uint32_t i = cnt; i > 0; ) {
		if (nd[t].len2) {
			--i; e->q_back[i] = 0; e->q_len[i] = nd[t].len2;
			--i; e->q_back[i] = LIT; e->q_len[i] = 1;
		}
		if (nd[t].len) {
			--i; e->q_back[i] = nd[t].back; e->q_len[i] = nd[t].len;
		}
		t -= nd[t].len + (nd[t].len2 ? 1 + nd[t].len2 : 0);
	}
	if (forced && pos + j < e->span_end) {
		/* decisions close to a forced cut were made without lookahead: commit only the symbols
		 * that end at or before node j - WTAIL (at least one), re-parse the rest */
		uint32_t acc = 0, keep = 0;
		while (keep < cnt && acc + e->q_len[keep] <= j - WTAIL) {
			acc += e->q_len[keep];
			++keep;
		}
		e->q_count = keep ? keep : 1;
	}
	return next_cached;
}

/* ---- cost-balanced spans (OUR definition; the device: k_span_est / k_span_cut) -------------------
 * A span costs the wavefront that codes it one step per position the optimal parser has to visit, and a
 * state reset costs a few hundred bytes of model learning whatever the data.  Positions covered by a
 * match of nice_len bytes or more are not visited (optimum_window takes such a match at once), so highly
 * compressible data is both cheap to parse and small when coded: spans are cut by estimated parser work,
 * not by input bytes, which equalises the running time of the spans AND bounds what the resets cost (the
 * coded bytes per visited position vary little between kinds of data).  The estimate of a chunk of
 * ORC_EST_CHUNK positions is a walk over the (span independent) match lists: a position whose longest
 * match reaches nice_len costs ORC_EST_LONG units and the walk jumps over the match (it may run past the
 * chunk end), any other position costs one unit. */
#define ORC_EST_LONG 4u
#define ORC_CHUNK_EST (16000u * 128u)   /* two-phase coder: a chunk ends when the summed prices reach this (1/16 bit); round 6: 16,000
                                         * bytes instead of 56,000 -- four times the lanes for the device's lane-per-chunk range coder */
#ifndef ORC_PREROLL
#define ORC_PREROLL 2048u         /* two-phase: bytes in front of a piece that are parsed twice (the device: XZAMD_PREROLL) */
#define ORC_TOK_PER_BYTE 10u      /* two-phase coder: token budget per input byte of an encode span (XZAMD_TOK_PER_BYTE) */
static uint32_t orc_tok_per_byte; /* tests: a smaller budget (orc_set_tok_per_byte; the device: XZAMD_TEST_TOK_PER_BYTE) */
void orc_set_tok_per_byte(uint32_t v) { orc_tok_per_byte = v >= 1 && v < ORC_TOK_PER_BYTE ? v : 0; }
#define ORC_WARM 16384u           /* two-phase: bytes in front of the pre-roll that train the price model by a greedy walk (XZAMD_WARM) */
#endif

/* The same walk also makes a rough estimate of the coded size in bits (a greedy parse: at a symbol
 * boundary take the longest match when it is >= 3 bytes long, or 2 bytes at a distance < 128, for 14 bits +
 * the bit length of the distance; else a literal, 6 bits): spans of highly compressible data must not end
 * before they have produced span_bits of it, or the resets would dominate their size. */
static void est_chunk(enc *e, uint32_t c0, uint32_t c1, uint32_t *work, uint32_t *bits)
{
	uint32_t w = 0, b = 0, x = c0, gnext = c0;
	while (x < c1) {
		uint32_t len = 0, dist = 0;
		if (x > 0) {
			find_sn(e, x);
			len = e->m_longest;
			if (e->m_count) dist = e->m_dist[e->m_count - 1];
		}
		if (x >= gnext) {
			if (len >= 3 || (len == 2 && dist < 128)) {
				uint32_t bl = 0;
				while (dist >> bl) ++bl;
				b += 14 + bl;
				gnext = x + len;
			} else {
				b += 6;
				gnext = x + 1;
			}
		}
		if (len >= e->prm.nice_len) {
			w += ORC_EST_LONG;
			x += len;
		} else {
			w += 1;
			x += 1;
		}
	}
	*work = w;
	*bits = b;
}

static uint32_t plan_spans_ex(enc *e, uint32_t *chunk_cost, uint32_t *span_start, uint32_t span_cap,
		uint32_t *enc_start, uint32_t enc_cap, uint32_t *n_enc)
{
	const uint32_t n = e->n;
	const uint32_t m = (n + ORC_EST_CHUNK - 1) / ORC_EST_CHUNK;
	const uint32_t T = e->prm.span_cost;
	const uint32_t min_len = e->prm.span_size ? e->prm.span_size : 65536u;
	const int two = e->prm.enc_bits != 0;
	/* two-phase: the first ORC_SEED_LEN bytes are the seed piece, the plan covers the rest */
	const uint32_t seed_chunks = two && n > ORC_SEED_LEN ? ORC_SEED_LEN / ORC_EST_CHUNK : 0;
	uint32_t *cc = chunk_cost ? chunk_cost : (uint32_t *)malloc((size_t)(m + 1) * 8);
	uint32_t *cb = cc + m;      /* chunk_cost: m work estimates, then m bit estimates */
	uint64_t total = 0, total_bits = 0, all_bits = 0;
	for (uint32_t c = 0; c < m; ++c) {
		const uint32_t c0 = c * ORC_EST_CHUNK, c1 = n - c0 < ORC_EST_CHUNK ? n : c0 + ORC_EST_CHUNK;
		est_chunk(e, c0, c1, &cc[c], &cb[c]);
		all_bits += cb[c];
		if (c >= seed_chunks) { total += cc[c]; total_bits += cb[c]; }
	}
	/* k spans of equal estimated work: k = floor(total / T), but no more than the Block's estimated coded size
	 * allows at span_bits per span, at least one; threshold = ceil(total / k) */
	uint64_t k = total / T;
	if (e->prm.span_bits) {
		/* (round 5 let span_bits grow up to 4x on Blocks below one estimated bit per byte, because a piece start cost 150 ...
		 * 280 bytes whatever the data; with the snapshots of round 6 it no longer does -- zero pages with islands of words
		 * +1.76 vs +1.79 %, a tar of headers +0.69 vs +0.71 % -- and the rule cost config C5 a third of its throughput: fewer,
		 * longer pieces than the GPU has wave slots.  Removed.) */
		const uint64_t kb = total_bits / e->prm.span_bits;
		if (kb < k) k = kb;
	}
	if (k == 0) k = 1;
	const uint64_t Tb = (total + k - 1) / k;
	/* encode spans (two-phase): ke of about equal estimated coded size, closed at piece ends */
	uint64_t ke = two ? all_bits / e->prm.enc_bits : 1;
	if (ke > n / ORC_ENC_MIN_LEN) ke = n / ORC_ENC_MIN_LEN;
	if (ke == 0) ke = 1;
	const uint64_t Eb = (all_bits + ke - 1) / ke;
	uint32_t ns = 0, start = 0, ne = 0, estart = 0;
	uint64_t acc = 0, accb = 0;
	if (n && span_start && ns < span_cap) span_start[0] = 0;
	if (n && enc_start && ne < enc_cap) enc_start[0] = 0;
	if (n) { ns = 1; ne = 1; }
	for (uint32_t c = 0; c + 1 < m; ++c) {
		acc += cc[c];
		accb += cb[c];
		const uint64_t len = (uint64_t)(c + 1 - start) * ORC_EST_CHUNK;
		if (len >= ORC_SPAN_MAX || (acc >= Tb && len >= min_len) || c + 1 == seed_chunks) {
			acc = 0;
			start = c + 1;
			if (span_start && ns < span_cap) span_start[ns] = start * ORC_EST_CHUNK;
			++ns;
			if (two && accb >= Eb && (uint64_t)(c + 1 - estart) * ORC_EST_CHUNK >= ORC_ENC_MIN_LEN) {
				accb = 0;
				estart = c + 1;
				if (enc_start && ne < enc_cap) enc_start[ne] = estart * ORC_EST_CHUNK;
				++ne;
			}
		}
	}
	if (n_enc) *n_enc = ne;
	if (!chunk_cost) free(cc);
	return ns;
}

static uint32_t plan_spans(enc *e, uint32_t *chunk_cost, uint32_t *span_start, uint32_t span_cap)
{
	return plan_spans_ex(e, chunk_cost, span_start, span_cap, NULL, 0, NULL);
}

/* ---- per-span chunk loop: lzma_encoder.c:313-436 + lzma2_encoder.c:135-259 - */
static int put(uint8_t *out, uint64_t cap, uint64_t *pos, const uint8_t *src, uint64_t n)
{
	if (cap - *pos < n)
		return -1;
	memcpy(out + *pos, src, n);
	*pos += n;
	return 0;
}

static int encode_span_(enc *e, uint32_t start, uint32_t end, int first_in_block,
		uint8_t *out, uint64_t cap, uint64_t *opos);
static int encode_span(enc *e, uint32_t start, uint32_t end, int first_in_block,
		uint8_t *out, uint64_t cap, uint64_t *opos)
{
#ifdef ORC_STATS
	const uint64_t n0 = st_nodes, s0 = st_syms, o0 = *opos;
	const int r = encode_span_(e, start, end, first_in_block, out, cap, opos);
	fprintf(stderr, "SPAN %u %u nodes %llu syms %llu out %llu\n", start, end - start, (unsigned long long)(st_nodes - n0),
			(unsigned long long)(st_syms - s0), (unsigned long long)(*opos - o0));
	return r;
#else
	return encode_span_(e, start, end, first_in_block, out, cap, opos);
#endif
}

static int encode_span_(enc *e, uint32_t start, uint32_t end, int first_in_block,
		uint8_t *out, uint64_t cap, uint64_t *opos)
{
	int need_props = 1, need_dict_reset = first_in_block, need_state_reset = 0;
	uint32_t cur = start;       /* next position to encode */
	int cached = 0;             /* round for `cur` already done (read_ahead == 1) */
	e->span_end = end;
	e->q_count = e->q_head = 0;
	lzma_state_reset(e);
	int initialized = !first_in_block;

	while (cur < end) {
		/* SEQ_INIT (lzma2_encoder.c:143-162) */
		if (need_state_reset) {
			lzma_state_reset(e);
			e->q_count = e->q_head = 0;
		}
		const uint32_t chunk_start = cur;
		e->cpos = 0;

		if (!initialized) {
			/* encode_init (lzma_encoder.c:267-293): first byte of a
			 * dictionary-reset stream is a literal coded with the
			 * initial contexts. */
			rc_bit(e, &e->probs[P_IS_MATCH], 0);
			rc_tree(e, e->probs + P_LITERAL, 8, e->in[0]);
			trace_sym(e, 0, LIT, 1);
			cur = 1;
			initialized = 1;
		}

		for (;;) {
			/* lzma_encoder.c:346-351 with limit from lzma2_encoder.c:167-181 */
			if (cur - chunk_start >= (1u << 21) - MATCH_LEN_MAX
					|| e->cpos + e->cache_size + 4 >= 65536 - 4097)
				break;
			if (cur >= end)
				break;
			uint32_t back, len;
			if (e->prm.parser == 0) {
				cached = optimum_fast(e, cur, cached, &back, &len);
			} else {
				if (e->q_head == e->q_count)
					cached = optimum_window(e, cur, cached);
				back = e->q_back[e->q_head];
				len = e->q_len[e->q_head];
				++e->q_head;
			}
			enc_symbol(e, cur, back, len);
			ST_SYM();
			cur += len;
		}
		rc_flush(e);

		uint32_t usize = cur - chunk_start;
		const uint32_t csize = e->cpos;
		uint8_t hdr[6];
		if (csize >= usize) {
			/* lzma2_encoder.c:205-214: store raw.  Fast parser: incl. the lookahead byte
			 * (read_ahead).  Optimal parser (ours): pending symbols are dropped and the
			 * window is re-parsed after the state reset. */
			if (e->prm.parser == 0 && cached) {
				usize += 1;
				cur += 1;
			}
			cached = 0;
			e->q_count = e->q_head = 0;
			hdr[0] = need_dict_reset ? 1 : 2;
			need_dict_reset = 0;
			hdr[1] = (uint8_t)((usize - 1) >> 8);
			hdr[2] = (uint8_t)(usize - 1);
			need_state_reset = 1;
			if (put(out, cap, opos, hdr, 3) || put(out, cap, opos, e->in + chunk_start, usize))
				return -1;
			if (e->trace) ++e->trace->chunks_uncompressed;
			continue;
		}
		/* lzma2_header_lzma: lzma2_encoder.c:54-106 */
		uint32_t hl = 0;
		if (need_props)
			hdr[hl] = need_dict_reset ? 0x80 + (3 << 5) : 0x80 + (2 << 5);
		else
			hdr[hl] = need_state_reset ? 0x80 + (1 << 5) : 0x80;
		hdr[hl++] += (uint8_t)((usize - 1) >> 16);
		hdr[hl++] = (uint8_t)((usize - 1) >> 8);
		hdr[hl++] = (uint8_t)(usize - 1);
		hdr[hl++] = (uint8_t)((csize - 1) >> 8);
		hdr[hl++] = (uint8_t)(csize - 1);
		if (need_props)
			hdr[hl++] = (uint8_t)((e->prm.pb * 5 + e->prm.lp) * 9 + e->prm.lc);
		need_props = need_state_reset = need_dict_reset = 0;
		if (put(out, cap, opos, hdr, hl) || put(out, cap, opos, e->cbuf, csize))
			return -1;
		if (e->trace) ++e->trace->chunks_lzma;
	}
	return 0;
}

/* ---- two-phase mode (OUR definition; the device: k_parse_pieces / k_encode_syms) --------------------
 * The optimal parser is the expensive part and needs many independent units to fill the GPU; the range coder
 * is cheap but every model reset costs output bytes.  So the two are decoupled: phase 1 parses independent
 * PIECES (own adaptive model for the prices, reset at the piece start, nothing coded) and records the symbols
 * in a form that does not depend on the coder state -- (length, distance) or literal; phase 2 codes the
 * recorded symbols of a whole ENCODE SPAN with one continuous model, choosing rep / short rep / match from
 * its own rep distances.  A recorded one-byte rep0 whose distance is not the coder's rep0 becomes a literal
 * (the byte equality it relied on does not matter to a literal). */
static void record_literal(enc *e, uint32_t pos, int first_of_piece)
{
	const uint32_t cur = e->in[pos], prev = pos ? e->in[pos - 1] : 0;
	/* the first symbol of a piece never carries a match byte: what precedes it in the parser (the pre-roll) is not
	 * what the coder has coded there, so the coder looks the byte up itself */
	const uint32_t matched = e->state >= 7 && !first_of_piece;
	const uint32_t mb = matched ? e->in[pos - e->reps[0] - 1] : 0;
	e->sy_len[pos] = 0;
	e->sy_dist[pos] = cur | (prev << 8) | (mb << 16) | (matched << 24);
}

/* Price model, coder state and rep distances at a piece start (round 6: what the carried model walk over the first
 * iteration's records hands to the second iteration -- the device: k_model_syms in snapshot mode writes the piece's slot of
 * a.prior / a.lit and a.snap_sr) */
typedef struct { uint16_t probs[P_TOTAL_MAX]; uint32_t state, reps[4]; } snap;

/* First iteration: only the first part of a piece is parsed: an eighth of it (part_end_of), at least ORC_PART_MIN bytes */
#define ORC_PART_MIN 16384u

/* Where the first part of the piece [a, pe) ends: behind the chunk at which an eighth of the piece's ESTIMATED WORK (the plan's
 * per-4-KiB estimates, est_chunk) has been seen, at least ORC_PART_MIN bytes.  By work, not by bytes: the pieces of a plan hold
 * about equal work, but a piece of mixed density may hold most of it in its first bytes, and the partial iteration is one launch
 * whose length is its heaviest part (the device, config C5: 667 ms per batch with byte-based parts of a 1,943 ms full parse). */
static uint32_t part_end_of(const uint32_t *cc, uint32_t a, uint32_t pe)
{
	if (pe - a <= ORC_PART_MIN) return pe;
	const uint32_t c0 = a / ORC_EST_CHUNK, c1 = (pe + ORC_EST_CHUNK - 1) / ORC_EST_CHUNK;
	uint64_t total = 0, acc = 0;
	for (uint32_t c = c0; c < c1; ++c) total += cc[c];
	const uint64_t target = (total + 7) / 8;
	/* the chunk boundary NEAREST to where the eighth is reached (a part that always ended behind the chunk in which it is
	 * reached would be half a chunk too long on average: 10 % more work for the usual 16 ... 32 KiB part) */
	for (uint32_t c = c0; c < c1; ++c) {
		const uint64_t b64 = (uint64_t)c * ORC_EST_CHUNK;
		if (acc + cc[c] / 2 >= target && b64 >= (uint64_t)a + ORC_PART_MIN)
			return (uint32_t)b64;
		acc += cc[c];
	}
	return pe;
}

/* Parses [start, end) of the piece that starts at `start` (end = the piece end, or the end of its first part in the first
 * iteration; symbols never cross `end`).  prior != NULL: the price model starts from the Block's prior + warm-up walk +
 * pre-roll (first iteration); sn != NULL: from the snapshot (second iteration; no walk, no pre-roll); both NULL: flat (the
 * seed piece).  Returns the summed prices of the recorded symbols (1/16 bit, probabilities before their update). */
static uint64_t parse_piece(enc *e, uint32_t start, uint32_t end, int first_in_block, const uint16_t *prior, const snap *sn)
{
	uint32_t cur = start;
	int cached = 0;
	e->span_end = end;
	e->q_count = e->q_head = 0;
	lzma_state_reset(e);
	if (prior)
		memcpy(e->probs, prior, sizeof(e->probs));
	if (sn) {
		memcpy(e->probs, sn->probs, sizeof(e->probs));
		e->state = sn->state;
		memcpy(e->reps, sn->reps, sizeof(e->reps));
		prior = NULL;
	}
	e->rc_off = 1;
#if ORC_WARM
	if (prior && start > ORC_PREROLL + ORC_SEED_LEN) {
		/* warm-up (round 5): the prior is what the Block's FIRST 64 KiB teach, and 2 KiB of pre-roll cannot re-train a
		 * model for data that has drifted since (measured on float32 arrays: +4 ... +6 % vs liblzma, all of it piece
		 * starts).  So the ORC_WARM bytes in front of the pre-roll -- never any of the seed piece, which the prior has
		 * seen -- are walked greedily first: at every symbol boundary the LONGEST entry of the position's match-list
		 * record is taken when it is cheap by a fixed rule (14 + bit length of the distance < 6 bits per byte covered, or
		 * it repeats rep0), else a literal; every symbol adapts the model exactly as a coded one would, nothing is
		 * recorded.  No dynamic program: ~1 % of a piece's parser steps on literal-heavy data, less elsewhere. */
		orc_trace *const tr = e->trace;
		e->trace = NULL;
		const uint32_t w1 = start - ORC_PREROLL;
		uint32_t x = w1 - ORC_SEED_LEN > ORC_WARM ? w1 - ORC_WARM : ORC_SEED_LEN;
		/* Rep distances that a continuous parse would be carrying.  Periodic numeric data (a float64 sine: +5.4 % vs liblzma
		 * before this) lives on a far distance D -- the same phase one period earlier -- that is only affordable as a REP
		 * (three or four bytes per use): the continuous parse finds it once and keeps it in its rep stack for the rest of
		 * the Block; a piece that starts with the rep stack of a 2 KiB pre-roll never gets it back.  So the walk remembers
		 * the distances of the last four candidates it REJECTED as too expensive (length >= 3); one that comes up again is
		 * put into the oldest rep slot -- nothing is coded, no probability moves -- and from then on counts as a rep
		 * distance: taken when it matches, coded as the prices say. */
		uint32_t rej[4] = { 0, 0, 0, 0 }, nrej = 0;
		/* What the walk may NOT teach: how often a match is a rep, and which one (is_rep / is_rep0 / is_rep1 / is_rep2, 48
		 * probabilities).  Its rep decisions are the least like the optimal parser's -- it takes whatever sits at a rep
		 * distance, never a short rep, and the far matches it takes all say "not a rep" -- while the seed piece's values come
		 * from a real parse of the same Block: they are put back when the walk ends (lines of hex ids, where every match of
		 * the true parse is a rep at the line length: +2.86 -> +0.75 % vs liblzma; ELF metadata +1.77 -> +0.95 %). */
		uint16_t keep_rep[P_IS_REP0_LONG - P_IS_REP];
		memcpy(keep_rep, e->probs + P_IS_REP, sizeof(keep_rep));
		while (x < w1) {
			find_sn(e, x);
			uint32_t len = e->m_longest;
			const uint32_t dist = e->m_count ? e->m_dist[e->m_count - 1] : 0;
			if (len > w1 - x) len = w1 - x;
			uint32_t bl = 0;
			while (dist >> bl) ++bl;
			const int in_reps = dist == e->reps[0] || dist == e->reps[1] || dist == e->reps[2] || dist == e->reps[3];
			const int take = len >= 2 && (14 + bl < 6 * len || in_reps);
			if (!take && len >= 3) {
				int seen = 0;
				for (uint32_t i = 0; i < nrej; ++i)
					if (rej[i] == dist) seen = 1;
				if (seen)
					e->reps[3] = dist;
				else if (nrej < 4)
					rej[nrej++] = dist;
				else {
					rej[0] = rej[1]; rej[1] = rej[2]; rej[2] = rej[3]; rej[3] = dist;
				}
			}
			if (take) {
				/* HOW the taken symbol is coded -- as a rep when its distance is one of the four rep distances, or as a
				 * match all the same -- is decided by the current prices, as the optimal parser decides it: data made of
				 * fixed-size records (relocation tables: "distance 24" thousands of times in a row) has two
				 * self-reinforcing ways to code the same copy, and a walk that always said "rep0" trained the piece's
				 * model into the other equilibrium than the one the parser (and so the coder's continuous model) lives
				 * in: +6.2 % vs liblzma on an ELF .rela.dyn section, +2.1 % with this rule (round 5). */
				uint32_t back = dist + 4;
				int ri = -1;
				for (int i = 0; i < 4; ++i)
					if (dist == e->reps[i]) { ri = i; break; }
				if (ri >= 0) {
					const uint32_t ps = x & ((1u << e->prm.pb) - 1), st = e->state;
					uint32_t prep = pr_bit(e, P_IS_REP + st, 1);
					if (ri == 0)
						prep += pr_bit(e, P_IS_REP0 + st, 0) + pr_bit(e, P_IS_REP0_LONG + st * 16 + ps, 1);
					else {
						prep += pr_bit(e, P_IS_REP0 + st, 1);
						if (ri == 1) prep += pr_bit(e, P_IS_REP1 + st, 0);
						else prep += pr_bit(e, P_IS_REP1 + st, 1) + pr_bit(e, P_IS_REP2 + st, (uint32_t)ri - 2);
					}
					prep += pr_len(e, P_REP_LEN, ps, len);
					const uint32_t pm = pr_bit(e, P_IS_REP + st, 0) + pr_len(e, P_MATCH_LEN, ps, len)
							+ pr_dist(e, dist, len < 6 ? len - 2 : 3);
					if (prep <= pm) back = (uint32_t)ri;
				}
				enc_symbol(e, x, back, len);
				x += len;
			} else {
				enc_symbol(e, x, LIT, 1);
				x += 1;
			}
		}
		memcpy(e->probs + P_IS_REP, keep_rep, sizeof(keep_rep));
		e->trace = tr;
	}
#endif
#if ORC_PREROLL
	if (prior && start > ORC_PREROLL) {
		/* pre-roll: the last ORC_PREROLL bytes in front of the piece are parsed once more, from the prior, and thrown
		 * away: the piece proper then starts with a price model that has seen the local data, and with rep distances
		 * and a coder state like the ones the previous piece ends with */
		orc_trace *const tr = e->trace;
		e->trace = NULL;                 /* the discarded symbols are not part of the parse */
		cur = start - ORC_PREROLL;
		e->span_end = start;
		while (cur < start) {
			if (e->q_head == e->q_count)
				cached = optimum_window(e, cur, cached);
			const uint32_t back = e->q_back[e->q_head], len = e->q_len[e->q_head];
			++e->q_head;
			enc_symbol(e, cur, back, len);
			cur += len;
		}
		e->span_end = end;
		e->q_count = e->q_head = 0;
		cached = 0;
		e->trace = tr;
	}
#endif
	e->est = 0;
	if (first_in_block && cur < end) {
		record_literal(e, 0, 1);
		rc_bit(e, &e->probs[P_IS_MATCH], 0);
		rc_tree(e, e->probs + P_LITERAL, 8, e->in[0]);
		trace_sym(e, 0, LIT, 1);
		cur = 1;
	}
	while (cur < end) {
		if (e->q_head == e->q_count)
			cached = optimum_window(e, cur, cached);
		const uint32_t back = e->q_back[e->q_head], len = e->q_len[e->q_head];
		++e->q_head;
		if (back == LIT) record_literal(e, cur, cur == start);
		else {
			/* bit 15: the parser chose a MATCH (not a rep).  The coder then codes a match even when the distance is one of
			 * its rep distances -- legal LZMA, and what the single-phase encoder and liblzma do when the match path is the
			 * cheaper one under the adapted model (relocation tables: "match, distance 24" thousands of times in a row).
			 * Round 4's coder turned every such match into a rep: other probabilities than the parser had priced and a rep
			 * stack that drifted away from the parser's -- +8.5 % vs liblzma on an ELF .rela.dyn section (round 5). */
			e->sy_len[cur] = (uint16_t)(len | (back >= 4 ? 0x8000u : 0u));
			e->sy_dist[cur] = back < 4 ? e->reps[back] : back - 4;
		}
		enc_symbol(e, cur, back, len);
		ST_SYM();
		cur += len;
	}
	e->rc_off = 0;
	return e->est;
}

/* ---- carried encode spans (round 6; OUR definition) -------------------------------------------------------------------
 * The reference codes a Block with ONE continuous model (lzma/lzma_encoder.c:313-436; reset only per lzma2_encoder.c:63-77).
 * Rounds 4-5 reset the coder's model at every encode span (>= 512 KiB) so that the spans could be walked in parallel; on
 * data whose model learns slowly (image pixels: thousands of literal contexts in use) each reset costs 7 ... 20 KB.  Now the
 * model is CARRIED from span to span -- exactly, and the spans are still walked in parallel on the device:
 *   1. k_model_bounds walks every span with two values per probability, lo (from 31) and hi (from 2017): the update rule is
 *      monotone, so the true value stays between them whatever the span's start model is; once lo == hi the value is known.
 *      Until then the slot's bits are logged (<= ORC_LOG_CAP per span and slot).
 *   2. k_model_chain, per Block and slot, span after span: start value of span k + 1 = lo if the slot merged in span k, else
 *      the logged bits replayed on its start value in span k.
 *   3. k_model_syms walks every span from its TRUE start model (tokens, chunks).
 * The coder state and the rep distances at a span start come from a LOOK-BACK over the piece in front of it (lookback()):
 * the state depends on the last few symbols only; rep distances the look-back has not seen stay UNKNOWN (ORC_REP_UNKNOWN:
 * never equal to a recorded distance, so such a symbol is coded as a match -- legal LZMA, the decoder does not care).
 * What cannot be carried falls back to round 5's reset: a slot that has not merged after ORC_LOG_CAP logged bits, a span that
 * ran out of token budget, a look-back that does not name the state -- from there on every span of the Block starts with a
 * state reset.  The oracle walks the spans one after another with the true model and the same bounds beside it, so that it
 * fails exactly where the device does.
 *
 * Two iterations of the parse.  The price model a piece starts from decides which of several self-consistent codings its
 * parser settles into; a model trained on other data than the piece's own neighbourhood costs up to 10 % on image pixels
 * (round-5 review).  So: iteration 1 parses only the first part of every piece (part_end_of(): an eighth of its estimated work, >= 16 KiB) from the
 * seed's prior + warm-up walk + pre-roll as round 5 did; the carried model walk over THOSE records (a continuous model over a
 * sample of the Block) leaves a snapshot at every piece start; iteration 2 parses every piece in full from its snapshot. */
#define ORC_REP_UNKNOWN 0xFFFFFFFFu
#define ORC_RAW_MIN_LEN 32768u     /* shortest piece that is stored raw (the device's chunk table: XZAMD_RAW_MIN_LEN) */

static uint32_t state_type(uint32_t s, int type)   /* type 0 literal, 1 match, 2 rep, 3 short rep (lzma_common.h:118-141) */
{
	switch (type) {
	case 0: return s <= 3 ? 0 : s <= 9 ? s - 3 : s - 6;
	case 1: return s < 7 ? 7 : 10;
	case 2: return s < 7 ? 8 : 11;
	default: return s < 7 ? 9 : 11;
	}
}

/* Coder state and rep distances behind the recorded symbols of [a, b) (a = a piece start, b = the end of its parsed
 * part), without knowing either at a: twelve candidate states run side by side (they agree after a few symbols), unknown
 * rep distances stay ORC_REP_UNKNOWN.  Returns 0, or -1 when the candidates still differ at b. */
static int lookback(const enc *e, uint32_t a, uint32_t b, uint32_t *state, uint32_t reps[4])
{
	uint32_t st[12], r[4] = { ORC_REP_UNKNOWN, ORC_REP_UNKNOWN, ORC_REP_UNKNOWN, ORC_REP_UNKNOWN };
	for (uint32_t i = 0; i < 12; ++i) st[i] = i;
	uint32_t cur = a;
	while (cur < b) {
		uint32_t len = e->sy_len[cur];
		const uint32_t d = e->sy_dist[cur];
		const uint32_t as_match = len >> 15;
		len &= 0x7FFFu;
		int type, ri = -1;
		if (len == 0) { type = 0; len = 1; }
		else if (as_match) type = 1;
		else if (len == 1) type = d == r[0] ? 3 : 0;
		else {
			for (int i = 0; i < 4; ++i)
				if (d == r[i]) { ri = i; break; }
			type = ri >= 0 ? 2 : 1;
		}
		if (type == 1) { r[3] = r[2]; r[2] = r[1]; r[1] = r[0]; r[0] = d; }
		else if (type == 2 && ri > 0) {
			for (int i = ri; i > 0; --i) r[i] = r[i - 1];
			r[0] = d;
		}
		for (uint32_t i = 0; i < 12; ++i) st[i] = state_type(st[i], type);
		cur += len;
	}
	for (uint32_t i = 1; i < 12; ++i)
		if (st[i] != st[0]) return -1;
	*state = st[0];
	memcpy(reps, r, 16);
	return 0;
}

static void bnd_reset(enc *e, int known)
{
	for (uint32_t i = 0; i < P_TOTAL_MAX; ++i) {
		e->blo[i] = known ? 1024 : 31;
		e->bhi[i] = known ? 1024 : 2017;
		e->bcnt[i] = 0;
	}
}

static int bnd_failed(const enc *e)
{
	const uint32_t total = P_LITERAL + (0x300u << (e->prm.lc + e->prm.lp));
	for (uint32_t i = 0; i < total; ++i)
		if (e->blo[i] != e->bhi[i] && e->bcnt[i] >= LOG_CAP_NOW) return 1;
	return 0;
}

/* the coder's reading of the record at cur (the device: k_model_syms): back = LIT / rep index / distance + 4 */
static uint32_t record_back(enc *e, uint32_t cur, uint32_t *len_out)
{
	uint32_t len = e->sy_len[cur];
	const uint32_t d = e->sy_dist[cur];
	const uint32_t as_match = len >> 15;            /* the parser's choice: a match, whatever the coder's reps are */
	len &= 0x7FFFu;
	uint32_t back;
	e->lit_rec = 0;
	if (len == 0) { back = LIT; len = 1; e->lit_rec = d | (1u << 31); }
	else if (as_match) back = d + 4;
	else if (len == 1) { back = d == e->reps[0] ? 0 : LIT; if (back == LIT) e->lit_rec = 0; }
	else if (d == e->reps[0]) back = 0;
	else if (d == e->reps[1]) back = 1;
	else if (d == e->reps[2]) back = 2;
	else if (d == e->reps[3]) back = 3;
	else back = d + 4;
	*len_out = len;
	return back;
}

/* The carried model walk over the FIRST iteration's records (only the first part of every piece has them): one continuous
 * model per Block, carried across the encode spans as the coder's is, nothing coded; snaps[j] = model, state and rep
 * distances when the walk reaches piece j (j >= 1).  The price model is the parser's: it runs with the parser's pb. */
static void snapshot_walk(enc *e, const uint32_t *ps, uint32_t np, const uint32_t *es, uint32_t ne, const uint32_t *pend, snap *snaps)
{
	uint32_t ke = 0;
	int failed = 0;
	lzma_state_reset(e);
	e->rc_off = 1;
	e->bnd_on = 1;
	bnd_reset(e, 1);
	for (uint32_t j = 0; j < np; ++j) {
		const uint32_t a = ps[j], b = pend[j];            /* (pend[0] = the seed piece's end: it has all its records) */
		if (ke + 1 < ne && a == es[ke + 1]) {
			/* a span boundary: can the model be carried into this span? */
			if (!failed && bnd_failed(e)) failed = 1;
			++ke;
			uint32_t st = 0, rp[4] = { 0, 0, 0, 0 };
			if (!failed && lookback(e, ps[j - 1], pend[j - 1], &st, rp)) failed = 1;
			if (failed) lzma_state_reset(e);
			else { e->state = st; memcpy(e->reps, rp, 16); }
			e->rc_off = 1;
			bnd_reset(e, failed);
		}
		if (j >= 1) {
			memcpy(snaps[j].probs, e->probs, sizeof(e->probs));
			snaps[j].state = e->state;
			for (int i = 0; i < 4; ++i)      /* a parser's rep distances must be real ones: unknown -> 0, as after a reset */
				snaps[j].reps[i] = e->reps[i] == ORC_REP_UNKNOWN ? 0 : e->reps[i];
		}
		uint32_t cur = a;
		if (j == 0 && cur < b) {
			rc_bit(e, &e->probs[P_IS_MATCH], 0);
			rc_tree(e, e->probs + P_LITERAL, 8, e->in[0]);
			cur = 1;
		}
		while (cur < b) {
			uint32_t len;
			const uint32_t back = record_back(e, cur, &len);
			enc_symbol(e, cur, back, len);
			e->lit_rec = 0;
			cur += len;
		}
	}
	e->bnd_on = 0;
	e->rc_off = 0;
}

/* test hook (orc_two_phase_debug): what the stages leave */
static orc_two_phase_dbg *tp_dbg;

/* phase 1 of a whole Block: the seed piece; iteration 1 (the first part of every other piece, from the seed's model);
 * the carried model walk over its records; iteration 2 (every piece in full, from its snapshot).  raw[k] = 1: the parser's
 * own price of piece k says it does not shrink -- the coder stores it (encode_block_syms). */
static int parse_block(enc *e, const uint32_t *piece_start, uint32_t np, const uint32_t *enc_start, uint32_t ne, uint8_t *raw,
		const uint32_t *chunk_work)
{
	const uint32_t n = e->n;
	e->sy_len = (uint16_t *)calloc((size_t)n + 1, 2);
	e->sy_dist = (uint32_t *)calloc((size_t)n + 1, 4);
	uint16_t *prior = (uint16_t *)malloc(sizeof(e->probs));
	snap *snaps = (snap *)malloc(sizeof(snap) * (np ? np : 1));
	uint32_t *pend = (uint32_t *)malloc(4 * (size_t)(np ? np : 1));
	if (!e->sy_len || !e->sy_dist || !prior || !snaps || !pend) { free(prior); free(snaps); free(pend); return -3; }
	for (uint32_t k = 0; k < np; ++k) {
		const uint32_t a = piece_start[k], pe = k + 1 < np ? piece_start[k + 1] : n;
		pend[k] = k == 0 ? pe : part_end_of(chunk_work, a, pe);
	}
	orc_trace *const tr = e->trace;
	/* pb = 3, 4 (lzma/lzma_common.h:32-37): the price model of the parse pieces is the parser's alone -- the coder runs
	 * its own continuous model with the real pb (encode_block_syms) -- and takes a pb = 2 view of the positions: the device
	 * parser's per-window price tables hold four position states.  The recorded symbols are valid under any pb. */
	const uint32_t pb_coder = e->prm.pb;
	if (e->prm.pb > 2) e->prm.pb = 2;
	uint64_t price0 = 0;
	e->trace = NULL;                /* the trace is the second iteration's (and the seed piece's) */
	for (uint32_t k = 0; k < np; ++k) {
		const uint32_t a = piece_start[k];
		if (k == 0) e->trace = tr;
		const uint64_t pr = parse_piece(e, a, pend[k], k == 0, k == 0 ? NULL : prior, NULL);
		if (k == 0) { memcpy(prior, e->probs, sizeof(e->probs)); price0 = pr; e->trace = NULL; }
	}
	snapshot_walk(e, piece_start, np, enc_start, ne, pend, snaps);
	/* further partial iterations (part_iters > 1): the first part of every piece again, from the snapshots; what a piece
	 * learns there reaches every later piece through the carried walk -- a Block walks out of the regime its first 64 KiB
	 * suggest a few pieces further with every iteration */
	for (uint32_t it = 1; it < (e->prm.part_iters ? e->prm.part_iters : 1u); ++it) {
		for (uint32_t k = 1; k < np; ++k) {
			parse_piece(e, piece_start[k], pend[k], 0, NULL, &snaps[k]);
		}
		snapshot_walk(e, piece_start, np, enc_start, ne, pend, snaps);
	}
	if (tp_dbg && tp_dbg->snap_sr)
		for (uint32_t k = 1; k < np; ++k) {
			tp_dbg->snap_sr[5 * k] = snaps[k].state;
			memcpy(tp_dbg->snap_sr + 5 * k + 1, snaps[k].reps, 16);
			if (tp_dbg->snap_probs) memcpy(tp_dbg->snap_probs + (size_t)k * P_LITERAL, snaps[k].probs, 2 * P_LITERAL);
		}
	e->trace = tr;
	for (uint32_t k = 0; k < np; ++k) {
		const uint32_t a = piece_start[k], pe = k + 1 < np ? piece_start[k + 1] : n;
		const uint64_t pr = k == 0 ? price0 : parse_piece(e, a, pe, 0, NULL, &snaps[k]);
		if (raw) raw[k] = pe - a >= ORC_RAW_MIN_LEN && pr / 128u >= pe - a;
		if (tp_dbg && tp_dbg->price) tp_dbg->price[k] = pr;
	}
	e->prm.pb = pb_coder;
	e->trace = tr;
	free(prior);
	free(snaps);
	free(pend);
	return 0;
}

static int put_raw(enc *e, uint32_t a, uint32_t b, int *need_dict_reset, uint8_t *out, uint64_t cap, uint64_t *opos)
{
	/* lzma2_encoder.c:110-131: uncompressed chunks of at most 64 KiB */
	while (a < b) {
		const uint32_t usize = b - a < 65536u ? b - a : 65536u;
		uint8_t hdr[3];
		hdr[0] = *need_dict_reset ? 1 : 2;
		*need_dict_reset = 0;
		hdr[1] = (uint8_t)((usize - 1) >> 8);
		hdr[2] = (uint8_t)(usize - 1);
		if (put(out, cap, opos, hdr, 3) || put(out, cap, opos, e->in + a, usize))
			return -1;
		if (e->trace) ++e->trace->chunks_uncompressed;
		a += usize;
	}
	return 0;
}

/* phase 2 of a whole Block: the recorded symbols through the coder, encode span after encode span (the device walks them in
 * parallel: k_model_bounds / k_model_chain / k_model_syms / k_rc_chunks). */
static int encode_block_syms(enc *e, const uint32_t *ps, uint32_t np, const uint32_t *es, uint32_t ne, const uint8_t *raw,
		uint8_t *out, uint64_t cap, uint64_t *opos)
{
	const uint32_t n = e->n;
	int need_dict_reset = 1, failed = 0, r = 0;
	uint32_t j = 0;                                       /* piece that holds `cur` */
	lzma_state_reset(e);
	e->bnd_on = 1;
	/* Chunk rule of the two-phase coder (OUR definition; the device: k_model_syms / k_rc_chunks).  The range coder of a
	 * chunk runs apart from the model pass that decides where chunks end, so that pass cannot look at the coded size:
	 * it sums the PRICES of the decisions instead (the parser's table, 1/16 bit each, probabilities before their
	 * update; 16 per direct bit).  A chunk ends in front of the first symbol at which the sum has reached
	 * ORC_CHUNK_EST (16,000 bytes: the coded size stays far below the format's 65,536) or 2 MiB - 273 bytes of input.
	 * Round 6: what is stored raw is decided per PIECE, by the parser's price of it (raw[]: the walk that finds the
	 * bounds has no exact prices to decide by); a state reset follows a stored piece as it follows a stored chunk in the
	 * reference (lzma2_encoder.c:205-214). */
	e->est_on = 1;
	for (uint32_t k = 0; k < ne && !r; ++k) {
		const uint32_t start = es[k], end = k + 1 < ne ? es[k + 1] : n;
		int need_props = k == 0, need_state_reset = 0;
		while (j + 1 < np && ps[j + 1] <= start) ++j;     /* ps[j] == start */
		if (k > 0) {
			/* carried, or round 5's reset (control 0xC0: state reset + properties, as every span start had) */
			uint32_t st = 0, rp[4] = { 0, 0, 0, 0 };
			if (!failed && bnd_failed(e)) failed = 1;
			const int prev_raw = raw[j - 1];
			if (!failed && !prev_raw && lookback(e, ps[j - 1], ps[j], &st, rp)) failed = 1;
			if (failed) { lzma_state_reset(e); need_props = 1; }
			else if (prev_raw) need_state_reset = 1;       /* the model was reset behind the stored piece: nothing to carry */
			else { e->state = st; memcpy(e->reps, rp, 16); }
			bnd_reset(e, failed || prev_raw);
			if (tp_dbg && tp_dbg->carry) tp_dbg->carry[k] = failed ? 0 : prev_raw ? 2 : 1;
		} else
			bnd_reset(e, 1);
		/* Token budget (the device's token buffer: ORC_TOK_PER_BYTE per input byte of the span + 4096, 64 spare): when it
		 * runs out -- data made of far three-byte matches needs 32 ... 41 decisions per 3 bytes -- the chunk is closed where
		 * it stands and the REST of the span is stored as raw chunks of 64 KiB (lzma2_encoder.c:110-131); no later span of
		 * the Block is carried. */
		const uint64_t tok_cap = (uint64_t)(end - start) * (orc_tok_per_byte ? orc_tok_per_byte : ORC_TOK_PER_BYTE) + 4096u - 64u;
		e->ntok = 0;
		uint32_t cur = start;
		while (cur < end && !r) {
			if (cur == ps[j] && raw[j]) {
				/* a stored piece */
				const uint32_t pe = j + 1 < np ? ps[j + 1] : n;
				r = put_raw(e, cur, pe, &need_dict_reset, out, cap, opos);
				need_state_reset = 1;
				cur = pe;
				if (j + 1 < np) ++j;
				continue;
			}
			if (need_state_reset) {
				lzma_state_reset(e);
				bnd_reset(e, 1);
			}
			const uint32_t chunk_start = cur;
			e->cpos = 0;
			e->est = 0;
			if (k == 0 && cur == 0) {
				rc_bit(e, &e->probs[P_IS_MATCH], 0);
				rc_tree(e, e->probs + P_LITERAL, 8, e->in[0]);
				cur = 1;
			}
			int tok_full = 0;
			for (;;) {
				if (cur - chunk_start >= (1u << 21) - MATCH_LEN_MAX || e->est >= ORC_CHUNK_EST)
					break;
				if (cur >= end)
					break;
				if (j + 1 < np && cur == ps[j + 1]) {
					++j;
					if (raw[j]) break;
				}
				if (e->ntok + 64u > tok_cap) { tok_full = 1; break; }
				uint32_t len;
				const uint32_t back = record_back(e, cur, &len);
				enc_symbol(e, cur, back, len);
				e->lit_rec = 0;
				cur += len;
			}
			rc_flush(e);
			const uint32_t usize = cur - chunk_start, csize = e->cpos;
			if (usize) {
				uint8_t hdr[6];
				if (csize > 65536) { r = -4; break; }              /* cannot happen: see ORC_CHUNK_EST */
				uint32_t hl = 0;
				if (need_props)
					hdr[hl] = need_dict_reset ? 0x80 + (3 << 5) : 0x80 + (2 << 5);
				else
					hdr[hl] = need_state_reset ? 0x80 + (1 << 5) : 0x80;
				hdr[hl++] += (uint8_t)((usize - 1) >> 16);
				hdr[hl++] = (uint8_t)((usize - 1) >> 8);
				hdr[hl++] = (uint8_t)(usize - 1);
				hdr[hl++] = (uint8_t)((csize - 1) >> 8);
				hdr[hl++] = (uint8_t)(csize - 1);
				if (need_props)
					hdr[hl++] = (uint8_t)((e->prm.pb * 5 + e->prm.lp) * 9 + e->prm.lc);
				need_props = need_state_reset = need_dict_reset = 0;
				if (put(out, cap, opos, hdr, hl) || put(out, cap, opos, e->cbuf, csize)) { r = -1; break; }
				if (e->trace) ++e->trace->chunks_lzma;
			}
			if (tok_full) {
				r = put_raw(e, cur, end, &need_dict_reset, out, cap, opos);
				cur = end;
				failed = 1;
			}
		}
	}
	e->est_on = 0;
	e->bnd_on = 0;
	return r;
}

static uint32_t hash_mask_for(uint32_t dict_size, uint32_t hash_bytes)
{
	/* lz/lz_encoder.c:306-327 */
	uint32_t hs = dict_size - 1;
	hs |= hs >> 1; hs |= hs >> 2; hs |= hs >> 4; hs |= hs >> 8;
	hs >>= 1;
	hs |= 0xFFFF;
	if (hs > (1u << 24)) {
		if (hash_bytes == 3)
			hs = (1u << 24) - 1;
		else
			hs >>= 1;
	}
	return hs;
}

static void enc_free(enc *e)
{
	if (!e) return;
	free(e->prev2); free(e->prev3); free(e->son); free(e->prev4); free(e->prev8); free(e->prev16); free(e->prev24); free(e->prev32); free(e->sa); free(e->sa_rank); free(e->cbuf); free(e->nodes); free(e->sy_len); free(e->sy_dist); free(e);
}

static enc *enc_new(const uint8_t *in, uint32_t n, const orc_enc_params *p)
{
	enc *e = (enc *)calloc(1, sizeof(*e));
	if (!e) return NULL;
	e->in = in;
	e->n = n;
	e->prm = *p;
	if (e->prm.nice_len < e->prm.mf)
		e->prm.nice_len = e->prm.mf;     /* lzma_encoder.c:479-480 */
	e->depth = p->depth ? p->depth : 4 + e->prm.nice_len / 4; /* lz_encoder.c:359-365 */
	e->hash_mask = hash_mask_for(p->dict_size, p->mf);
	e->cyclic_size = p->dict_size + 1;       /* lz_encoder.c:254 */
	e->prev2 = (uint32_t *)calloc((size_t)n + 1, 4);
	e->prev3 = (uint32_t *)calloc((size_t)n + 1, 4);
	e->son = (uint32_t *)calloc((size_t)n + 1, 4);
	if (p->sa_window) {
		e->prev4 = (uint32_t *)calloc((size_t)n + 1, 4);
		e->prev8 = (uint32_t *)calloc((size_t)n + 1, 4);
		e->prev16 = (uint32_t *)calloc((size_t)n + 1, 4);
		e->prev24 = (uint32_t *)calloc((size_t)n + 1, 4);
		e->prev32 = (uint32_t *)calloc((size_t)n + 1, 4);
		e->sa = (uint32_t *)calloc((size_t)n + 1, 4);
		e->sa_rank = (uint32_t *)calloc((size_t)n + 1, 4);
	}
	e->cbuf = (uint8_t *)malloc(1 << 17);
	e->nodes = (node *)calloc(WMAX_CAP + MATCH_LEN_MAX + 2, sizeof(node));
	e->wmax = e->prm.nice_len > 128 ? WMAX_LONG : WMAX_STD;
	if (!e->prev2 || !e->prev3 || !e->son || !e->cbuf || !e->nodes
			|| (p->sa_window && (!e->prev4 || !e->prev8 || !e->prev16 || !e->prev24 || !e->prev32 || !e->sa || !e->sa_rank))
			|| build_links(e) || (p->sa_window && build_sa(e))) {
		enc_free(e);
		return NULL;
	}
#ifdef ORC_XPREV
	if (p->sa_window) {
		static const uint32_t ks[] = { ORC_XPREV };
		e->xprev_n = (int)(sizeof(ks) / sizeof(ks[0]));
		const uint32_t hb = 22, hm = (1u << hb) - 1;
		uint32_t *head = (uint32_t *)malloc((size_t)(hm + 1) * 4);
		for (int xi = 0; xi < e->xprev_n; ++xi) {
			const uint32_t K = ks[xi];
			e->xprev[xi] = (uint32_t *)calloc((size_t)n + 1, 4);
			memset(head, 0, (size_t)(hm + 1) * 4);
			for (uint32_t x = 0; x + K <= n; ++x) {
				uint64_t h = 1469598103934665603ull;
				for (uint32_t i = 0; i < K; ++i) h = (h ^ in[x + i]) * 1099511628211ull;
				const uint32_t b = (uint32_t)(h >> 20) & hm;
				const uint32_t q1 = head[b];
				if (q1 && memcmp(in + q1 - 1, in + x, K) == 0) e->xprev[xi][x] = x - (q1 - 1);
				head[b] = x + 1;
			}
		}
		free(head);
	}
#endif
	price_table_init(e);
	return e;
}

static int encode_block_impl(const uint8_t *in, uint32_t n, const orc_enc_params *p,
		uint8_t *out, uint64_t cap, uint64_t *out_size, orc_trace *trace, uint16_t *sym_len, uint32_t *sym_dist);

int orc_lzma2_encode_block(const uint8_t *in, uint32_t n, const orc_enc_params *p,
		uint8_t *out, uint64_t cap, uint64_t *out_size, orc_trace *trace)
{
	return encode_block_impl(in, n, p, out, cap, out_size, trace, NULL, NULL);
}

int orc_lzma2_encode_block_syms(const uint8_t *in, uint32_t n, const orc_enc_params *p,
		uint8_t *out, uint64_t cap, uint64_t *out_size, uint16_t *sym_len, uint32_t *sym_dist)
{
	if (!p->enc_bits || !sym_len || !sym_dist)
		return -2;
	return encode_block_impl(in, n, p, out, cap, out_size, NULL, sym_len, sym_dist);
}

static int encode_block_impl(const uint8_t *in, uint32_t n, const orc_enc_params *p,
		uint8_t *out, uint64_t cap, uint64_t *out_size, orc_trace *trace, uint16_t *sym_len, uint32_t *sym_dist)
{
	if ((p->mf != 3 && p->mf != 4) || p->lc + p->lp > 4 || p->pb > 4
			|| (p->sa_window && (p->mf != 4 || p->sa_window > SA_WMAX))
			|| p->parser > 1)
		return -2;
	enc *e = enc_new(in, n, p);
	if (!e) return -3;
	e->trace = trace;
	uint64_t opos = 0;
	int r = 0;
	if (p->enc_bits && p->span_cost && p->sa_window && p->parser == 1) {
		const uint32_t scap = n / 4096 + 2;
		uint32_t *ss = (uint32_t *)malloc((size_t)scap * 8), *es = ss + scap, ne = 0;
		uint32_t *cost = (uint32_t *)malloc(((size_t)n / ORC_EST_CHUNK + 2) * 8);
		const uint32_t np = plan_spans_ex(e, cost, ss, scap, es, scap, &ne);
		uint8_t *raw = (uint8_t *)calloc(np + 1, 1);
		r = raw && cost ? parse_block(e, ss, np, es, ne, raw, cost) : -3;
		free(cost);
		if (!r && sym_len && sym_dist) {
			memcpy(sym_len, e->sy_len, (size_t)n * 2);
			memcpy(sym_dist, e->sy_dist, (size_t)n * 4);
		}
		e->trace = NULL;            /* the trace is the parser's */
		if (!r && n) r = encode_block_syms(e, ss, np, es, ne, raw, out, cap, &opos);
		free(raw);
		e->trace = trace;
		free(ss);
	} else if (p->span_cost && p->sa_window) {
		const uint32_t scap = n / 4096 + 2;
		uint32_t *ss = (uint32_t *)malloc((size_t)scap * 4);
		const uint32_t ns = plan_spans(e, NULL, ss, scap);
		for (uint32_t k = 0; k < ns && !r; ++k)
			r = encode_span(e, ss[k], k + 1 < ns ? ss[k + 1] : n, k == 0, out, cap, &opos);
		free(ss);
	} else {
		const uint32_t span = p->span_size ? p->span_size : (n ? n : 1);
		for (uint32_t s = 0; s < n && !r; s += span) {
			const uint32_t end = n - s < span ? n : s + span;
			r = encode_span(e, s, end, s == 0, out, cap, &opos);
		}
	}
	if (!r) {
		/* end marker: lzma2_encoder.c:146-149 */
		if (opos < cap) out[opos++] = 0x00; else r = -1;
	}
	*out_size = opos;
	enc_free(e);
	return r;
}

int orc_mf_dump(const uint8_t *in, uint32_t n, const orc_enc_params *p,
		const uint32_t *pos_list, const uint32_t *end_list, uint32_t npos,
		uint32_t max_pairs, uint32_t *counts, uint32_t *pairs, uint32_t *longest)
{
	enc *e = enc_new(in, n, p);
	if (!e) return -3;
	static const uint32_t zero_reps[4] = { 0, 0, 0, 0 };
	for (uint32_t i = 0; i < npos; ++i) {
		const uint32_t pos = pos_list[i];
		if (pos >= n || pos == 0) { enc_free(e); return -4; }
		e->span_end = end_list ? end_list[i] : n;
		do_round(e, pos, zero_reps);
		counts[i] = e->m_count;
		longest[i] = e->m_longest;
		for (uint32_t k = 0; k < e->m_count && k < max_pairs; ++k) {
			pairs[(size_t)i * max_pairs * 2 + 2 * k] = e->m_len[k];
			pairs[(size_t)i * max_pairs * 2 + 2 * k + 1] = e->m_dist[k];
		}
	}
	enc_free(e);
	return 0;
}

uint32_t orc_span_plan(const uint8_t *in, uint32_t n, const orc_enc_params *p, uint32_t *chunk_cost,
		uint32_t *span_start, uint32_t span_cap)
{
	if (!p->sa_window || !p->span_cost)
		return 0;
	enc *e = enc_new(in, n, p);
	if (!e) return 0;
	const uint32_t ns = plan_spans(e, chunk_cost, span_start, span_cap);
	enc_free(e);
	return ns;
}

uint32_t orc_piece_plan(const uint8_t *in, uint32_t n, const orc_enc_params *p, uint32_t *chunk_cost,
		uint32_t *span_start, uint32_t span_cap, uint32_t *enc_start, uint32_t enc_cap, uint32_t *n_enc)
{
	if (!p->sa_window || !p->span_cost || !p->enc_bits)
		return 0;
	enc *e = enc_new(in, n, p);
	if (!e) return 0;
	const uint32_t ns = plan_spans_ex(e, chunk_cost, span_start, span_cap, enc_start, enc_cap, n_enc);
	enc_free(e);
	return ns;
}

int orc_two_phase_debug(const uint8_t *in, uint32_t n, const orc_enc_params *p, orc_two_phase_dbg *d)
{
	uint64_t osz = 0;
	uint8_t *out = (uint8_t *)malloc((size_t)n + n / 8 + 4096);
	if (!out) return -3;
	tp_dbg = d;
	const int r = encode_block_impl(in, n, p, out, (uint64_t)n + n / 8 + 4096, &osz, NULL, NULL, NULL);
	tp_dbg = NULL;
	free(out);
	return r;
}

int orc_parse_dump(const uint8_t *in, uint32_t n, const orc_enc_params *p, uint16_t *sym_len, uint32_t *sym_dist)
{
	if (!p->sa_window || !p->span_cost || !p->enc_bits || p->parser != 1)
		return -2;
	enc *e = enc_new(in, n, p);
	if (!e) return -3;
	const uint32_t scap = n / 4096 + 2;
	uint32_t *ss = (uint32_t *)malloc((size_t)scap * 8), *es = ss + scap, ne = 0;
	uint32_t *cost = (uint32_t *)malloc(((size_t)n / ORC_EST_CHUNK + 2) * 8);
	const uint32_t np = plan_spans_ex(e, cost, ss, scap, es, scap, &ne);
	const int r = parse_block(e, ss, np, es, ne, NULL, cost);
	free(cost);
	if (!r) {
		memcpy(sym_len, e->sy_len, (size_t)n * 2);
		memcpy(sym_dist, e->sy_dist, (size_t)n * 4);
	}
	free(ss);
	enc_free(e);
	return r;
}

/* Debug/test hooks for the device parity tests: the suffix order and the per-position match-list
 * records exactly as the GPU's k_find_sn / k_find_exact store them (packed format: 7 entries
 * length << 23 | distance-1 sorted by length, then count | len2a << 8 | len2b << 16; entries past
 * the count are reported as 0). */
int orc_sa_dump(const uint8_t *in, uint32_t n, uint32_t depth, uint32_t *sa_out, uint32_t *rank_out)
{
	orc_enc_params p;
	memset(&p, 0, sizeof(p));
	p.dict_size = 1u << 23; p.lc = 3; p.pb = 2; p.nice_len = 64; p.mf = 4; p.depth = 1; p.sa_window = 1; p.parser = 1;
	p.sa_depth = depth;
	enc *e = enc_new(in, n, &p);
	if (!e) return -3;
	memcpy(sa_out, e->sa, (size_t)n * 4);
	memcpy(rank_out, e->sa_rank, (size_t)n * 4);
	enc_free(e);
	return 0;
}

int orc_list_dump(const uint8_t *in, uint32_t n, const orc_enc_params *p, uint32_t *words)
{
	enc *e = enc_new(in, n, p);
	if (!e) return -3;
	static const uint32_t zero_reps[4] = { 0, 0, 0, 0 };
	const uint32_t span = p->span_size ? p->span_size : (n ? n : 1);
	for (uint32_t x = 0; x < n; ++x) {
		const uint32_t se = (x / span + 1) * span;
		e->span_end = se < n && se > x ? se : n;
		if (p->sa_window)
			e->span_end = n;        /* the suffix-neighbourhood finder's records are span independent */
		uint32_t *w = words + (size_t)x * 8;
		memset(w, 0, 32);
		if (x == 0 && 0) continue;
		e->m_count = 0; e->m_longest = 0; e->m_len2[0] = e->m_len2[1] = 0;
		if (x > 0 || p->sa_window) {
			/* position 0 has no earlier data; the exact finder's prefetch-free form needs x >= 1 */
			if (x > 0) do_round(e, x, zero_reps);
		}
		for (uint32_t k = 0; k < e->m_count; ++k) {
			const uint32_t len = k + 1 == e->m_count ? e->m_longest : e->m_len[k];
			w[k] = (len << 23) | e->m_dist[k];
		}
		w[7] = e->m_count | (e->m_len2[0] << 8) | (e->m_len2[1] << 16);
	}
	enc_free(e);
	return 0;
}

int orc_preset(uint32_t preset, orc_enc_params *p, uint32_t *mode_normal)
{
	/* lzma/lzma_encoder_presets.c:17-63 */
	const uint32_t level = preset & 0x1F;
	const uint32_t extreme = preset & 0x80000000u;
	if (level > 9 || (preset & ~(0x1Fu | 0x80000000u)))
		return 1;
	static const uint8_t dict_pow2[10] = { 18, 20, 21, 22, 22, 23, 23, 24, 25, 26 };
	memset(p, 0, sizeof(*p));
	p->dict_size = 1u << dict_pow2[level];
	p->lc = 3; p->lp = 0; p->pb = 2;
	uint32_t normal;
	if (level <= 3) {
		normal = 0;
		p->nice_len = level <= 1 ? 128 : 273;
		p->mf = level == 0 ? 3 : 4;
		static const uint8_t depths[4] = { 4, 8, 24, 48 };
		p->depth = depths[level];
	} else {
		normal = 1;
		p->mf = 0x14;       /* BT4: not restated by this oracle */
		p->nice_len = level == 4 ? 16 : (level == 5 ? 32 : 64);
		p->depth = 0;
	}
	if (extreme) {
		normal = 1;
		p->mf = 0x14;
		if (level == 3 || level == 5) {
			p->nice_len = 192;
			p->depth = 0;
		} else {
			p->nice_len = 273;
			p->depth = 512;
		}
	}
	if (mode_normal) *mode_normal = normal;
	return 0;
}
