/*
 * lzma_fast_enc.c -- TEST INFRASTRUCTURE ONLY (oracle).
 *
 * CPU restatement of the reference's fast-mode LZMA2 encode path for ONE
 * Block -- the code a worker thread of lzma_stream_encoder_mt runs for
 * presets 0-3:
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
 *   lzma/lzma2_encoder.c:54-259  chunk headers, uncompressed-chunk fallback
 *
 * It is written for a whole-Block-resident window (positions are absolute
 * offsets in the Block; no sliding, no cyclic buffers) and codes bits
 * directly instead of queueing them -- the same layout the HIP kernels use --
 * but every decision is the reference's, so with span_size == 0 the output is
 * byte-identical to liblzma's raw LZMA2 encoder (pinned in
 * tests/test_oracle_encoder.py).
 *
 * span_size != 0 is the GPU production mode: the Block is cut into spans that
 * are parsed and entropy-coded independently (LZMA state reset + properties at
 * each span start, control byte 0xC0; lzma2_decoder.c:84-111 makes that
 * legal) while the dictionary and the hash chains stay Block-global.  Matches
 * may not cross a span end.  This is our definition, not the reference's; the
 * HIP path must reproduce it bit-exactly.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

#define MATCH_LEN_MAX 273u
#define LIT 0xFFFFFFFFu

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
	const uint8_t *in;
	uint32_t n;
	orc_enc_params prm;
	uint32_t depth, hash_mask, cyclic_size;
	/* match finder: positions stored +1, 0 = empty
	 * (EMPTY_HASH_VALUE, lz_encoder_mf.c:82-85) */
	uint32_t *head2, *head3, *head4, *son;
	uint32_t mf_pos;            /* next position to insert */
	uint32_t span_end;          /* exclusive end for avail computations */
	/* matches of the last find */
	uint32_t m_len[MATCH_LEN_MAX + 1], m_dist[MATCH_LEN_MAX + 1];
	uint32_t m_count, m_longest;
	/* lzma state */
	uint16_t probs[P_TOTAL_MAX];
	uint32_t state, reps[4];
	uint32_t read_ahead;        /* 0/1: lookahead find already done */
	/* range coder */
	uint64_t low;
	uint64_t cache_size;
	uint32_t range;
	uint8_t cache;
	uint8_t *cbuf;              /* chunk payload buffer */
	uint32_t cpos;
	orc_trace *trace;
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

/* ---- match finder ------------------------------------------------------- */
static uint32_t cmplen(const uint8_t *a, const uint8_t *b, uint32_t len, uint32_t limit)
{
	/* common/memcmplen.h:47: first `len` bytes are known equal */
	while (len < limit && a[len] == b[len])
		++len;
	return len;
}

/* Insert position p into the tables without searching (hc3/hc4 skip). */
static void mf_insert(enc *e, uint32_t p)
{
	const uint32_t *T = crc_table0();
	const uint8_t *cur = e->in + p;
	const uint32_t hb = e->prm.mf;
	/* lz_encoder_mf.c:419-422 / :331-334: fewer than hash-bytes left in the
	 * BLOCK -> pending, never inserted.  (nice_len >= hash bytes always,
	 * lzma_encoder.c:479-480.) */
	if (e->n - p < hb)
		return;
	const uint32_t temp = T[cur[0]] ^ cur[1];
	const uint32_t h2 = temp & 0x3FF;
	uint32_t prev;
	if (hb == 3) {
		const uint32_t h = (temp ^ ((uint32_t)cur[2] << 8)) & e->hash_mask;
		prev = e->head3[h];
		e->head2[h2] = p + 1;
		e->head3[h] = p + 1;
	} else {
		const uint32_t h3 = (temp ^ ((uint32_t)cur[2] << 8)) & 0xFFFF;
		const uint32_t h = (temp ^ ((uint32_t)cur[2] << 8) ^ (T[cur[3]] << 5)) & e->hash_mask;
		prev = e->head4[h];
		e->head2[h2] = p + 1;
		e->head3[h3] = p + 1;
		e->head4[h] = p + 1;
	}
	e->son[p] = prev;
}

/* lzma_mf_find at position p == e->mf_pos. Fills m_*; advances mf_pos. */
static void mf_find(enc *e)
{
	const uint32_t *T = crc_table0();
	const uint32_t p = e->mf_pos;
	const uint8_t *cur = e->in + p;
	const uint32_t hb = e->prm.mf;
	const uint32_t nice = e->prm.nice_len;
	const uint32_t avail = e->span_end - p;
	uint32_t count = 0;
	e->m_count = 0;
	e->m_longest = 0;
	e->mf_pos = p + 1;

	/* header(): lz_encoder_mf.c:190-201 */
	uint32_t len_limit = avail;
	if (nice <= len_limit) {
		len_limit = nice;
	} else if (len_limit < hb) {
		/* pending: no matches. In span mode the position may still have
		 * to enter the Block-global tables. */
		mf_insert(e, p);
		return;
	}

	const uint32_t temp = T[cur[0]] ^ cur[1];
	const uint32_t h2 = temp & 0x3FF;
	uint32_t cur_match, len_best;
	int done = 0;
	if (hb == 3) {
		/* lz_encoder_mf.c:305-335 */
		const uint32_t h = (temp ^ ((uint32_t)cur[2] << 8)) & e->hash_mask;
		const uint32_t s2 = e->head2[h2];
		cur_match = e->head3[h];
		e->head2[h2] = p + 1;
		e->head3[h] = p + 1;
		len_best = 2;
		const uint32_t delta2 = p + 1 - s2;
		if (s2 != 0 && delta2 < e->cyclic_size && cur[-(int64_t)delta2] == cur[0]) {
			len_best = cmplen(cur - delta2, cur, len_best, len_limit);
			e->m_len[0] = len_best;
			e->m_dist[0] = delta2 - 1;
			count = 1;
			if (len_best == len_limit)
				done = 1;
		}
	} else {
		/* lz_encoder_mf.c:366-413 */
		const uint32_t h3 = (temp ^ ((uint32_t)cur[2] << 8)) & 0xFFFF;
		const uint32_t h = (temp ^ ((uint32_t)cur[2] << 8) ^ (T[cur[3]] << 5)) & e->hash_mask;
		const uint32_t s2 = e->head2[h2], s3 = e->head3[h3];
		cur_match = e->head4[h];
		e->head2[h2] = p + 1;
		e->head3[h3] = p + 1;
		e->head4[h] = p + 1;
		/* An empty slot gives delta = pos - 0 >= cyclic_size in the
		 * reference (positions start at cyclic_size, lz_encoder.c:395). */
		uint32_t delta2 = s2 ? p + 1 - s2 : 0xFFFFFFFFu;
		const uint32_t delta3 = s3 ? p + 1 - s3 : 0xFFFFFFFFu;
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
	e->son[p] = cur_match;

	if (!done) {
		/* hc_find_func: lz_encoder_mf.c:250-287 */
		uint32_t depth = e->depth;
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

static void mf_skip(enc *e, uint32_t amount)
{
	while (amount--) {
		mf_insert(e, e->mf_pos);
		++e->mf_pos;
	}
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

static inline void rc_bit(enc *e, uint16_t *prob, uint32_t bit)
{
	if (e->range < (1u << 24)) {
		rc_shift_low(e);
		e->range <<= 8;
	}
	uint32_t p = *prob;
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

static void enc_literal(enc *e, uint32_t pos)
{
	const uint8_t cur = e->in[pos];
	const uint32_t prev = pos ? e->in[pos - 1] : 0;
	const uint32_t lc = e->prm.lc;
	const uint32_t mask = (0x100u << e->prm.lp) - (0x100u >> lc);
	uint16_t *sub = e->probs + P_LITERAL + 3u * ((((pos << 8) + prev) & mask) << lc);
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

static void enc_symbol(enc *e, uint32_t pos, uint32_t back, uint32_t len)
{
	const uint32_t ps = pos & ((1u << e->prm.pb) - 1);
	uint16_t *P = e->probs;
	if (e->trace) {
		orc_trace *t = e->trace;
		if (t->sym && t->sym_count < t->sym_cap) {
			t->sym[t->sym_count].pos = pos;
			t->sym[t->sym_count].back = back;
			t->sym[t->sym_count].len = len;
		}
		++t->sym_count;
	}
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
		}
		return;
	}
	rc_bit(e, &P[P_IS_REP + e->state], 0);
	const uint32_t dist = back - 4;
	e->state = e->state < 7 ? 7 : 10;
	enc_length(e, P_MATCH_LEN, ps, len);
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
		}
	}
	e->reps[3] = e->reps[2];
	e->reps[2] = e->reps[1];
	e->reps[1] = e->reps[0];
	e->reps[0] = dist;
}

/* ---- parser: lzma/lzma_encoder_optimum_fast.c:20-169 ----------------------- */
#define change_pair(small, big) (((big) >> 7) > (small))

static void optimum_fast(enc *e, uint32_t pos, uint32_t *back_res, uint32_t *len_res)
{
	const uint32_t nice = e->prm.nice_len;
	uint32_t len_main, count;
	if (e->read_ahead == 0) {
		mf_find(e);
		++e->read_ahead;
	}
	len_main = e->m_longest;
	count = e->m_count;

	const uint8_t *buf = e->in + pos;
	const uint32_t rem = e->span_end - pos;
	const uint32_t buf_avail = rem < MATCH_LEN_MAX ? rem : MATCH_LEN_MAX;
	if (buf_avail < 2) {
		*back_res = LIT; *len_res = 1;
		return;
	}

	uint32_t rep_len = 0, rep_index = 0;
	for (uint32_t i = 0; i < 4; ++i) {
		const uint8_t *bb = buf - e->reps[i] - 1;
		if (buf[0] != bb[0] || buf[1] != bb[1])
			continue;
		const uint32_t len = cmplen(buf, bb, 2, buf_avail);
		if (len >= nice) {
			*back_res = i; *len_res = len;
			mf_skip(e, len - 1);
			e->read_ahead += len - 1;
			return;
		}
		if (len > rep_len) {
			rep_index = i;
			rep_len = len;
		}
	}

	if (len_main >= nice) {
		*back_res = e->m_dist[count - 1] + 4;
		*len_res = len_main;
		mf_skip(e, len_main - 1);
		e->read_ahead += len_main - 1;
		return;
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
			mf_skip(e, rep_len - 1);
			e->read_ahead += rep_len - 1;
			return;
		}
	}

	if (len_main < 2 || buf_avail <= 2) {
		*back_res = LIT; *len_res = 1;
		return;
	}

	/* lookahead: matches of the next byte */
	mf_find(e);
	++e->read_ahead;
	if (e->m_longest >= 2) {
		const uint32_t nl = e->m_longest;
		const uint32_t new_dist = e->m_dist[e->m_count - 1];
		if ((nl >= len_main && new_dist < back_main)
				|| (nl == len_main + 1 && !change_pair(back_main, new_dist))
				|| (nl > len_main + 1)
				|| (nl + 1 >= len_main && len_main >= 3 && change_pair(new_dist, back_main))) {
			*back_res = LIT; *len_res = 1;
			return;
		}
	}

	++buf;
	const uint32_t limit = len_main - 1 > 2 ? len_main - 1 : 2;
	for (uint32_t i = 0; i < 4; ++i) {
		if (memcmp(buf, buf - e->reps[i] - 1, limit) == 0) {
			*back_res = LIT; *len_res = 1;
			return;
		}
	}

	*back_res = back_main + 4;
	*len_res = len_main;
	mf_skip(e, len_main - 2);
	e->read_ahead += len_main - 2;
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

static int encode_span(enc *e, uint32_t start, uint32_t end, int first_in_block,
		uint8_t *out, uint64_t cap, uint64_t *opos)
{
	int need_props = 1, need_dict_reset = first_in_block, need_state_reset = 0;
	uint32_t cur = start;       /* == read_pos - read_ahead */
	e->span_end = end;
	e->read_ahead = 0;
	lzma_state_reset(e);
	/* catch the match finder up to the span start (Block-global tables) */
	if (e->mf_pos < start)
		mf_skip(e, start - e->mf_pos);
	int initialized = !first_in_block;

	while (cur < end) {
		/* SEQ_INIT (lzma2_encoder.c:143-162) */
		if (need_state_reset)
			lzma_state_reset(e);
		const uint32_t chunk_start = cur;
		e->cpos = 0;

		if (!initialized) {
			/* encode_init (lzma_encoder.c:267-293): first byte of a
			 * dictionary-reset stream is a literal coded with the
			 * initial contexts. */
			mf_skip(e, 1);
			rc_bit(e, &e->probs[P_IS_MATCH], 0);
			rc_tree(e, e->probs + P_LITERAL, 8, e->in[0]);
			if (e->trace) {
				orc_trace *t = e->trace;
				if (t->sym && t->sym_count < t->sym_cap) {
					t->sym[t->sym_count].pos = 0;
					t->sym[t->sym_count].back = LIT;
					t->sym[t->sym_count].len = 1;
				}
				++t->sym_count;
			}
			cur = 1;
			initialized = 1;
		}

		for (;;) {
			/* lzma_encoder.c:346-351 with limit from lzma2_encoder.c:167-181 */
			if (cur - chunk_start >= (1u << 21) - MATCH_LEN_MAX
					|| e->cpos + e->cache_size + 4 >= 65536 - 4097)
				break;
			/* lzma_encoder.c:354-360 (finishing) */
			if (cur >= end && e->read_ahead == 0)
				break;
			if (cur >= end)
				break; /* cannot happen: read_ahead implies cur < end */
			uint32_t back, len;
			optimum_fast(e, cur, &back, &len);
			enc_symbol(e, cur, back, len);
			e->read_ahead -= len;
			cur += len;
		}
		rc_flush(e);

		uint32_t usize = cur - chunk_start;
		const uint32_t csize = e->cpos;
		uint8_t hdr[6];
		if (csize >= usize) {
			/* lzma2_encoder.c:205-214: store raw, incl. the lookahead byte */
			usize += e->read_ahead;
			cur += e->read_ahead;
			e->read_ahead = 0;
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
	e->head2 = (uint32_t *)calloc(1024, 4);
	e->head3 = (uint32_t *)calloc(p->mf == 3 ? (size_t)e->hash_mask + 1 : 65536, 4);
	e->head4 = p->mf == 4 ? (uint32_t *)calloc((size_t)e->hash_mask + 1, 4) : NULL;
	e->son = (uint32_t *)calloc((size_t)n + 1, 4);
	e->cbuf = (uint8_t *)malloc(1 << 17);
	return e;
}

static void enc_free(enc *e)
{
	free(e->head2); free(e->head3); free(e->head4); free(e->son); free(e->cbuf); free(e);
}

int orc_lzma2_encode_block(const uint8_t *in, uint32_t n, const orc_enc_params *p,
		uint8_t *out, uint64_t cap, uint64_t *out_size, orc_trace *trace)
{
	if ((p->mf != 3 && p->mf != 4) || p->lc + p->lp > 4 || p->pb > 4)
		return -2;
	enc *e = enc_new(in, n, p);
	if (!e) return -3;
	e->trace = trace;
	uint64_t opos = 0;
	int r = 0;
	const uint32_t span = p->span_size ? p->span_size : (n ? n : 1);
	for (uint32_t s = 0; s < n && !r; s += span) {
		const uint32_t end = n - s < span ? n : s + span;
		r = encode_span(e, s, end, s == 0, out, cap, &opos);
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
	for (uint32_t i = 0; i < npos; ++i) {
		const uint32_t pos = pos_list[i];
		if (pos < e->mf_pos || pos >= n) { enc_free(e); return -4; }
		mf_skip(e, pos - e->mf_pos);
		e->span_end = end_list ? end_list[i] : n;
		mf_find(e);
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
