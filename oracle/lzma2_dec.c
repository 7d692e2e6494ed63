/*
 * lzma2_dec.c -- TEST INFRASTRUCTURE ONLY (oracle).
 *
 * CPU restatement of the reference DECODE path, used as the verifier for
 * everything the HIP encoder emits (bit-exact round trip) and to extract the
 * parse (symbol sequence) from any LZMA2 payload.  The decoder sources are the
 * authoritative definition of a valid chunk:
 *   lzma/lzma2_decoder.c:54-232  chunk grammar, reset rules
 *   lzma/lzma_decoder.c:286-700  symbol decoding, state machine
 *   rangecoder/range_decoder.h   rc init (5 bytes, first must be 0), normalize,
 *                                "code == 0 at chunk end" (:139-140)
 *   common/stream_decoder.c, block_decoder.c, index_hash.c: container checks
 * Restated as a whole-buffer decoder (the output buffer is the dictionary).
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

#define PROB_INIT 1024
#define NUM_STATES 12
#define POS_STATES_MAX 16
#define LITERAL_CODER_SIZE 0x300
#define LEN_LOW 8
#define LEN_MID 8
#define LEN_HIGH 256
#define DIST_SLOTS 64
#define DIST_MODEL_START 4
#define DIST_MODEL_END 14
#define FULL_DISTANCES 128
#define ALIGN_SIZE 16

typedef struct {
	uint16_t choice, choice2;
	uint16_t low[POS_STATES_MAX][LEN_LOW];
	uint16_t mid[POS_STATES_MAX][LEN_MID];
	uint16_t high[LEN_HIGH];
} len_dec;

typedef struct {
	uint16_t literal[LITERAL_CODER_SIZE << 4];
	uint16_t is_match[NUM_STATES][POS_STATES_MAX];
	uint16_t is_rep[NUM_STATES], is_rep0[NUM_STATES], is_rep1[NUM_STATES], is_rep2[NUM_STATES];
	uint16_t is_rep0_long[NUM_STATES][POS_STATES_MAX];
	uint16_t dist_slot[4][DIST_SLOTS];
	uint16_t dist_special[FULL_DISTANCES - DIST_MODEL_END];
	uint16_t dist_align[ALIGN_SIZE];
	len_dec match_len, rep_len;
	uint32_t state, rep[4];
	uint32_t lc, lp, pb;
	/* range decoder */
	const uint8_t *in;
	uint64_t in_pos, in_end;
	uint32_t range, code;
	int error;
} lzma_dec;

static void probs_reset(lzma_dec *d)
{
	/* lzma/lzma_decoder.c:1052-1100 (lzma_decoder_reset) */
	uint16_t *p = d->literal;
	for (size_t i = 0; i < sizeof(d->literal) / 2; ++i) p[i] = PROB_INIT;
#define FILL(a) do { uint16_t *q = (uint16_t *)(a); for (size_t i = 0; i < sizeof(a) / 2; ++i) q[i] = PROB_INIT; } while (0)
	FILL(d->is_match); FILL(d->is_rep); FILL(d->is_rep0); FILL(d->is_rep1); FILL(d->is_rep2);
	FILL(d->is_rep0_long); FILL(d->dist_slot); FILL(d->dist_special); FILL(d->dist_align);
	len_dec *l[2] = { &d->match_len, &d->rep_len };
	for (int k = 0; k < 2; ++k) {
		l[k]->choice = l[k]->choice2 = PROB_INIT;
		FILL(l[k]->low); FILL(l[k]->mid); FILL(l[k]->high);
	}
#undef FILL
	d->state = 0;
	d->rep[0] = d->rep[1] = d->rep[2] = d->rep[3] = 0;
}

static inline void rc_norm(lzma_dec *d)
{
	/* range_decoder.h rc_normalize: lazy, before each bit */
	if (d->range < (1u << 24)) {
		if (d->in_pos >= d->in_end) { d->error = 1; return; }
		d->range <<= 8;
		d->code = (d->code << 8) | d->in[d->in_pos++];
	}
}

static inline uint32_t rc_bit(lzma_dec *d, uint16_t *prob)
{
	rc_norm(d);
	uint32_t bound = (d->range >> 11) * *prob;
	if (d->code < bound) {
		d->range = bound;
		*prob += (2048 - *prob) >> 5;
		return 0;
	}
	d->range -= bound;
	d->code -= bound;
	*prob -= *prob >> 5;
	return 1;
}

static inline uint32_t rc_tree(lzma_dec *d, uint16_t *probs, uint32_t bits)
{
	uint32_t m = 1;
	for (uint32_t i = 0; i < bits; ++i)
		m = (m << 1) | rc_bit(d, &probs[m]);
	return m - (1u << bits);
}

static inline uint32_t rc_tree_rev(lzma_dec *d, uint16_t *probs, uint32_t bits)
{
	uint32_t m = 1, r = 0;
	for (uint32_t i = 0; i < bits; ++i) {
		uint32_t b = rc_bit(d, &probs[m]);
		m = (m << 1) | b;
		r |= b << i;
	}
	return r;
}

static inline uint32_t rc_direct(lzma_dec *d, uint32_t bits)
{
	uint32_t r = 0;
	while (bits--) {
		rc_norm(d);
		d->range >>= 1;
		d->code -= d->range;
		uint32_t t = 0u - (d->code >> 31);
		d->code += d->range & t;
		r = (r << 1) + (t + 1);
	}
	return r;
}

static uint32_t len_decode(lzma_dec *d, len_dec *l, uint32_t ps)
{
	if (!rc_bit(d, &l->choice))
		return 2 + rc_tree(d, l->low[ps], 3);
	if (!rc_bit(d, &l->choice2))
		return 2 + 8 + rc_tree(d, l->mid[ps], 3);
	return 2 + 16 + rc_tree(d, l->high, 8);
}

static void trace_add(orc_trace *t, uint32_t pos, uint32_t back, uint32_t len)
{
	if (!t) return;
	if (t->sym && t->sym_count < t->sym_cap) {
		t->sym[t->sym_count].pos = pos;
		t->sym[t->sym_count].back = back;
		t->sym[t->sym_count].len = len;
	}
	++t->sym_count;
}

/* Decode exactly usize bytes of one LZMA chunk into out[*opos ...]. */
static int lzma_chunk(lzma_dec *d, uint8_t *out, uint64_t dict_start, uint64_t *opos,
		uint64_t out_cap, uint32_t usize, uint32_t dict_size, orc_trace *tr)
{
	/* rc init: range_decoder.h:85-109 -- first byte must be 0x00, then 4
	 * code bytes. */
	if (d->in_end - d->in_pos < 5) return -10;
	if (d->in[d->in_pos] != 0) return -11;
	d->code = 0;
	for (int i = 1; i < 5; ++i)
		d->code = (d->code << 8) | d->in[d->in_pos + i];
	d->in_pos += 5;
	d->range = 0xFFFFFFFFu;

	uint64_t pos = *opos;
	const uint64_t end = pos + usize;
	if (end > out_cap) return -12;
	const uint32_t pb_mask = (1u << d->pb) - 1;
	const uint32_t lit_mask = (0x100u << d->lp) - (0x100u >> d->lc);

	while (pos < end) {
		const uint32_t upos = (uint32_t)(pos - dict_start); /* offset in Block */
		const uint32_t ps = upos & pb_mask;
		if (!rc_bit(d, &d->is_match[d->state][ps])) {
			/* literal: lzma_decoder.c:297-330 */
			uint32_t prev = pos > dict_start ? out[pos - 1] : 0;
			uint16_t *sub = d->literal + 3u * ((((upos << 8) + prev) & lit_mask) << d->lc);
			uint32_t sym = 1;
			if (d->state < 7) {
				do { sym = (sym << 1) | rc_bit(d, &sub[sym]); } while (sym < 0x100);
			} else {
				if (pos - dict_start <= d->rep[0]) return -13;
				uint32_t mb = out[pos - d->rep[0] - 1];
				uint32_t off = 0x100;
				do {
					mb <<= 1;
					uint32_t mbit = mb & off;
					uint32_t b = rc_bit(d, &sub[off + mbit + sym]);
					sym = (sym << 1) | b;
					off &= b ? mbit : ~mbit;
				} while (sym < 0x100);
			}
			out[pos] = (uint8_t)sym;
			trace_add(tr, upos, 0xFFFFFFFFu, 1);
			++pos;
			d->state = d->state <= 3 ? 0 : (d->state <= 9 ? d->state - 3 : d->state - 6);
			if (d->error) return -14;
			continue;
		}
		uint32_t len, back;
		if (!rc_bit(d, &d->is_rep[d->state])) {
			/* match: lzma_decoder.c:340-480 */
			len = len_decode(d, &d->match_len, ps);
			uint32_t ds = len < 6 ? len - 2 : 3;
			uint32_t slot = rc_tree(d, d->dist_slot[ds], 6);
			uint32_t dist;
			if (slot < DIST_MODEL_START) {
				dist = slot;
			} else {
				uint32_t fb = (slot >> 1) - 1;
				dist = (2 | (slot & 1)) << fb;
				if (slot < DIST_MODEL_END) {
					dist += rc_tree_rev(d, d->dist_special + dist - slot - 1, fb);
				} else {
					dist += rc_direct(d, fb - 4) << 4;
					dist += rc_tree_rev(d, d->dist_align, 4);
				}
			}
			if (dist == 0xFFFFFFFFu) return -15; /* EOPM not allowed in LZMA2 */
			d->rep[3] = d->rep[2]; d->rep[2] = d->rep[1]; d->rep[1] = d->rep[0];
			d->rep[0] = dist;
			d->state = d->state < 7 ? 7 : 10;
			back = dist + 4;
		} else {
			if (pos == dict_start) return -16;
			if (!rc_bit(d, &d->is_rep0[d->state])) {
				if (!rc_bit(d, &d->is_rep0_long[d->state][ps])) {
					/* short rep */
					if (pos - dict_start <= d->rep[0]) return -17;
					out[pos] = out[pos - d->rep[0] - 1];
					trace_add(tr, upos, 0, 1);
					++pos;
					d->state = d->state < 7 ? 9 : 11;
					if (d->error) return -14;
					continue;
				}
				back = 0;
			} else {
				uint32_t dist;
				if (!rc_bit(d, &d->is_rep1[d->state])) {
					dist = d->rep[1];
					back = 1;
				} else {
					if (!rc_bit(d, &d->is_rep2[d->state])) {
						dist = d->rep[2];
						back = 2;
					} else {
						dist = d->rep[3];
						d->rep[3] = d->rep[2];
						back = 3;
					}
					d->rep[2] = d->rep[1];
				}
				d->rep[1] = d->rep[0];
				d->rep[0] = dist;
			}
			len = len_decode(d, &d->rep_len, ps);
			d->state = d->state < 7 ? 8 : 11;
		}
		if (d->error) return -14;
		/* dict_is_distance_valid (lz/lz_decoder.h:195-198) + dict_size */
		if (pos - dict_start <= d->rep[0] || d->rep[0] >= dict_size) return -18;
		if (pos + len > end) return -19; /* chunk must end on a symbol boundary */
		trace_add(tr, upos, back, len);
		const uint64_t src = pos - d->rep[0] - 1;
		for (uint32_t i = 0; i < len; ++i)
			out[pos + i] = out[src + i];
		pos += len;
	}
	*opos = pos;
	/* lzma_decoder.c:669-675 + range_decoder.h:139-140: the range decoder
	 * must be "finished" (code == 0) when the chunk's uncompressed size is
	 * reached.  A pending normalization is applied first (rc_normalize at
	 * the top of the next symbol in the reference's loop). */
	rc_norm(d);
	if (d->error) return -14;
	if (d->code != 0) return -20;
	return 0;
}

static int lzma2_decode_ex(const uint8_t *in, uint64_t in_size, uint32_t dict_size,
		uint8_t *out, uint64_t out_cap, uint64_t *out_size, orc_trace *tr,
		uint64_t *in_used)
{
	lzma_dec *d = (lzma_dec *)calloc(1, sizeof(*d));
	if (!d) return -1;
	int need_dict_reset = 1, need_props = 1;
	uint64_t ip = 0, op = 0, dict_start = 0;
	int ret = -2;
	d->in = in;
	for (;;) {
		if (ip >= in_size) { ret = -3; break; }
		uint8_t control = in[ip++];
		if (control == 0x00) { ret = (in_used || ip == in_size) ? 0 : -4; break; }
		/* lzma2_decoder.c:67-127 */
		if (control >= 0xE0 || control == 0x01) {
			need_props = 1;
			need_dict_reset = 0;
			if (tr) ++tr->dict_resets;
			/* lzma2_decoder.c:121-127 -> lz_decoder dict_reset: history is
			 * dropped and the position counter restarts */
			dict_start = op;
		} else if (need_dict_reset) { ret = -6; break; }
		if (control >= 0x80) {
			if (in_size - ip < 4) { ret = -3; break; }
			uint32_t usize = ((uint32_t)(control & 0x1F) << 16) + ((uint32_t)in[ip] << 8) + in[ip + 1] + 1;
			uint32_t csize = ((uint32_t)in[ip + 2] << 8) + in[ip + 3] + 1;
			ip += 4;
			if (control >= 0xC0) {
				if (ip >= in_size) { ret = -3; break; }
				uint8_t props = in[ip++];
				if (props > (4 * 5 + 4) * 9 + 8) { ret = -7; break; }
				d->pb = props / (9 * 5);
				props -= (uint8_t)(d->pb * 9 * 5);
				d->lp = props / 9;
				d->lc = props - d->lp * 9;
				if (d->lc + d->lp > 4) { ret = -7; break; }
				need_props = 0;
				probs_reset(d);
				if (tr) { ++tr->prop_resets; ++tr->state_resets; }
			} else if (need_props) { ret = -8; break; }
			else if (control >= 0xA0) {
				probs_reset(d);
				if (tr) ++tr->state_resets;
			}
			if (in_size - ip < csize) { ret = -3; break; }
			d->in_pos = ip;
			d->in_end = ip + csize;
			d->error = 0;
			int r = lzma_chunk(d, out, dict_start, &op, out_cap, usize, dict_size, tr);
			if (r) { ret = r; break; }
			if (d->in_pos != d->in_end) { ret = -21; break; }
			ip += csize;
			if (tr) ++tr->chunks_lzma;
		} else {
			if (control > 2) { ret = -9; break; }
			if (in_size - ip < 2) { ret = -3; break; }
			uint32_t usize = ((uint32_t)in[ip] << 8) + in[ip + 1] + 1;
			ip += 2;
			if (in_size - ip < usize || out_cap - op < usize) { ret = -3; break; }
			memcpy(out + op, in + ip, usize);
			ip += usize;
			op += usize;
			if (tr) ++tr->chunks_uncompressed;
		}
	}
	*out_size = op;
	if (in_used) *in_used = ip;
	free(d);
	return ret;
}

int orc_lzma2_decode(const uint8_t *in, uint64_t in_size, uint32_t dict_size,
		uint8_t *out, uint64_t out_cap, uint64_t *out_size, orc_trace *tr)
{
	return lzma2_decode_ex(in, in_size, dict_size, out, out_cap, out_size, tr, NULL);
}

/* ---- .xz Stream decode with every integrity check ------------------------ */
static int vli_get(const uint8_t *in, uint64_t n, uint64_t *pos, uint64_t *v)
{
	*v = 0;
	for (int i = 0; i < 9; ++i) {
		if (*pos >= n) return -1;
		uint8_t b = in[(*pos)++];
		*v |= (uint64_t)(b & 0x7F) << (7 * i);
		if (!(b & 0x80)) {
			if (b == 0 && i > 0) return -1;
			return 0;
		}
	}
	return -1;
}

static uint32_t get32le(const uint8_t *p)
{
	return p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

int orc_xz_decode(const uint8_t *in, uint64_t n, uint8_t *out, uint64_t out_cap,
		uint64_t *out_size, uint64_t *nblocks_out)
{
	static const uint8_t magic[6] = { 0xFD, '7', 'z', 'X', 'Z', 0x00 };
	*out_size = 0;
	if (n < 32 || memcmp(in, magic, 6)) return -100;
	if (in[6] != 0 || in[7] > 15) return -101;
	if (get32le(in + 8) != orc_crc32(in + 6, 2, 0)) return -102;
	const int check = in[7];
	const uint32_t cs = orc_check_size(check);
	uint64_t pos = 12, op = 0, nb = 0;
	uint64_t cap_rec = 1024;
	uint64_t *rec = (uint64_t *)malloc(cap_rec * 16);
	int ret = 0;
	while (pos < n && in[pos] != 0x00) {
		uint32_t hs = ((uint32_t)in[pos] + 1) * 4;
		if (pos + hs > n) { ret = -103; break; }
		if (get32le(in + pos + hs - 4) != orc_crc32(in + pos, hs - 4, 0)) { ret = -104; break; }
		uint8_t flags = in[pos + 1];
		if (flags & 0x3C) { ret = -105; break; }
		uint64_t p = pos + 2, csize = UINT64_MAX, usize = UINT64_MAX;
		if (flags & 0x40) if (vli_get(in, pos + hs - 4, &p, &csize)) { ret = -106; break; }
		if (flags & 0x80) if (vli_get(in, pos + hs - 4, &p, &usize)) { ret = -106; break; }
		uint32_t nf = (flags & 3) + 1;
		uint32_t dict_size = 0;
		int ok = 1;
		for (uint32_t f = 0; f < nf; ++f) {
			uint64_t id, psz;
			if (vli_get(in, pos + hs - 4, &p, &id) || vli_get(in, pos + hs - 4, &p, &psz)) { ok = 0; break; }
			if (id == 0x21 && psz == 1 && f == nf - 1) {
				uint8_t b = in[p];
				if (b > 40) { ok = 0; break; }
				dict_size = b == 40 ? 0xFFFFFFFFu : ((2u | (b & 1)) << (b / 2 + 11));
			} else { ok = 0; break; } /* verifier handles plain LZMA2 chains only */
			p += psz;
		}
		if (!ok) { ret = -107; break; }
		while (p < pos + hs - 4) if (in[p++] != 0) { ok = 0; break; }
		if (!ok) { ret = -108; break; }
		pos += hs;
		uint64_t produced = 0;
		if (csize == UINT64_MAX) {
			/* single-threaded encoder output (stream_encoder.c:68-69): sizes
			 * absent, the payload ends at its end marker */
			uint64_t used = 0;
			int r = lzma2_decode_ex(in + pos, n - pos, dict_size, out + op, out_cap - op, &produced, NULL, &used);
			if (r) { ret = r; break; }
			csize = used;
		} else {
			if (pos + csize > n) { ret = -110; break; }
			int r = orc_lzma2_decode(in + pos, csize, dict_size, out + op, out_cap - op, &produced, NULL);
			if (r) { ret = r; break; }
		}
		if (usize != UINT64_MAX && usize != produced) { ret = -111; break; }
		pos += csize;
		while (pos & 3) { if (pos >= n || in[pos] != 0) { ok = 0; break; } ++pos; }
		if (!ok) { ret = -112; break; }
		if (pos + cs > n) { ret = -113; break; }
		if (check == ORC_CHECK_CRC64) {
			uint64_t c = orc_crc64(out + op, produced, 0);
			if (get32le(in + pos) != (uint32_t)c || get32le(in + pos + 4) != (uint32_t)(c >> 32)) { ret = -114; break; }
		} else if (check == ORC_CHECK_CRC32) {
			if (get32le(in + pos) != orc_crc32(out + op, produced, 0)) { ret = -114; break; }
		}
		pos += cs;
		if (nb == cap_rec) { cap_rec *= 2; rec = (uint64_t *)realloc(rec, cap_rec * 16); }
		rec[2 * nb] = hs + csize + cs;
		rec[2 * nb + 1] = produced;
		++nb;
		op += produced;
	}
	if (!ret) {
		/* Index */
		uint64_t istart = pos, cnt;
		++pos;
		if (vli_get(in, n, &pos, &cnt) || cnt != nb) ret = -120;
		for (uint64_t i = 0; !ret && i < nb; ++i) {
			uint64_t a, b;
			if (vli_get(in, n, &pos, &a) || vli_get(in, n, &pos, &b)) ret = -121;
			else if (a != rec[2 * i] || b != rec[2 * i + 1]) ret = -122;
		}
		while (!ret && ((pos - istart) & 3)) { if (pos >= n || in[pos] != 0) ret = -123; ++pos; }
		if (!ret && (pos + 4 > n || get32le(in + pos) != orc_crc32(in + istart, pos - istart, 0))) ret = -124;
		pos += 4;
		uint64_t isize = pos - istart;
		if (!ret) {
			if (pos + 12 != n) ret = -125;
			else if (get32le(in + pos) != orc_crc32(in + pos + 4, 6, 0)) ret = -126;
			else if (((uint64_t)get32le(in + pos + 4) + 1) * 4 != isize) ret = -127;
			else if (in[pos + 8] != 0 || in[pos + 9] != check) ret = -128;
			else if (in[pos + 10] != 'Y' || in[pos + 11] != 'Z') ret = -129;
		}
	}
	free(rec);
	*out_size = op;
	if (nblocks_out) *nblocks_out = nb;
	return ret;
}
