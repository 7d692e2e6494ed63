/* kernels_api.h -- internal ABI between the plain-C host layer (xzamd_host.c) and the HIP
 * translation unit (lzma_kernels.hip).  Plain C types only; not part of the public C ABI
 * (that is include/xz_amd.h). */
#ifndef XZAMD_KERNELS_API_H
#define XZAMD_KERNELS_API_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Arguments of the span encoder (one wavefront per span). All offsets are byte offsets into
 * the batch input `in` (n < 2^31). */
typedef struct {
	const uint8_t *in;
	const uint32_t *rank;        /* rank[pos]   = slot of pos in the sorted bucket order */
	const uint32_t *sorted_pos;  /* slot -> pos | first_of_bucket << 31 */
	const uint32_t *prev2;       /* distance to previous position with equal hash2, 0 = none */
	const uint32_t *prev3;       /* same for hash3 (HC4 only) */
	uint8_t *scratch;            /* span slot s = block * max_spb + k covering [start, end) writes at
	                                scratch + align16(start + start / 8) + s * XZAMD_SPAN_SLACK, at most end - start +
	                                (end - start) / 8 + 4096 bytes */
	/* span plan: Block b has span_cnt[b] spans, its k-th one is span_tab[2 * (b * max_spb + k)] (first byte) ..
	 * span_tab[2 * (b * max_spb + k) + 1] (end, exclusive), offsets into the batch.  Written by the host (spans of
	 * span_size bytes) or by xzk_span_plan (cost-balanced spans). */
	const uint32_t *span_tab;
	const uint32_t *span_cnt;
	uint32_t max_spb;            /* span slots per Block */
	const uint32_t *order;       /* optional: workgroup i codes span slot order[i] (longest estimated work first, so that a
	                                launch ends with its short spans); NULL = slot i */
	uint32_t *span_bytes;        /* out: bytes produced per span slot */
	uint32_t *lit;               /* literal-coder probabilities: 6144 x u32 per span */
	/* per-position match lists (parser != 0), written by xzk_find_matches: 8 x u32 per position =
	 * 7 entries sorted by length + trailer (count | len2 of the longest << 8 | len2 of the second << 16) */
	const uint16_t *mlen;        /* 8 x u16 per position, lengths of the entries (list_packed == 0 only) */
	const uint32_t *mdist;
	uint32_t list_packed;        /* 1: entries are length << 23 | distance-1 and mlen is unused (dict_size <= 8 MiB) */
	uint16_t *mtop;              /* per position, written by k_find_sn for the span plan: length of the longest entry
	                                (incl. the > nice_len extension) | bit length of its zero-based distance << 9; 0 = no match */
	uint32_t *trace;             /* optional debug: 4 x u32 per symbol (span,pos,back,len) */
	uint32_t *trace_count;
	uint32_t trace_cap;
	uint32_t *err;               /* 8 x u32: [0] != 0 -> a span hit an internal consistency check */
	uint32_t n;
	uint32_t block_size;
	uint32_t span_size;          /* k_find_exact only (its lists end at the span end): spans of span_size bytes */
	uint32_t dict_size, nice_len, depth, hash_bytes;
	uint32_t lc, lp, pb;
	uint32_t sa_window;          /* suffix-neighbourhood finder: slots examined on either side (0 = exact HC3/HC4 finder) */
	uint32_t parser;             /* 0 = optimum_fast, 1 = windowed optimal parser */
	/* Two-phase mode (oracle: parse_piece / encode_syms): the spans of the plan are parse PIECES (xzk_parse_pieces: the
	 * optimal parser with an adaptive price model that codes nothing; the seed piece = slot 0 of every Block starts flat
	 * and leaves the prior of the others), the recorded symbols are range-coded per ENCODE SPAN (xzk_encode_syms). */
	uint16_t *sym_len;           /* per position, valid at symbol starts: 0 = literal, else the match / rep length (1 = short rep);
	                                bit 15: the parser chose a MATCH (the coder codes a match even at one of its rep distances) */
	uint32_t *sym_dist;          /* zero-based distance; literal: byte | previous byte << 8 | match byte << 16 | (parser state >= 7) << 24 */
	uint32_t *prior;             /* XZAMD_PRIOR_WORDS x u32 per PIECE slot (b * max_spb + k): the non-literal probabilities of the
	                                piece's price model (the literal coders: `lit`, same slots); slot 0 of a Block = its seed
	                                piece, whose model is the prior the other pieces of the Block copy */
	const uint32_t *enc_tab;     /* encode spans: (first byte, end) per slot b * max_esb + j, offsets into the batch */
	const uint32_t *enc_cnt;     /* encode spans per Block */
	uint32_t max_esb;            /* encode span slots per Block */
	uint32_t enc_bits;           /* != 0: two-phase plan (seed cut + encode spans of about enc_bits estimated bits) */
	/* Coder of the two-phase mode (xzk_encode_syms): the model pass of an encode span writes one 16-bit TOKEN per binary
	 * decision (probability before its update | bit << 12, or bit << 12 | 1 << 15 for a direct bit) and one xzamd_chunk
	 * per LZMA2 chunk; the range coder then runs one LANE per chunk. */
	uint16_t *tok;               /* tokens of encode-span slot s (first byte st): from XZAMD_TOK_BASE(st, s) on */
	struct xzamd_chunk *chunks;  /* chunk slots of encode-span slot s: from XZAMD_CHUNK_BASE(st, s) on; usize 0 = unused */
	uint32_t tok_limit;          /* 0 = XZAMD_TOK_PER_BYTE; tests (XZAMD_TEST_TOK_PER_BYTE): a smaller token budget per input byte,
	                                to reach the "out of tokens: the rest of the span goes out raw" path on ordinary data */
	/* Round 6 (oracle: "carried encode spans"): two parse iterations and a coder model that is carried from encode span to
	 * encode span.  A partial iteration parses only the first part (part_tab) of every piece but the seed -- the first from
	 * the prior + walk + pre-roll, later ones and the final, full iteration from the snapshot the carried model walk over the
	 * records of the iteration before left in the piece's slot of prior / lit (state and rep distances: snap_sr). */
	uint32_t iter;               /* XZAMD_ITER_* */
	uint32_t *pinfo;             /* XZAMD_PINFO_WORDS x u32 per piece slot, two halves of 8: [0..7] written by iter 1, [8..15] by iter 2, the
	                                seed piece (parsed once) writes both: [0] coder state behind the piece's recorded symbols |
	                                XZAMD_PI_STATE_OK (the twelve candidate states agree), [1..4] the coder's rep distances there
	                                (XZAMD_REP_UNKNOWN: not named by the piece), [5] 1 = the parser's price of the piece says it does not
	                                shrink: stored raw (full parses only) */
	uint32_t *part_tab;          /* per piece slot: where its first part ends (xzk_span_plan: an eighth of the piece's estimated work, at least
	                                XZAMD_PART_MIN bytes; the seed piece: its end) -- what a partial iteration parses */
	uint32_t *snap_sr;           /* 8 x u32 per piece slot: [0] state, [1..4] rep distances a piece of iter 2 starts with */
	/* the carried model walk (k_model_bounds / k_model_chain / k_model_syms), per encode-span slot */
	uint32_t *cb_bnd;            /* model_slots_pad x u32: lo | hi << 11 | logged bits << 22 of every probability over the span */
	uint32_t *cb_log;            /* model_slots_pad x XZAMD_LOG_WORDS x u32: the bits of a probability until lo == hi */
	uint32_t *cb_hdr;            /* 1 x u32: XZAMD_CB_* flags of the span */
	uint16_t *cb_start;          /* model_slots_pad x u16: the model at the span start (k_model_chain) */
	uint32_t *cb_carry;          /* 1 = the span continues the model of the span in front of it (k_model_chain) */
	uint32_t model_slots_pad;    /* probabilities of the model, rounded up to 64 */
	uint32_t log_cap;            /* 0 = XZAMD_LOG_CAP; tests (XZAMD_TEST_LOG_CAP): fewer logged bits per span and probability, to reach the
	                                "not merged: the rest of the Block is not carried" fall-back on ordinary data */
} xzamd_span_args;
#define XZAMD_ITER_PARTIAL 1u       /* parse only the first part (part_tab) of every piece but the seed */
#define XZAMD_ITER_SNAP 2u          /* every piece but the seed starts from its snapshot (else: the seed's prior + walk + pre-roll) */
#define XZAMD_PINFO_WORDS 16u
#define XZAMD_PI_STATE_OK 0x80000000u
#define XZAMD_REP_UNKNOWN 0xFFFFFFFFu
#define XZAMD_PART_MIN 16384u       /* shortest first part of a piece (oracle: ORC_PART_MIN) */
#define XZAMD_LOG_WORDS 32u         /* 1023 logged bits per span and probability (oracle: ORC_LOG_CAP) */
#define XZAMD_LOG_CAP 1023u
#define XZAMD_CB_BAD_END 1u         /* a probability has not merged within XZAMD_LOG_CAP logged bits, or the span ran out of tokens */
#define XZAMD_CB_KNOWN_START 2u     /* first span of its Block, or behind a stored piece: the model is flat there */
#define XZAMD_CB_BAD_START 4u       /* the piece in front does not name the coder state */
/* One LZMA2 chunk of the two-phase coder.  Its bytes (chunk header included) are written at
 * scratch + XZAMD_CHUNK_OUT(in_start, its slot index). */
typedef struct xzamd_chunk {
	uint32_t in_start;           /* first input byte (offset into the batch) */
	uint32_t usize;              /* input bytes; 0 = unused slot */
	uint32_t tok_lo, tok_hi;     /* index of its first token */
	uint32_t ntok;
	uint32_t csize;              /* out: bytes at its scratch location, header included (raw: 3 + usize) */
	uint32_t flags;              /* XZAMD_CH_* */
	uint32_t pad_;
} xzamd_chunk;
#define XZAMD_CH_RAW 1u             /* stored uncompressed (lzma2_encoder.c:205-214) */
#define XZAMD_CH_PROPS 2u           /* header carries the properties byte */
#define XZAMD_CH_DICT_RESET 4u
#define XZAMD_CH_STATE_RESET 8u
#define XZAMD_CHUNK_EST (16000u * 128u)  /* a chunk ends when the summed prices of its decisions reach this (1/16 bit): oracle ORC_CHUNK_EST.
                                          * Round 6: 16,000 bytes (56,000 before): the range coder runs one LANE per chunk, and ~7,000 chunks per batch
                                          * were 110 wavefronts on a 256-CU GPU; a chunk costs ~10 bytes (header + coder flush): +0.06 % */
#define XZAMD_TOK_PER_BYTE 10u      /* token capacity: a literal is 9 decisions */
#define XZAMD_TOK_BASE(st, slot) ((uint64_t)(st) * XZAMD_TOK_PER_BYTE + (uint64_t)(slot) * 4096u)
/* chunk slots: an LZMA chunk holds > 15 KiB of input (its prices sum to 16,000 bytes, and a byte costs at most ~8.2 bits) unless
 * its span ends or a stored piece (>= 32 KiB: XZAMD_RAW_MIN_LEN; 64 KiB per raw chunk) cuts it short -- one slot per 8 KiB of
 * input covers every mix of the two */
#define XZAMD_CHUNK_BASE(st, slot) (((st) >> 13) + 2u * (slot))
#define XZAMD_CHUNK_CAP(len) (((len) >> 13) + 2u)                      /* chunk slots of a span of len bytes */
#define XZAMD_CHUNK_SLOTS(n, nslots) (((n) >> 13) + 2u * (nslots) + 2u)
#define XZAMD_RAW_MIN_LEN 32768u    /* shortest piece that is stored raw when its price says so (oracle: ORC_RAW_MIN_LEN) */
#define XZAMD_CHUNK_OUT(in_start, cidx) (((((uint64_t)(in_start) + ((in_start) >> 3)) + 15) & ~15ull) + (uint64_t)(cidx) * 32u)
#define XZAMD_PRIOR_WORDS 1856u     /* u32 each, >= the 1846 non-literal probabilities (a multiple of 64) */
#define XZAMD_SEED_LEN 65536u       /* two-phase: the first piece of every Block (oracle: ORC_SEED_LEN) */
#ifndef XZAMD_ENC_MIN_LEN
#define XZAMD_ENC_MIN_LEN (256u << 10)  /* shortest encode span (but the last of a Block).  Round 6: 256 KiB (512 before): the spans no longer
                                         * reset the coder's model, so their length is a matter of parallelism alone -- twice the wavefronts
                                         * for the coder's walks: 4 GiB on one GPU 559 -> 567 MB/s, one rank's 512 MiB of an 8-GPU run 424 -> 446 */
#endif
#define XZAMD_WARM 16384u          /* two-phase: bytes in front of the pre-roll walked greedily to train the price model (oracle: ORC_WARM) */
#define XZAMD_PREROLL 2048u         /* two-phase: bytes in front of a piece that are parsed twice (oracle: ORC_PREROLL) */
#define XZAMD_SPAN_SLACK 4112u      /* 4096 + 16 bytes of scratch per span slot on top of 9/8 of the input */
#define XZAMD_EST_CHUNK 4096u       /* positions per work estimate of the span plan */
#define XZAMD_SPAN_MAX (1u << 20)   /* longest cost-balanced span (oracle: ORC_SPAN_MAX) */

/* One gather segment of the final assembly. kind 0: src is an offset into the span scratch,
 * 1: into the literal-bytes buffer prepared by the host, 2: into the batch input (raw). */
typedef struct {
	uint64_t src;
	uint64_t dst;
	uint64_t len;
	uint32_t kind;
	uint32_t pad_;
} xzamd_copy_seg;

int xzk_sort_temp_bytes(uint32_t n, uint32_t end_bit, uint64_t *bytes);
int xzk_build_chains(const uint8_t *d_in, uint32_t n, uint32_t block_size, uint32_t nblocks,
		uint32_t hash_bytes, uint32_t hash_mask, uint32_t hash_bits, uint32_t sa_depth,
		uint32_t *keys_a, uint32_t *keys_b, uint32_t *vals_a, uint32_t *vals_b,
		void *sort_tmp, uint64_t sort_tmp_bytes,
		uint32_t *rank, uint32_t *sorted_pos, uint32_t *prev2, uint32_t *prev3,
		uint32_t *prev4, uint64_t *rp8, uint64_t *rp16, uint64_t *key64_a, uint64_t *key64_b,
		uint32_t *sa, uint32_t *sa_rank, uint32_t *prev24, uint32_t *prev32, void *stream);
int xzk_sa_temp_bytes(uint32_t n, uint64_t *bytes);
/* part: 0 = every position; 1 = the positions of the seed regions (the first XZAMD_SEED_LEN bytes of every Block, in
 * whole runs of 256), 2 = all the others (suffix-neighbourhood finder with block_size >= XZAMD_SEED_LEN + 512 only):
 * the two-phase mode parses the seed pieces underneath launch 2 */
int xzk_find_matches(const xzamd_span_args *a, const uint32_t *sa, const uint32_t *sa_rank, const uint32_t *prev4,
		const uint64_t *rp8, const uint64_t *rp16, const uint32_t *prev24, const uint32_t *prev32,
		uint16_t *mlen, uint32_t *mdist, int part, void *stream);
/* Cost-balanced span plan of a batch from its match lists (oracle: plan_spans): est[0 .. 2 * nblocks * cpb) receives
 * the per-chunk work and bit estimates (cpb = chunks per Block), totals[0 .. nblocks] the per-Block work and, last,
 * the batch total (u64 each); then span_tab / span_cnt as described in xzamd_span_args.  The work target of a span is
 * cost_min: the plan of a Block depends on the Block and the options only, never on the batch or the GPU.  With
 * a->enc_bits (two-phase): the first XZAMD_SEED_LEN bytes of a Block are a span of their own and enc_tab / enc_cnt receive
 * the encode spans (a->max_esb slots per Block). */
int xzk_span_plan(const xzamd_span_args *a, uint32_t nblocks, uint32_t *est, unsigned long long *totals,
		uint32_t *span_tab, uint32_t *span_cnt, uint32_t cost_min, uint32_t bits_min, uint32_t min_len,
		uint32_t *enc_tab, uint32_t *enc_cnt,
		uint32_t *order_bufs, void *sort_tmp, uint64_t sort_tmp_bytes, uint32_t **order_out, void *stream);
/* order_bufs: 4 x (nblocks * max_spb) u32 of scratch; *order_out = where the launch order ends up (inside order_bufs) */
int xzk_span_encode(const xzamd_span_args *a, uint32_t nslots, uint32_t waves, uint32_t *counter, void *stream);
/* Two-phase mode.  xzk_parse_pieces: phase 0 = the seed pieces (one wavefront per Block), phase 1 = every other piece
 * (nslots = nblocks * max_spb slots in a->order, heaviest first; persistent with `waves` wavefronts when waves != 0).
 * xzk_encode_syms: the model pass (one wavefront per encode-span slot: tokens + chunk table, raw chunks copied) and the
 * range coder (one lane per chunk slot); the chunk table must be zero on entry (the call clears it). */
int xzk_parse_pieces(const xzamd_span_args *a, uint32_t nblocks, int phase, uint32_t waves, uint32_t *counter, void *stream);
int xzk_encode_syms(const xzamd_span_args *a, uint32_t nblocks, void *stream);
/* The carried model walk over the records of parse iteration 1 (sub-sampled: the first part of every piece): k_model_bounds,
 * k_model_chain, then k_model_syms in snapshot mode -- every piece but the seed finds the price model it starts iteration 2
 * from in its slot of a->prior / a->lit, state and rep distances in a->snap_sr. */
int xzk_model_snapshots(const xzamd_span_args *a, uint32_t nblocks, void *stream);
/* wavefronts of the span kernel variant for (parser, nice_len) one CU holds at once */
int xzk_span_occupancy(int parser, uint32_t nice_len, int *waves_per_cu);
/* x86 BCJ encoder: d_out = filtered copy of d_in, every Block filtered independently (simple/x86.c). */
int xzk_x86_bcj(const uint8_t *d_in, uint8_t *d_out, uint32_t n, uint32_t block_size, uint32_t nblocks, void *stream);
/* ARM64 BCJ (kind 0x0A) / delta (kind 3, dist 1..256) encoders, every Block filtered independently. */
int xzk_prefilter(const uint8_t *d_in, uint8_t *d_out, uint32_t n, uint32_t block_size, uint32_t nblocks, uint32_t kind, uint32_t dist,
		void *stream);
/* SHA-256 of every Block (32 bytes each). */
int xzk_sha256_blocks(const uint8_t *d_in, uint32_t n, uint32_t block_size, uint32_t nblocks, uint8_t *d_out32, void *stream);
/* Block checks: CRC64 (crc32 = 0) or CRC32 (crc32 = 1, zero-extended into d_block_crc). */
int xzk_crc_blocks(const uint8_t *d_in, uint32_t n, uint32_t block_size, uint32_t nblocks,
		uint32_t strip, int crc32, uint64_t *d_strip_crc, uint64_t *d_block_crc, void *stream);
int xzk_assemble(const xzamd_copy_seg *d_segs, uint32_t nsegs, const uint8_t *d_scratch,
		const uint8_t *d_lits, const uint8_t *d_in, uint8_t *d_out, void *stream);

/* ---- LZMA2 Block decoder (lzma_decode.hip) ---- */
typedef struct {
	uint64_t cpos;       /* offset of the Block's LZMA2 data (behind the Block Header) in the stream */
	uint64_t csize;      /* its size (Compressed Size incl. the 0x00 end marker) */
	uint64_t upos;       /* where the Block's bytes go in the output */
	uint64_t usize;      /* Uncompressed Size */
	uint32_t dict_size;
	uint32_t nunits;     /* out (k_dec_scan) */
	uint32_t error;      /* out: 0 = fine */
	uint32_t pad_;
} xzamd_dec_block;

typedef struct {
	uint64_t cpos;       /* stream offset of the unit's first chunk header */
	uint64_t upos;       /* offset of its first byte inside the Block */
	uint32_t block;
	uint32_t dbase;      /* Block offset of the last dictionary reset at or before the unit */
} xzamd_dec_unit;

int xzk_dec_scan(const uint8_t *d_xz, xzamd_dec_block *d_blocks, uint32_t nblocks, xzamd_dec_unit *d_units,
		uint32_t units_cap, int split, void *stream);
int xzk_dec_units(const uint8_t *d_xz, const xzamd_dec_block *d_blocks, uint32_t nblocks, const xzamd_dec_unit *d_units,
		uint32_t units_cap, const uint32_t *d_unit_first, uint32_t total_units, uint8_t *d_out, const uint8_t *d_expected,
		uint16_t *d_lit_pool, uint32_t waves, uint32_t *d_counter, uint32_t *d_block_err, void *stream);
int xzk_dec_compare(const uint8_t *a, const uint8_t *b, uint64_t n, unsigned long long *d_mismatches, void *stream);

int xzk_malloc(void **p, uint64_t bytes);
int xzk_free(void *p);
int xzk_host_alloc(void **p, uint64_t bytes);
int xzk_host_free(void *p);
int xzk_h2d(void *d, const void *h, uint64_t bytes, void *stream);
int xzk_d2h(void *h, const void *d, uint64_t bytes, void *stream);
int xzk_memset(void *d, int v, uint64_t bytes, void *stream);
int xzk_sync(void *stream);
int xzk_set_device(int dev);
int xzk_get_device(int *dev);
int xzk_device_count(int *n);
int xzk_cu_count(int dev, int *cus);
int xzk_stream_create(void **stream);
int xzk_stream_create_low(void **stream);      /* lowest priority of the device */
int xzk_stream_wait_event(void *stream, void *ev);
int xzk_stream_destroy(void *stream);
int xzk_event_create(void **ev);
int xzk_event_destroy(void *ev);
int xzk_event_record(void *ev, void *stream);
int xzk_event_elapsed_ms(void *a, void *b, float *ms);
int xzk_event_query(void *ev);      /* 0 = everything recorded before the event has completed */
const char *xzk_error_string(int e);
int xzk_mem_info(uint64_t *free_b, uint64_t *total_b);

#ifdef __cplusplus
}
#endif
#endif
