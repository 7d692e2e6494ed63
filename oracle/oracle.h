/*
 * oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference algorithms on the hot path
 *  and of the container around it.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call
 * anything in oracle/.  The product (xz_amd/csrc) never does.
 *
 * Pinning (tests/test_oracle_*.py, CPU-only):
 *   - container pieces: byte-compared against the real reference library
 *     (oracle/_ref/liblzma_ref.so) and the reference's own KATs
 *     (tests/test_vli.c:20-46, tests/test_check.c:69-139);
 *   - LZMA2 decoder: decodes the reference's tests/files/good-*.xz fixtures
 *     (copied as data into tests/golden/) and everything liblzma_ref produces;
 *   - fast-mode encoder (HC3/HC4 + optimum_fast + range coder + LZMA2
 *     chunking): output is BYTE-IDENTICAL to the reference's raw LZMA2
 *     encoder for presets 0-3 on every test corpus;
 *   - x86 BCJ encoder: byte-identical to the reference filter's output.
 * PARITY UNPINNED (by the reference) for the two algorithms that are OURS and
 * have no counterpart to compare bytes with: build_sa / find_sn (suffix-
 * neighbourhood finder, the BT4 successor) and optimum_window (windowed optimal
 * parser with the reference's compound edges) -- what presets 4-9 run.  Their
 * pins are indirect: every stream they produce must decode bit-exactly through
 * the REAL reference decoder, the symbol coder / range coder / chunker
 * underneath them are the pinned ones, and their compressed size is held to a
 * stated tolerance: <= 1.03 x liblzma's at the same preset and Block size
 * (tests/test_oracle_encoder.py, tests/test_gpu_parity.py).
 * Reference citations are file:line relative to /root/reference.
 */
#ifndef XZ_AMD_ORACLE_H
#define XZ_AMD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* Container (src/liblzma/common, src/liblzma/check)                    */
/* ------------------------------------------------------------------ */
uint32_t orc_crc32(const uint8_t *buf, size_t n, uint32_t crc);
uint64_t orc_crc64(const uint8_t *buf, size_t n, uint64_t crc);
/* CRC64 of a concatenation from the parts' CRCs (used to check the GPU's
 * piecewise CRC): crc(A||B) from crc(A), crc(B), len(B). */
uint64_t orc_crc64_combine(uint64_t crc_a, uint64_t crc_b, uint64_t len_b);

uint32_t orc_vli_size(uint64_t v);
uint32_t orc_vli_encode(uint64_t v, uint8_t *out);

#define ORC_CHECK_NONE 0
#define ORC_CHECK_CRC32 1
#define ORC_CHECK_CRC64 4
uint32_t orc_check_size(int check);

void orc_stream_header(uint8_t out[12], int check);
void orc_stream_footer(uint8_t out[12], uint64_t index_size, int check);
uint8_t orc_lzma2_dict_byte(uint32_t dict_size);
uint64_t orc_block_bound(uint64_t uncompressed_size);
/* Header size as lzma_block_header_size() computes it for one LZMA2 filter
 * (+ optional x86 BCJ filter without start offset). */
uint32_t orc_block_header_size(uint64_t csize, uint64_t usize, int with_x86);
void orc_block_header_encode(uint8_t *out, uint32_t header_size,
		uint64_t csize, uint64_t usize, uint8_t dict_byte, int with_x86);
/* Index field. Returns its size (multiple of 4). out may be NULL. */
uint64_t orc_index_encode(uint8_t *out, uint64_t nblocks,
		const uint64_t *unpadded, const uint64_t *uncompressed);
/* Whole-Block uncompressed fallback (block_buffer_encoder.c:88-162). Writes
 * header + chunks + 0x00 + padding + check; returns total bytes, sets
 * *unpadded. */
uint64_t orc_block_uncomp_encode(const uint8_t *in, uint64_t n, int check,
		uint8_t *out, uint64_t *unpadded);

/* Assemble a whole .xz Stream the way stream_encoder_mt does from per-Block
 * raw LZMA2 payloads (each already ending with the 0x00 end marker).
 * block_size = the lzma_mt.block_size the header sizes are derived from. */
uint64_t orc_xz_frame(const uint8_t *const *payloads, const uint64_t *payload_sizes,
		const uint8_t *const *inputs, const uint64_t *input_sizes,
		uint64_t nblocks, uint64_t block_size, uint32_t dict_size,
		int check, uint8_t *out, uint64_t out_cap);

/* x86 BCJ encoder over one whole Block, in place (simple/x86.c:26-118, fresh state). */
void orc_x86_encode(uint8_t *buf, uint64_t size);

/* ------------------------------------------------------------------ */
/* Decoder (verifier)                                                   */
/* ------------------------------------------------------------------ */
typedef struct {
	uint32_t pos;   /* uncompressed offset in the Block */
	uint32_t back;  /* 0..3 rep index, dist+4, or 0xFFFFFFFF literal */
	uint32_t len;
} orc_symbol;

typedef struct {
	orc_symbol *sym;     /* optional, capacity sym_cap */
	uint64_t sym_cap;
	uint64_t sym_count;  /* total symbols seen (may exceed cap) */
	uint64_t chunks_lzma, chunks_uncompressed, state_resets, prop_resets, dict_resets;
} orc_trace;

/* Raw LZMA2 decode. Returns 0 on success (end marker seen, all input used). */
int orc_lzma2_decode(const uint8_t *in, uint64_t in_size, uint32_t dict_size,
		uint8_t *out, uint64_t out_cap, uint64_t *out_size, orc_trace *trace);
/* Full .xz single-stream decode with all CRC/size/index checks.
 * Returns 0 on success; negative codes identify the failing check.
 * nblocks_out (optional) receives the number of Blocks. */
int orc_xz_decode(const uint8_t *in, uint64_t in_size,
		uint8_t *out, uint64_t out_cap, uint64_t *out_size,
		uint64_t *nblocks_out);

/* ------------------------------------------------------------------ */
/* Fast-mode encoder (HC3/HC4, optimum_fast, rc, LZMA2 chunker)         */
/* ------------------------------------------------------------------ */
typedef struct {
	uint32_t dict_size;
	uint32_t lc, lp, pb;
	uint32_t nice_len;
	uint32_t mf;        /* 3 = HC3, 4 = HC4 */
	uint32_t depth;     /* 0 = reference default (4 + nice_len/4) */
	uint32_t span_size; /* 0 = whole Block is one span (== reference);
	                       else independent state-reset spans (GPU mode) */
	uint32_t sa_window; /* 0 = exact HC3/HC4; else the suffix-neighbourhood finder: recency records among
	                       `sa_window` (<= 5) slots on either side in 32-byte-prefix suffix order, plus the
	                       nearest equal hash2 / hash4 and equal 8 / 16 bytes (`depth` unused) */
	uint32_t parser;    /* 0 = optimum_fast (reference); 1 = windowed optimal parser (ours) */
	uint32_t sa_depth;  /* suffix-neighbourhood finder: bytes of prefix the suffix order compares: 32 (0 = 32), 64,
	                       128 or 256 (one more rank-doubling round each) */
	uint32_t span_cost; /* != 0 (needs sa_window): cost-balanced spans instead of spans of span_size bytes.  Every
	                       ORC_EST_CHUNK bytes get an estimate of the parser's work from a walk over the match lists
	                       (one unit per position the parser visits); a Block of estimated work `total` is cut into
	                       k = max(1, min(total / span_cost, bits / span_bits)) spans: a span ends at the first chunk
	                       boundary where its estimate reaches ceil(total / k) and it is >= span_size bytes long (span_size 0: 64 KiB) */
	uint32_t span_bits; /* with span_cost: a Block of estimated coded size `bits` (greedy parse over the match lists) gets
	                       at most bits / span_bits spans: the number of state resets is bounded by the Block's OUTPUT */
	uint32_t enc_bits;  /* != 0 (needs span_cost and parser 1): TWO-PHASE encode (OUR definition; the device: k_parse_pieces /
	                       k_encode_syms).  The spans of the plan become parse PIECES: phase 1 runs the optimal parser over
	                       every piece on its own, with an adaptive price model that codes nothing, and records the symbols in a
	                       coder-independent form (length, distance) / literal; symbols never cross a piece end.  The first
	                       ORC_SEED_LEN bytes of the Block are a piece of their own, parsed from the flat model; the model they
	                       leave is the PRIOR every other piece of the Block starts its prices from.  Phase 2 range-codes the
	                       recorded symbols with ONE continuous model per ENCODE SPAN (state reset only there), choosing rep /
	                       short rep / match from its own rep distances.  Encode spans end at piece ends: a Block of estimated
	                       coded size `bits` gets ke = max(1, min(n / ORC_ENC_MIN_LEN, bits / enc_bits)) of them, closed at the
	                       first piece end where the estimate reaches ceil(bits / ke) and the span is >= ORC_ENC_MIN_LEN long */
	uint32_t part_iters; /* two-phase: partial parse iterations in front of the full one (0 = 1; the device: xzamd_lzma_options.part_iters) */
} orc_enc_params;
#define ORC_EST_CHUNK 4096u
#define ORC_SPAN_MAX (1u << 20)    /* longest piece (round 6: 16 MiB let one wavefront walk 16 MiB of a highly compressible Block) */
#define ORC_SEED_LEN 65536u
#define ORC_ENC_MIN_LEN (256u << 10)

/* The span plan of one Block under p->span_cost: chunk_cost (optional, 2 * ceil(n / ORC_EST_CHUNK) entries: work
 * estimates, then bit estimates) receives the per-chunk estimates, span_start (optional, capacity span_cap) the first byte of every span; returns the
 * number of spans (0 on error). */
uint32_t orc_span_plan(const uint8_t *in, uint32_t n, const orc_enc_params *p, uint32_t *chunk_cost,
		uint32_t *span_start, uint32_t span_cap);
/* Two-phase mode: the same plan plus the encode spans (enc_start: first byte of each, capacity enc_cap; *n_enc their
 * number).  Returns the number of pieces. */
uint32_t orc_piece_plan(const uint8_t *in, uint32_t n, const orc_enc_params *p, uint32_t *chunk_cost,
		uint32_t *span_start, uint32_t span_cap, uint32_t *enc_start, uint32_t enc_cap, uint32_t *n_enc);
/* Two-phase mode: the recorded parse of one Block as the device stores it: per position (valid at symbol starts)
 * sym_len (0 = literal, else the match length) and sym_dist (zero-based distance; literal: byte | previous byte << 8 |
 * match byte << 16 | (parser state >= 7) << 24). */
int orc_parse_dump(const uint8_t *in, uint32_t n, const orc_enc_params *p, uint16_t *sym_len, uint32_t *sym_dist);
/* Two-phase mode, stage by stage (device parity tests): every pointer optional.  snap_sr: 5 x u32 per piece (state, rep
 * distances a piece starts iteration 2 with; piece 0 unused), snap_probs: the 1846 non-literal probabilities of that model per
 * piece, price: the parser's price of every piece (1/16 bit), carry: per encode span 0 = reset + properties, 1 = carried,
 * 2 = flat start behind a stored piece (span 0 unused). */
typedef struct {
	uint32_t *snap_sr;
	uint16_t *snap_probs;
	uint64_t *price;
	uint32_t *carry;
} orc_two_phase_dbg;
/* tests: at most v logged bits per encode span and probability (0 = ORC_LOG_CAP); the device: XZAMD_TEST_LOG_CAP */
void orc_set_log_cap(uint32_t v);
int orc_two_phase_debug(const uint8_t *in, uint32_t n, const orc_enc_params *p, orc_two_phase_dbg *d);
/* Two-phase mode: orc_lzma2_encode_block and orc_parse_dump in one pass (full-size parity tests). */
int orc_lzma2_encode_block_syms(const uint8_t *in, uint32_t n, const orc_enc_params *p,
		uint8_t *out, uint64_t out_cap, uint64_t *out_size, uint16_t *sym_len, uint32_t *sym_dist);

/* tests only: a smaller token budget per input byte for the two-phase coder (0 / >= 10: the default) */
void orc_set_tok_per_byte(uint32_t v);
int orc_preset(uint32_t preset, orc_enc_params *p, uint32_t *mode_normal);

/* Encode one Block's raw LZMA2 payload (incl. 0x00 end marker).
 * Returns 0 on success. trace optional. */
int orc_lzma2_encode_block(const uint8_t *in, uint32_t n, const orc_enc_params *p,
		uint8_t *out, uint64_t out_cap, uint64_t *out_size, orc_trace *trace);

/* Match-finder dump: for every position pos_list[i] report what
 * lzma_mf_find() would return there when positions are consumed one at a
 * time (lz_encoder_mf.c:22-79).  `end` = exclusive end used for avail
 * (Block end, or span end in span mode).  Output: for each i,
 * counts[i] pairs stored at pairs[i*max_pairs*2 ...] as (len,dist), and
 * longest[i] = the (possibly extended) longest length. */
int orc_mf_dump(const uint8_t *in, uint32_t n, const orc_enc_params *p,
		const uint32_t *pos_list, const uint32_t *end_list, uint32_t npos,
		uint32_t max_pairs, uint32_t *counts, uint32_t *pairs, uint32_t *longest);

/* Device-parity debug hooks (OUR structures): suffix order of one Block (slot -> position and
 * position -> slot) and the 8-word match-list record of every position, as the GPU stores them. */
int orc_sa_dump(const uint8_t *in, uint32_t n, uint32_t depth, uint32_t *sa_out, uint32_t *rank_out);
int orc_list_dump(const uint8_t *in, uint32_t n, const orc_enc_params *p, uint32_t *words);

/* ------------------------------------------------------------------ */
/* Synthetic corpora                                     */
/* ------------------------------------------------------------------ */
/* lorem-LCG: tests/create_compress_files.c:110-152 continued to n bytes. */
void orc_corpus_lorem(uint8_t *out, uint64_t n);
/* tests/create_compress_files.c:82-105 */
void orc_corpus_abc(uint8_t *out, uint64_t n);
void orc_corpus_random(uint8_t *out, uint64_t n);

#ifdef __cplusplus
}
#endif
#endif
