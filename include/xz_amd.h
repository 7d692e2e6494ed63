/*
 * xz_amd.h -- public C ABI of the MI355X-native LZMA2 Block encoder (libxz_amd.so).
 *
 * Plain C: pointers and sizes only, no HIP or torch types in any signature.
 * `stream` arguments are a hipStream_t passed as void* (NULL = the library's
 * own stream).
 *
 * Two levels, both replacing reference interfaces (paths relative to the
 * XZ Utils 5.8.3 tree):
 *
 *  (1) Device-resident batch encode -- what every worker thread of the
 *      reference does for one Block, done here for all Blocks at once:
 *        worker_encode()            src/liblzma/common/stream_encoder_mt.c:219-361
 *        block_encode()             src/liblzma/common/block_encoder.c:47-135
 *        lzma2_encode()             src/liblzma/lzma/lzma2_encoder.c:135-259
 *        lzma_lzma_encode()         src/liblzma/lzma/lzma_encoder.c:313-436
 *        lzma_mf_hc3/hc4_find/skip  src/liblzma/lz/lz_encoder_mf.c:305-441
 *      -> xzamd_stream_encode_device()
 *
 *  (2) The liblzma streaming API itself (declared in xz_amd_lzma.h):
 *        lzma_stream_encoder_mt()   stream_encoder_mt.c:1196
 *        lzma_code()/lzma_end()     common/common.c:203,379
 *        lzma_get_progress()        common/common.c:406
 */
#ifndef XZ_AMD_H
#define XZ_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct xzamd_ctx xzamd_ctx;

/* Return codes: the lzma_ret values of api/lzma/base.h:55-271 (same numbers). */
#define XZAMD_OK              0
#define XZAMD_STREAM_END      1
#define XZAMD_UNSUPPORTED_CHECK 3
#define XZAMD_MEM_ERROR       5
#define XZAMD_OPTIONS_ERROR   8
#define XZAMD_DATA_ERROR      9
#define XZAMD_BUF_ERROR      10
#define XZAMD_PROG_ERROR     11
/* HIP runtime / kernel failure (mapped to LZMA_PROG_ERROR at the lzma_* level). */
#define XZAMD_DEVICE_ERROR  100

#define XZAMD_CHECK_NONE   0
#define XZAMD_CHECK_CRC32  1
#define XZAMD_CHECK_CRC64  4
#define XZAMD_CHECK_SHA256 10

#define XZAMD_MF_HC3 0x03   /* lzma_match_finder values, api/lzma/lzma12.h:58-112 */
#define XZAMD_MF_HC4 0x04
#define XZAMD_MF_BT4 0x14

#define XZAMD_MODE_FAST   1 /* lzma_mode, api/lzma/lzma12.h:138-157 */
#define XZAMD_MODE_NORMAL 2

#define XZAMD_PRESET_EXTREME 0x80000000u

/* One span per Block: output is byte-identical to the reference MT encoder
 * for fast-mode HC3/HC4 chains (presets 0-3). */
#define XZAMD_SPAN_WHOLE_BLOCK 0xFFFFFFFFu
#define XZAMD_SPAN_DEFAULT 0u
/* Same as XZAMD_SPAN_DEFAULT (kept for callers of the round-2 interface: the span plan is derived from the
 * batch and the GPU's wave slots either way). */
#define XZAMD_SPAN_AUTO 1u

/* LZMA2 options: the encoder-relevant subset of lzma_options_lzma
 * (api/lzma/lzma12.h:216-525) plus the GPU span size. */
typedef struct {
	uint32_t dict_size;
	uint32_t lc, lp, pb;
	uint32_t mode;       /* as requested by the preset (informational) */
	uint32_t nice_len;
	uint32_t mf;         /* requested match finder */
	uint32_t depth;      /* 0 = reference default (lz_encoder.c:359-365) */
	/* --- what the device path actually runs --- */
	uint32_t gpu_mf;     /* XZAMD_MF_HC3 / XZAMD_MF_HC4 */
	uint32_t gpu_nice_len;
	uint32_t gpu_depth;  /* candidates taken from the main (3/4-byte hash) chain (exact finder only) */
	uint32_t span_size;  /* bytes per independently coded span; XZAMD_SPAN_*.  With the optimal parser (gpu_parser = 1) an EXPLICIT
	                        span size (also: XZAMD_SPAN_KIB in the environment) selects the single-phase kernel -- every span codes
	                        with its own model, 8 wavefronts per CU: a test and measurement mode, several times slower than the
	                        default (XZAMD_SPAN_DEFAULT / _AUTO: cost-balanced parse pieces + the two-phase encode) */
	uint32_t gpu_sa_window; /* 0 = exact HC3/HC4 semantics of the reference; else the suffix-neighbourhood
	                        finder (the BT4 successor): recency records among this many slots (<= 5) on
	                        either side of a position in 32-byte-prefix suffix order, plus the nearest
	                        equal hash2 / hash4 and equal 8 / 16 bytes; needs gpu_mf = HC4 and
	                        gpu_parser = 1 */
	uint32_t gpu_parser; /* 0 = lzma_lzma_optimum_fast semantics, 1 = windowed optimal parser (384-node DP over
	                        per-position match lists incl. the reference's compound edges; pb > 2 needs the two-phase
	                        mode, i.e. no explicit span size, else XZAMD_OPTIONS_ERROR) */
	uint32_t bcj;        /* 0 = chain {LZMA2}; XZAMD_BCJ_* / XZAMD_FILTER_DELTA(d) = that filter in front of LZMA2
	                        (the FIRST filter of the chain when bcj2 / bcj3 below name more) */
	uint32_t gpu_sa_depth; /* suffix-neighbourhood finder: prefix bytes the suffix order compares, 32 / 64 / 128 / 256
	                        (0 = 32); one rank-doubling round more per step.  Presets: 32 for nice_len <= 32, 64 up to
	                        nice_len 64, 256 above (what lz_encoder_mf.c:450-512 orders by is the whole suffix) */
	uint32_t span_cost;  /* != 0 with span_size XZAMD_SPAN_DEFAULT / _AUTO and the optimal parser: cost-balanced spans.
	                        The match lists give every 4 KiB an estimate of the parser's work (one unit per position it
	                        has to visit: positions under a match of nice_len bytes or more are skipped); a Block is cut
	                        into spans of equal estimated work of about span_cost units, each at least 64 KiB long (a batch that
	                        fills the GPU: 0.8 .. 2 x span_cost, chosen so that the rounds of the launch are full; the target
	                        used is reported in xzamd_stats.span_cost_used).  0: spans of span_size bytes */
	uint32_t span_bits;  /* with span_cost: a Block whose estimated coded size is `bits` (greedy parse over the match lists)
	                        gets at most bits / span_bits spans (no span longer than 1 MiB): a piece start costs output bytes
	                        whatever the data, so what bounds their number is the Block's OUTPUT (highly compressible Blocks:
	                        fewer, longer spans) */
	uint32_t enc_span_bits; /* != 0 with cost-balanced spans: TWO-PHASE encode.  The spans of the plan are
	                        parse PIECES: the optimal parser runs over each with an adaptive price model that codes nothing
	                        (the first 64 KiB of a Block are the seed piece, parsed from the flat model; what it leaves is
	                        the prior of the partial iteration; the full one starts every piece from a snapshot, part_iters) and
	                        records (length, distance) / literal symbols; the coder then walks them per ENCODE SPAN, in parallel,
	                        with ONE continuous model per Block (the model is carried from span to span exactly; a
	                        state reset only at the Block start, behind a stored piece, or where the carry falls back).
	                        A Block of estimated coded size `bits` gets max(1, min(size / 256 KiB, bits / enc_span_bits))
	                        encode spans, closed at piece ends.  0: single phase, every span of the plan resets the state */
	uint32_t bcj2, bcj3; /* second and third filter in front of LZMA2, same values as `bcj`, applied in that order (a chain
	                        holds at most 4 filters, LZMA2 last: common/filter_common.c:250-334); bcj3 needs bcj2 needs bcj */
	uint32_t part_iters; /* two-phase: PARTIAL parse iterations in front of the full one (0 = XZAMD_PART_ITERS_DEFAULT, at most 8).
	                        Each parses the first eighth (of its estimated work; >= 16 KiB) of every piece -- the first from the seed's prior, the others
	                        from the snapshots of the walk before -- and the carried model walk over its records leaves every
	                        piece the price model the next iteration starts from.  One costs about 15 % of the
	                        parse; more of them walk record tables out of the regime the Block's first 64 KiB suggest */
} xzamd_lzma_options;
#define XZAMD_PART_ITERS_DEFAULT 1u
#define XZAMD_PART_ITERS_MAX 8u
#define XZAMD_PREFILTERS_MAX 3u
#define XZAMD_SPAN_COST_DEFAULT 131072u   /* text: 128 KiB spans */
#define XZAMD_SPAN_BITS_DEFAULT 400000u
#ifndef XZAMD_ENC_SPAN_BITS_DEFAULT
#define XZAMD_ENC_SPAN_BITS_DEFAULT 800000u    /* about 100 KB of output per encode span (round 5: 1,600,000) */
#endif
#define XZAMD_SPAN_MIN_LEN 65536u         /* shortest cost-balanced span */
#define XZAMD_BCJ_X86 4u    /* LZMA_FILTER_X86, api/lzma/bcj.h:20 */
#define XZAMD_BCJ_ARM64 0x0Au   /* LZMA_FILTER_ARM64, api/lzma/bcj.h (simple/arm64.c), start offset 0 */
#define XZAMD_BCJ_RISCV 0x0Bu   /* LZMA_FILTER_RISCV (simple/riscv.c) */
#define XZAMD_BCJ_POWERPC 5u    /* LZMA_FILTER_POWERPC (simple/powerpc.c) */
#define XZAMD_BCJ_IA64 6u       /* LZMA_FILTER_IA64 (simple/ia64.c) */
#define XZAMD_BCJ_ARM 7u        /* LZMA_FILTER_ARM (simple/arm.c) */
#define XZAMD_BCJ_ARMTHUMB 8u   /* LZMA_FILTER_ARMTHUMB (simple/armthumb.c) */
#define XZAMD_BCJ_SPARC 9u      /* LZMA_FILTER_SPARC (simple/sparc.c) */
#define XZAMD_FILTER_DELTA(dist) (3u | (((uint32_t)(dist) - 1u) << 8))   /* LZMA_FILTER_DELTA, dist 1..256 (delta/delta_encoder.c) */
#define XZAMD_SA_WINDOW_MAX 5u

/* lzma_lzma_preset() (lzma/lzma_encoder_presets.c:17-63) + the device mapping.
 * Returns nonzero for an invalid preset. */
int xzamd_lzma_preset(xzamd_lzma_options *opt, uint32_t preset);
/* For hand-made option sets: fill in what a BT2/BT3/BT4 + normal-mode request runs on the device (suffix-
 * neighbourhood finder, suffix depth from gpu_nice_len, windowed optimal parser, cost-balanced spans). */
void xzamd_sn_defaults(xzamd_lzma_options *opt);
/* NULL when the device path runs this option set, else the reason it does not (static string). */
const char *xzamd_options_check(const xzamd_lzma_options *opt);
/* Give back the device / pinned buffers lzma_end() keeps parked for the next stream of the process. */
void xzamd_release_parked(void);
/* lzma_mt_block_size() for a plain LZMA2 chain (lzma2_encoder.c:403-413). */
uint64_t xzamd_mt_block_size(const xzamd_lzma_options *opt);
/* lzma_block_buffer_bound64() (common/block_buffer_encoder.c:55-69). */
uint64_t xzamd_block_buffer_bound(uint64_t uncompressed_size);
/* Upper bound for the whole .xz Stream produced from in_size bytes. */
uint64_t xzamd_stream_buffer_bound(uint64_t in_size, uint64_t block_size);

/* Context = one GPU (device ordinal, -1 = current) + its work buffers. */
int xzamd_ctx_create(xzamd_ctx **ctx, int device);
void xzamd_ctx_destroy(xzamd_ctx *ctx);
/* Bytes of input processed per device batch (default 2 GiB - 1 MiB; must be < 2 GiB: positions are 31-bit).
 */
int xzamd_ctx_set_batch_bytes(xzamd_ctx *ctx, uint64_t bytes);
const char *xzamd_last_error(const xzamd_ctx *ctx);
int xzamd_ctx_device(const xzamd_ctx *ctx);          /* device ordinal the context lives on */

typedef struct {
	uint64_t in_bytes, out_bytes;
	uint64_t blocks, spans, batches;
	uint64_t blocks_stored;      /* Blocks that took the uncompressed fallback */
	/* HIP-event time per stage, summed over batches (milliseconds) */
	float ms_chains;             /* hash keys + radix sorts + link kernels */
	float ms_encode;             /* k_find (when the optimal parser runs) + k_span_encode */
	float ms_crc;
	float ms_assemble;
	float ms_total;              /* first launch -> last kernel done */
	uint32_t encode_launches;
	float ms_find;               /* the k_find part of ms_encode */
	uint32_t span_size;          /* bytes per span the last encode used (0: cost-balanced spans) */
	float ms_find_overlapped;    /* k_find of batches whose structure + lists were made underneath the previous span kernel */
	uint32_t span_cost_used;     /* cost-balanced spans: work target of the LAST batch (>= span_cost) */
	float ms_plan;               /* span plan (k_span_est + k_span_cut) */
	uint32_t wave_slots;         /* span wavefronts the GPU holds at once (CUs x occupancy of the span kernel) */
	uint64_t enc_spans;          /* two-phase: encode spans (= state resets + 1 per Block); `spans` counts the parse pieces */
	float ms_seed;               /* two-phase: the seed pieces (one wavefront per Block, part of ms_encode) */
	float ms_parse;              /* two-phase: every other piece */
	float ms_code;               /* two-phase: the coder's walk (bounds, chain, tokens) + the range coder, second stream */
	float ms_iter1;              /* two-phase: the part of ms_parse spent in the partial iteration(s) and the carried walk over their records;
	                                ms_parse - ms_iter1 = the full parse (k_parse_pieces, iteration 2): the dominant kernel */
} xzamd_stats;
/* Stats of the last xzamd_stream_encode_device call of the context.  Under the lzma_* front end consecutive jobs of a
 * worker are pipelined (the back end of a job's last batch finishes underneath the next job's front end): there the stage
 * times, blocks, spans and batches of that carried batch are booked to the call that finishes it, i.e. the figures are
 * running per-context totals shifted by one batch, not per-job figures. */
void xzamd_get_stats(const xzamd_ctx *ctx, xzamd_stats *out);

/* Encode in_size bytes resident in device memory (d_in) into a complete,
 * standards-conformant .xz Stream in device memory (d_out): Stream Header,
 * one Block per block_size bytes (Block Header with both sizes, LZMA2 data,
 * padding, Check), Index, Stream Footer -- exactly the layout
 * lzma_stream_encoder_mt produces (stream_encoder_mt.c:717-883).
 *
 * flags: XZAMD_F_BLOCKS_ONLY emits only the Blocks (no header/index/footer)
 * and reports per-Block sizes through `binfo` so several GPUs can each encode
 * a shard and one of them frames the Stream.
 *
 * block_size 0 = xzamd_mt_block_size(opt). check = XZAMD_CHECK_*.
 * Returns XZAMD_OK or an error code. */
#define XZAMD_F_BLOCKS_ONLY 1u

typedef struct {
	uint64_t unpadded_size;      /* Index record field 1 (block_util.c:45-76) */
	uint64_t uncompressed_size;  /* Index record field 2 */
	uint64_t out_offset;         /* where the Block starts in d_out */
	uint64_t total_size;         /* header + data + padding + check */
} xzamd_block_info;

int xzamd_stream_encode_device(xzamd_ctx *ctx,
		const void *d_in, uint64_t in_size, uint64_t block_size,
		const xzamd_lzma_options *opt, int check, uint32_t flags,
		void *d_out, uint64_t out_cap, uint64_t *out_size,
		xzamd_block_info *binfo, uint64_t binfo_cap, uint64_t *nblocks,
		void *stream);

/* Device .xz decoder (the reference's stream_decoder_mt.c / lzma2_decoder.c / lzma_decoder.c path, SURVEY.md
 * 8f.3): decode a single-Stream .xz file resident in device memory (filter chain {LZMA2}; checks none / CRC32 /
 * CRC64 / SHA-256 are verified, any other Check id is XZAMD_UNSUPPORTED_CHECK) into d_out.  Blocks decode in parallel, one wavefront each.  With d_expected (the original
 * data, device memory, expected_size bytes: a Stream of another uncompressed size is XZAMD_DATA_ERROR) it is a VERIFICATION decode: every chunk chain that starts with a state
 * reset + properties (our spans) is its own unit, history is read from d_expected, and the decoded bytes are
 * compared with it afterwards (*mismatches).  Returns XZAMD_OK, 7 (LZMA_FORMAT_ERROR: not an .xz Stream),
 * XZAMD_OPTIONS_ERROR (unsupported chain), XZAMD_DATA_ERROR (corrupt / mismatch), XZAMD_BUF_ERROR (out_cap). */
int xzamd_stream_decode_device(xzamd_ctx *ctx, const void *d_xz, uint64_t xz_size, void *d_out, uint64_t out_cap,
		uint64_t *out_size, const void *d_expected, uint64_t expected_size, uint64_t *mismatches, uint64_t *nblocks,
		void *stream);

/* Host-side framing helpers for the multi-GPU path: Stream Header (12 bytes),
 * Index + Stream Footer from gathered Block records. Return bytes written. */
uint64_t xzamd_frame_header(uint8_t *out, int check);
uint64_t xzamd_frame_index_footer(uint8_t *out, uint64_t out_cap, int check,
		const uint64_t *unpadded, const uint64_t *uncompressed, uint64_t nblocks);

/* Debug hook for the parity tests: when set, the next encode records every
 * LZMA symbol (span, pos, back, len as 4 x u32) into a device buffer of
 * `cap` symbols; xzamd_trace_read copies it out. Not for production use. */
int xzamd_trace_enable(xzamd_ctx *ctx, uint32_t cap);
int xzamd_trace_read(xzamd_ctx *ctx, uint32_t *out, uint32_t cap, uint32_t *count);

/* Debug hook for the parity tests: copy a work buffer of the LAST batch of the last encode to the host
 * (suffix order slot -> position, position -> slot, the 8 x u32 match-list records, their u16 lengths). */
#define XZAMD_DEBUG_SA 1
#define XZAMD_DEBUG_SA_RANK 2
#define XZAMD_DEBUG_LISTS 3
#define XZAMD_DEBUG_LIST_LENS 4
#define XZAMD_DEBUG_SPAN_TAB 5      /* span plan: (first byte, end) per span slot, slots = Block * (block_size / 64 KiB + 2) + k */
#define XZAMD_DEBUG_SPAN_CNT 6      /* spans per Block */
#define XZAMD_DEBUG_LITP 8          /* literal-coder slices of the span slots (timing builds leave a per-span record there) */
#define XZAMD_DEBUG_SPAN_EST 7      /* per 4 KiB chunk: work estimates of every Block, then the bit estimates */
#define XZAMD_DEBUG_SYM_LEN 9       /* two-phase: u16 per position, valid at symbol starts (0 = literal) */
#define XZAMD_DEBUG_SYM_DIST 10     /* two-phase: u32 per position (distance / literal bytes) */
#define XZAMD_DEBUG_ENC_TAB 11      /* two-phase: (first byte, end) per encode-span slot, slots = Block * (block_size / 256 KiB + 1) + j */
#define XZAMD_DEBUG_ENC_CNT 12      /* encode spans per Block */
#define XZAMD_DEBUG_PINFO 13        /* two-phase: 16 x u32 per piece slot ([0..7] iteration 1, [8..15] iteration 2: state | ok << 31, rep distances, raw) */
#define XZAMD_DEBUG_SNAP_SR 14      /* 8 x u32 per piece slot: state and rep distances a piece starts iteration 2 with */
#define XZAMD_DEBUG_PRIOR 15        /* 1856 x u32 per piece slot: the non-literal probabilities of the piece's price model (after the run: as adapted) */
#define XZAMD_DEBUG_CARRY 16        /* the coder's walk, per encode-span slot: 0 = reset + properties, 1 = carried, 2 = flat start */
#define XZAMD_DEBUG_CB_HDR 17       /* the coder's bounds walk, per encode-span slot: XZAMD_CB_* flags */
#define XZAMD_DEBUG_CB_START 18     /* the coder's walk: the model at the span start, u16 per probability (padded to 64) */
int xzamd_debug_fetch(xzamd_ctx *ctx, int what, void *host_out, uint64_t bytes);

/* Seeded synthetic corpora used by bench.py and the tests (host memory). */
void xzamd_corpus_lorem(uint8_t *out, uint64_t n);                 /* tests/create_compress_files.c:110-152 continued */
void xzamd_corpus_text(uint8_t *out, uint64_t n, uint64_t seed, int threads);   /* Zipf/Markov "enwik-style" */
/* ustar stream of the source trees under `roots` (':'-separated directories of the box,
 * walked in strcmp order), cycled to n bytes with a per-cycle byte-level perturbation.  Returns the number of
 * files in one cycle (0: nothing readable, out is zero-filled). */
uint64_t xzamd_corpus_tar(uint8_t *out, uint64_t n, const char *roots, uint64_t seed);
#define XZAMD_TAR_ROOTS "/opt/rocm/include:/usr/include:/usr/lib/python3:/usr/lib/python3.10"

const char *xzamd_version(void);

#ifdef __cplusplus
}
#endif
#endif
