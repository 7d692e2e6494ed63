// lzma_decode.hip -- MI355X (gfx950) LZMA2 Block decoder: the "other side" of the encode path
// (SURVEY.md 8f.3).  Replaces, for every Block of an .xz Stream resident in HBM, what a worker of the
// reference's threaded decoder runs (common/stream_decoder_mt.c -> block_decoder.c ->
//   lzma/lzma2_decoder.c:54-232   chunk grammar, reset rules
//   lzma/lzma_decoder.c:286-700   symbol decoding, state machine, literal / length / distance trees
//   rangecoder/range_decoder.h    rc init (5 bytes, first must be 0), normalisation, "code == 0" at chunk end).
//
// Parallelism.  LZMA decoding is a serial bit-by-bit recurrence, so a decode unit is one wavefront whose
// control flow is wave-uniform (the scalar unit decodes, the 64 lanes copy match bytes).  Units:
//   * any Stream:            one unit per Block (Blocks never reference each other, doc/faq.txt:156-196);
//   * verification decode:   the caller also passes the ORIGINAL data.  Every LZMA2 chunk that resets the
//                            coder state (control >= 0xA0: our span starts) begins an independent unit,
//                            because match bytes that reach back before the unit are read from the
//                            original instead of from a neighbour's unfinished output.  The decoded bytes
//                            are written out and compared with the original afterwards, so a stream that
//                            only decodes correctly "by luck of the oracle" cannot pass.
// k_dec_scan walks the chunk headers of every Block (grammar checks of lzma2_decoder.c:67-127) and lists
// the unit starts; k_dec_units decodes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels_api.h"

namespace {

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ void wave_sync()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

enum : uint32_t {      // error codes (xzamd_dec_block.error)
    DEC_OK = 0, DEC_BAD_CONTROL = 1, DEC_NEED_DICT_RESET = 2, DEC_NEED_PROPS = 3, DEC_BAD_PROPS = 4,
    DEC_TRUNCATED = 5, DEC_SIZE_MISMATCH = 6, DEC_BAD_DISTANCE = 7, DEC_RC_INIT = 8, DEC_RC_END = 9,
    DEC_CHUNK_OVERRUN = 10, DEC_UNIT_OVERFLOW = 11, DEC_MISMATCH = 12
};

// One thread per Block: walk the LZMA2 chunk chain, validate the grammar, list the units.
__global__ __launch_bounds__(64) void k_dec_scan(const uint8_t* __restrict__ xz, xzamd_dec_block* __restrict__ blocks,
        uint32_t nblocks, xzamd_dec_unit* __restrict__ units, uint32_t units_cap, int split)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    xzamd_dec_block& B = blocks[b];
    const uint8_t* p = xz + B.cpos;
    uint64_t c = 0, u = 0;
    uint32_t nunits = 0, err = DEC_OK;
    uint64_t dbase = 0;                     // Block offset of the last dictionary reset
    bool need_dict_reset = true, need_props = true, ended = false;
    xzamd_dec_unit* U = units + (uint64_t)b * units_cap;
    while (c < B.csize) {
        const uint32_t ctl = p[c];
        if (ctl == 0x00) { ++c; ended = true; break; }
        bool starts_unit = nunits == 0;
        uint64_t hs, cs, us;
        if (ctl >= 0x80) {
            if (c + 5 > B.csize) { err = DEC_TRUNCATED; break; }
            us = ((((uint64_t)ctl & 0x1F) << 16) | ((uint64_t)p[c + 1] << 8) | p[c + 2]) + 1;
            cs = (((uint64_t)p[c + 3] << 8) | p[c + 4]) + 1;
            hs = 5;
            if (ctl >= 0xE0) { need_dict_reset = false; dbase = u; }
            if (need_dict_reset) { err = DEC_NEED_DICT_RESET; break; }
            if (ctl >= 0xC0) {
                if (c + 6 > B.csize) { err = DEC_TRUNCATED; break; }
                if (p[c + 5] > (4 * 5 + 4) * 9 + 8) { err = DEC_BAD_PROPS; break; }
                hs = 6;
                need_props = false;
            }
            if (need_props) { err = DEC_NEED_PROPS; break; }
            if (ctl >= 0xC0 && split) starts_unit = true;      // carries the properties and resets the state: self-contained
        } else if (ctl == 0x01 || ctl == 0x02) {
            if (c + 3 > B.csize) { err = DEC_TRUNCATED; break; }
            if (ctl == 0x01) { need_dict_reset = false; need_props = true; dbase = u; }     // lzma2_decoder.c:121-127
            if (need_dict_reset) { err = DEC_NEED_DICT_RESET; break; }
            us = (((uint64_t)p[c + 1] << 8) | p[c + 2]) + 1;
            cs = us;
            hs = 3;
        } else { err = DEC_BAD_CONTROL; break; }
        if (c + hs + cs > B.csize || u + us > B.usize) { err = DEC_SIZE_MISMATCH; break; }
        if (starts_unit) {
            if (nunits == units_cap) { err = DEC_UNIT_OVERFLOW; break; }
            U[nunits].cpos = B.cpos + c;
            U[nunits].upos = u;
            U[nunits].block = b;
            U[nunits].dbase = (uint32_t)dbase;
            ++nunits;
        }
        c += hs + cs;
        u += us;
    }
    if (err == DEC_OK && (!ended || c != B.csize || u != B.usize)) err = DEC_SIZE_MISMATCH;
    B.nunits = nunits;
    B.error = err;
}

// ---- range decoder + LZMA symbol decoder, wave-uniform -------------------------------------------------
enum : uint32_t {
    D_IS_MATCH = 0, D_IS_REP = 192, D_IS_REP0 = 204, D_IS_REP1 = 216, D_IS_REP2 = 228, D_IS_REP0_LONG = 240,
    D_DIST_SLOT = 432, D_DIST_SPECIAL = 688, D_DIST_ALIGN = 802, D_MATCH_LEN = 818,
    L_CHOICE = 0, L_CHOICE2 = 1, L_LOW = 2, L_MID = 2 + 128, L_HIGH = 2 + 256, L_SIZE = 2 + 256 + 256,
    D_REP_LEN = D_MATCH_LEN + L_SIZE, D_TOTAL = D_REP_LEN + L_SIZE          // 1846
};

struct Rd {
    const uint8_t* in;      // chunk payload
    uint32_t pos, end;      // next byte / payload size
    uint32_t range, code;
    bool over;              // read past the payload
};

__device__ __forceinline__ uint32_t rd_byte(Rd& r)
{
    if (r.pos >= r.end) { r.over = true; return 0; }
    return uni(r.in[r.pos++]);
}
__device__ __forceinline__ void rd_norm(Rd& r)
{
    if (r.range < (1u << 24)) { r.range <<= 8; r.code = (r.code << 8) | rd_byte(r); }
}
__device__ __forceinline__ uint32_t rd_bit(Rd& r, uint16_t* probs, uint32_t idx)
{
    rd_norm(r);
    uint32_t p = uni(probs[idx]);
    const uint32_t bound = (r.range >> 11) * p;
    uint32_t bit;
    if (r.code < bound) { r.range = bound; p += (2048 - p) >> 5; bit = 0; }
    else { r.range -= bound; r.code -= bound; p -= p >> 5; bit = 1; }
    if (threadIdx.x == 0) probs[idx] = (uint16_t)p;
    wave_sync();
    return bit;
}
__device__ __forceinline__ uint32_t rd_tree(Rd& r, uint16_t* probs, uint32_t base, uint32_t nbits)
{
    uint32_t m = 1;
    for (uint32_t i = 0; i < nbits; ++i) m = (m << 1) | rd_bit(r, probs, base + m);
    return m - (1u << nbits);
}
__device__ __forceinline__ uint32_t rd_tree_rev(Rd& r, uint16_t* probs, uint32_t base, uint32_t nbits)
{
    uint32_t m = 1, sym = 0;
    for (uint32_t i = 0; i < nbits; ++i) {
        const uint32_t b = rd_bit(r, probs, base + m);
        m = (m << 1) | b;
        sym |= b << i;
    }
    return sym;
}
__device__ __forceinline__ uint32_t rd_direct(Rd& r, uint32_t nbits)
{
    uint32_t v = 0;
    for (uint32_t i = 0; i < nbits; ++i) {
        rd_norm(r);
        r.range >>= 1;
        const uint32_t t = (r.code >= r.range) ? 1u : 0u;
        if (t) r.code -= r.range;
        v = (v << 1) | t;
    }
    return v;
}
__device__ __forceinline__ uint32_t rd_len(Rd& r, uint16_t* probs, uint32_t base, uint32_t ps)
{
    if (!rd_bit(r, probs, base + L_CHOICE)) return 2 + rd_tree(r, probs, base + L_LOW + ps * 8, 3);
    if (!rd_bit(r, probs, base + L_CHOICE2)) return 10 + rd_tree(r, probs, base + L_MID + ps * 8, 3);
    return 18 + rd_tree(r, probs, base + L_HIGH, 8);
}

// literal-coder probabilities in global memory (one slice per resident wavefront): uniform accesses
__device__ __forceinline__ uint32_t rd_bit_g(Rd& r, uint16_t* g, uint32_t idx)
{
    rd_norm(r);
    uint32_t p = uni(__hip_atomic_load(g + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const uint32_t bound = (r.range >> 11) * p;
    uint32_t bit;
    if (r.code < bound) { r.range = bound; p += (2048 - p) >> 5; bit = 0; }
    else { r.range -= bound; r.code -= bound; p -= p >> 5; bit = 1; }
    if (threadIdx.x == 0) __hip_atomic_store(g + idx, (uint16_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return bit;
}

__device__ __forceinline__ uint8_t dict_byte(const uint8_t* src, uint64_t off)
{
    return __hip_atomic_load(src + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Decode the units [u0, u1) of one Block in order.  `hist` = where bytes before the current position are
// read from: the output itself (plain decode) or the original data (verification decode).
__device__ __noinline__ void decode_units(const uint8_t* __restrict__ xz, const xzamd_dec_block& B, const xzamd_dec_unit* U,
        uint32_t u0, uint32_t u1, uint8_t* __restrict__ out, const uint8_t* hist, bool plain, uint16_t* probs, uint16_t* lit,
        uint32_t* err_out)
{
    const uint32_t lane = threadIdx.x;
    uint8_t* const bout = out + B.upos;                 // this Block's output
    const uint8_t* const bhist = hist + B.upos;         // this Block's history source
    uint32_t err = DEC_OK;
    uint32_t lc = 0, lp = 0, pb = 0, state = 0, rep0 = 0, rep1 = 0, rep2 = 0, rep3 = 0;
    for (uint32_t ui = u0; ui < u1 && err == DEC_OK; ++ui) {
        uint64_t c = U[ui].cpos - B.cpos;               // offset of the unit's first chunk inside the Block's data
        uint64_t u = U[ui].upos;
        uint64_t dbase = U[ui].dbase;                   // bytes before it are outside the dictionary (lz_decoder.h:195-198)
        const uint64_t c_end = ui + 1 < B.nunits ? U[ui + 1].cpos - B.cpos : B.csize;
        const uint8_t* p = xz + B.cpos;
        while (c < c_end && err == DEC_OK) {
            const uint32_t ctl = uni(p[c]);
            if (ctl == 0x00) { ++c; break; }
            if (ctl == 0x01 || ctl >= 0xE0) dbase = u;      // dictionary reset
            if (ctl < 0x80) {
                // uncompressed chunk: copy
                const uint32_t us = ((uni(p[c + 1]) << 8) | uni(p[c + 2])) + 1;
                for (uint32_t i = lane; i < us; i += 64) bout[u + i] = p[c + 3 + i];
                if (plain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                c += 3 + us; u += us;
                continue;
            }
            const uint32_t us = (((ctl & 0x1F) << 16) | (uni(p[c + 1]) << 8) | uni(p[c + 2])) + 1;
            const uint32_t cs = ((uni(p[c + 3]) << 8) | uni(p[c + 4])) + 1;
            uint32_t hs = 5;
            if (ctl >= 0xC0) {
                uint32_t d = uni(p[c + 5]);
                pb = d / 45; d -= pb * 45; lp = d / 9; lc = d - lp * 9;
                if (lc + lp > 4) { err = DEC_BAD_PROPS; break; }
                hs = 6;
            }
            if (ctl >= 0xA0) {
                // state reset (lzma_decoder.c:1052-1100)
                for (uint32_t i = lane; i < D_TOTAL; i += 64) probs[i] = 1024;
                const uint32_t nl = 0x300u << (lc + lp);
                for (uint32_t i = lane; i < nl; i += 64) lit[i] = 1024;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                wave_sync();
                state = 0; rep0 = rep1 = rep2 = rep3 = 0;
            }
            Rd r;
            r.in = p + c + hs; r.pos = 0; r.end = cs; r.over = false;
            // rc init: range_decoder.h:85-100 (first byte must be 0, then 4 bytes of code)
            if (cs < 5 || uni(r.in[0]) != 0) { err = DEC_RC_INIT; break; }
            r.pos = 1; r.range = 0xFFFFFFFFu; r.code = 0;
            for (int i = 0; i < 4; ++i) r.code = (r.code << 8) | rd_byte(r);
            const uint64_t u_end = u + us;
            const uint32_t pbm = (1u << pb) - 1, lpm = (1u << lp) - 1;
            while (u < u_end) {
                const uint32_t ps = (uint32_t)(u - dbase) & pbm;      // positions count from the last dictionary reset
                if (!rd_bit(r, probs, D_IS_MATCH + state * 16 + ps)) {
                    // literal (lzma_decoder.c:330-400)
                    const uint32_t prev = u > dbase ? (plain ? dict_byte(bout, u - 1) : bhist[u - 1]) : 0u;
                    uint16_t* sub = lit + 0x300u * ((((uint32_t)(u - dbase) & lpm) << lc) + (uni(prev) >> (8 - lc)));
                    uint32_t sym = 1;
                    if (state < 7) {
                        do { sym = (sym << 1) | rd_bit_g(r, sub, sym); } while (sym < 0x100);
                    } else {
                        uint32_t mb = uni(plain ? dict_byte(bout, u - rep0 - 1) : bhist[u - rep0 - 1]);
                        uint32_t off = 0x100;
                        do {
                            mb <<= 1;
                            const uint32_t mbit = mb & off;
                            const uint32_t b = rd_bit_g(r, sub, off + mbit + sym);
                            sym = (sym << 1) | b;
                            off &= b ? mbit : ~mbit;
                        } while (sym < 0x100);
                    }
                    if (lane == 0) bout[u] = (uint8_t)sym;
                    if (plain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    state = state <= 3 ? 0 : (state <= 9 ? state - 3 : state - 6);
                    ++u;
                    continue;
                }
                uint32_t len;
                if (rd_bit(r, probs, D_IS_REP + state)) {
                    if (!rd_bit(r, probs, D_IS_REP0 + state)) {
                        if (!rd_bit(r, probs, D_IS_REP0_LONG + state * 16 + ps)) {
                            // short rep
                            state = state < 7 ? 9 : 11;
                            len = 1;
                            goto copy;
                        }
                    } else {
                        uint32_t d;
                        if (!rd_bit(r, probs, D_IS_REP1 + state)) { d = rep1; }
                        else {
                            if (!rd_bit(r, probs, D_IS_REP2 + state)) { d = rep2; }
                            else { d = rep3; rep3 = rep2; }
                            rep2 = rep1;
                        }
                        rep1 = rep0;
                        rep0 = d;
                    }
                    state = state < 7 ? 8 : 11;
                    len = rd_len(r, probs, D_REP_LEN, ps);
                } else {
                    rep3 = rep2; rep2 = rep1; rep1 = rep0;
                    len = rd_len(r, probs, D_MATCH_LEN, ps);
                    state = state < 7 ? 7 : 10;
                    const uint32_t ds = len < 6 ? len - 2 : 3;
                    const uint32_t slot = rd_tree(r, probs, D_DIST_SLOT + ds * 64, 6);
                    if (slot < 4) rep0 = slot;
                    else {
                        const uint32_t fb = (slot >> 1) - 1;
                        rep0 = (2 | (slot & 1)) << fb;
                        if (slot < 14) rep0 += rd_tree_rev(r, probs, D_DIST_SPECIAL + rep0 - slot - 1, fb);
                        else {
                            rep0 += rd_direct(r, fb - 4) << 4;
                            rep0 += rd_tree_rev(r, probs, D_DIST_ALIGN, 4);
                        }
                    }
                }
            copy:
                // dict_is_distance_valid (lzma_decoder.c:530,549): inside the data so far and the dictionary
                if ((uint64_t)rep0 >= u - dbase || rep0 >= B.dict_size) { err = DEC_BAD_DISTANCE; break; }
                if (u + len > u_end) { err = DEC_CHUNK_OVERRUN; break; }
                {
                    const uint32_t period = rep0 + 1;
                    for (uint32_t i = lane; i < len; i += 64) {
                        // bytes the match copies from itself repeat with period rep0 + 1
                        const uint64_t so = u - period + (i % period);
                        bout[u + i] = plain ? dict_byte(bout, so) : bhist[so];
                    }
                    if (plain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                u += len;
            }
            if (err != DEC_OK) break;
            // chunk end: all payload used, code back to 0 (range_decoder.h:139-140, lzma2_decoder.c:196-212)
            rd_norm(r);
            if (r.over) { err = DEC_TRUNCATED; break; }
            if (r.pos != r.end || r.code != 0) { err = DEC_RC_END; break; }
            c += hs + cs;
        }
    }
    if (err != DEC_OK && lane == 0) atomicCAS(err_out, 0u, err);
}

// mode 0: one wavefront per Block; mode 1 (verification): one wavefront per unit, history = original data
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_dec_units(const uint8_t* __restrict__ xz, const xzamd_dec_block* __restrict__ blocks,
        uint32_t nblocks, const xzamd_dec_unit* __restrict__ units, uint32_t units_cap, const uint32_t* __restrict__ unit_first,
        uint32_t total_units, uint8_t* __restrict__ out, const uint8_t* __restrict__ expected,
        uint16_t* __restrict__ lit_pool, uint32_t* __restrict__ counter, uint32_t* __restrict__ block_err)
{
    __shared__ uint16_t probs[D_TOTAL + 2];
    uint16_t* lit = lit_pool + (uint64_t)blockIdx.x * (0x300u << 4);
    const uint32_t total = expected ? total_units : nblocks;
    for (;;) {
        uint32_t w = 0;
        if (threadIdx.x == 0) w = atomicAdd(counter, 1u);
        w = uni(w);
        if (w >= total) break;
        uint32_t bi = w, k0 = 0, k1 = 0;
        if (expected) {
            // unit w of the flattened list: find its Block (unit_first = exclusive prefix sum of nunits)
            uint32_t lo = 0, hi = nblocks;
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (unit_first[mid] <= w) lo = mid; else hi = mid; }
            bi = lo;
            k0 = w - unit_first[lo];
            k1 = k0 + 1;
        } else {
            k1 = blocks[w].nunits;
        }
        const xzamd_dec_block& B = blocks[bi];
        if (B.error == DEC_OK)
            decode_units(xz, B, units + (uint64_t)bi * units_cap, k0, k1, out, expected ? expected : out, expected == nullptr,
                    probs, lit, block_err + bi);
        __builtin_amdgcn_s_waitcnt(0);
        wave_sync();
    }
}

// verification: count bytes that differ from the original
__global__ __launch_bounds__(256) void k_dec_compare(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint64_t n,
        unsigned long long* __restrict__ mismatches)
{
    uint64_t bad = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 16;
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n; i += stride) {
        if (i + 16 <= n) {
            uint4 x, y;
            __builtin_memcpy(&x, a + i, 16);
            __builtin_memcpy(&y, b + i, 16);
            bad += (x.x != y.x) + (x.y != y.y) + (x.z != y.z) + (x.w != y.w);
        } else {
            for (uint64_t k = i; k < n; ++k) bad += a[k] != b[k];
        }
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

} // namespace

extern "C" {

int xzk_dec_scan(const uint8_t* d_xz, xzamd_dec_block* d_blocks, uint32_t nblocks, xzamd_dec_unit* d_units,
        uint32_t units_cap, int split, void* stream_)
{
    if (nblocks == 0) return 0;
    hipLaunchKernelGGL(k_dec_scan, dim3((nblocks + 63) / 64), dim3(64), 0, (hipStream_t)stream_, d_xz, d_blocks, nblocks,
            d_units, units_cap, split);
    return (int)hipGetLastError();
}

int xzk_dec_units(const uint8_t* d_xz, const xzamd_dec_block* d_blocks, uint32_t nblocks, const xzamd_dec_unit* d_units,
        uint32_t units_cap, const uint32_t* d_unit_first, uint32_t total_units, uint8_t* d_out, const uint8_t* d_expected,
        uint16_t* d_lit_pool, uint32_t waves, uint32_t* d_counter, uint32_t* d_block_err, void* stream_)
{
    const uint32_t total = d_expected ? total_units : nblocks;
    if (total == 0) return 0;
    const uint32_t grid = waves < total ? waves : total;
    hipLaunchKernelGGL(k_dec_units, dim3(grid), dim3(64), 0, (hipStream_t)stream_, d_xz, d_blocks, nblocks, d_units, units_cap,
            d_unit_first, total_units, d_out, d_expected, d_lit_pool, d_counter, d_block_err);
    return (int)hipGetLastError();
}

int xzk_dec_compare(const uint8_t* a, const uint8_t* b, uint64_t n, unsigned long long* d_mismatches, void* stream_)
{
    if (n == 0) return 0;
    uint64_t g = (n / 16 + 255) / 256;
    if (g > 65536) g = 65536;
    if (g == 0) g = 1;
    hipLaunchKernelGGL(k_dec_compare, dim3((uint32_t)g), dim3(256), 0, (hipStream_t)stream_, a, b, n, d_mismatches);
    return (int)hipGetLastError();
}

} // extern "C"
