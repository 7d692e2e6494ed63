// lzma_kernels.hip -- MI355X (gfx950 / CDNA4) device side of the LZMA2 Block encoder.
//
// Replaces, for a whole batch of .xz Blocks resident in HBM, what one worker
// thread of the reference runs per Block (stream_encoder_mt.c:219-298):
//
//   lz/lz_encoder_mf.c      HC3/HC4 match finder      -> k_h2_* / k_hash_keys + radix sort + k_link_*
//                                                       (Block-global "sorted bucket" chains)
//                                                       + wave-parallel candidate evaluation (do_round)
//   lz/lz_encoder_mf.c      BT4 (presets 4-9)         -> suffix-neighbourhood finder: k_sa_* (suffix order of every
//                                                       Block by radix sorts + rank doubling), k_find_sn
//   lzma/lzma_encoder_optimum_fast.c    parser        -> fast parser block of span_encode_one (wave-uniform)
//   lzma/lzma_encoder_optimum_normal.c  parser        -> optimum_window() (windowed DP, lane = length / candidate)
//   lzma/lzma_encoder.c     symbol coder               -> encode_symbol()
//   rangecoder/range_encoder.h  range coder            -> struct RC
//   lzma/lzma2_encoder.c    chunk framing              -> span_encode_one epilogue
//   simple/*.c, delta/      filters in front of LZMA2  -> k_x86_bcj, k_riscv_bcj, k_arm64_bcj, k_bcj_simple, k_delta
//   check/crc64_fast.c, crc32, sha256.c                -> k_crc_strips / k_crc_fold, k_sha256_blocks
//   block_encoder.c / stream_encoder_mt.c assembly     -> k_assemble (gather of span outputs)
//
// Design: the sequential insert-then-search structures of the
// reference are replaced by parse-independent ones built in parallel.  Hash
// chains: every position's masked hash is radix-sorted (key = block<<hash_bits
// | hash, stable, so positions ascend inside a bucket); the `depth` slots before
// a position's slot ARE the hash chain the reference would walk (same candidates,
// same order, collisions included), so one coalesced read fetches the whole
// chain and 64 lanes evaluate all candidates at once.  BT4's tree is the
// Cartesian tree of (suffix order, insertion time): its search path is the set
// of recency records around a position in suffix order, which a sort provides.
// Nothing is scattered: whatever a sort produces is brought back to position
// order by two radix passes and one LDS step (invert_perm).  The LZMA state
// machine + range coder is serial by construction; parallelism there comes from
// cutting each Block into spans that are entropy-coded independently (LZMA2
// state-reset chunks), one wavefront per span, probability model in LDS.
//
// No MFMA: there is no dense contraction anywhere on this path.

#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <stdint.h>
#include "kernels_api.h"

namespace {

constexpr uint32_t MATCH_LEN_MAX = 273;
constexpr uint32_t LITERAL = 0xFFFFFFFFu;
constexpr uint32_t NO_DELTA = 0xFFFFFFFFu;

// Probability model layout (u16 each), 7990 entries at lc+lp<=... we size for lc+lp <= 3
// (8 literal coders = 6144) which covers every preset (lc=3, lp=0).  Same order as the oracle.
enum : uint32_t {
    P_IS_MATCH = 0,
    P_IS_REP = P_IS_MATCH + 12 * 16,
    P_IS_REP0 = P_IS_REP + 12,
    P_IS_REP1 = P_IS_REP0 + 12,
    P_IS_REP2 = P_IS_REP1 + 12,
    P_IS_REP0_LONG = P_IS_REP2 + 12,
    P_DIST_SLOT = P_IS_REP0_LONG + 12 * 16,
    P_DIST_SPECIAL = P_DIST_SLOT + 4 * 64,
    P_DIST_ALIGN = P_DIST_SPECIAL + 114,
    P_MATCH_LEN = P_DIST_ALIGN + 16,
    LEN_CHOICE = 0, LEN_CHOICE2 = 1, LEN_LOW = 2, LEN_MID = 2 + 16 * 8, LEN_HIGH = 2 + 32 * 8,
    LEN_CODER_SIZE = 2 + 32 * 8 + 256,
    P_REP_LEN = P_MATCH_LEN + LEN_CODER_SIZE,
    P_LITERAL = P_REP_LEN + LEN_CODER_SIZE,       // 1846
    P_TOTAL = P_LITERAL + (0x300 << 3)            // 7990
};
static_assert(P_TOTAL <= 8192, "model must fit the 16 KiB LDS slice");

// Make LDS stores of some lanes visible to later LDS loads of other lanes of the same wavefront:
// LDS hand-off between lanes of the wave.  DS instructions of one wave execute in issue order, so a
// ds_write is visible to any later ds_read of the same wave without waiting; what has to be
// prevented is the COMPILER moving accesses across the hand-off.  A compiler-only barrier: a fence
// builtin would also drain vmcnt, i.e. wait for every prefetch in flight at each hand-off.
__device__ __forceinline__ void wave_sync()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// Sum over the 64 lanes on the DPP network (no LDS traffic): row-wise shifts, then the two row broadcasts; lane 63 has it.
__device__ __forceinline__ uint32_t wave_sum_dpp(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);     // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);     // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);     // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);     // row_shr:8   -> lane 15 of a row = row sum
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true);     // row_bcast:15 into rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true);     // row_bcast:31 into rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// A wave-uniform value the compiler must keep in a scalar register from here on (no instruction is emitted when it is
// there already): without the pin a loop-carried uniform chain can end up on the vector ALU as a whole.
__device__ __forceinline__ void pin_s(uint32_t& v) { asm volatile("" : "+s"(v)); }
__device__ __forceinline__ void pin_s(uint64_t& v) { asm volatile("" : "+s"(v)); }
__device__ __forceinline__ uint32_t lane_of(uint32_t v, uint32_t l) { return __builtin_amdgcn_readlane(v, uni(l)); }

// CRC32 table[0] entry for one byte (lz_encoder_hash.h:31-39 uses lzma_crc32_table[0]).
__device__ __forceinline__ uint32_t crc_t0(uint32_t b)
{
    uint32_t r = b;
#pragma unroll
    for (int k = 0; k < 8; ++k) r = (r >> 1) ^ ((r & 1) ? 0xEDB88320u : 0u);
    return r;
}

// ------------------------------------------------------------------------------------------
// Match-finder structure build
// ------------------------------------------------------------------------------------------

// One thread per input byte: one of the hash keys of lz_encoder_hash.h:55-75 (which = 2: hash2,
// 3: hash3 of HC4, 0: the main chain hash), prefixed with the Block number so one sort serves
// every Block of the batch.  Positions with fewer than hash_bytes left in their Block are never
// inserted by the reference (lz_encoder_mf.c:190-201 "pending"): they get the sentinel bucket
// `nblocks`.  vals = iota (the position itself).
__global__ __launch_bounds__(256) void k_hash_keys(const uint8_t* __restrict__ in, uint32_t n,
        uint32_t block_size, uint32_t nblocks, uint32_t hash_bytes, uint32_t hash_mask, uint32_t hash_bits,
        uint32_t which, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    __shared__ uint32_t T[256];
    T[threadIdx.x] = crc_t0(threadIdx.x);
    __syncthreads();
    const uint32_t kbits = which == 2 ? 10u : (which == 3 ? 16u : hash_bits);
    const uint32_t need = hash_bytes;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < n; g += stride) {
        const uint32_t b = g / block_size;
        const uint32_t bend = min(n, (b + 1) * block_size);   // n < 2^31, no overflow
        const uint32_t avail = bend - g;
        uint32_t key = nblocks << kbits;
        if (avail >= need) {
            const uint32_t c0 = in[g], c1 = in[g + 1], c2 = in[g + 2];
            const uint32_t temp = T[c0] ^ c1;
            uint32_t h;
            if (which == 2) h = temp & 0x3FF;
            else if (which == 3) h = (temp ^ (c2 << 8)) & 0xFFFF;
            else if (hash_bytes == 3) h = (temp ^ (c2 << 8)) & hash_mask;
            else h = (temp ^ (c2 << 8) ^ (T[in[g + 3]] << 5)) & hash_mask;
            key = (b << kbits) | h;
        }
        keys[g] = key;
        vals[g] = g;
    }
}

// After the stable sort by key_main: publish the bucket order.
//   sorted_pos[i] = position | (first-of-bucket << 31)      rank[position] = i
__global__ __launch_bounds__(256) void k_link_main(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
        uint32_t n, uint32_t* __restrict__ sorted_pos, uint32_t* __restrict__ rank)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t p = vals[i];
        const uint32_t first = (i == 0 || keys[i - 1] != keys[i]) ? 0x80000000u : 0u;
        sorted_pos[i] = p | first;
        rank[p] = i;
    }
}

// After the stable sort by key2 / key3 / key4: prev[position] = distance to the previous position of the same
// bucket (the value the reference's hash head table holds when `position` is reached, expressed as delta),
// 0 = none.
// hash2 needs no sort: its table has 1024 entries per Block, so the "previous position with the same hash" is
// found the way the reference does it -- a head table walked in text order -- with the Block cut into segments of
// H2_SEG positions, one wavefront each:
//   k_h2_last   last inserted position (+1) of every hash value inside the segment      (LDS table, ds_max)
//   k_h2_scan   per Block: exclusive running maximum of those tables over its segments   (the carry-in heads)
//   k_h2_prev   the segment again, 64 positions a step: a lane's predecessor is the highest lower lane of the
//               step with the same hash (ten ballots give every lane its set of peers), else the table entry;
//               the last lane of each peer set then updates the table.
// Positions with fewer than hash_bytes left in their Block are not inserted and get 0 (they are never looked up).
constexpr uint32_t H2_SEG = 65536;

__device__ __forceinline__ uint32_t h2_of(const uint8_t* __restrict__ in, const uint32_t* T, uint32_t p)
{
    return (T[in[p]] ^ in[p + 1]) & 0x3FFu;
}

__global__ __launch_bounds__(64) void k_h2_last(const uint8_t* __restrict__ in, uint32_t n, uint32_t block_size,
        uint32_t segs_per_block, uint32_t hash_bytes, uint32_t* __restrict__ seg_tab)
{
    __shared__ uint32_t T[256];
    __shared__ uint32_t tab[1024];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 256; i += 64) T[i] = crc_t0(i);
    for (uint32_t i = lane; i < 1024; i += 64) tab[i] = 0;
    const uint32_t b = blockIdx.x / segs_per_block, sg = blockIdx.x - b * segs_per_block;
    const uint32_t bs = b * block_size;
    const uint32_t bend = min(n, bs + block_size);
    const uint32_t s0 = bs + sg * H2_SEG;
    const uint32_t s1 = min(bend, s0 + H2_SEG);
    wave_sync();
    for (uint32_t x = s0 + lane; x < s1; x += 64)
        if (bend - x >= hash_bytes) atomicMax(&tab[h2_of(in, T, x)], x + 1);
    wave_sync();
    uint32_t* out = seg_tab + (uint64_t)blockIdx.x * 1024;
    for (uint32_t i = lane; i < 1024; i += 64) out[i] = tab[i];
}

__global__ __launch_bounds__(256) void k_h2_scan(uint32_t* __restrict__ seg_tab, uint32_t segs_per_block)
{
    // thread = (Block, hash value): exclusive running maximum along the Block's segments
    const uint32_t b = blockIdx.x >> 2, h = ((blockIdx.x & 3) << 8) | threadIdx.x;
    uint32_t* t = seg_tab + (uint64_t)b * segs_per_block * 1024 + h;
    uint32_t run = 0;
    for (uint32_t s = 0; s < segs_per_block; ++s) {
        const uint32_t v = t[(uint64_t)s * 1024];
        t[(uint64_t)s * 1024] = run;
        run = max(run, v);
    }
}

__global__ __launch_bounds__(64) void k_h2_prev(const uint8_t* __restrict__ in, uint32_t n, uint32_t block_size,
        uint32_t segs_per_block, uint32_t hash_bytes, const uint32_t* __restrict__ seg_tab, uint32_t* __restrict__ prev2)
{
    __shared__ uint32_t T[256];
    __shared__ uint32_t tab[1024];
    const uint32_t lane = threadIdx.x;
    const uint32_t* carry = seg_tab + (uint64_t)blockIdx.x * 1024;
    for (uint32_t i = lane; i < 256; i += 64) T[i] = crc_t0(i);
    for (uint32_t i = lane; i < 1024; i += 64) tab[i] = carry[i];
    const uint32_t b = blockIdx.x / segs_per_block, sg = blockIdx.x - b * segs_per_block;
    const uint32_t bs = b * block_size;
    const uint32_t bend = min(n, bs + block_size);
    const uint32_t s0 = bs + sg * H2_SEG;
    const uint32_t s1 = min(bend, s0 + H2_SEG);
    const uint64_t below = (1ull << lane) - 1;
    wave_sync();
    for (uint32_t x0 = s0; x0 < s1; x0 += 64) {
        const uint32_t x = x0 + lane;
        const bool ins = x < s1 && bend - x >= hash_bytes;
        const uint32_t h = ins ? h2_of(in, T, x) : 0u;
        uint64_t peers = __builtin_amdgcn_ballot_w64(ins);
#pragma unroll
        for (uint32_t k = 0; k < 10; ++k) {
            const uint64_t m = __builtin_amdgcn_ballot_w64(((h >> k) & 1u) != 0);
            peers &= ((h >> k) & 1u) ? m : ~m;
        }
        const uint32_t head = tab[h];                       // head before this step
        const uint64_t lower = peers & below;
        uint32_t d = 0;
        if (lower) d = lane - (63u - (uint32_t)__builtin_clzll(lower));
        else if (head) d = x + 1 - head;
        if (x < s1) prev2[x] = ins ? d : 0u;
        wave_sync();
        if (ins && (peers >> lane) == 1ull) tab[h] = x + 1;   // last lane of its peer set
        wave_sync();
    }
}

// The same distances, written in sorted order (d[i] belongs to position vals[i]).  A scattered 4-byte store
// costs a whole sector and 1.4 G of them run at ~30 G/s; sorting the (position, distance) pairs back by
// position (invert_perm below: two streaming radix passes and an LDS step) is more than twice as fast.
__global__ __launch_bounds__(256) void k_link_prev_seq(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
        uint32_t n, uint32_t* __restrict__ d)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        d[i] = (i > 0 && keys[i - 1] == keys[i]) ? vals[i] - vals[i - 1] : 0u;
}

__global__ __launch_bounds__(256) void k_iota(uint32_t* __restrict__ v, uint32_t n)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) v[i] = i;
}

// Last step of an inversion (invert_perm below).  The (position, value) pairs arrive sorted by position >> 15;
// the positions are a permutation of 0..n-1, so bucket b is exactly the pairs of positions [b << 15, (b + 1) << 15)
// and sits in exactly those slots.  One workgroup per bucket places the values in LDS by the low 15 bits and
// writes the 128 KiB out linearly: the last 15 key bits cost one read and one coalesced write instead of two
// radix passes.  In place is fine (a bucket reads and writes the same slots, reads first).
constexpr uint32_t INV_LOW = 15;
__global__ __launch_bounds__(1024) void k_inv_low_u32(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
        uint32_t n, uint32_t* __restrict__ out)
{
    __shared__ uint32_t lds_inv[1u << INV_LOW];             // 128 KiB of the 160 KiB a CU has
    const uint32_t base = blockIdx.x << INV_LOW;
    const uint32_t cnt = min(n - base, 1u << INV_LOW);
    for (uint32_t i = threadIdx.x; i < cnt; i += 1024) lds_inv[keys[base + i] & ((1u << INV_LOW) - 1)] = vals[base + i];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cnt; i += 1024) out[base + i] = lds_inv[i];
}

// 64-bit values: the two halves go to two arrays (out_lo, out_hi), one LDS round each
__global__ __launch_bounds__(1024) void k_inv_low_u64(const uint32_t* __restrict__ keys, const uint64_t* __restrict__ vals,
        uint32_t n, uint32_t* __restrict__ out_lo, uint32_t* __restrict__ out_hi)
{
    __shared__ uint32_t lds_inv[1u << INV_LOW];             // 128 KiB of the 160 KiB a CU has
    const uint32_t base = blockIdx.x << INV_LOW;
    const uint32_t cnt = min(n - base, 1u << INV_LOW);
    for (uint32_t i = threadIdx.x; i < cnt; i += 1024) lds_inv[keys[base + i] & ((1u << INV_LOW) - 1)] = (uint32_t)vals[base + i];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cnt; i += 1024) out_lo[base + i] = lds_inv[i];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cnt; i += 1024) lds_inv[keys[base + i] & ((1u << INV_LOW) - 1)] = (uint32_t)(vals[base + i] >> 32);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cnt; i += 1024) out_hi[base + i] = lds_inv[i];
}

// ------------------------------------------------------------------------------------------
// Suffix order of every Block by its first 32 bytes (oracle: build_sa) -- the structure behind the
// suffix-neighbourhood finder.  Order inside a Block: four 8-byte chunks compared as big-endian
// numbers (bytes past the Block end read as zero, a chunk that starts past the end sorts lowest),
// ties by position.  Built by stable LSD radix sorts (rocprim onesweep, HBM-bound):
//   round 0   sort (chunk(p), p), then stably by Block number (so the slots of a Block are exactly
//             its positions' range);
//   round h   h = 8, 16: rank[p] = 1 + first slot of p's group of equal keys; sort by
//             (rank[p], rank[p + h]) (0 = past the Block end): doubles the compared prefix.
// Group starts come from a max-scan over "slot if the key differs from its left neighbour".
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sa_chunk_keys(const uint8_t* __restrict__ in, uint32_t n, uint32_t block_size,
        uint64_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < n; g += stride) {
        const uint32_t b = g / block_size;
        const uint32_t bend = min(n, (b + 1) * block_size);
        const uint32_t avail = bend - g;
        uint64_t v = 0;
        if (avail >= 8) {
            uint64_t t;
            __builtin_memcpy(&t, in + g, 8);
            v = __builtin_bswap64(t);
        } else {
            for (uint32_t i = 0; i < avail; ++i) v |= (uint64_t)in[g + i] << (56 - 8 * i);
        }
        keys[g] = v;
        vals[g] = g;
    }
}

// grp[i] = i where the 64-bit key differs from its left neighbour (or, with block_size != 0, where a Block
// starts: the slots of a Block are its positions' range), else 0 (input of the max-scan)
__global__ __launch_bounds__(256) void k_sa_flags64(const uint64_t* __restrict__ keys, uint32_t n, uint32_t block_size,
        uint32_t* __restrict__ grp)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const bool first = i != 0 && (keys[i] != keys[i - 1] || (block_size != 0 && i % block_size == 0));
        grp[i] = first ? i : 0u;
    }
}

// round 0, second step: pack (position, chunk group) as the value, Block number as the key
__global__ __launch_bounds__(256) void k_sa_block_keys(const uint32_t* __restrict__ pos, const uint32_t* __restrict__ grp,
        uint32_t n, uint32_t block_size, uint32_t* __restrict__ bkeys, uint64_t* __restrict__ bvals)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t p = pos[i];
        bkeys[i] = p / block_size;
        bvals[i] = (uint64_t)p | ((uint64_t)grp[i] << 32);
    }
}

// after the Block sort: positions out, group flags from (Block, chunk group)
__global__ __launch_bounds__(256) void k_sa_block_unpack(const uint32_t* __restrict__ bkeys, const uint64_t* __restrict__ bvals,
        uint32_t n, uint32_t* __restrict__ pos, uint32_t* __restrict__ grp)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t v = bvals[i];
        pos[i] = (uint32_t)v;
        const bool first = i == 0 || bkeys[i] != bkeys[i - 1] || (uint32_t)(bvals[i - 1] >> 32) != (uint32_t)(v >> 32);
        grp[i] = (first && i != 0) ? i : 0u;
    }
}

// Slot order: rk[i] = (group start + 1, distance to the left neighbour inside the group or 0) of position pos[i].
// The second word is a by-product of the sort round: inside a group of equal keys positions ascend, so the left
// neighbour of a group member is the nearest earlier position with the same 8 (round 0) / 16 (round 1) bytes.
// The pairs are then brought to position order (invert_perm): two arrays indexed by position.
__global__ __launch_bounds__(256) void k_sa_rank_seq(const uint32_t* __restrict__ pos, const uint32_t* __restrict__ grp,
        uint32_t n, uint2* __restrict__ rk)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t g = grp[i];
        rk[i] = make_uint2(g + 1, g != i ? pos[i] - pos[i - 1] : 0u);
    }
}

// the same for the rounds that only want the rank (sa_depth > 32)
__global__ __launch_bounds__(256) void k_sa_rank_only_seq(const uint32_t* __restrict__ grp, uint32_t n, uint32_t* __restrict__ rk)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) rk[i] = grp[i] + 1;
}

// ---- rank doubling on the unresolved slots only -------------------------------------------------------------
// After the 16-byte round most positions of ordinary data are alone in their group (text: 88 %, after 32 bytes
// 99.95 %): their slot is final.  A round then only has to order the slots that still share a group with another one:
//   k_sa_unres     u[i] = 1 when slot i lies in a group of >= 2 slots            (exclusive scan -> compact index)
//   k_sa_compact   (key, position, slot) of the unresolved slots, in slot order; key = (rank, rank of p + h) as
//                  k_sa_pair_keys_pos makes it
//   radix sort of those m elements (stable: equal keys keep ascending positions, as in the full round)
//   k_sa_newgrp    group starts of the sorted elements (max-scan over "slot where the key changes")
//   k_sa_writeback the j-th sorted element goes to the j-th unresolved slot (the groups are contiguous slot ranges in
//                  ascending order, so the sorted sequence enumerates them in place); position, group start and the
//                  by-position rank of the moved elements are updated
__global__ __launch_bounds__(256) void k_sa_unres(const uint32_t* __restrict__ grp, uint32_t n, uint32_t* __restrict__ u)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        u[i] = (grp[i] != i || (i + 1 < n && grp[i + 1] == i)) ? 1u : 0u;
}

__global__ void k_sa_count(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ grp, uint32_t n, uint32_t* __restrict__ count)
{
    if (blockIdx.x == 0 && threadIdx.x == 0)
        *count = idx[n - 1] + ((grp[n - 1] != n - 1) ? 1u : 0u);      // the last slot has no right neighbour
}

__global__ __launch_bounds__(256) void k_sa_compact(const uint32_t* __restrict__ pos, const uint32_t* __restrict__ grp,
        const uint32_t* __restrict__ idx, const uint32_t* __restrict__ rank, uint32_t n, uint32_t block_size, uint32_t h,
        uint32_t sbits, uint64_t* __restrict__ ckey, uint32_t* __restrict__ cval, uint32_t* __restrict__ cslot)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t g = grp[i];
        if (!(g != i || (i + 1 < n && grp[i + 1] == i))) continue;
        const uint32_t j = idx[i];
        const uint32_t p = pos[i];
        const uint32_t bs = (p / block_size) * block_size;
        const uint32_t bend = min(n, bs + block_size);
        const uint32_t second = p + h < bend ? rank[p + h] - bs : 0u;
        ckey[j] = ((uint64_t)(g + 1) << sbits) | second;
        cval[j] = p;
        cslot[j] = i;
    }
}

__global__ __launch_bounds__(256) void k_sa_newgrp(const uint64_t* __restrict__ ckey, const uint32_t* __restrict__ cslot,
        uint32_t m, uint32_t* __restrict__ gs)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride)
        gs[j] = (j == 0 || ckey[j] != ckey[j - 1]) ? cslot[j] : 0u;
}

__global__ __launch_bounds__(256) void k_sa_writeback(const uint32_t* __restrict__ cval, const uint32_t* __restrict__ cslot,
        const uint32_t* __restrict__ gs, uint32_t m, uint32_t* __restrict__ pos, uint32_t* __restrict__ grp,
        uint32_t* __restrict__ rank)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
        const uint32_t sl = cslot[j], p = cval[j], g = gs[j];
        pos[sl] = p;
        grp[sl] = g;
        rank[p] = g + 1;
    }
}

// By-product of a sorted round: inside a run of equal keys the positions ascend, so the left neighbour of a member is the
// nearest earlier position of its group; d[position] = distance to it (0: first of its group).  A scatter of m elements.
__global__ __launch_bounds__(256) void k_sa_prev_scatter(const uint64_t* __restrict__ key, const uint32_t* __restrict__ val,
        uint32_t m, uint32_t* __restrict__ d)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
        const uint32_t p = val[j];
        d[p] = (j > 0 && key[j] == key[j - 1]) ? p - val[j - 1] : 0u;
    }
}

// doubling key of every position, in position order: (rank[p], rank[p + h]) with 0 for a second half that
// starts past the Block end; vals = iota.  The second rank is taken relative to the Block (sbits = bits of
// block_size + 1), so the key is 31 + sbits bits wide instead of 62: one radix pass less for Blocks up to 32 MiB.  (The radix sort is stable and the members of a group ascend by
// position in slot order too, so feeding it in position order gives the same result as slot order -- without
// the random gather of rank[p + h].)
__global__ __launch_bounds__(256) void k_sa_pair_keys_pos(const uint32_t* __restrict__ rank, uint32_t n, uint32_t block_size,
        uint32_t h, uint32_t sbits, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        const uint32_t b = p / block_size;
        const uint32_t bs = b * block_size;
        const uint32_t bend = min(n, bs + block_size);
        // the second half lies in the same Block, whose slots are [bs, bend): relative rank 1..block_size
        const uint32_t second = p + h < bend ? rank[p + h] - bs : 0u;
        keys[p] = ((uint64_t)rank[p] << sbits) | second;
        vals[p] = p;
    }
}

// final: sa[i] = position of slot i; slot[i] = i (sorted by position afterwards: sa_rank)
__global__ __launch_bounds__(256) void k_sa_final_seq(const uint32_t* __restrict__ pos, uint32_t n,
        uint32_t* __restrict__ sa, uint32_t* __restrict__ slot)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        sa[i] = pos[i];
        slot[i] = i;
    }
}

// ------------------------------------------------------------------------------------------
// Range coder (rangecoder/range_encoder.h:136-263), coding directly instead of queueing.
// All members are wave-uniform.
// ------------------------------------------------------------------------------------------
struct RC {
    uint64_t low;
    uint32_t range;
    uint32_t cache_size;
    uint32_t cache;
    uint32_t cpos;      // payload bytes written for the current chunk
    uint8_t* out;       // payload base of the current chunk
    // token sink of the two-phase model pass (rc_emit<.., TOK>): the decisions are written out instead of coded
    uint16_t* tok;      // next free token of the encode span
    uint32_t est;       // summed prices of the current chunk's decisions, 1/16 bit
    const uint8_t* ptab;
    uint32_t* log;      // k_model_bounds (rc_emit<.., BND>): the logged bits of the span, XZAMD_LOG_WORDS per probability
    uint32_t log_cap;   // at most this many per probability (XZAMD_LOG_CAP)
#ifdef XZAMD_TIMING
    uint64_t tm_run;    // cycles inside rc_run (profiling builds only)
    uint64_t tm_bits;
#endif

    __device__ __forceinline__ void reset()
    {
        low = 0; range = 0xFFFFFFFFu; cache_size = 1; cache = 0;
    }
    __device__ __forceinline__ void shift_low()
    {
        if ((uint32_t)low < 0xFF000000u || (uint32_t)(low >> 32) != 0) {
            const uint32_t carry = (uint32_t)(low >> 32);
            const uint32_t lane = threadIdx.x;
            // byte 0 = cache + carry, then cache_size-1 bytes of 0xFF + carry
            if (cache_size == 1) {                   // the usual case: one pending byte
                if (lane == 0) out[cpos] = (uint8_t)(cache + carry);
            } else {
                for (uint32_t i = lane; i < cache_size; i += 64)
                    out[cpos + i] = (uint8_t)((i == 0 ? cache : 0xFFu) + carry);
            }
            cpos += cache_size;
            cache_size = 0;
            cache = (uint32_t)(low >> 24) & 0xFF;
        }
        ++cache_size;
        low = (low & 0x00FFFFFFu) << 8;
    }
    __device__ __forceinline__ void normalize()
    {
        if (range < (1u << 24)) { shift_low(); range <<= 8; }
    }
    __device__ __forceinline__ void bit(uint16_t* probs, uint32_t idx, uint32_t b)
    {
        normalize();
        uint32_t p = uni(probs[idx]);
        const uint32_t bound = (range >> 11) * p;
        if (!b) { range = bound; p += (2048 - p) >> 5; }
        else { low += bound; range -= bound; p -= p >> 5; }
        probs[idx] = (uint16_t)p;
    }
    __device__ __forceinline__ void tree(uint16_t* probs, uint32_t base, uint32_t nbits, uint32_t sym)
    {
        uint32_t m = 1;
        do {
            const uint32_t b = (sym >> --nbits) & 1;
            bit(probs, base + m, b);
            m = (m << 1) + b;
        } while (nbits);
    }
    __device__ __forceinline__ void direct(uint32_t value, uint32_t nbits)
    {
        do {
            normalize();
            range >>= 1;
            if ((value >> --nbits) & 1) low += range;
        } while (nbits);
    }
    __device__ __forceinline__ void flush()
    {
        normalize();                     // the queue loop normalizes before the first RC_FLUSH
        for (int i = 0; i < 5; ++i) shift_low();
        reset();
    }
};

// ------------------------------------------------------------------------------------------
// Wave-cooperative byte comparison
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t ld64(const uint8_t* p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// Per-lane: length of the common prefix of in[q..] and in[x..], at most lim.  Lanes diverge.
// 32 bytes per trip: every loop iteration is a dependent HBM/L2 round trip, so the trip count
// (not the byte count) is what a long match costs.
__device__ __forceinline__ uint32_t lane_cmplen(const uint8_t* __restrict__ in, uint32_t q, uint32_t x, uint32_t lim)
{
    uint32_t len = 0;
    while (len + 32 <= lim) {
        const uint8_t* a = in + q + len;
        const uint8_t* b = in + x + len;
        const uint64_t a0 = ld64(a), a1 = ld64(a + 8), a2 = ld64(a + 16), a3 = ld64(a + 24);
        const uint64_t b0 = ld64(b), b1 = ld64(b + 8), b2 = ld64(b + 16), b3 = ld64(b + 24);
        uint64_t d = a0 ^ b0;
        if (d) return len + (uint32_t)(__builtin_ctzll(d) >> 3);
        d = a1 ^ b1;
        if (d) return len + 8 + (uint32_t)(__builtin_ctzll(d) >> 3);
        d = a2 ^ b2;
        if (d) return len + 16 + (uint32_t)(__builtin_ctzll(d) >> 3);
        d = a3 ^ b3;
        if (d) return len + 24 + (uint32_t)(__builtin_ctzll(d) >> 3);
        len += 32;
    }
    while (len + 8 <= lim) {
        const uint64_t d = ld64(in + q + len) ^ ld64(in + x + len);
        if (d) return len + (uint32_t)(__builtin_ctzll(d) >> 3);
        len += 8;
    }
    while (len < lim && in[q + len] == in[x + len]) ++len;
    return len;
}

// Whole wave: extend a known common prefix `len` of in[a..], in[b..] up to lim (64 bytes/step).
__device__ __forceinline__ uint32_t wave_cmplen(const uint8_t* __restrict__ in, uint32_t a, uint32_t b,
        uint32_t len, uint32_t lim)
{
    const uint32_t lane = threadIdx.x;
    while (len < lim) {
        const uint32_t off = len + lane;
        const bool mism = off >= lim || in[a + off] != in[b + off];
        const uint64_t m = __ballot(mism);
        if (m) { len += (uint32_t)__builtin_ctzll(m); break; }
        len += 64;
    }
    return len < lim ? len : lim;
}

__device__ __forceinline__ uint32_t prefix_max_incl(uint32_t v)
{
    const uint32_t lane = threadIdx.x;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const uint32_t o = __shfl_up(v, s);
        if (lane >= (uint32_t)s) v = max(v, o);
    }
    return v;
}

// ------------------------------------------------------------------------------------------
// One "round": everything lzma_mf_find() would report at position x (lz_encoder_mf.c:22-79,
// HC3 :305-335, HC4 :366-413, chain walk :250-287) plus the four rep-match lengths at x.
// Lane roles: 0 = hash2 candidate, 1 = hash3 candidate, 2 = own slot (bucket flag only),
// 3..2+depth = chain candidates in chain order, 60..63 = rep0..rep3.
// ------------------------------------------------------------------------------------------
struct Round {
    uint64_t mask;      // recorded matches, in matches[] order (ascending lane)
    uint32_t L;         // per lane: recorded length (lanes 0..58), rep length (60..63)
    uint32_t D;         // per lane: distance (zero based)
    uint32_t longest;   // lzma_mf_find() return value (incl. the > nice_len extension)
};

struct Env {
    const uint8_t* __restrict__ in;
    const uint32_t* __restrict__ rank;
    const uint32_t* __restrict__ sorted_pos;
    const uint32_t* __restrict__ prev2;
    const uint32_t* __restrict__ prev3;
    uint32_t nice, depth, hb, cyclic;
    uint32_t block_end;
    uint32_t n_last;                            // last valid byte offset of the batch (prefetch clamp)
    // match lists written by k_find_sn / k_find_exact (one 32-byte record per position), read by the list-driven parser
    const uint16_t* __restrict__ mlen;
    const uint32_t* __restrict__ mdist;
    uint32_t packed;                            // lists are one u32 per entry: length << 23 | distance-1 (dict <= 8 MiB)
};
constexpr uint32_t LIST_K = 7;                // entries kept per position (the LIST_K longest)
constexpr uint32_t LIST_W = 8;                // words per position: LIST_K entries + trailer (count | len2 of the two longest)
constexpr uint32_t LEN2_MAX = 127;            // cap of the rep0 run recorded with the two longest entries

// ------------------------------------------------------------------------------------------
// Software prefetch of the parse-independent per-position data.  Rounds mostly visit consecutive
// positions (lookahead of the fast parser, every node of an optimal-parser window), so while the
// wave works on position x the loads for x+1 (chain slots) and x+2 (rank / prev links) are already
// in flight: two of the three dependent HBM round trips of a round leave the critical path.
// ------------------------------------------------------------------------------------------
struct PreA { uint32_t rk, d2, d3; };
struct Pre {
    uint32_t pos;       // position (a, ent) belong to; `an` belongs to pos + 1
    bool valid;
    PreA a;
    uint32_t ent;       // per lane: chain slot entry for this lane's role
    PreA an;
};

__device__ __forceinline__ PreA load_a(const Env& e, uint32_t x)
{
    x = x < e.n_last ? x : e.n_last;
    PreA a;
    a.rk = e.rank[x];
    a.d2 = e.prev2[x];
    a.d3 = e.hb == 4 ? e.prev3[x] : 0;
    return a;
}

__device__ __forceinline__ uint32_t load_ent(const Env& e, const PreA& a)
{
    const uint32_t lane = threadIdx.x;
    uint32_t ent = 0x80000000u;                      // out of range == "bucket start"
    const bool in_chain = lane >= 2 && lane <= 2 + e.depth;
    if (in_chain && a.rk >= lane - 2) ent = e.sorted_pos[a.rk - (lane - 2)];
    return ent;
}

__device__ __forceinline__ void fetch(const Env& e, Pre& P, uint32_t x, PreA& a, uint32_t& ent)
{
    if (!(P.valid && P.pos == x)) {                  // cold start: two dependent round trips
        P.a = load_a(e, x);
        P.ent = load_ent(e, P.a);
        P.an = load_a(e, x + 1);
    }
    a = P.a;
    ent = P.ent;
    const PreA an = P.an;                            // issued one round ago
    P.a = an;
    P.ent = load_ent(e, an);                         // for x + 1, consumed next round
    P.an = load_a(e, x + 2);
    P.pos = x + 1;
    P.valid = true;
}

template <bool REPS = true>
__device__ __forceinline__ void do_round(const Env& e, Pre& P, uint32_t x, uint32_t end,
        uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, Round& R)
{
    PreA pa;
    uint32_t pent;
    fetch(e, P, x, pa, pent);
    const uint32_t lane = threadIdx.x;
    const uint32_t avail = end - x;
    const uint32_t buf_avail = avail < MATCH_LEN_MAX ? avail : MATCH_LEN_MAX;
    uint32_t len_limit = avail;
    bool mf_ok = true;
    if (e.nice <= len_limit) len_limit = e.nice;
    else if (len_limit < e.hb) mf_ok = false;            // "pending": no matches reported

    uint32_t q = 0, lim = 0;
    bool chain_valid = false;
    if (mf_ok) {
        const uint32_t d2 = pa.d2;
        const uint32_t d3 = pa.d3;
        if (lane == 0) {
            if (d2 != 0 && d2 < e.cyclic) { q = x - d2; lim = len_limit; }
        } else if (lane == 1) {
            if (d3 != 0 && d3 != d2 && d3 < e.cyclic) { q = x - d3; lim = len_limit; }
        }
        const uint32_t ent = pent;
        const bool in_chain = lane >= 2 && lane <= 2 + e.depth;
        const uint64_t flags = __ballot(in_chain && (ent >> 31)) >> 2;   // bit j = flag of slot rank-j
        if (lane >= 3 && in_chain) {
            const uint32_t j = lane - 2;                 // candidate number 1..depth
            const bool same_bucket = (flags & ((1ull << j) - 1)) == 0;
            const uint32_t qp = ent & 0x7FFFFFFFu;
            if (same_bucket && x - qp < e.cyclic) { chain_valid = true; q = qp; lim = len_limit; }
        }
    }
    if (REPS && lane >= 60) {
        const uint32_t rep = lane == 60 ? r0 : lane == 61 ? r1 : lane == 62 ? r2 : r3;
        q = x - rep - 1;
        lim = buf_avail;
    }

    uint32_t L = lane_cmplen(e.in, q, x, lim);
    const uint32_t D = x - q - 1;

    const uint32_t L0 = lane_of(L, 0), L1 = lane_of(L, 1);
    const bool has2 = L0 >= 1;              // first byte equal => (by the hash) >= 2 bytes equal
    const bool has3 = L1 >= 1;              // lane 1 only active for HC4
    uint32_t best;
    bool done;
    if (e.hb == 4) {
        best = has3 ? L1 : (has2 ? L0 : 1);
        done = (has2 || has3) && best == len_limit;
        if (best < 3) best = 3;
    } else {
        best = has2 ? L0 : 2;
        done = has2 && best == len_limit;
    }
    const uint32_t X = chain_valid ? L : 0;
    const uint32_t incl = prefix_max_incl(X);
    uint32_t excl = __shfl_up(incl, 1);
    if (lane == 0) excl = 0;
    bool rec = chain_valid && !done && X > max(best, excl);
    if (lane == 0) { rec = has2; if (has2 && has3) L = 2; }
    if (lane == 1) rec = has3;
    R.mask = mf_ok ? __ballot(rec) : 0ull;
    R.L = L;
    R.D = D;
    uint32_t longest = 0;
    if (R.mask) {
        const uint32_t top = 63 - (uint32_t)__builtin_clzll(R.mask);
        longest = lane_of(L, top);
        if (longest == e.nice) {
            const uint32_t dist = lane_of(D, top);
            longest = wave_cmplen(e.in, x, x - dist - 1, longest, buf_avail);
        }
    }
    R.longest = longest;
}

// ------------------------------------------------------------------------------------------
// LZMA symbol coder (lzma/lzma_encoder.c:23-263)
// ------------------------------------------------------------------------------------------
#ifdef XZAMD_LIT16
typedef uint16_t plit_t_fwd;
#else
typedef uint32_t plit_t_fwd;
#endif
struct Lz {
    uint32_t state;
    uint32_t rep0, rep1, rep2, rep3;
    uint32_t lc, lp, pb;
    uint32_t cnt_len, cnt_match, cnt_align;   // coded lengths / matches / align-coded matches since refresh
    plit_t_fwd* lit;                          // literal coder probabilities of this span (global memory, L2-resident)
    uint32_t* gp;                             // parse pieces (PG): every other probability of the piece's model, u32 each, global
};

__device__ __forceinline__ uint32_t dist_slot_of(uint32_t d)
{
    if (d <= 4) return d;
    const uint32_t i = 31 - __builtin_clz(d);
    return (i + i) + ((d >> (i - 1)) & 1);
}

// Literal-coder probabilities (6144 of the 7990 model entries at lc=3) live in global memory, one
// 24 KiB slice per span, so the LDS footprint of a wavefront drops from 16 KiB to 3.6 KiB and the
// CU can hold 2x (and more) wavefronts: the kernel is issue-latency bound per wave and throughput
// scales with resident waves (measured: halving occupancy costs 1.8x).  A literal touches 8 of them:
// one gather and one scatter per symbol, L1 bypassed (sc1) so the wave reads back its own updates
// from L2.
// XZAMD_LIT16 (measurement build): the literal coders as u16 instead of u32 -- half the L2 footprint of a piece's model
#ifdef XZAMD_LIT16
typedef uint16_t plit_t;
#define PLIT_PER_U4 8u
#define PLIT_FLAT4 make_uint4(0x04000400u, 0x04000400u, 0x04000400u, 0x04000400u)
__device__ __forceinline__ uint32_t lit_load(const uint16_t* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lit_store(uint16_t* p, uint32_t v)
{
    __hip_atomic_store(p, (uint16_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#else
typedef uint32_t plit_t;
#define PLIT_PER_U4 4u
#define PLIT_FLAT4 make_uint4(1024u, 1024u, 1024u, 1024u)
#endif
__device__ __forceinline__ uint32_t lit_load(const uint32_t* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lit_store(uint32_t* p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- wave-parallel symbol coder -------------------------------------------------------------
// An LZMA symbol is a list of <= 48 (probability, bit) pairs whose probability INDICES are known
// up front (lzma_encoder.c:23-263 walks them one rc_bit at a time).  Lane k of the wavefront
// owns pair k: all probabilities are fetched with ONE LDS gather, adapted in parallel (the update
// depends only on the old value and the bit) and written back with ONE scatter; only the
// range/low recurrence (range_encoder.h:196-257) runs serially, on scalar registers, through a
// single copy of the normalise/shift_low code.
enum : uint32_t { SEG_BIT = 0, SEG_TREE = 1, SEG_REV = 2, SEG_DIRECT = 3, SEG_MATCHED = 4 };

struct SegSel {            // per lane: which segment this lane's pair belongs to
    uint32_t type, base, sym, i, n;
    bool hit;
};

__device__ __forceinline__ void seg_add(SegSel& s, uint32_t& off, uint32_t n, uint32_t type, uint32_t base, uint32_t sym)
{
    const uint32_t k = threadIdx.x;
    const bool h = k - off < n;                 // off <= k < off + n (unsigned wrap)
    s.type = h ? type : s.type;
    s.base = h ? base : s.base;
    s.sym = h ? sym : s.sym;
    s.i = h ? k - off : s.i;
    s.n = h ? n : s.n;
    s.hit = s.hit || h;
    off += n;
}

// serial part: t = p (12 bits) | bit << 12 | direct << 13, one entry per lane 0..n-1
// Entries [d0, d1) are direct bits (rc_direct), all others probability-coded; the latter are
// written without data-dependent branches (the bit only selects via a mask).
template <bool PIN>
__device__ __forceinline__ void rc_run(RC& rc, uint32_t packed, uint32_t n, uint32_t d0, uint32_t d1)
{
#ifdef XZAMD_TIMING
    const uint64_t t0_ = __builtin_amdgcn_s_memtime();
    rc.tm_bits += n;
#endif
    uint32_t k = 0;
    for (;;) {
        const uint32_t stop = k < d0 ? d0 : n;
        for (; k < stop; ++k) {
            const uint32_t t = lane_of(packed, k);
            rc.normalize();
            const uint32_t m = 0u - ((t >> 12) & 1u);          // all ones for a 1 bit
            const uint32_t bound = (rc.range >> 11) * (t & 0xFFFu);
            rc.low += bound & m;
            rc.range = bound + ((rc.range - 2 * bound) & m);   // 1: range - bound, 0: bound
            if constexpr (PIN) { pin_s(rc.range); pin_s(rc.low); }
        }
        if (k >= n) break;
        for (; k < d1; ++k) {
            const uint32_t t = lane_of(packed, k);
            rc.normalize();
            rc.range >>= 1;
            rc.low += rc.range & (0u - ((t >> 12) & 1u));
            if constexpr (PIN) { pin_s(rc.range); pin_s(rc.low); }
        }
    }
#ifdef XZAMD_TIMING
    rc.tm_run += __builtin_amdgcn_s_memtime() - t0_;
#endif
}

// CODE: run the range coder (false: the parse pieces of the two-phase mode only adapt their price model);
// LITG: literal-coder probabilities live in global memory (u32 each, `lit`), else in LDS behind P_LITERAL (u16);
// TOK: the model pass of the two-phase coder: one 16-bit token per decision goes to rc.tok and the price of the
// decisions (the parser's table, probability before its update) is added to rc.est.
// PG: the whole model lives in global memory (the parse pieces: `gp` holds what LDS holds elsewhere).
// BND: the bounds walk of the carried model (k_model_walk<1, 2>): `probs` is a u32 array, lo | hi << 11 | logged << 22 per
// probability; both bounds take the update, and while they differ the bit is logged for k_model_chain (the log is zero on entry:
// only 1 bits are written).  Tokens are only counted.
template <bool CODE, bool LITG, bool TOK = false, bool PG = false, bool BND = false>
__device__ __forceinline__ void rc_emit(RC& rc, uint16_t* probs, plit_t* lit, uint32_t* gp, const SegSel& s, uint32_t total,
        uint32_t d0, uint32_t d1)
{
    uint32_t idx = 0, bit = 0;
    bool direct = false;
    if (s.hit) {
        const uint32_t i = s.i, n = s.n, sym = s.sym;
        if (s.type == SEG_BIT) {
            idx = s.base; bit = sym & 1;
        } else if (s.type == SEG_TREE) {            // rc_bittree: MSB first
            bit = (sym >> (n - 1 - i)) & 1;
            idx = s.base + ((1u << i) | (sym >> (n - i)));
        } else if (s.type == SEG_REV) {             // rc_bittree_reverse: LSB first
            bit = (sym >> i) & 1;
            const uint32_t low = sym & ((1u << i) - 1);
            idx = s.base + ((1u << i) | (__builtin_bitreverse32(low) >> ((32 - i) & 31)));   // i == 0: low == 0
        } else if (s.type == SEG_DIRECT) {
            bit = (sym >> (n - 1 - i)) & 1;
            direct = true;
        } else {                                    // literal_matched (lzma_encoder.c:23-41)
            const uint32_t cur = sym & 0xFF, mb = sym >> 8;
            bit = (cur >> (7 - i)) & 1;
            const uint32_t pre = (0x100u | cur) >> (8 - i);
            const bool same = (mb >> (8 - i)) == (cur >> (8 - i));
            idx = s.base + (same ? 0x100u + (((mb >> (7 - i)) & 1) << 8) + pre : pre);
        }
    }
    if constexpr (BND) {
        uint32_t* const M = reinterpret_cast<uint32_t*>(probs);
        if (s.hit && !direct) {
            // (LITG: the literal coders' bounds live in the span's slice of cb_bnd in global memory, read and written past L1)
            const bool glob = LITG && idx >= P_LITERAL;
            const uint32_t v = glob ? lit_load(reinterpret_cast<const uint32_t*>(lit) + (idx - P_LITERAL)) : M[idx];
            uint32_t lo = v & 0x7FFu, hi = (v >> 11) & 0x7FFu, nb = v >> 22;
            if (lo != hi && nb < rc.log_cap) {
                if (bit) atomicOr(rc.log + (uint64_t)idx * XZAMD_LOG_WORDS + (nb >> 5), 1u << (nb & 31u));
                ++nb;
            }
            lo = bit ? lo - (lo >> 5) : lo + ((2048u - lo) >> 5);
            hi = bit ? hi - (hi >> 5) : hi + ((2048u - hi) >> 5);
            const uint32_t nv = lo | (hi << 11) | (nb << 22);
            if (glob) lit_store(reinterpret_cast<uint32_t*>(lit) + (idx - P_LITERAL), nv);
            else M[idx] = nv;
        }
        rc.tok += total;
        return;
    }
    uint32_t p = 0;
    if (s.hit && !direct) {
        if ((LITG || PG) && idx >= P_LITERAL) {
            plit_t* g = lit + (idx - P_LITERAL);
            p = lit_load(g);
            lit_store(g, bit ? p - (p >> 5) : p + ((2048 - p) >> 5));
        } else if (PG) {
            uint32_t* g = gp + idx;
            p = lit_load(g);
            lit_store(g, bit ? p - (p >> 5) : p + ((2048 - p) >> 5));
        } else {
            p = probs[idx];
            probs[idx] = (uint16_t)(bit ? p - (p >> 5) : p + ((2048 - p) >> 5));
        }
    }
    if constexpr (TOK) {
        const uint32_t k = threadIdx.x;
        uint32_t price = 0;
        if (k < total) {
            rc.tok[k] = (uint16_t)(p | (bit << 12) | (direct ? 0x8000u : 0u));
            price = direct ? 16u : (uint32_t)rc.ptab[(p ^ ((0u - bit) & 0x7FFu)) >> 4];
        }
        rc.est += wave_sum_dpp(price);
        rc.tok += total;
    } else if constexpr (PG) {
        // the parser's price of its own parse (rc.ptab set: the recorded symbols of a piece; oracle: parse_piece's return value)
        if (rc.ptab != nullptr) {
            const uint32_t price = threadIdx.x < total ? (direct ? 16u : (uint32_t)rc.ptab[(p ^ ((0u - bit) & 0x7FFu)) >> 4]) : 0u;
            rc.est += wave_sum_dpp(price);
        }
    } else if constexpr (CODE) {
        const uint32_t packed = p | (bit << 12) | (direct ? 0x2000u : 0u);
        rc_run<!LITG>(rc, packed, total, d0, d1);
    }
}

// upos = offset of the symbol inside the Block; lit3 (literals only) = byte | previous byte << 8 | match byte << 16
// (the match byte is read only in states >= 7)
template <bool CODE, bool LITG, bool TOK = false, bool PG = false, bool BND = false>
__device__ __forceinline__ void encode_symbol_t(RC& rc, uint16_t* probs, Lz& z, uint32_t upos, uint32_t back, uint32_t len,
        uint32_t lit3)
{
    if (PG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the previous symbol's probability scatter
    const uint32_t ps = upos & ((1u << z.pb) - 1);
    SegSel s;
    s.type = 0; s.base = 0; s.sym = 0; s.i = 0; s.n = 0; s.hit = false;
    uint32_t off = 0;
    if (back == LITERAL) {
        if (LITG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // previous literal's probability scatter
        seg_add(s, off, 1, SEG_BIT, P_IS_MATCH + z.state * 16 + ps, 0);
        const uint32_t cur = lit3 & 0xFFu;
        const uint32_t prev = (lit3 >> 8) & 0xFFu;
        const uint32_t mask = (0x100u << z.lp) - (0x100u >> z.lc);
        const uint32_t sub = P_LITERAL + 3u * ((((upos << 8) + prev) & mask) << z.lc);
        if (z.state < 7) {
            z.state = z.state <= 3 ? 0 : z.state - 3;
            seg_add(s, off, 8, SEG_TREE, sub, cur);
        } else {
            z.state = z.state <= 9 ? z.state - 3 : z.state - 6;
            const uint32_t mb = (lit3 >> 16) & 0xFFu;
            seg_add(s, off, 8, SEG_MATCHED, sub, cur | (mb << 8));
        }
        rc_emit<CODE, LITG, TOK, PG, BND>(rc, probs, z.lit, z.gp, s, off, off, off);
        return;
    }
    seg_add(s, off, 1, SEG_BIT, P_IS_MATCH + z.state * 16 + ps, 1);
    uint32_t len_base, dir0 = ~0u, dir1 = ~0u;
    if (back < 4) {
        seg_add(s, off, 1, SEG_BIT, P_IS_REP + z.state, 1);
        if (back == 0) {
            seg_add(s, off, 1, SEG_BIT, P_IS_REP0 + z.state, 0);
            seg_add(s, off, 1, SEG_BIT, P_IS_REP0_LONG + z.state * 16 + ps, len != 1);
        } else {
            seg_add(s, off, 1, SEG_BIT, P_IS_REP0 + z.state, 1);
            uint32_t dist;
            if (back == 1) {
                seg_add(s, off, 1, SEG_BIT, P_IS_REP1 + z.state, 0);
                dist = z.rep1;
            } else {
                seg_add(s, off, 1, SEG_BIT, P_IS_REP1 + z.state, 1);
                seg_add(s, off, 1, SEG_BIT, P_IS_REP2 + z.state, back - 2);
                if (back == 3) { dist = z.rep3; z.rep3 = z.rep2; }
                else dist = z.rep2;
                z.rep2 = z.rep1;
            }
            z.rep1 = z.rep0;
            z.rep0 = dist;
        }
        if (len == 1) {
            z.state = z.state < 7 ? 9 : 11;
            rc_emit<CODE, LITG, TOK, PG, BND>(rc, probs, z.lit, z.gp, s, off, off, off);
            return;
        }
        len_base = P_REP_LEN;
        z.state = z.state < 7 ? 8 : 11;
        ++z.cnt_len;
    } else {
        seg_add(s, off, 1, SEG_BIT, P_IS_REP + z.state, 0);
        len_base = P_MATCH_LEN;
        z.state = z.state < 7 ? 7 : 10;
        ++z.cnt_len;
        ++z.cnt_match;
    }
    // length (lzma_encoder.c:106-134)
    {
        const uint32_t l = len - 2;
        if (l < 8) {
            seg_add(s, off, 1, SEG_BIT, len_base + LEN_CHOICE, 0);
            seg_add(s, off, 3, SEG_TREE, len_base + LEN_LOW + ps * 8, l);
        } else if (l < 16) {
            seg_add(s, off, 1, SEG_BIT, len_base + LEN_CHOICE, 1);
            seg_add(s, off, 1, SEG_BIT, len_base + LEN_CHOICE2, 0);
            seg_add(s, off, 3, SEG_TREE, len_base + LEN_MID + ps * 8, l - 8);
        } else {
            seg_add(s, off, 1, SEG_BIT, len_base + LEN_CHOICE, 1);
            seg_add(s, off, 1, SEG_BIT, len_base + LEN_CHOICE2, 1);
            seg_add(s, off, 8, SEG_TREE, len_base + LEN_HIGH, l - 16);
        }
    }
    if (back >= 4) {
        // distance (lzma_encoder.c:142-181)
        const uint32_t dist = back - 4;
        const uint32_t slot = dist_slot_of(dist);
        const uint32_t ds = len < 6 ? len - 2 : 3;
        seg_add(s, off, 6, SEG_TREE, P_DIST_SLOT + ds * 64, slot);
        if (slot >= 4) {
            const uint32_t fb = (slot >> 1) - 1;
            const uint32_t base = (2 | (slot & 1)) << fb;
            const uint32_t red = dist - base;
            if (slot < 14) {
                seg_add(s, off, fb, SEG_REV, P_DIST_SPECIAL + base - slot - 1, red);
            } else {
                dir0 = off;
                seg_add(s, off, fb - 4, SEG_DIRECT, 0, red >> 4);
                dir1 = off;
                seg_add(s, off, 4, SEG_REV, P_DIST_ALIGN, red & 15);
                ++z.cnt_align;
            }
        }
        z.rep3 = z.rep2; z.rep2 = z.rep1; z.rep1 = z.rep0; z.rep0 = dist;
    }
    if (dir0 == ~0u) dir0 = dir1 = off;
    rc_emit<CODE, LITG, TOK, PG, BND>(rc, probs, z.lit, z.gp, s, off, dir0, dir1);
}

// the bytes a literal at global offset g needs, for encode_symbol_t
__device__ __forceinline__ uint32_t literal_bytes(const uint8_t* __restrict__ in, uint32_t g, uint32_t upos, const Lz& z)
{
    const uint32_t cur = uni(in[g]);
    const uint32_t prev = upos ? uni(in[g - 1]) : 0;
    const uint32_t mb = z.state >= 7 ? uni(in[g - z.rep0 - 1]) : 0;
    return cur | (prev << 8) | (mb << 16);
}

// g = global offset of the byte, upos = its offset inside the Block (single-phase kernels: code while parsing)
__device__ __forceinline__ void encode_symbol(RC& rc, uint16_t* probs, Lz& z, const uint8_t* __restrict__ in,
        uint32_t g, uint32_t upos, uint32_t back, uint32_t len)
{
    encode_symbol_t<true, true>(rc, probs, z, upos, back, len, back == LITERAL ? literal_bytes(in, g, upos, z) : 0u);
}

__device__ __forceinline__ bool change_pair(uint32_t small_dist, uint32_t big_dist)
{
    return (big_dist >> 7) > small_dist;
}

// ------------------------------------------------------------------------------------------
// Suffix-neighbourhood match finder (the GPU successor of BT4) and the windowed optimal parser.
// Semantics are defined by oracle/lzma_fast_enc.c (build_sa / find_sn / optimum_window); the code
// below is their wave-parallel form and must stay bit-exact with them.
// ------------------------------------------------------------------------------------------
// Optimal-parser window: nodes 0..WM, a template parameter of the parser.  384 for every option set since round 5 (LDS per
// wave <= 10 KiB -> 16 waves per CU); rounds 1-4 ran 232 nodes, and 360 at 12 waves per CU for nice_len > 128.
constexpr uint32_t WMAX_STD = 384;
constexpr uint32_t PRICE_INF = 1u << 30;

#ifdef XZAMD_TIMING
#define TM_BEGIN(v) const uint64_t v = __builtin_amdgcn_s_memtime()
#define TM_END(w, k, v) do { if (threadIdx.x == 0) (w).tm[k] += __builtin_amdgcn_s_memtime() - (v); } while (0)
#define TM_COUNT(w, k) do { if (threadIdx.x == 0) (w).tm[k] += 1; } while (0)
#else
#define TM_BEGIN(v) do { } while (0)
#define TM_END(w, k, v) do { } while (0)
#define TM_COUNT(w, k) do { } while (0)
#endif

// Per-wave LDS carve for the list-based paths.
struct Work {
    // optimal parser only
    uint32_t* n_price;  // [WM+1] node price; after backtracking: out-edge `back`
    uint32_t* n_info;   // [WM+1] in-len (9) | state (4) << 9 | out-len (9) << 13 | in-edge kind (3) << 22
                        //          kind: 0..3 = rep index, 4 = match (its distance is the node's rep0), 5 = literal
    uint4* n_reps4;     // [WM+1] rep distances of the node (complete when the parser reaches it)
    uint16_t* dsp;      // [4*64]  dist-slot price (+ direct bits for slot >= 14)
    uint16_t* xt;       // [128]   price of the footer bits of distances < 128 (slots 4..13, 0 below)
    uint16_t* ap;       // [16]    align price; == xt + 128, so one table: index dist < 128 ? dist : 128 + (dist & 15)
    uint8_t* ptab;      // [128]   bit price table (price_tablegen.c:31-58)
    uint32_t* err;      // debug/consistency word block (global)
    uint32_t ptv;       // the same price table in registers: lane l (< 32) holds entries 4l..4l+3
#ifdef XZAMD_TIMING
    unsigned long long* tm;   // [16] cycle accumulators in LDS (profiling builds only)
#endif
};

struct RoundL {
    uint32_t rp[4];     // list-driven parser: the four rep-match lengths
    uint32_t rl[4];     // the same, 0 when < 2 (usable rep matches)
    uint32_t pre;       // compound precheck: some rep source has two equal bytes right behind its first mismatch
    uint32_t mlo, mhi;  // lane i < 4: mismatch mask of rep source i over the 64-byte row at x (bit o: offset o differs or is past the end)
    uint32_t l2a, l2b;  // rep0 run behind the byte after the longest / second longest match (list trailer)
    uint32_t L;         // per lane; lanes 60..63 = rep lengths (in-kernel finders)
    uint32_t SL, SD;    // kept matches sorted by length: lane r holds entry r (length, zero-based distance)
    uint32_t cnt;       // number of entries
    uint32_t longest;   // incl. the > nice_len extension
};

__device__ __forceinline__ uint64_t rm_of(const RoundL& R, uint32_t i)
{
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)R.mhi, (int)i) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)R.mlo, (int)i);
}

// lane `dst` receives `v` from this lane (all lanes must execute; unused senders target lane 63)
__device__ __forceinline__ uint32_t lane_scatter(uint32_t dst, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), (int)v);
}

// ---- list-driven rounds ------------------------------------------------------------------------
// The match finder is parse independent (find and skip both insert), so k_find_t runs it for every
// position of the batch as a separate, fully parallel kernel.  The parser then streams the lists:
// positions are visited strictly in order, so the record of x+1 is always in flight while x is
// priced.  Only the four rep-match lengths depend on the parse; lanes 60..63 measure them here.
// (the record in flight is kept as it was loaded -- one register for the packed form, two for the other -- and taken apart
// when its round is worked out, not when it is loaded: it is live across the whole node in front)
struct ListPre { uint32_t pos; bool valid; uint32_t v, l16; };

// One 32-byte record per position (see k_find_sn): lanes 0..6 = entries, lane 7 = trailer.
__device__ __forceinline__ void lists_load(const Env& e, uint32_t x, uint32_t& v, uint32_t& l16)
{
    const uint32_t lane = threadIdx.x;
    x = x < e.n_last ? x : e.n_last;
    const uint64_t base = (uint64_t)x * LIST_W + (lane & (LIST_W - 1));
    v = e.mdist[base];                             // every lane loads (lanes >= LIST_W repeat the record): no exec masking
    l16 = e.packed ? 0u : (uint32_t)e.mlen[base];
}

// lane LIST_K of `tr` holds the trailer
__device__ __forceinline__ void lists_split(const Env& e, uint32_t v, uint32_t l16, uint32_t& sl, uint32_t& sd, uint32_t& tr)
{
    tr = v;
    if (e.packed) {
        sl = v >> 23; sd = v & 0x7FFFFFu;
    } else {
        sd = v;
        sl = l16;
    }
}

// The five byte rows of a round (lane = byte offset): the text at x and the four rep sources.  They can be issued as
// soon as the node's rep distances are final -- for node j + 1 that is right after the literal / short-rep edge out of
// node j, every other edge out of j is longer -- and are only waited for when the round is worked out.
struct Rows { uint32_t cx, c0, c1, c2, c3; };

__device__ __forceinline__ void rows_issue(const Env& e, uint32_t x, uint32_t end,
        uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, Rows& W)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t avail = end - x;
    const uint32_t buf_avail = avail < MATCH_LEN_MAX ? avail : MATCH_LEN_MAX;
    const uint32_t off = lane < buf_avail ? lane : 0u;      // lanes past the end re-read byte 0 (masked by the round)
    W.cx = e.in[x + off];
    W.c0 = e.in[x - r0 - 1 + off]; W.c1 = e.in[x - r1 - 1 + off];
    W.c2 = e.in[x - r2 - 1 + off]; W.c3 = e.in[x - r3 - 1 + off];
}

// A node with the rep distances of the node in front of it (reached by a literal, or by the same match or rep at another
// length: 57 % of the nodes of the bench text): its four mismatch masks are that node's shifted down by one byte, and only
// offset 63 is new -- ONE load (lanes 0..3: byte 63 of the four rep sources, every other lane: byte 63 of the text) in place
// of the five rows.  Offset 63 is dead when fewer than 64 bytes are left (the masks' `dead` bits cover it).
__device__ __forceinline__ void rows_issue_top(const Env& e, uint32_t x, uint32_t end,
        uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, Rows& W)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t off = end - x > 63u ? 63u : 0u;
    const uint32_t back = lane == 0 ? r0 + 1 : lane == 1 ? r1 + 1 : lane == 2 ? r2 + 1 : lane == 3 ? r3 + 1 : 0u;
    W.cx = e.in[x + off - back];
}

template <bool SHIFT = false>
__device__ __forceinline__ void round_lists_rows(const Env& e, ListPre& LP, uint32_t x, uint32_t end,
        uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, const Rows& W, RoundL& R, const bool shift = false)
{
    const uint32_t lane = threadIdx.x;
    if (!(LP.valid && LP.pos == x)) lists_load(e, x, LP.v, LP.l16);
    uint32_t sl, sd, tv;
    lists_split(e, LP.v, LP.l16, sl, sd, tv);
    lists_load(e, x + 1, LP.v, LP.l16);
    LP.pos = x + 1;
    LP.valid = true;
    const uint32_t avail = end - x;
    const uint32_t buf_avail = avail < MATCH_LEN_MAX ? avail : MATCH_LEN_MAX;
    // Rep-match lengths: one 64-byte row of the text against the four rep sources, lane = byte offset, a
    // ballot each (bit o: offset o differs or lies past the end).  Everything that follows from a mask --
    // length, "usable" length, the compound precheck -- is then worked out by lane i for rep i with vector
    // instructions (the scalar unit is what bounds this kernel), and only the results go back to scalars.
    {
        const uint64_t dead = buf_avail < 64 ? ~0ull << buf_avail : 0ull;
        uint32_t lo, hi;
        if (SHIFT && shift) {
            // the masks of the node in front, one byte further on (lane i: rep i); bit 63 from the one load of rows_issue_top
            const uint32_t nb = W.cx != lane_of(W.cx, 4) ? 0x80000000u : 0u;          // lanes 0..3 can differ
            lo = (R.mlo >> 1) | (R.mhi << 31) | (uint32_t)dead;
            hi = (R.mhi >> 1) | nb | (uint32_t)(dead >> 32);
        } else {
            const uint32_t cx = W.cx, c0 = W.c0, c1 = W.c1, c2 = W.c2, c3 = W.c3;
            const uint64_t m0 = __builtin_amdgcn_ballot_w64(c0 != cx) | dead, m1 = __builtin_amdgcn_ballot_w64(c1 != cx) | dead;
            const uint64_t m2 = __builtin_amdgcn_ballot_w64(c2 != cx) | dead, m3 = __builtin_amdgcn_ballot_w64(c3 != cx) | dead;
            lo = lane == 0 ? (uint32_t)m0 : lane == 1 ? (uint32_t)m1 : lane == 2 ? (uint32_t)m2 : (uint32_t)m3;
            hi = lane == 0 ? (uint32_t)(m0 >> 32) : lane == 1 ? (uint32_t)(m1 >> 32)
                    : lane == 2 ? (uint32_t)(m2 >> 32) : (uint32_t)(m3 >> 32);
        }
        R.mlo = lo; R.mhi = hi;
        const uint32_t l = lo ? (uint32_t)__builtin_ctz(lo) : hi ? 32u + (uint32_t)__builtin_ctz(hi) : 64u;
        const uint32_t lu = l >= 2 ? l : 0u;
        // two equal bytes behind the first mismatch <=> bits l+1, l+2 clear (bit l is set)
        const uint64_t mm = ((uint64_t)hi << 32) | lo;
        const uint32_t t3 = (uint32_t)(mm >> (l & 63u)) & 7u;
        const bool okl = (l - 2u < 60u) || (lane == 0 && l == 0);
        R.pre = __builtin_amdgcn_ballot_w64(lane < 4 && okl && t3 == 1u) != 0ull;
        const uint64_t full = __builtin_amdgcn_ballot_w64(lane < 4 && l >= 64u);
        R.rp[0] = lane_of(l, 0); R.rp[1] = lane_of(l, 1); R.rp[2] = lane_of(l, 2); R.rp[3] = lane_of(l, 3);
        R.rl[0] = lane_of(lu, 0); R.rl[1] = lane_of(lu, 1); R.rl[2] = lane_of(lu, 2); R.rl[3] = lane_of(lu, 3);
        if (full) {                                             // a rep match of >= 64 bytes: the slow path
            if (full & 1) R.rl[0] = R.rp[0] = wave_cmplen(e.in, x, x - r0 - 1, 64, buf_avail);
            if (full & 2) R.rl[1] = R.rp[1] = wave_cmplen(e.in, x, x - r1 - 1, 64, buf_avail);
            if (full & 4) R.rl[2] = R.rp[2] = wave_cmplen(e.in, x, x - r2 - 1, 64, buf_avail);
            if (full & 8) R.rl[3] = R.rp[3] = wave_cmplen(e.in, x, x - r3 - 1, 64, buf_avail);
        }
    }
    const uint32_t tr = lane_of(tv, LIST_K);
    const uint32_t cnt = tr & 0xFFu;
    uint32_t slc = sl, l2a = (tr >> 8) & 0xFFu, l2b = (tr >> 16) & 0xFFu;
    uint32_t longest = cnt ? lane_of(sl, cnt - 1) : 0;
    if (avail < MATCH_LEN_MAX + 1 + LEN2_MAX) {
        // The records are made for the whole Block; a span may not reference bytes behind its end, so near the end
        // the parser clamps what it reads (oracle: do_round): lengths to the bytes left, the rep0 run of the
        // "match + literal + rep0" edge to what follows match + literal.
        const uint32_t second = cnt >= 2 ? lane_of(sl, cnt - 2) : 0;
        l2a = longest + 1 >= avail ? 0u : min(l2a, avail - longest - 1);
        l2b = second + 1 >= avail ? 0u : min(l2b, avail - second - 1);
        slc = min(sl, avail);
        longest = min(longest, avail);
    }
    R.SL = slc;
    R.SD = sd;
    R.cnt = cnt;
    R.l2a = l2a;
    R.l2b = l2b;
    R.longest = longest;
}

__device__ __forceinline__ void round_lists(const Env& e, ListPre& LP, uint32_t x, uint32_t end,
        uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, RoundL& R)
{
    Rows W;
    rows_issue(e, x, end, r0, r1, r2, r3, W);
    round_lists_rows(e, LP, x, end, r0, r1, r2, r3, W, R);
}

// ---- prices (rangecoder/price.h:28-92) ------------------------------------------------------
// The non-literal probabilities of a model behind one accessor: `const uint16_t*` (LDS: the single-phase kernels) or GProbs
// (global memory, u32 each, read past L1 like the literal coders: the parse pieces, whose LDS goes to the DP nodes).
struct GProbs {
    uint32_t* p;
    __device__ __forceinline__ uint32_t operator[](uint32_t i) const { return lit_load(p + i); }
};

template <class PR>
__device__ __forceinline__ uint32_t pr_bit(PR probs, const uint8_t* ptab, uint32_t idx, uint32_t bit)
{
    return ptab[(probs[idx] ^ ((0u - bit) & 0x7FFu)) >> 4];
}
template <class PR>
__device__ __forceinline__ uint32_t pr_tree(PR probs, const uint8_t* ptab, uint32_t base,
        uint32_t nbits, uint32_t sym)
{
    uint32_t price = 0;
    sym += 1u << nbits;
    do {
        const uint32_t bit = sym & 1;
        sym >>= 1;
        price += pr_bit(probs, ptab, base + sym, bit);
    } while (sym != 1);
    return price;
}
template <class PR>
__device__ __forceinline__ uint32_t pr_tree_rev(PR probs, const uint8_t* ptab, uint32_t base,
        uint32_t nbits, uint32_t sym)
{
    uint32_t price = 0, m = 1;
    do {
        const uint32_t bit = sym & 1;
        sym >>= 1;
        price += pr_bit(probs, ptab, base + m, bit);
        m = (m << 1) + bit;
    } while (--nbits);
    return price;
}
// price of one length under a length coder (length_update_prices, lzma_encoder.c:77-102, evaluated for a single length)
template <class PR>
__device__ __forceinline__ uint32_t pr_len_one(PR probs, const uint8_t* ptab, uint32_t base, uint32_t ps, uint32_t len)
{
    len -= 2;
    if (len < 8)
        return pr_bit(probs, ptab, base + LEN_CHOICE, 0) + pr_tree(probs, ptab, base + LEN_LOW + ps * 8, 3, len);
    len -= 8;
    if (len < 8)
        return pr_bit(probs, ptab, base + LEN_CHOICE, 1) + pr_bit(probs, ptab, base + LEN_CHOICE2, 0)
                + pr_tree(probs, ptab, base + LEN_MID + ps * 8, 3, len);
    return pr_bit(probs, ptab, base + LEN_CHOICE, 1) + pr_bit(probs, ptab, base + LEN_CHOICE2, 1)
            + pr_tree(probs, ptab, base + LEN_HIGH, 8, len - 8);
}

// Length price tables live in registers, lane = length: lane holds lengths 2 + lane + 64*it.
// The low/mid trees depend on pos_state but cover only lengths 2..17 (lanes 0..15 of it == 0);
// the high tree is shared by all pos_states.  Low 16 bits = match length coder, high 16 = rep.
struct LenTab { uint32_t lo[4]; uint32_t hi[5]; };

// tmp: 256 words of LDS that are dead while the tables are rebuilt (the parser's node array)
template <class PR>
__device__ __forceinline__ void refresh_len_tables(PR probs, const uint8_t* ptab, LenTab& t, uint32_t nps,
        uint32_t* tmp)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t m1 = pr_bit(probs, ptab, P_MATCH_LEN + LEN_CHOICE, 1), r1 = pr_bit(probs, ptab, P_REP_LEN + LEN_CHOICE, 1);
    {
        const uint32_t mh = m1 + pr_bit(probs, ptab, P_MATCH_LEN + LEN_CHOICE2, 1);
        const uint32_t rh = r1 + pr_bit(probs, ptab, P_REP_LEN + LEN_CHOICE2, 1);
#pragma unroll 1
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t sym = lane + 64u * k;
            const uint32_t pm = mh + pr_tree(probs, ptab, P_MATCH_LEN + LEN_HIGH, 8, sym);
            const uint32_t prr = rh + pr_tree(probs, ptab, P_REP_LEN + LEN_HIGH, 8, sym);
            tmp[sym] = pm | (prr << 16);
        }
    }
    wave_sync();
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        const uint32_t len = 2 + lane + 64u * it;
        t.hi[it] = (len >= 18 && len <= MATCH_LEN_MAX) ? tmp[len - 18] : 0;
    }
    wave_sync();
    const uint32_t m0 = pr_bit(probs, ptab, P_MATCH_LEN + LEN_CHOICE, 0), r0 = pr_bit(probs, ptab, P_REP_LEN + LEN_CHOICE, 0);
    const uint32_t mm = m1 + pr_bit(probs, ptab, P_MATCH_LEN + LEN_CHOICE2, 0);
    const uint32_t rm = r1 + pr_bit(probs, ptab, P_REP_LEN + LEN_CHOICE2, 0);
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        uint32_t v = 0;
        if (lane < 16 && (uint32_t)ps < nps) {
            const uint32_t sub = (lane < 8 ? LEN_LOW : LEN_MID) + (uint32_t)ps * 8, sym = lane & 7;
            const uint32_t pm = (lane < 8 ? m0 : mm) + pr_tree(probs, ptab, P_MATCH_LEN + sub, 3, sym);
            const uint32_t prr = (lane < 8 ? r0 : rm) + pr_tree(probs, ptab, P_REP_LEN + sub, 3, sym);
            v = pm | (prr << 16);
        }
        t.lo[ps] = v;
    }
}

template <class PR>
__device__ __forceinline__ void refresh_dist_tables(PR probs, const Work& w)
{
    const uint32_t lane = threadIdx.x;
#pragma unroll 1
    for (uint32_t i = lane; i < 256; i += 64) {
        const uint32_t ds = i >> 6, slot = i & 63;
        uint32_t pr = pr_tree(probs, w.ptab, P_DIST_SLOT + ds * 64, 6, slot);
        if (slot >= 14) pr += (((slot >> 1) - 1) - 4) << 4;
        w.dsp[i] = (uint16_t)pr;
    }
    // footer of distances < 128: the reverse bit tree of slots 4..13 (lzma_encoder.c:160-168); the
    // slot part is in dsp, so  price(dist < 128) = dsp[ds][slot] + xt[dist]  (the sum the reference
    // keeps in dist_prices[], optimum_normal.c:132-183)
#pragma unroll 1
    for (uint32_t d = lane; d < 128; d += 64) {
        uint32_t pr = 0;
        if (d >= 4) {
            const uint32_t slot = dist_slot_of(d);
            const uint32_t fb = (slot >> 1) - 1;
            const uint32_t base = (2 | (slot & 1)) << fb;
            pr = pr_tree_rev(probs, w.ptab, P_DIST_SPECIAL + base - slot - 1, fb, d - base);
        }
        w.xt[d] = (uint16_t)pr;
    }
    wave_sync();
}

template <class PR>
__device__ __forceinline__ void refresh_align_table(PR probs, const Work& w)
{
    const uint32_t lane = threadIdx.x;
    if (lane < 16) w.ap[lane] = (uint16_t)pr_tree_rev(probs, w.ptab, P_DIST_ALIGN, 4, lane);
    wave_sync();
}

// ---- literal prices (get_literal_price, optimum_normal.c:21-53) --------------------------------
// Probabilities do not change inside a parser window, so everything that depends only on the input
// bytes is priced once per 64 nodes, lane = node.  With N_i / A_i / B_i the price of bit i of the
// byte in the plain coder, in the matched coder when the match byte has the same bit, and in the
// matched coder when it differs, a literal costs
//     plain (state < 7):                          sum N_i
//     matched, first differing bit k of (byte ^ match byte):  sum_{i<k} A_i + B_k + sum_{i>k} N_i
//     matched, match byte == byte (k = 8):        sum A_i
// i.e. ten values per node, none of which depends on the path; the node later picks one by k.
// 24 independent gathers per lane, one memory round trip per 64 nodes.
struct LitChunk { uint32_t v[5]; };     // (V0,V1) (V2,V3) (V4,V5) (V6,V7) 16 bits each; V8 | plain << 11 | byte << 22

__device__ __forceinline__ void lit_chunk(const uint8_t* __restrict__ in, const uint8_t* ptab, const Lz& z,
        uint32_t x0, uint32_t block_start, uint32_t span_end, LitChunk& c)
{
    const uint32_t lane = threadIdx.x;
    uint32_t x = x0 + lane;
    if (x >= span_end) x = span_end - 1;
    const uint32_t cur = in[x];
    const uint32_t prev = x > block_start ? in[x - 1] : 0;
    const uint32_t upos = x - block_start;
    const uint32_t mask = (0x100u << z.lp) - (0x100u >> z.lc);
    const plit_t* sub = z.lit + 3u * ((((upos << 8) + prev) & mask) << z.lc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the coder's probability updates must have landed
    uint32_t pn[8], pa[8], pb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t pre = (0x100u | cur) >> (8 - i);
        const uint32_t bit = (cur >> (7 - i)) & 1;
        pn[i] = lit_load(sub + pre);
        pa[i] = lit_load(sub + 0x100u + (bit << 8) + pre);
        pb[i] = lit_load(sub + 0x100u + ((bit ^ 1u) << 8) + pre);
    }
    uint32_t N[8], A[8], B[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t bit = (cur >> (7 - i)) & 1;
        const uint32_t flip = (0u - bit) & 0x7FFu;
        N[i] = ptab[(pn[i] ^ flip) >> 4];
        A[i] = ptab[(pa[i] ^ flip) >> 4];
        B[i] = ptab[(pb[i] ^ flip) >> 4];
    }
    uint32_t sufN[9];                      // sufN[k] = sum_{i>=k} N_i
    sufN[8] = 0;
#pragma unroll
    for (int i = 7; i >= 0; --i) sufN[i] = sufN[i + 1] + N[i];
    uint32_t V[9], preA = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        V[k] = preA + B[k] + sufN[k + 1];
        preA += A[k];
    }
    V[8] = preA;
    c.v[0] = V[0] | (V[1] << 16);
    c.v[1] = V[2] | (V[3] << 16);
    c.v[2] = V[4] | (V[5] << 16);
    c.v[3] = V[6] | (V[7] << 16);
    c.v[4] = V[8] | (sufN[0] << 11) | (cur << 22);
}

// price of the literal at chunk node jj; mb = match byte (only used when state >= 7)
__device__ __forceinline__ uint32_t lit_price(const LitChunk& c, uint32_t jj, uint32_t state, uint32_t mb)
{
    const uint32_t w4 = lane_of(c.v[4], jj);
    if (state < 7) return (w4 >> 11) & 0x7FFu;
    const uint32_t d = ((w4 >> 22) ^ mb) & 0xFFu;
    if (d == 0) return w4 & 0x7FFu;
    const uint32_t k = (uint32_t)__builtin_clz(d) - 24;          // first differing bit, MSB first
    const uint32_t r = k < 2 ? lane_of(c.v[0], jj) : k < 4 ? lane_of(c.v[1], jj) : k < 6 ? lane_of(c.v[2], jj) : lane_of(c.v[3], jj);
    return (k & 1) ? r >> 16 : r & 0xFFFFu;
}

// Relax all rep / match lengths out of node j (lane = length).  A lane that improves its target
// node also writes the node's coder state and rep distances (they follow from this edge), so a node
// is complete the moment the parser reaches it.  Written as selects, not branches: per-lane `if`s
// cost the compiler an exec-mask save/restore and a branch each.
__device__ __forceinline__ void relax_lengths(const Work& w, const LenTab& lt, uint32_t PS, uint32_t SL, uint32_t SD, uint32_t j, uint32_t reach,
        uint32_t longest, uint32_t cnt, uint32_t rl0, uint32_t rl1, uint32_t rl2, uint32_t rl3,
        uint32_t prep0, uint32_t prep1, uint32_t prep2, uint32_t prep3, uint32_t pmatch,
        uint32_t s, uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t info_rep = (s < 7 ? 8u : 11u) << 9, info_match = (s < 7 ? 7u : 10u) << 9;
    const uint32_t lo_ps = PS == 0 ? lt.lo[0] : PS == 1 ? lt.lo[1] : PS == 2 ? lt.lo[2] : lt.lo[3];
    const uint32_t SLx = lane + 1 < cnt ? SL : 0xFFFFu;      // list lengths with everything from the last entry on = infinity
    const bool any_rep = (rl0 | rl1 | rl2 | rl3) != 0;
    // not unrolled: one pass covers 64 lengths and is all that nearly every node needs; five copies of
    // the body would only cost instruction-cache space
#pragma unroll 1
    for (uint32_t it = 0; it < 5; ++it) {
        if (2 + 64u * it > reach) break;
        const uint32_t l = 2 + lane + 64u * it;
        uint32_t idx_m = 0;                       // first entry whose length reaches l (or the last one)
#pragma unroll
        for (uint32_t k = 0; k + 1 < LIST_K; ++k) idx_m += (lane_of(SLx, k) < l) ? 1u : 0u;   // entries >= cnt-1 hold "infinity"
        const uint32_t dist_m = __shfl(SD, idx_m);
        const uint32_t cur = w.n_price[j + l];
        const uint32_t hv = it == 0 ? lt.hi[0] : it == 1 ? lt.hi[1] : it == 2 ? lt.hi[2] : it == 3 ? lt.hi[3] : lt.hi[4];
        const uint32_t lv = (it == 0 && lane < 16) ? lo_ps : hv;
        const uint32_t lpm = lv & 0xFFFFu, lpr = lv >> 16;
        // distance price = slot price (+ direct bits) + footer / align price: two table reads
        const uint32_t ds = l < 6 ? l - 2 : 3;
        const uint32_t dnz = dist_m < 4 ? 4u : dist_m;                     // branch-free slot (fastpos.h:78-86)
        const uint32_t di = 31 - (uint32_t)__builtin_clz(dnz);
        const uint32_t slot = dist_m < 4 ? dist_m : 2 * di + ((dnz >> (di - 1)) & 1);
        const uint32_t p_dist = (uint32_t)w.dsp[ds * 64 + slot] + w.xt[dist_m < 128 ? dist_m : 128 + (dist_m & 15)];
        const uint32_t c4 = l <= longest ? pmatch + lpm + p_dist : PRICE_INF;
        uint32_t best = cur, bb = 0, n0 = r0;
        if (any_rep) {                              // uniform: most nodes of a text have no usable rep match
            const uint32_t c0 = rl0 >= l ? prep0 + lpr : PRICE_INF;
            const uint32_t c1 = rl1 >= l ? prep1 + lpr : PRICE_INF;
            const uint32_t c2 = rl2 >= l ? prep2 + lpr : PRICE_INF;
            const uint32_t c3 = rl3 >= l ? prep3 + lpr : PRICE_INF;
            { const bool t = c0 < best; best = t ? c0 : best; bb = t ? 0u : bb; }
            { const bool t = c1 < best; best = t ? c1 : best; bb = t ? 1u : bb; n0 = t ? r1 : n0; }
            { const bool t = c2 < best; best = t ? c2 : best; bb = t ? 2u : bb; n0 = t ? r2 : n0; }
            { const bool t = c3 < best; best = t ? c3 : best; bb = t ? 3u : bb; n0 = t ? r3 : n0; }
        }
        { const bool t = c4 < best; best = t ? c4 : best; bb = t ? dist_m + 4 : bb; n0 = t ? dist_m : n0; }
        const uint32_t n1 = bb == 0 ? r1 : r0;
        const uint32_t n2 = bb <= 1 ? r2 : r1;
        const uint32_t n3 = bb <= 2 ? r3 : r2;
        if (l <= reach && best < cur) {
            w.n_price[j + l] = best;
            w.n_info[j + l] = l | (bb < 4 ? info_rep | (bb << 22) : info_match | (4u << 22));
            w.n_reps4[j + l] = make_uint4(n0, n1, n2, n3);
        }
    }
}

// ---- compound edges "X + literal + rep0" (lzma_encoder_optimum_normal.c:562-597, 635-687, 728-790) ----
// Seven candidates per node, one lane each: lane 0 = "literal + rep0" (X empty), lanes 1..4 = rep0..rep3
// at its full length, lanes 5 / 6 = the longest / second longest match of the list at its full length.
// Geometry (lengths, target node) depends only on the round; it is fixed before any edge of the node
// is relaxed so that the window end covers the targets.  The rep0 run behind the literal is limited to
// the 64-byte row at x for lanes 0..4 (the rep compare masks) and comes from the list trailer for 5 / 6.
struct Compound {
    uint32_t L1, l2, T;     // per lane: length of X, rep0 run, target node (valid lanes)
    uint32_t dist;          // per lane: distance X leaves in rep0 (lanes 1..6)
    uint64_t mask;          // valid candidate lanes
};

__device__ __forceinline__ uint32_t mask_run_after(uint64_t m, uint32_t first)
{
    // equal bytes behind the mismatch at offset `first` inside the row (m: bit o = offset o differs / past the end)
    const uint64_t rest = first >= 63 ? 0ull : (m >> (first + 1));
    return rest ? (uint32_t)__builtin_ctzll(rest) : 63u - min(first, 63u);
}

__device__ __forceinline__ void compound_setup(const RoundL& RL, uint32_t j, uint32_t room, uint32_t buf_avail,
        uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, Compound& c, const uint32_t WMAX)
{
    const uint32_t lane = threadIdx.x;
    // uniform per-candidate geometry
    uint32_t L1[7], l2[7];
    bool ok[7];
    L1[0] = 0;
    ok[0] = RL.rp[0] == 0;                                     // rep0's byte differs here (buf_avail >= 1 inside a span)
    l2[0] = ok[0] ? mask_run_after(rm_of(RL, 0), 0) : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t l = RL.rp[i];
        L1[1 + i] = l;
        ok[1 + i] = l >= 2 && l <= room && l < buf_avail && l < 64;
        l2[1 + i] = ok[1 + i] ? mask_run_after(rm_of(RL, i), l) : 0;
    }
    const uint32_t cnt = RL.cnt;
    L1[5] = cnt >= 1 ? RL.longest : 0;
    L1[6] = cnt >= 2 ? lane_of(RL.SL, cnt >= 2 ? cnt - 2 : 0) : 0;
    l2[5] = RL.l2a; l2[6] = RL.l2b;
    ok[5] = cnt >= 1 && L1[5] >= 2 && L1[5] <= room && L1[5] <= 61;
    ok[6] = cnt >= 2 && L1[6] >= 2 && L1[6] <= room && L1[6] <= 61;
    const uint32_t dm0 = cnt >= 1 ? lane_of(RL.SD, cnt - 1) : 0, dm1 = cnt >= 2 ? lane_of(RL.SD, cnt - 2) : 0;
    uint64_t m = 0;
    uint32_t vL1 = 0, vl2 = 0, vT = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const uint32_t T = j + L1[k] + 1 + l2[k];
        const bool v = ok[k] && l2[k] >= 2 && T <= WMAX;
        if (v) m |= 1ull << k;
        vL1 = lane == (uint32_t)k ? L1[k] : vL1;
        vl2 = lane == (uint32_t)k ? l2[k] : vl2;
        vT = lane == (uint32_t)k ? T : vT;
    }
    c.L1 = vL1; c.l2 = vl2; c.T = vT;
    c.dist = lane == 1 ? r0 : lane == 2 ? r1 : lane == 3 ? r2 : lane == 4 ? r3 : lane == 5 ? dm0 : dm1;
    c.mask = m;
}

// One window of the optimal parser (oracle: optimum_window).  Returns with the chosen symbol path
// stored as out-edges: node t -> (n_price[t] = back, out-len in n_info[t]); q_end = last node to code
// (a window cut by the node limit only commits the symbols that end WTAIL nodes before the cut).
constexpr uint32_t WTAIL = 32;
template <uint32_t WMAX, class PR>
__device__ __forceinline__ bool optimum_window(const Env& e, const Work& w, ListPre& P, PR probs, const Lz& z, LenTab& lt,
        const uint8_t* __restrict__ in, uint32_t pos, uint32_t block_start, uint32_t span_end, bool cached,
        RoundL& RL, uint32_t& q_end)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t pbm = (1u << z.pb) - 1;
    // Bit-price combinations that depend only on (state, pos_state): lane = state * 4 + pos_state
    // holds all seven of them; a node fetches its set with four readlanes.
    uint32_t c01 = 0, c23 = 0, c45 = 0, c6 = 0;
    {
        const uint32_t s = lane >> 2, ps = lane & 3;
        if (lane < 48 && ps <= pbm) {
            const uint8_t* pt = w.ptab;
            const uint32_t m0 = pr_bit(probs, pt, P_IS_MATCH + s * 16 + ps, 0), m1 = pr_bit(probs, pt, P_IS_MATCH + s * 16 + ps, 1);
            const uint32_t e0 = pr_bit(probs, pt, P_IS_REP + s, 0), e1 = pr_bit(probs, pt, P_IS_REP + s, 1);
            const uint32_t g0 = pr_bit(probs, pt, P_IS_REP0 + s, 0), g1 = pr_bit(probs, pt, P_IS_REP0 + s, 1);
            const uint32_t l0 = pr_bit(probs, pt, P_IS_REP0_LONG + s * 16 + ps, 0), l1 = pr_bit(probs, pt, P_IS_REP0_LONG + s * 16 + ps, 1);
            const uint32_t h0 = pr_bit(probs, pt, P_IS_REP1 + s, 0), h1 = pr_bit(probs, pt, P_IS_REP1 + s, 1);
            const uint32_t k0 = pr_bit(probs, pt, P_IS_REP2 + s, 0), k1 = pr_bit(probs, pt, P_IS_REP2 + s, 1);
            const uint32_t prep = m1 + e1, p11 = prep + g1 + h1;
            c01 = m0 | ((prep + g0 + l0) << 16);                 // literal | short rep
            c23 = (prep + g0 + l1) | ((prep + g1 + h0) << 16);   // rep0 | rep1
            c45 = (p11 + k0) | ((p11 + k1) << 16);               // rep2 | rep3
            c6 = m1 + e0;                                        // match
        }
    }
    for (uint32_t t = 1 + lane; t <= WMAX; t += 64) w.n_price[t] = PRICE_INF;     // once per window, not per node
    if (lane == 0) {
        w.n_price[0] = 0;
        w.n_info[0] = z.state << 9;
    }
    wave_sync();
    uint32_t n_end = 0, j = 0;
    bool next_cached = false, forced = false;
    LitChunk lc;                          // lane = node - (j & ~63): the node's literal prices
    lc.v[0] = lc.v[1] = lc.v[2] = lc.v[3] = lc.v[4] = 0;
    const uint32_t lmask = (0x100u << z.lp) - (0x100u >> z.lc);
    // node 0: the coder's state; its round may be cached from the previous window
    uint32_t s = z.state, r0 = z.rep0, r1 = z.rep1, r2 = z.rep2, r3 = z.rep3, Pj = 0;
    uint32_t b_mb = in[pos - r0 - 1];               // rep0's byte at the node: issued early, used only for a matched literal
    lit_chunk(in, w.ptab, z, pos, block_start, span_end, lc);
    if (!cached) round_lists(e, P, pos, span_end, r0, r1, r2, r3, RL);
    uint32_t longest = RL.longest;
    {
        // a window that opens with a >= nice_len rep or match is that symbol (optimum_normal.c:281-306)
        uint32_t sb = LITERAL, sl = 0;
        if (RL.rl[0] >= e.nice) { sb = 0; sl = RL.rl[0]; }
        else if (RL.rl[1] >= e.nice) { sb = 1; sl = RL.rl[1]; }
        else if (RL.rl[2] >= e.nice) { sb = 2; sl = RL.rl[2]; }
        else if (RL.rl[3] >= e.nice) { sb = 3; sl = RL.rl[3]; }
        else if (longest >= e.nice) { sb = lane_of(RL.SD, RL.cnt - 1) + 4; sl = longest; }
        if (sl) {
            wave_sync();
            if (lane == 0) { w.n_price[0] = sb; w.n_info[0] = (z.state << 9) | (sl << 13); }
            wave_sync();
            q_end = sl;
            return false;
        }
    }
    // The loop is rotated: its body prices the edges out of node j, whose state and round are already there;
    // the state and the round of node j + 1 are fetched at the bottom.
    for (;;) {
        const uint32_t x = pos + j;
        TM_COUNT(w, 9);
        TM_BEGIN(t_bits);
        const uint32_t rp0 = RL.rp[0];
        uint32_t rl0 = RL.rl[0], rl1 = RL.rl[1], rl2 = RL.rl[2], rl3 = RL.rl[3];
        const uint32_t room = WMAX - j;
        const uint32_t avail = span_end - x;
        const uint32_t buf_avail = avail < MATCH_LEN_MAX ? avail : MATCH_LEN_MAX;
        // compound candidates: geometry now, prices after the plain edges.  A cheap necessary condition first
        // (two equal bytes behind the first mismatch of a rep source, or a run recorded with the list): text
        // rarely has any, and then none of the compound code runs.
        Compound cp;
        cp.mask = 0; cp.L1 = cp.l2 = cp.T = cp.dist = 0;
        {
            const bool any = (RL.l2a | RL.l2b) >= 2 || RL.pre;
            if (any) compound_setup(RL, j, room, buf_avail, r0, r1, r2, r3, cp, WMAX);
        }
        uint32_t cT_max = 0;
        for (uint64_t mm = cp.mask; mm; mm &= mm - 1) cT_max = max(cT_max, lane_of(cp.T, (uint32_t)__builtin_ctzll(mm)));
        // the three bytes each compound literal needs (lane = candidate 1..6): the byte itself, its
        // predecessor (literal context) and the match byte X leaves behind
        uint32_t c_bytes = 0;
        if (((cp.mask >> lane) & 1) && lane >= 1) {
            const uint8_t* pl = in + x + cp.L1;
            c_bytes = (uint32_t)pl[0] | ((uint32_t)*(pl - cp.dist - 1) << 8) | ((uint32_t)*(pl - 1) << 16);
        }

        uint32_t rmax = max(max(rl0, rl1), max(rl2, rl3));
        uint32_t reach = max(longest, rmax);
        if (reach > room) {                         // only the last nodes of a window can reach past its end
            longest = min(longest, room);
            rl0 = min(rl0, room); rl1 = min(rl1, room); rl2 = min(rl2, room); rl3 = min(rl3, room);
            rmax = min(rmax, room);
            reach = room;
        }
        n_end = max(max(max(n_end, j + reach), j + 1), cT_max);     // (every node price was set to infinity at the window start)

        const uint32_t upos = x - block_start;
        const uint32_t ps = upos & pbm;
        const uint32_t ci = s * 4 + ps;
        const uint32_t v01 = lane_of(c01, ci), v23 = lane_of(c23, ci), v45 = lane_of(c45, ci), v6 = lane_of(c6, ci);
        TM_END(w, 2, t_bits);

        // compound literal prices: lane = candidate * 8 + bit (candidates 1..6), one gather of the eight
        // probabilities each matched literal would use (literal_matched, lzma_encoder.c:23-41), issued
        // here and summed after the plain edges
        TM_BEGIN(t_cg);
        uint32_t cg_p = 0, cg_flip = 0;
        const bool cg_any = (cp.mask & 0x7Eull) != 0;
        if (cg_any) {
            const uint32_t packed = c_bytes;
            const uint32_t cand = lane >> 3, bit_i = lane & 7;
            const uint32_t pk = (uint32_t)__shfl((int)packed, (int)cand);
            const uint32_t cL = (uint32_t)__shfl((int)cp.L1, (int)cand);
            const bool act = cand >= 1 && cand <= 6 && ((cp.mask >> cand) & 1);
            const uint32_t cur = pk & 0xFFu, mb = (pk >> 8) & 0xFFu, prev = (pk >> 16) & 0xFFu;
            const uint32_t up = upos + cL;
            const uint32_t sub = 3u * ((((up << 8) + prev) & lmask) << z.lc);
            const uint32_t pre = (0x100u | cur) >> (8 - bit_i);
            const bool same = (mb >> (8 - bit_i)) == (cur >> (8 - bit_i));
            const uint32_t idx = same ? 0x100u + (((mb >> (7 - bit_i)) & 1u) << 8) + pre : pre;
            const uint32_t bit = (cur >> (7 - bit_i)) & 1u;
            cg_flip = (0u - bit) & 0x7FFu;
            if (act) cg_p = lit_load(z.lit + sub + idx);
        }
        TM_END(w, 15, t_cg);

        TM_BEGIN(t_lit);
        // literal and short rep -> node j+1
        const uint32_t lp = lit_price(lc, j & 63, s, s < 7 ? 0u : uni(b_mb));
        const uint32_t plit = Pj + (v01 & 0xFFFFu) + lp;
        {
            uint32_t best = uni(w.n_price[j + 1]), bb = 0, ns = 0;
            bool upd = false;
            if (plit < best) { best = plit; bb = LITERAL; upd = true; ns = s <= 3 ? 0 : (s <= 9 ? s - 3 : s - 6); }
            if (rp0 >= 1) {
                const uint32_t psr = Pj + (v01 >> 16);
                if (psr < best) { best = psr; bb = 0; upd = true; ns = s < 7 ? 9u : 11u; }
            }
            if (upd && lane == 0) {
                w.n_price[j + 1] = best; w.n_info[j + 1] = 1 | (ns << 9) | ((bb == LITERAL ? 5u : 0u) << 22);
                w.n_reps4[j + 1] = make_uint4(r0, r1, r2, r3);
            }
            wave_sync();
        }
        // Node j + 1 is final now (every other edge out of j is longer): its rows go out here and are in flight
        // while the length and compound edges of node j are priced.
        Rows NW;
        uint32_t n_mb = 0;
        const bool rows_early = j + 1 != n_end;
        if (rows_early) {
            const uint4 nr = w.n_reps4[j + 1];
            n_mb = in[x - nr.x];                         // in[(x + 1) - rep0 - 1]
            rows_issue(e, x + 1, span_end, nr.x, nr.y, nr.z, nr.w, NW);
        }
        TM_END(w, 3, t_lit);
        TM_BEGIN(t_relax);
        const uint32_t prep0 = Pj + (v23 & 0xFFFFu), prep1 = Pj + (v23 >> 16), prep2 = Pj + (v45 & 0xFFFFu), prep3 = Pj + (v45 >> 16);
        const uint32_t pmatch = Pj + v6;
        if (reach >= 2) {
            relax_lengths(w, lt, ps & 3, RL.SL, RL.SD, j, reach, longest, RL.cnt, rl0, rl1, rl2, rl3,
                    prep0, prep1, prep2, prep3, pmatch, s, r0, r1, r2, r3);
            wave_sync();
        }
        TM_END(w, 4, t_relax);

        TM_BEGIN(t_cp);
        if (cp.mask) {
            TM_COUNT(w, 13);
            // ---- price the compound candidates, lane = candidate ----
            // length prices: LenTab holds length 2 + l + 64 * it in lane l (low half match, high half rep)
            const uint32_t lo_ps = (ps & 3) == 0 ? lt.lo[0] : (ps & 3) == 1 ? lt.lo[1] : (ps & 3) == 2 ? lt.lo[2] : lt.lo[3];
            const uint32_t i1 = lane < 7 && cp.L1 >= 2 ? cp.L1 - 2 : 0u;          // L1 <= 63: first pass of the table
            const uint32_t t_lo = (uint32_t)__shfl((int)lo_ps, (int)(i1 & 63)), t_hi0 = (uint32_t)__shfl((int)lt.hi[0], (int)(i1 & 63));
            const uint32_t lenX = i1 < 16 ? t_lo : t_hi0;                           // X's length price (match | rep << 16)
            const uint32_t i2 = lane < 7 && cp.l2 >= 2 ? cp.l2 - 2 : 0u;            // l2 <= 127
            const uint32_t psn = (upos + cp.L1 + 1) & pbm & 3;
            const uint32_t u0 = (uint32_t)__shfl((int)lt.lo[0], (int)(i2 & 63)), u1 = (uint32_t)__shfl((int)lt.lo[1], (int)(i2 & 63));
            const uint32_t u2 = (uint32_t)__shfl((int)lt.lo[2], (int)(i2 & 63)), u3 = (uint32_t)__shfl((int)lt.lo[3], (int)(i2 & 63));
            const uint32_t h0v = (uint32_t)__shfl((int)lt.hi[0], (int)(i2 & 63)), h1v = (uint32_t)__shfl((int)lt.hi[1], (int)(i2 & 63));
            const uint32_t ulo = psn == 0 ? u0 : psn == 1 ? u1 : psn == 2 ? u2 : u3;
            const uint32_t len2p = (i2 < 16 ? ulo : i2 < 64 ? h0v : h1v) >> 16;     // rep length price of the rep0 run
            // state chain: X -> literal -> long rep0
            const bool isrep = lane >= 1 && lane <= 4;
            const uint32_t sX = lane == 0 ? s : isrep ? (s < 7 ? 8u : 11u) : (s < 7 ? 7u : 10u);
            const uint32_t sL = lane == 0 ? (s <= 3 ? 0u : (s <= 9 ? s - 3 : s - 6)) : isrep ? 5u : 4u;   // state after the literal
            const uint32_t psl = (upos + cp.L1) & pbm & 3;
            const uint32_t m0X = (uint32_t)__shfl((int)c01, (int)(sX * 4 + psl)) & 0xFFFFu;               // is_match = 0 before the literal
            const uint32_t rafter = ((uint32_t)__shfl((int)c23, (int)(sL * 4 + psn)) & 0xFFFFu) + len2p;  // rep0 behind the literal
            // price of X
            uint32_t pX;
            {
                const uint32_t dist_m = cp.dist;
                const uint32_t l = cp.L1;
                const uint32_t ds = l < 6 ? (l >= 2 ? l - 2 : 0u) : 3u;
                const uint32_t dnz = dist_m < 4 ? 4u : dist_m;
                const uint32_t di = 31 - (uint32_t)__builtin_clz(dnz);
                const uint32_t slot = dist_m < 4 ? dist_m : 2 * di + ((dnz >> (di - 1)) & 1);
                const bool ism = lane == 5 || lane == 6;
                const uint32_t p_dist = ism ? (uint32_t)w.dsp[ds * 64 + slot] + w.xt[dist_m < 128 ? dist_m : 128 + (dist_m & 15)] : 0u;
                const uint32_t prep_i = lane == 1 ? prep0 : lane == 2 ? prep1 : lane == 3 ? prep2 : prep3;
                pX = ism ? pmatch + (lenX & 0xFFFFu) + p_dist : prep_i + (lenX >> 16);
            }
            // matched-literal prices: sum the eight bit prices of each candidate's group of lanes
            uint32_t litp = 0;
            if (cg_any) {
                uint32_t bp = w.ptab[(cg_p ^ cg_flip) >> 4];
                const uint32_t cand = lane >> 3;
                if (!(cand >= 1 && cand <= 6 && ((cp.mask >> cand) & 1))) bp = 0;
                bp += (uint32_t)__shfl_xor((int)bp, 1);
                bp += (uint32_t)__shfl_xor((int)bp, 2);
                bp += (uint32_t)__shfl_xor((int)bp, 4);
                litp = (uint32_t)__shfl((int)bp, (int)((lane < 7 ? lane : 0u) * 8));
            }
            const uint32_t cprice = lane == 0 ? plit + rafter : pX + m0X + litp + rafter;
            // apply in candidate order (strict <, like every other edge)
            for (uint64_t mm = cp.mask; mm; mm &= mm - 1) {
                const uint32_t c = (uint32_t)__builtin_ctzll(mm);
                const uint32_t T = lane_of(cp.T, c), pr = lane_of(cprice, c);
                const uint32_t curp = uni(w.n_price[T]);
                if (pr < curp) {
                    const uint32_t cL = lane_of(cp.L1, c), cl2 = lane_of(cp.l2, c), cd = lane_of(cp.dist, c);
                    const uint32_t kind = c == 0 ? 5u : c <= 4 ? c - 1 : 4u;
                    uint4 nr;
                    if (c == 0) nr = make_uint4(r0, r1, r2, r3);
                    else if (c == 1) nr = make_uint4(r0, r1, r2, r3);
                    else if (c == 2) nr = make_uint4(r1, r0, r2, r3);
                    else if (c == 3) nr = make_uint4(r2, r0, r1, r3);
                    else if (c == 4) nr = make_uint4(r3, r0, r1, r2);
                    else nr = make_uint4(cd, r0, r1, r2);
                    if (lane == 0) {
                        w.n_price[T] = pr;
                        w.n_info[T] = cL | (8u << 9) | (kind << 22) | (cl2 << 25);
                        w.n_reps4[T] = nr;
                    }
                }
                wave_sync();
            }
        }
        TM_END(w, 14, t_cp);
        ++j;
        if (j == n_end) { forced = j >= WMAX; break; }
        // state and round of the next node
        {
            TM_BEGIN(t_derive);
            const uint4 rr = w.n_reps4[j];
            s = (uni(w.n_info[j]) >> 9) & 15;
            Pj = uni(w.n_price[j]);
            r0 = uni(rr.x); r1 = uni(rr.y); r2 = uni(rr.z); r3 = uni(rr.w);
            TM_END(w, 0, t_derive);
            TM_BEGIN(t_round);
            const uint32_t xn = pos + j;
            b_mb = n_mb;                                   // rows and rep0 byte of this node: issued by its predecessor
            if ((j & 63) == 0) lit_chunk(in, w.ptab, z, xn, block_start, span_end, lc);
            round_lists_rows(e, P, xn, span_end, r0, r1, r2, r3, NW, RL);
            longest = RL.longest;
            TM_END(w, 1, t_round);
            if (longest >= e.nice) { next_cached = true; break; }
        }
    }
    TM_BEGIN(t_back);
    // backtrack from node j: turn in-edges into out-edges.  A compound in-edge becomes three (two when
    // X is empty) out-edges through its intermediate nodes.  `cut` = largest symbol boundary that lies
    // at least WTAIL nodes before a forced window end, `first` = end of the first symbol.
    if (lane == 0) {
        const uint32_t limit = j >= WTAIL ? j - WTAIL : 0;
        uint32_t t = j, cut = 0, first = j;
        while (t > 0) {
            const uint32_t info = w.n_info[t];
            const uint32_t ilen = info & 0x1FF, kind = (info >> 22) & 7, cl2 = info >> 25;
            const uint32_t bk = kind < 4 ? kind : kind == 4 ? w.n_reps4[t].x + 4 : LITERAL;
            const uint32_t pv = t - ilen - (cl2 ? 1 + cl2 : 0);
            if (cut == 0 && t <= limit) cut = t;
            first = t;
            if (cl2) {
                const uint32_t a = pv + ilen;                 // node of the literal, a + 1 = start of the rep0 run
                w.n_price[a + 1] = 0;                          // rep0
                w.n_info[a + 1] = cl2 << 13;
                if (cut == 0 && a + 1 <= limit) cut = a + 1;
                first = a + 1;
                if (ilen) {
                    w.n_price[a] = LITERAL;
                    w.n_info[a] = 1u << 13;
                    if (cut == 0 && a <= limit) cut = a;
                    first = a;
                    w.n_price[pv] = bk;
                    w.n_info[pv] = (w.n_info[pv] & 0xFFC01FFFu) | (ilen << 13);
                } else {
                    w.n_price[pv] = LITERAL;                   // X empty: the literal leaves pv itself
                    w.n_info[pv] = (w.n_info[pv] & 0xFFC01FFFu) | (1u << 13);
                }
            } else {
                w.n_price[pv] = bk;
                w.n_info[pv] = (w.n_info[pv] & 0xFFC01FFFu) | (ilen << 13);   // keep in-len, state, in-edge kind and len2 of pv
            }
            t = pv;
        }
        w.n_info[WMAX + 1] = (forced && pos + j < span_end) ? (cut ? cut : first) : j;   // spare word behind the node arrays
    }
    wave_sync();
    q_end = uni(w.n_info[WMAX + 1]);
    TM_END(w, 5, t_back);
    TM_COUNT(w, 11);
    return next_cached;
}

// ------------------------------------------------------------------------------------------
// Span encoder: one wavefront per span.  FINDER 0 = exact HC3/HC4 of the reference evaluated inside
// the kernel (fast parser), 2 = per-position match lists written by k_find_sn / k_find_exact;
// OPT selects the parser (false = optimum_fast of the reference).
// ------------------------------------------------------------------------------------------
template <int FINDER, bool OPT, uint32_t WMAX>      // FINDER: 0 = exact HC3/HC4 in-kernel, 2 = lists from the batch finders
#ifndef XZAMD_WAVES_FAST
#define XZAMD_WAVES_FAST 4
#endif
#ifndef XZAMD_WALK_LITG_DEFAULT
#define XZAMD_WALK_LITG_DEFAULT true
#endif
#ifndef XZAMD_WAVES_OPT
#define XZAMD_WAVES_OPT 4
#endif
__device__ __forceinline__ void span_encode_one(const xzamd_span_args& a, const uint32_t span)
{
    constexpr bool LISTS = FINDER == 2;
    static_assert(!LISTS || OPT, "the fast parsers run with the in-kernel finders");
    // One LDS pool, carved by hand (a single __shared__ object: no aliasing or ordering surprises).
    constexpr uint32_t W_PROBS = 928;                                    // 1856 x u16 >= P_LITERAL (1846): all but the literal coders
    constexpr uint32_t W_LIST = 0;
    constexpr uint32_t W_NODES = OPT ? 6 * (WMAX + 1) + 2 : 0;           // reps[4], price, info (16-byte multiple)
    constexpr uint32_t W_TABS = OPT ? (128 + 72 + 32) : 0;               // dsp, xt + ap (u16), ptab (u8)
#ifndef XZAMD_LDS_PAD_WORDS
#define XZAMD_LDS_PAD_WORDS 0        /* occupancy experiments only */
#endif
    __shared__ __attribute__((aligned(16))) uint32_t pool[W_PROBS + W_LIST + W_NODES + W_TABS + XZAMD_LDS_PAD_WORDS];
#ifdef XZAMD_TIMING
    __shared__ unsigned long long tm_lds[16];
    if (threadIdx.x < 16) tm_lds[threadIdx.x] = 0;
#endif
    uint16_t* const probs = reinterpret_cast<uint16_t*>(pool);
    const uint32_t lane = threadIdx.x;
    const uint32_t blk = span / a.max_spb;
    const uint32_t k = span - blk * a.max_spb;
    if (k >= a.span_cnt[blk]) return;                  // an unused slot of the span plan
    const uint32_t block_start = blk * a.block_size;
    const uint32_t block_end = min(a.n, block_start + a.block_size);
    const uint32_t span_start = uni(a.span_tab[2 * span]), span_end = uni(a.span_tab[2 * span + 1]);
    // scratch of the slot: 9/8 of the input bytes in front of it plus XZAMD_SPAN_SLACK per slot (kernels_api.h)
    uint8_t* const outp = a.scratch + ((((uint64_t)span_start + (span_start >> 3)) + 15) & ~15ull) + (uint64_t)span * XZAMD_SPAN_SLACK;
    const uint32_t span_cap = (span_end - span_start) + ((span_end - span_start) >> 3) + 4096;
    const uint8_t* __restrict__ in = a.in;

    Env e;
    e.in = in; e.rank = a.rank; e.sorted_pos = a.sorted_pos; e.prev2 = a.prev2; e.prev3 = a.prev3;
    e.nice = a.nice_len; e.depth = a.depth; e.hb = a.hash_bytes; e.cyclic = a.dict_size + 1;
    e.block_end = block_end; e.n_last = a.n - 1;
    e.mlen = a.mlen; e.mdist = a.mdist; e.packed = a.list_packed;
    ListPre LP;
    LP.valid = false; LP.pos = 0; LP.v = LP.l16 = 0;
    Pre P;
    P.valid = false; P.pos = 0; P.ent = 0;
    P.a.rk = P.a.d2 = P.a.d3 = 0; P.an = P.a;

    Work w{};
    if constexpr (OPT) {
        uint32_t* nb = pool + W_PROBS + W_LIST;
        w.n_reps4 = reinterpret_cast<uint4*>(nb);           // 16-byte aligned: W_PROBS * 4 is a multiple of 16
        w.n_price = nb + 4 * (WMAX + 1);
        w.n_info = nb + 5 * (WMAX + 1);
        uint32_t* tb = nb + W_NODES;
        w.dsp = reinterpret_cast<uint16_t*>(tb);            // 256 x u16 = 128 words
        w.xt = reinterpret_cast<uint16_t*>(tb + 128);       // 128 x u16 = 64 words
        w.ap = w.xt + 128;                                  // 16 x u16 = 8 words, contiguous with xt
        w.ptab = reinterpret_cast<uint8_t*>(tb + 200);      // 128 x u8 = 32 words
        w.err = a.err;
#ifdef XZAMD_TIMING
        w.tm = tm_lds;
#endif
        // bit price table (price_tablegen.c:31-58)
        for (uint32_t t = lane; t < 128; t += 64) {
            uint32_t wv = t * 16 + 8, bit_count = 0;
            for (int jj = 0; jj < 4; ++jj) {
                wv *= wv;
                bit_count <<= 1;
                while (wv >= (1u << 16)) { wv >>= 1; ++bit_count; }
            }
            w.ptab[t] = (uint8_t)((11 << 4) - 15 - bit_count);
        }
        wave_sync();
        w.ptv = reinterpret_cast<const uint32_t*>(w.ptab)[lane & 31];
    }

    Lz z;
    z.lc = a.lc; z.lp = a.lp; z.pb = a.pb;
    z.cnt_len = z.cnt_match = z.cnt_align = 0;
    const uint32_t lit_size = 0x300u << (a.lc + a.lp);       // literal coders of this span (lc + lp <= 4)
    z.lit = reinterpret_cast<plit_t*>(a.lit) + (uint64_t)span * lit_size;
    z.gp = nullptr;
    RC rc;
    rc.cpos = 0; rc.out = outp; rc.reset();

    bool need_props = true, need_dict_reset = (span_start == block_start), need_state_reset = true;
    bool initialized = (span_start != block_start);
    uint32_t cur = span_start;
    uint32_t read_ahead = 0;        // exact/fast path bookkeeping (as in the reference)
    bool cached = false;            // list paths: the round for `cur` has been done
    uint32_t out_off = 0;
    Round R;            // exact path: cached find at `cur` when read_ahead == 1
    R.mask = 0; R.L = 0; R.D = 0; R.longest = 0;
    RoundL RL;
    RL.L = 0; RL.SL = 0; RL.SD = 0; RL.cnt = 0; RL.longest = 0; RL.l2a = RL.l2b = 0;
    RL.rp[0] = RL.rp[1] = RL.rp[2] = RL.rp[3] = 0;
    RL.mlo = RL.mhi = 0;
    RL.rl[0] = RL.rl[1] = RL.rl[2] = RL.rl[3] = 0;
    RL.pre = 0;
    LenTab lt;
    uint32_t q_pos = 0, q_end = 0;  // pending path of the optimal parser (nodes in LDS)
    bool tables_valid = false;
#ifdef XZAMD_TIMING
    uint64_t tm_round1 = 0, tm_round2 = 0, tm_encode = 0, tm_rounds = 0, tm_syms = 0;
    const uint64_t tm_start = __builtin_amdgcn_s_memtime();
#endif

    while (cur < span_end) {
        if (need_state_reset) {
            // lzma_lzma_encoder_reset (lzma_encoder.c:529-598)
            uint32_t* p32 = reinterpret_cast<uint32_t*>(probs);
            for (uint32_t i = lane; i < 928; i += 64) p32[i] = 0x04000400u;
            {
                uint4* l4 = reinterpret_cast<uint4*>(z.lit);
                const uint4 v = PLIT_FLAT4;
                for (uint32_t i = lane; i < lit_size / PLIT_PER_U4; i += 64) l4[i] = v;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            wave_sync();
            z.state = 0; z.rep0 = z.rep1 = z.rep2 = z.rep3 = 0;
            tables_valid = false;
            q_pos = q_end = 0;
            // the flag stays set until an LZMA chunk header has announced the reset
        }
        const uint32_t chunk_start = cur;
        const uint32_t hl = need_props ? 6 : 5;
        rc.out = outp + out_off + hl;
        rc.cpos = 0;
        rc.reset();

        if (!initialized) {
            // encode_init (lzma_encoder.c:267-293)
            encode_symbol(rc, probs, z, in, block_start, 0, LITERAL, 1);
            cur = block_start + 1;
            initialized = true;
        }

        for (;;) {
            // lzma_encoder.c:346-351 (limit from lzma2_encoder.c:167-181)
            if (cur - chunk_start >= (1u << 21) - MATCH_LEN_MAX
                    || rc.cpos + rc.cache_size + 4 >= 65536 - 4097)
                break;
            if (cur >= span_end)
                break;

            uint32_t back = LITERAL, len = 1;
            if constexpr (OPT) {
                if (q_pos == q_end) {
                    // price-table refresh policy (oracle: refresh_tables)
                    TM_BEGIN(t_refresh);
                    if (!tables_valid || z.cnt_len >= 64) { refresh_len_tables(probs, w.ptab, lt, 1u << z.pb, reinterpret_cast<uint32_t*>(w.n_reps4)); z.cnt_len = 0; }
                    if (!tables_valid || z.cnt_match >= 128) { refresh_dist_tables(probs, w); z.cnt_match = 0; }
                    if (!tables_valid || z.cnt_align >= 16) { refresh_align_table(probs, w); z.cnt_align = 0; }
                    tables_valid = true;
                    TM_END(w, 7, t_refresh);
                    // Node 0's round is done here so that a window opening with a >= nice_len rep or
                    // match (optimum_window's j == 0 rule; the common case in binary data) is coded
                    // straight away, without building the window's price tables.
                    if (!cached) {
                        round_lists(e, LP, cur, span_end, z.rep0, z.rep1, z.rep2, z.rep3, RL);
                        cached = true;
                    }
                    uint32_t sb = LITERAL, sl = 0;
                    if (RL.rp[0] >= e.nice) { sb = 0; sl = RL.rp[0]; }
                    else if (RL.rp[1] >= e.nice) { sb = 1; sl = RL.rp[1]; }
                    else if (RL.rp[2] >= e.nice) { sb = 2; sl = RL.rp[2]; }
                    else if (RL.rp[3] >= e.nice) { sb = 3; sl = RL.rp[3]; }
                    else if (RL.longest >= e.nice) { sb = lane_of(RL.SD, RL.cnt - 1) + 4; sl = RL.longest; }
                    if (sl) {
                        back = sb; len = sl;
                        cached = false;
                        q_pos = q_end = 0;
                    } else if (RL.cnt == 0 && RL.rp[0] == 0 && RL.rp[1] < 2 && RL.rp[2] < 2 && RL.rp[3] < 2
                            && mask_run_after(rm_of(RL, 0), 0) < 2) {
                        // nothing but a literal can leave node 0 (no match, no rep of two bytes, not even
                        // a short rep, no "literal + rep0" compound): the window would be that literal
                        // (incompressible data lives here)
                        cached = false;
                        q_pos = q_end = 0;
                    } else {
                        cached = optimum_window<WMAX>(e, w, LP, probs, z, lt, in, cur, block_start, span_end, cached, RL, q_end);
                        q_pos = 0;
                        if (q_end == 0) {            // consistency failure reported by the parser
                            if (lane == 0) a.span_bytes[span] = 0;
                            return;
                        }
                    }
                }
                if (q_pos != q_end) {
                    back = uni(w.n_price[q_pos]);
                    len = (uni(w.n_info[q_pos]) >> 13) & 0x1FF;
                    q_pos += len;
                }
            } else {
            // ---------------- lzma_lzma_optimum_fast (optimum_fast.c:20-169) ----------------
#ifdef XZAMD_TIMING
            const uint64_t tt0 = __builtin_amdgcn_s_memtime();
#endif
            if (read_ahead == 0) {
                do_round(e, P, cur, span_end, z.rep0, z.rep1, z.rep2, z.rep3, R);
                read_ahead = 1;
#ifdef XZAMD_TIMING
                ++tm_rounds;
#endif
            }
#ifdef XZAMD_TIMING
            const uint64_t tt1 = __builtin_amdgcn_s_memtime();
            tm_round1 += tt1 - tt0;
#endif
            {
                const uint32_t rem = span_end - cur;
                const uint32_t buf_avail = rem < MATCH_LEN_MAX ? rem : MATCH_LEN_MAX;
                uint32_t len_main = R.longest;
                uint64_t m = R.mask;
                bool decided = false;
                if (buf_avail < 2) decided = true;          // literal

                uint32_t rep_len = 0, rep_index = 0;
                if (!decided) {
                    for (uint32_t i = 0; i < 4; ++i) {
                        const uint32_t rl = lane_of(R.L, 60 + i);
                        if (rl < 2) continue;
                        if (rl >= e.nice) {
                            back = i; len = rl; read_ahead += rl - 1; decided = true;
                            break;
                        }
                        if (rl > rep_len) { rep_index = i; rep_len = rl; }
                    }
                }
                if (!decided && len_main >= e.nice) {
                    const uint32_t top = 63 - (uint32_t)__builtin_clzll(m);
                    back = lane_of(R.D, top) + 4; len = len_main; read_ahead += len_main - 1;
                    decided = true;
                }
                uint32_t back_main = 0;
                if (!decided) {
                    if (len_main >= 2) {
                        uint32_t top = 63 - (uint32_t)__builtin_clzll(m);
                        back_main = lane_of(R.D, top);
                        while (__builtin_popcountll(m) > 1) {
                            const uint64_t m2 = m & ~(1ull << top);
                            const uint32_t t2 = 63 - (uint32_t)__builtin_clzll(m2);
                            const uint32_t l2 = lane_of(R.L, t2);
                            if (len_main != l2 + 1) break;
                            const uint32_t d2 = lane_of(R.D, t2);
                            if (!change_pair(d2, back_main)) break;
                            m = m2; top = t2; len_main = l2; back_main = d2;
                        }
                        if (len_main == 2 && back_main >= 0x80) len_main = 1;
                    }
                    if (rep_len >= 2) {
                        if (rep_len + 1 >= len_main
                                || (rep_len + 2 >= len_main && back_main > (1u << 9))
                                || (rep_len + 3 >= len_main && back_main > (1u << 15))) {
                            back = rep_index; len = rep_len; read_ahead += rep_len - 1;
                            decided = true;
                        }
                    }
                }
                if (!decided && (len_main < 2 || buf_avail <= 2)) decided = true;   // literal
                if (!decided) {
                    // lookahead: matches (and rep lengths) of the next byte
#ifdef XZAMD_TIMING
                    const uint64_t tl0 = __builtin_amdgcn_s_memtime();
#endif
                    do_round(e, P, cur + 1, span_end, z.rep0, z.rep1, z.rep2, z.rep3, R);
#ifdef XZAMD_TIMING
                    tm_round2 += __builtin_amdgcn_s_memtime() - tl0;
                    ++tm_rounds;
#endif
                    ++read_ahead;
                    const uint32_t nl = R.longest;
                    bool lit = false;
                    if (nl >= 2) {
                        const uint32_t top = 63 - (uint32_t)__builtin_clzll(R.mask);
                        const uint32_t new_dist = lane_of(R.D, top);
                        if ((nl >= len_main && new_dist < back_main)
                                || (nl == len_main + 1 && !change_pair(back_main, new_dist))
                                || (nl > len_main + 1)
                                || (nl + 1 >= len_main && len_main >= 3 && change_pair(new_dist, back_main)))
                            lit = true;
                    }
                    if (!lit) {
                        const uint32_t limit = len_main - 1 > 2 ? len_main - 1 : 2;
                        for (uint32_t i = 0; i < 4; ++i)
                            if (lane_of(R.L, 60 + i) >= limit) lit = true;
                    }
                    if (!lit) { back = back_main + 4; len = len_main; read_ahead += len_main - 2; }
                }
            }
            read_ahead -= len;
            // ------------------------------------------------------------------------------
            }
            if (len == 0 || len > MATCH_LEN_MAX || cur + len > span_end
                    || (back != LITERAL && back >= 4 && back - 4 >= cur - block_start)
                    || out_off + hl + rc.cpos + 64 > span_cap) {
                // internal consistency failure: report instead of corrupting memory
                if (lane == 0 && a.err) {
                    if (atomicCAS(a.err, 0u, 1u) == 0u) {
                        a.err[1] = span; a.err[2] = cur - block_start; a.err[3] = back; a.err[4] = len;
                        a.err[5] = q_pos; a.err[6] = q_end; a.err[7] = out_off + rc.cpos;
                    }
                }
                if (lane == 0) a.span_bytes[span] = 0;
                return;
            }
#ifdef XZAMD_TIMING
            const uint64_t te0 = __builtin_amdgcn_s_memtime();
#endif
            encode_symbol(rc, probs, z, in, cur, cur - block_start, back, len);
#ifdef XZAMD_TIMING
            tm_encode += __builtin_amdgcn_s_memtime() - te0;
            ++tm_syms;
            if constexpr (OPT) { if (lane == 0) { tm_lds[6] += __builtin_amdgcn_s_memtime() - te0; tm_lds[10] += 1; } }
#endif
            if (a.trace && lane == 0) {
                // debug only: symbol stream for the parity tests (compared with the oracle's parse)
                const uint32_t ti = atomicAdd(a.trace_count, 1u);
                if (ti < a.trace_cap) {
                    a.trace[4 * ti] = span;
                    a.trace[4 * ti + 1] = cur - block_start;
                    a.trace[4 * ti + 2] = back;
                    a.trace[4 * ti + 3] = len;
                }
            }
            cur += len;
        }
        rc.flush();

        uint32_t usize = cur - chunk_start;
        const uint32_t csize = rc.cpos;
        uint8_t* const hdr = outp + out_off;
        if (csize >= usize) {
            // lzma2_encoder.c:205-214: store the chunk raw.  Fast parser: including the lookahead
            // byte; optimal parser: pending symbols are dropped and re-parsed after the reset.
            if constexpr (!OPT) {
                usize += read_ahead;
                cur += read_ahead;
            }
            read_ahead = 0;
            cached = false;
            q_pos = q_end = 0;
            // the discarded range-coder bytes were stored by lane 0; make sure they have landed
            // before other lanes overwrite the same addresses
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (uint32_t i = lane; i < usize; i += 64) hdr[3 + i] = in[chunk_start + i];
            if (lane == 0) {
                hdr[0] = need_dict_reset ? 1 : 2;
                hdr[1] = (uint8_t)((usize - 1) >> 8);
                hdr[2] = (uint8_t)(usize - 1);
            }
            need_dict_reset = false;
            need_state_reset = true;
            out_off += 3 + usize;
            continue;
        }
        // lzma2_header_lzma (lzma2_encoder.c:54-106)
        if (lane == 0) {
            uint32_t c;
            if (need_props) c = need_dict_reset ? 0xE0 : 0xC0;
            else c = need_state_reset ? 0xA0 : 0x80;
            hdr[0] = (uint8_t)(c + ((usize - 1) >> 16));
            hdr[1] = (uint8_t)((usize - 1) >> 8);
            hdr[2] = (uint8_t)(usize - 1);
            hdr[3] = (uint8_t)((csize - 1) >> 8);
            hdr[4] = (uint8_t)(csize - 1);
            if (need_props) hdr[5] = (uint8_t)((a.pb * 5 + a.lp) * 9 + a.lc);
        }
        need_props = false; need_dict_reset = false; need_state_reset = false;
        out_off += hl + csize;
    }
    if (lane == 0) a.span_bytes[span] = out_off;
#ifdef XZAMD_TIMING
    if constexpr (OPT) {
        if (lane == 0 && a.err) {
            tm_lds[8] = __builtin_amdgcn_s_memtime() - tm_start;
            {   // per-span record for tools/gpu_span_times.py, in the (now dead) literal-coder slice of the span
                uint32_t hw_id;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
                uint32_t xcc_id;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
                uint32_t* zl = reinterpret_cast<uint32_t*>(z.lit);
                zl[0] = 0x54494D45u; zl[1] = (uint32_t)tm_lds[8]; zl[2] = (uint32_t)(tm_lds[8] >> 32);
                zl[3] = hw_id; zl[4] = xcc_id; zl[5] = (uint32_t)tm_lds[9]; zl[6] = (uint32_t)(tm_start >> 10);
                zl[7] = span_end - span_start;
            }
            unsigned long long* g = reinterpret_cast<unsigned long long*>(a.err + 16);
            for (int i = 0; i < 12; ++i) atomicAdd(g + i, tm_lds[i]);
            atomicMax(g + 12, tm_lds[8]);
            atomicAdd(g + 13, tm_lds[13]);
            atomicAdd(g + 14, tm_lds[14]);
            atomicAdd(g + 15, tm_lds[15]);
        }
    }
    if (lane == 0 && span == 0 && a.err) {
        const uint64_t tot = __builtin_amdgcn_s_memtime() - tm_start;
        a.err[8] = (uint32_t)(tot >> 8); a.err[9] = (uint32_t)(tm_round1 >> 8); a.err[10] = (uint32_t)(tm_round2 >> 8);
        a.err[11] = (uint32_t)(tm_encode >> 8); a.err[12] = (uint32_t)tm_rounds; a.err[13] = (uint32_t)tm_syms;
    }
#endif
}

// Persistent form: a launch has at most as many wavefronts as the GPU holds at once (the host passes the
// count); each pulls span numbers from a counter until none is left, so a launch is not a sequence of rounds
// of equally long spans and fewer resident wavefronts can be asked for (leaving room for other streams).
template <int FINDER, bool OPT, uint32_t WMAX = WMAX_STD>
__global__ __launch_bounds__(64)
__attribute__((amdgpu_waves_per_eu(OPT ? 2 : XZAMD_WAVES_FAST, OPT ? 2 : XZAMD_WAVES_FAST)))      // OPT: the single-phase optimal kernel keeps its model in LDS (14 KiB per wave); a test mode
void k_span_encode_t(xzamd_span_args a, uint32_t nspans, uint32_t* __restrict__ counter)
{
    for (;;) {
        uint32_t s = blockIdx.x;            // counter == nullptr: one wavefront per span
        if (counter != nullptr) {
            if (threadIdx.x == 0) s = atomicAdd(counter, 1u);
            s = uni(s);
        }
        if (s >= nspans) break;
        span_encode_one<FINDER, OPT, WMAX>(a, a.order ? a.order[s] : s);
        if (counter == nullptr) break;
        __builtin_amdgcn_s_waitcnt(0);      // this span's stores are out before the LDS pool is reused
        wave_sync();
    }
}

// ------------------------------------------------------------------------------------------
// Two-phase mode (oracle: parse_piece / parse_block / snapshot_walk / encode_block_syms).  The optimal parser is the
// expensive part and needs thousands of independent units to fill the GPU; the coder is cheap, but every reset of its
// model costs output bytes -- and a parser that prices with the wrong model chooses badly.  So the two are decoupled:
//   k_parse_pieces   one wavefront per PIECE of the span plan: optimum_window with an adaptive price model that codes
//                    nothing.  The first XZAMD_SEED_LEN bytes of every Block are the seed piece (phase 0, flat model).
//                    Iteration 1 (XZAMD_ITER_PARTIAL): the first eighth of every other piece, from the seed's model + a
//                    greedy warm-up walk + a pre-roll; iteration 2 (XZAMD_ITER_SNAP): every piece in full from the snapshot
//                    the carried model walk over iteration 1's records left in its slot.  Symbols are recorded per position
//                    in a coder-independent form: (length, distance) or literal bytes.
//   k_model_walk     one wavefront per ENCODE SPAN (>= 256 KiB of input): the recorded symbols through the coder's model
//                    (all of it in LDS), rep / short rep / match chosen from the coder's own rep distances.  <1>, <2>: the
//                    bounds of every probability whatever the span's start model is; k_model_chain: the true model at every
//                    span start -- ONE continuous model per Block; <0>: tokens and the LZMA2 chunk table; <3>: snapshots.
//   k_rc_chunks      one LANE per LZMA2 chunk: the range coder over the chunk's tokens.
// ------------------------------------------------------------------------------------------
template <uint32_t WMAX, bool PACKED>
__device__ __forceinline__ void parse_piece_one(const xzamd_span_args& a, const uint32_t span)
{
    // LDS per wavefront: the DP nodes and the price tables -- 10,176 bytes at WMAX = 384 -> 16 wavefronts per CU.  The
    // probabilities of the piece's price model live in global memory (a.prior / a.lit, one slot per piece, L2-resident
    // while the piece runs): a parser window reads them once (bit-price table, length / distance tables on refresh), a
    // recorded symbol gathers and scatters <= 48 of them -- the 3.7 KiB they took in LDS buy 152 more nodes per window.
    constexpr uint32_t W_NODES = 6 * (WMAX + 1) + 2;                     // reps[4], price, info (16-byte multiple)
    constexpr uint32_t W_TABS = 128 + 72 + 32;                           // dsp, xt + ap (u16), ptab (u8)
    __shared__ __attribute__((aligned(16))) uint32_t pool[W_NODES + W_TABS];
#ifdef XZAMD_TIMING
    __shared__ unsigned long long tm_lds[16];
    if (threadIdx.x < 16) tm_lds[threadIdx.x] = 0;
    const uint64_t tm_start = __builtin_amdgcn_s_memtime();
#endif
    const uint32_t lane = threadIdx.x;
    const uint32_t blk = span / a.max_spb;
    const uint32_t k = span - blk * a.max_spb;
    const uint32_t block_start = blk * a.block_size;
    const uint32_t block_end = min(a.n, block_start + a.block_size);
    uint32_t span_start, span_end;
    uint32_t piece_end;
    if (k == 0) {
        // the seed piece is the same whatever the plan says (k_span_cut: seed_chunks), so it can run before the plan exists
        span_start = block_start;
        span_end = block_end - block_start > XZAMD_SEED_LEN ? block_start + XZAMD_SEED_LEN : block_end;
        piece_end = span_end;
    } else {
        if (k >= a.span_cnt[blk]) return;              // an unused slot of the span plan
        span_start = uni(a.span_tab[2 * span]); span_end = uni(a.span_tab[2 * span + 1]);
        piece_end = span_end;
        // a partial iteration parses only the first part of the piece (oracle: part_len); symbols never cross its end
        if (a.iter & XZAMD_ITER_PARTIAL) span_end = uni(a.part_tab[span]);
    }
    // from the snapshot: the price model, coder state and rep distances come from the carried model walk over the records of
    // the partial iteration before (k_model_walk<3>: this piece's slot of a.prior / a.lit, a.snap_sr) -- no prior, no walk, no pre-roll
    const bool from_snap = k != 0 && (a.iter & XZAMD_ITER_SNAP) != 0;
    const uint8_t* __restrict__ in = a.in;

    Env e;
    e.in = in; e.rank = a.rank; e.sorted_pos = a.sorted_pos; e.prev2 = a.prev2; e.prev3 = a.prev3;
    e.nice = a.nice_len; e.depth = a.depth; e.hb = a.hash_bytes; e.cyclic = a.dict_size + 1;
    e.block_end = block_end; e.n_last = a.n - 1;
    // (the list format is a template parameter: the packed records -- every dictionary up to 8 MiB -- need neither the branch
    // nor the pointer to the length array)
    e.mlen = PACKED ? nullptr : a.mlen; e.mdist = a.mdist; e.packed = PACKED ? 1u : 0u;
    ListPre LP;
    LP.valid = false; LP.pos = 0; LP.v = LP.l16 = 0;

    Work w{};
#ifdef XZAMD_TIMING
    w.tm = tm_lds;
#endif
    {
        uint32_t* nb = pool;
        w.n_reps4 = reinterpret_cast<uint4*>(nb);
        w.n_price = nb + 4 * (WMAX + 1);
        w.n_info = nb + 5 * (WMAX + 1);
        uint32_t* tb = nb + W_NODES;
        w.dsp = reinterpret_cast<uint16_t*>(tb);
        w.xt = reinterpret_cast<uint16_t*>(tb + 128);
        w.ap = w.xt + 128;
        w.ptab = reinterpret_cast<uint8_t*>(tb + 200);
        w.err = a.err;
        for (uint32_t t = lane; t < 128; t += 64) {     // bit price table (price_tablegen.c:31-58)
            uint32_t wv = t * 16 + 8, bit_count = 0;
            for (int jj = 0; jj < 4; ++jj) {
                wv *= wv;
                bit_count <<= 1;
                while (wv >= (1u << 16)) { wv >>= 1; ++bit_count; }
            }
            w.ptab[t] = (uint8_t)((11 << 4) - 15 - bit_count);
        }
        wave_sync();
        w.ptv = reinterpret_cast<const uint32_t*>(w.ptab)[lane & 31];
    }

    Lz z;
    // pb = 3, 4: the piece's price model -- which is the parser's alone, the coder keeps its own (k_model_syms, the real
    // pb) -- takes a pb = 2 view of the positions; the bit-price and length tables of the window cover four position
    // states (oracle: parse_block).  The recorded symbols are valid under any pb.
    z.lc = a.lc; z.lp = a.lp; z.pb = a.pb < 2u ? a.pb : 2u;
    z.cnt_len = z.cnt_match = z.cnt_align = 0;
    const uint32_t lit_size = 0x300u << (a.lc + a.lp);
    z.lit = reinterpret_cast<plit_t*>(a.lit) + (uint64_t)span * lit_size;
    z.gp = a.prior + (uint64_t)span * XZAMD_PRIOR_WORDS;
    z.state = 0; z.rep0 = z.rep1 = z.rep2 = z.rep3 = 0;
    RC rc;                                              // never codes: encode_symbol_t<false, .> only adapts the model
    rc.cpos = 0; rc.out = nullptr; rc.reset();
    rc.ptab = nullptr; rc.est = 0; rc.tok = nullptr; rc.log = nullptr; rc.log_cap = 0;
    const GProbs probs{z.gp};
    uint16_t* const no_lds = nullptr;                   // encode_symbol_t<.., PG = true> never touches its LDS argument

    // price model: flat for the seed piece (slot 0 of the Block), else the model the seed left in ITS slot
    {
        uint4* g4 = reinterpret_cast<uint4*>(z.gp);
        uint4* l4 = reinterpret_cast<uint4*>(z.lit);
        if (k == 0) {
            const uint4 v = make_uint4(1024u, 1024u, 1024u, 1024u);
            for (uint32_t i = lane; i < XZAMD_PRIOR_WORDS / 4; i += 64) g4[i] = v;
            const uint4 vl = PLIT_FLAT4;
            for (uint32_t i = lane; i < lit_size / PLIT_PER_U4; i += 64) l4[i] = vl;
        } else if (from_snap) {
            const uint32_t* sr = a.snap_sr + (uint64_t)span * 8u;
            z.state = uni(sr[0]); z.rep0 = uni(sr[1]); z.rep1 = uni(sr[2]); z.rep2 = uni(sr[3]); z.rep3 = uni(sr[4]);
        } else {
            const uint4* p4 = reinterpret_cast<const uint4*>(a.prior + (uint64_t)blk * a.max_spb * XZAMD_PRIOR_WORDS);
            for (uint32_t i = lane; i < XZAMD_PRIOR_WORDS / 4; i += 64) g4[i] = p4[i];
            const uint4* s4 = reinterpret_cast<const uint4*>(reinterpret_cast<const plit_t*>(a.lit) + (uint64_t)blk * a.max_spb * lit_size);
            for (uint32_t i = lane; i < lit_size / PLIT_PER_U4; i += 64) l4[i] = s4[i];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_sync();
    }

    uint32_t cur = span_start;
    // Pre-roll (oracle: parse_piece, ORC_PREROLL): every piece but the seed first parses the XZAMD_PREROLL bytes in front
    // of it once more, from the prior, and throws the symbols away: the piece proper then starts with prices that have seen
    // the local data and with rep distances / a coder state like the ones the previous piece ends with.  `lim` is what the
    // parser may not cross: the piece start while pre-rolling, then the piece end.
    uint32_t lim = span_end;
    bool rec = true;
    if (k != 0 && !from_snap && span_start - block_start > XZAMD_PREROLL) {
        cur = span_start - XZAMD_PREROLL;
        lim = span_start;
        rec = false;
    }
    // Warm-up (oracle: parse_piece, ORC_WARM): the prior is what the Block's first 64 KiB teach; data that has drifted since
    // (float arrays, PCM, ...) needs more than the pre-roll to re-train the model, so the XZAMD_WARM bytes in front of the
    // pre-roll -- never the seed piece's -- are walked greedily first: at every symbol boundary the longest entry of the
    // position's list record when it is cheap by a fixed rule, else a literal; each symbol adapts the model, nothing is
    // recorded, no prices, no DP.  64 positions per trip: lane = position (its longest entry, its byte, the byte before it,
    // the byte at rep0), the walk itself runs on readlanes.
    if (k != 0 && !from_snap && span_start - block_start > XZAMD_PREROLL + XZAMD_SEED_LEN) {
        const uint32_t w1 = span_start - XZAMD_PREROLL;
        uint32_t x0 = w1 - block_start - XZAMD_SEED_LEN > XZAMD_WARM ? w1 - XZAMD_WARM : block_start + XZAMD_SEED_LEN;
        // the distances of the last four candidates the walk rejected as too expensive: one that comes up again is a distance a
        // continuous parse would be carrying in its rep stack (periodic numeric data) and goes into the oldest rep slot -- nothing
        // is coded (oracle: parse_piece)
        uint32_t rj0 = 0, rj1 = 0, rj2 = 0, rj3 = 0, nrej = 0;
        // what the walk may not teach: is_rep / is_rep0 / is_rep1 / is_rep2 (48 probabilities) keep the seed piece's values -- its
        // rep decisions are the least like the optimal parser's (oracle: parse_piece)
        static_assert(P_IS_REP0_LONG - P_IS_REP == 48, "the rep-choice probabilities");
        const uint32_t keep_rep = lane < 48 ? lit_load(z.gp + P_IS_REP + lane) : 0u;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // in hand before the walk's first scatter
        while (x0 < w1) {
            const uint32_t px = min(x0 + lane, w1 - 1);
            const uint64_t rb = (uint64_t)px * LIST_W;
            const uint32_t tr = e.mdist[rb + LIST_K];
            const uint32_t cnt = tr & 0xFFu;
            const uint32_t ev = cnt ? e.mdist[rb + cnt - 1] : 0u;
            uint32_t L = 0, D = 0;
            if (cnt) {
                if (e.packed) { L = ev >> 23; D = ev & 0x7FFFFFu; }
                else { D = ev; L = e.mlen[rb + cnt - 1]; }
            }
            const uint32_t tc = in[px], tp = in[px - 1];
            uint32_t tm = in[px - z.rep0 - 1];
            uint32_t i = 0;
            const uint32_t rowlen = min(64u, w1 - x0);
            while (i < rowlen) {
                const uint32_t x = x0 + i;
                uint32_t len = min(lane_of(L, i), w1 - x);
                const uint32_t dist = lane_of(D, i);
                const uint32_t bl = dist ? 32u - (uint32_t)__builtin_clz(dist) : 0u;         // bit length of the distance
                const bool in_reps = dist == z.rep0 || dist == z.rep1 || dist == z.rep2 || dist == z.rep3;
                const bool take = len >= 2 && (14 + bl < 6 * len || in_reps);
                if (!take && len >= 3) {
                    const bool seen = (nrej > 0 && rj0 == dist) || (nrej > 1 && rj1 == dist) || (nrej > 2 && rj2 == dist) || (nrej > 3 && rj3 == dist);
                    if (seen) z.rep3 = dist;
                    else if (nrej < 4) { rj0 = nrej == 0 ? dist : rj0; rj1 = nrej == 1 ? dist : rj1; rj2 = nrej == 2 ? dist : rj2; rj3 = nrej == 3 ? dist : rj3; ++nrej; }
                    else { rj0 = rj1; rj1 = rj2; rj2 = rj3; rj3 = dist; }
                }
                if (take) {
                    // rep or match?  When the distance is one of the rep distances the current prices decide, as they do in
                    // the optimal parser (oracle: parse_piece -- data made of fixed-size records has two self-reinforcing
                    // ways to code the same copy; the walk must train the model into the one the parser lives in)
                    uint32_t back = dist + 4;
                    const uint32_t ri = dist == z.rep0 ? 0u : dist == z.rep1 ? 1u : dist == z.rep2 ? 2u : dist == z.rep3 ? 3u : 4u;
                    if (ri < 4) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the model updates so far
                        const uint32_t ps = (x - block_start) & ((1u << z.pb) - 1), st = z.state;
                        const uint8_t* pt = w.ptab;
                        uint32_t prep = pr_bit(probs, pt, P_IS_REP + st, 1);
                        if (ri == 0) prep += pr_bit(probs, pt, P_IS_REP0 + st, 0) + pr_bit(probs, pt, P_IS_REP0_LONG + st * 16 + ps, 1);
                        else {
                            prep += pr_bit(probs, pt, P_IS_REP0 + st, 1);
                            if (ri == 1) prep += pr_bit(probs, pt, P_IS_REP1 + st, 0);
                            else prep += pr_bit(probs, pt, P_IS_REP1 + st, 1) + pr_bit(probs, pt, P_IS_REP2 + st, ri - 2);
                        }
                        prep += pr_len_one(probs, pt, P_REP_LEN, ps, len);
                        uint32_t pm = pr_bit(probs, pt, P_IS_REP + st, 0) + pr_len_one(probs, pt, P_MATCH_LEN, ps, len);
                        {
                            const uint32_t slot = dist_slot_of(dist);
                            pm += pr_tree(probs, pt, P_DIST_SLOT + (len < 6 ? len - 2 : 3u) * 64, 6, slot);
                            if (slot >= 4) {
                                const uint32_t fb = (slot >> 1) - 1;
                                const uint32_t base = (2 | (slot & 1)) << fb;
                                const uint32_t red = dist - base;
                                if (slot < 14) pm += pr_tree_rev(probs, pt, P_DIST_SPECIAL + base - slot - 1, fb, red);
                                else pm += ((fb - 4) << 4) + pr_tree_rev(probs, pt, P_DIST_ALIGN, 4, red & 15);
                            }
                        }
                        if (uni(prep) <= uni(pm)) back = ri;
                    }
                    const bool rep = back == 0;                              // rep0 stays what it was
                    encode_symbol_t<false, true, false, true>(rc, no_lds, z, x - block_start, back, len, 0u);
                    i += len;
                    if (!rep && i < rowlen) tm = in[max(px, x0 + i) - z.rep0 - 1];   // rep0 changed: the match bytes of the rest of the row
                                                                                     // (lanes behind the walk would reach in front of the Block)
                } else {
                    const uint32_t l3 = lane_of(tc, i) | (lane_of(tp, i) << 8) | (lane_of(tm, i) << 16);
                    encode_symbol_t<false, true, false, true>(rc, no_lds, z, x - block_start, LITERAL, 1, l3);
                    i += 1;
                }
            }
            x0 += i;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane < 48) lit_store(z.gp + P_IS_REP + lane, keep_rep);
    }
    bool cached = false;
    RoundL RL;
    RL.L = 0; RL.SL = 0; RL.SD = 0; RL.cnt = 0; RL.longest = 0; RL.l2a = RL.l2b = 0;
    RL.rp[0] = RL.rp[1] = RL.rp[2] = RL.rp[3] = 0;
    RL.mlo = RL.mhi = 0;
    RL.rl[0] = RL.rl[1] = RL.rl[2] = RL.rl[3] = 0;
    RL.pre = 0;
    LenTab lt;
    uint32_t q_pos = 0, q_end = 0;
    bool tables_valid = false;

    // What the coder will read out of this piece's records WITHOUT knowing its state and rep distances at the piece start
    // (oracle: lookback): twelve candidate states side by side (lane = candidate; they agree after a few symbols), rep
    // distances the piece does not name stay XZAMD_REP_UNKNOWN.  The walk of the encode span that starts behind this piece
    // begins with them (k_model_walk).
    uint32_t tl_st = lane < 12 ? lane : 0u;
    uint32_t tl_r0 = XZAMD_REP_UNKNOWN, tl_r1 = XZAMD_REP_UNKNOWN, tl_r2 = XZAMD_REP_UNKNOWN, tl_r3 = XZAMD_REP_UNKNOWN;
    if (rec) rc.ptab = w.ptab;                           // the price of the piece: its recorded symbols only
    if (span_start == block_start) {
        // encode_init (lzma_encoder.c:267-293): the first byte of a Block is a literal in the initial contexts
        const uint32_t l3 = literal_bytes(in, block_start, 0, z);
        if (lane == 0) { a.sym_len[block_start] = 0; a.sym_dist[block_start] = l3; }
        encode_symbol_t<false, true, false, true>(rc, no_lds, z, 0, LITERAL, 1, l3);
        tl_st = tl_st <= 3 ? 0u : tl_st <= 9 ? tl_st - 3 : tl_st - 6;
        cur = block_start + 1;
    }
    for (;;) {
        if (cur >= lim) {
            if (rec) break;
            rec = true;                                  // the pre-roll is over: the piece proper
            rc.ptab = w.ptab; rc.est = 0;
            lim = span_end;
            cached = false;
            q_pos = q_end = 0;
            continue;
        }
        uint32_t back = LITERAL, len = 1;
        if (q_pos == q_end) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the model updates of the symbols so far: the prices read them
            TM_BEGIN(t_refresh);
            if (!tables_valid || z.cnt_len >= 64) { refresh_len_tables(probs, w.ptab, lt, 1u << z.pb, reinterpret_cast<uint32_t*>(w.n_reps4)); z.cnt_len = 0; }
            if (!tables_valid || z.cnt_match >= 128) { refresh_dist_tables(probs, w); z.cnt_match = 0; }
            if (!tables_valid || z.cnt_align >= 16) { refresh_align_table(probs, w); z.cnt_align = 0; }
            tables_valid = true;
            TM_END(w, 7, t_refresh);
            if (!cached) {
                round_lists(e, LP, cur, lim, z.rep0, z.rep1, z.rep2, z.rep3, RL);
                cached = true;
            }
            uint32_t sb = LITERAL, sl = 0;
            if (RL.rp[0] >= e.nice) { sb = 0; sl = RL.rp[0]; }
            else if (RL.rp[1] >= e.nice) { sb = 1; sl = RL.rp[1]; }
            else if (RL.rp[2] >= e.nice) { sb = 2; sl = RL.rp[2]; }
            else if (RL.rp[3] >= e.nice) { sb = 3; sl = RL.rp[3]; }
            else if (RL.longest >= e.nice) { sb = lane_of(RL.SD, RL.cnt - 1) + 4; sl = RL.longest; }
            if (sl) {
                back = sb; len = sl;
                cached = false;
                q_pos = q_end = 0;
            } else if (RL.cnt == 0 && RL.rp[0] == 0 && RL.rp[1] < 2 && RL.rp[2] < 2 && RL.rp[3] < 2
                    && mask_run_after(rm_of(RL, 0), 0) < 2) {
                cached = false;                         // nothing but a literal can leave this node
                q_pos = q_end = 0;
            } else {
                cached = optimum_window<WMAX>(e, w, LP, probs, z, lt, in, cur, block_start, lim, cached, RL, q_end);
                q_pos = 0;
                if (q_end == 0) {                       // consistency failure reported by the parser
                    if (lane == 0 && a.err) atomicCAS(a.err, 0u, 2u);
                    return;
                }
            }
        }
        if (q_pos != q_end) {
            back = uni(w.n_price[q_pos]);
            len = (uni(w.n_info[q_pos]) >> 13) & 0x1FF;
            q_pos += len;
        }
        if (len == 0 || len > MATCH_LEN_MAX || cur + len > lim
                || (back != LITERAL && back >= 4 && back - 4 >= cur - block_start)) {
            if (lane == 0 && a.err) {
                if (atomicCAS(a.err, 0u, 1u) == 0u) {
                    a.err[1] = span; a.err[2] = cur - block_start; a.err[3] = back; a.err[4] = len;
                    a.err[5] = q_pos; a.err[6] = q_end; a.err[7] = 0;
                }
            }
            return;
        }
        uint32_t l3 = 0;
        if (back == LITERAL) {
            l3 = literal_bytes(in, cur, cur - block_start, z) | (z.state >= 7 ? 1u << 24 : 0u);
            // the match byte of the record is the coder's only when the coder has coded the symbol in front of it, i.e. not
            // for the first symbol of a piece (the pre-roll in front of it is the parser's alone): there the coder fetches
            // the byte itself
            const uint32_t l3r = cur == span_start ? (l3 & 0xFFFFu) : l3;
            if (lane == 0 && rec) { a.sym_len[cur] = 0; a.sym_dist[cur] = l3r; }
        } else if (lane == 0 && rec) {
            // bit 15: the parser chose a MATCH -- the coder then codes a match even when the distance is one of its rep
            // distances (oracle: parse_piece)
            a.sym_len[cur] = (uint16_t)(len | (back >= 4 ? 0x8000u : 0u));
            a.sym_dist[cur] = back >= 4 ? back - 4 : back == 0 ? z.rep0 : back == 1 ? z.rep1 : back == 2 ? z.rep2 : z.rep3;
        }
        if (rec) {
            // the coder's reading of the record (k_model_walk) with unknown rep distances: 0 literal, 1 match, 2 rep, 3 short rep
            uint32_t type = 0;
            if (back != LITERAL) {
                const uint32_t dd = back >= 4 ? back - 4 : back == 0 ? z.rep0 : back == 1 ? z.rep1 : back == 2 ? z.rep2 : z.rep3;
                if (back >= 4) type = 1;
                else if (len == 1) type = dd == tl_r0 ? 3u : 0u;
                else {
                    const uint32_t ri = dd == tl_r0 ? 0u : dd == tl_r1 ? 1u : dd == tl_r2 ? 2u : dd == tl_r3 ? 3u : 4u;
                    type = ri < 4 ? 2u : 1u;
                    if (ri == 1) { tl_r1 = tl_r0; tl_r0 = dd; }
                    else if (ri == 2) { tl_r2 = tl_r1; tl_r1 = tl_r0; tl_r0 = dd; }
                    else if (ri == 3) { tl_r3 = tl_r2; tl_r2 = tl_r1; tl_r1 = tl_r0; tl_r0 = dd; }
                }
                if (type == 1) { tl_r3 = tl_r2; tl_r2 = tl_r1; tl_r1 = tl_r0; tl_r0 = dd; }
            }
            tl_st = type == 0 ? (tl_st <= 3 ? 0u : tl_st <= 9 ? tl_st - 3 : tl_st - 6)
                    : tl_st < 7 ? (type == 1 ? 7u : type == 2 ? 8u : 9u) : (type == 1 ? 10u : 11u);
        }
        {
            TM_BEGIN(t_sym);
            encode_symbol_t<false, true, false, true>(rc, no_lds, z, cur - block_start, back, len, l3);
            TM_END(w, 6, t_sym);
            TM_COUNT(w, 10);
        }
        if (a.trace && lane == 0 && rec) {
            const uint32_t ti = atomicAdd(a.trace_count, 1u);
            if (ti < a.trace_cap) {
                a.trace[4 * ti] = span;
                a.trace[4 * ti + 1] = cur - block_start;
                a.trace[4 * ti + 2] = back;
                a.trace[4 * ti + 3] = len;
            }
        }
        cur += len;
    }
    // (the seed piece leaves the prior of the Block where it is: its own slot of a.prior / a.lit)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
        // what the walk of the encode span behind this piece starts from, and whether the coder stores the piece raw: its
        // parser's own price says it does not shrink (oracle: parse_block)
        const uint32_t s0 = lane_of(tl_st, 0);
        const bool agree = __builtin_amdgcn_ballot_w64(lane < 12 && tl_st != s0) == 0;
        if (lane < 2 && a.pinfo) {
            // lane 0: the half the walk over iteration 1's records reads, lane 1: the half the coder's walk reads (the seed
            // piece is parsed once and serves both)
            const bool mine = k == 0 || (lane == 0) == ((a.iter & XZAMD_ITER_PARTIAL) != 0);
            if (mine) {
                uint32_t* pi = a.pinfo + (uint64_t)span * XZAMD_PINFO_WORDS + lane * (XZAMD_PINFO_WORDS / 2);
                pi[0] = s0 | (agree ? XZAMD_PI_STATE_OK : 0u);
                pi[1] = tl_r0; pi[2] = tl_r1; pi[3] = tl_r2; pi[4] = tl_r3;
                const uint32_t plen = piece_end - span_start;
                pi[5] = (span_end == piece_end && plen >= XZAMD_RAW_MIN_LEN && rc.est / 128u >= plen) ? 1u : 0u;
            }
        }
    }
#ifdef XZAMD_TIMING
    if (lane == 0 && a.err) {
        tm_lds[8] = __builtin_amdgcn_s_memtime() - tm_start;
        unsigned long long* g = reinterpret_cast<unsigned long long*>(a.err + 16);
        for (int i = 0; i < 12; ++i) atomicAdd(g + i, tm_lds[i]);
        atomicMax(g + 12, tm_lds[8]);
        atomicAdd(g + 13, tm_lds[13]);
        atomicAdd(g + 14, tm_lds[14]);
        atomicAdd(g + 15, tm_lds[15]);
    }
#endif
}

template <uint32_t WMAX = WMAX_STD, bool PACKED = true>
__global__ __launch_bounds__(64)
__attribute__((amdgpu_waves_per_eu(XZAMD_WAVES_OPT, XZAMD_WAVES_OPT)))
void k_parse_pieces(xzamd_span_args a, uint32_t nslots, int phase, uint32_t* __restrict__ counter)
{
    for (;;) {
        uint32_t s = blockIdx.x;
        if (counter != nullptr) {
            if (threadIdx.x == 0) s = atomicAdd(counter, 1u);
            s = uni(s);
        }
        if (s >= nslots) break;
        if (phase == 0) {
            parse_piece_one<WMAX, PACKED>(a, s * a.max_spb);                 // s = Block: its seed piece
        } else {
            const uint32_t slot = a.order ? a.order[s] : s;
            if (slot % a.max_spb != 0) parse_piece_one<WMAX, PACKED>(a, slot);
        }
        if (counter == nullptr) break;
        __builtin_amdgcn_s_waitcnt(0);
        wave_sync();
    }
}

// ---- phase 2: the recorded symbols of one encode span through the coder's model ----
struct SymRow {
    uint32_t base;          // position of lane 0 of `l` / `d`
    uint32_t l, d;          // records of positions base + lane
    uint32_t nl, nd;        // records of positions base + 64 + lane (in flight)
};

__device__ __forceinline__ void symrow_load(const xzamd_span_args& a, uint32_t pos, uint32_t& l, uint32_t& d)
{
    uint32_t x = pos + threadIdx.x;
    x = x < a.n ? x : a.n - 1;
    l = a.sym_len[x];
    d = a.sym_dist[x];
}

// The walk of one encode span over its recorded symbols (oracle: encode_block_syms / snapshot_walk), in four forms:
//   MODE 0  k_model_syms      the coder's model pass: from the span's TRUE start model (k_model_chain) every binary decision
//                             leaves a token, chunks are cut by the summed prices of their decisions (the range coder runs
//                             later, one lane per chunk: k_rc_chunks), stored pieces are copied
//   MODE 1  k_model_bounds    the same walk with two values per probability, lo (from 31) and hi (from 2017): whatever the
//                             start model is, the true value stays between them (the update is monotone); until they meet the
//                             slot's bits are logged for k_model_chain
//   MODE 2  bounds, and MODE 3 snapshots, over the records of parse iteration 1 (only the first part of a piece, a.part_tab, of a
//                             piece have them): the model every piece starts iteration 2 from goes into ITS slot of
//                             a.prior / a.lit (u32 each), coder state and rep distances into a.snap_sr
// LITG: the literal coders (6,144 of the 7,990 probabilities at lc = 3) live in global memory -- the span's slice of cb_bnd,
// u32 each, L2-resident while the span is walked -- and LDS holds the 1,846 others: 7.4 KiB (bounds) / 3.8 KiB per wavefront instead
// of 32 / 16 KiB, so that a CU holds several times the wavefronts (these walks are bound by the latency of their dependent
// instructions; the single-phase span kernels made the same trade).
template <int MODE, bool LITG>
__global__ __launch_bounds__(64) void k_model_walk(xzamd_span_args a, uint32_t nslots)
{
    constexpr bool BND = MODE == 1 || MODE == 2, SUB = MODE >= 2, TOKM = MODE == 0;
    extern __shared__ __attribute__((aligned(16))) uint32_t enc_pool[];
    uint16_t* const probs = reinterpret_cast<uint16_t*>(enc_pool);
    const uint32_t lane = threadIdx.x;
    const uint32_t slot = blockIdx.x;
    if (slot >= nslots) return;
    const uint32_t blk = slot / a.max_esb;
    const uint32_t k = slot - blk * a.max_esb;
    if (k >= a.enc_cnt[blk]) return;
    const uint32_t block_start = blk * a.block_size;
    const uint32_t span_start = uni(a.enc_tab[2 * slot]), span_end = uni(a.enc_tab[2 * slot + 1]);
    const uint8_t* __restrict__ in = a.in;
    const uint32_t nprob = P_LITERAL + (0x300u << (a.lc + a.lp));
    const uint32_t nlds = LITG ? (uint32_t)P_LITERAL : nprob;                  // probabilities held in LDS
    const uint32_t model_words = BND ? nlds : (nlds + 1) / 2;
    uint8_t* const ptab = reinterpret_cast<uint8_t*>(enc_pool + model_words);
    // LITG: the literal coders' working array (bounds, or plain probabilities: k_model_chain has read the bounds by then)
    uint32_t* const litw = a.cb_bnd + (uint64_t)slot * a.model_slots_pad + P_LITERAL;
    const uint32_t nlit = nprob - P_LITERAL;
    if (!BND) {
        for (uint32_t t = lane; t < 128; t += 64) {         // bit price table (price_tablegen.c:31-58)
            uint32_t wv = t * 16 + 8, bit_count = 0;
            for (int jj = 0; jj < 4; ++jj) {
                wv *= wv;
                bit_count <<= 1;
                while (wv >= (1u << 16)) { wv >>= 1; ++bit_count; }
            }
            ptab[t] = (uint8_t)((11 << 4) - 15 - bit_count);
        }
    }
    // the piece that starts at the span start (encode spans are closed at piece ends)
    const uint32_t pcnt = uni(a.span_cnt[blk]);
    const uint32_t* __restrict__ ptb = a.span_tab + 2ull * blk * a.max_spb;
    uint32_t pj = 0;
    for (uint32_t j0 = 0; j0 < pcnt; j0 += 64) {
        const uint32_t jj = j0 + lane;
        const uint64_t m = __builtin_amdgcn_ballot_w64(jj < pcnt && ptb[2 * jj] == span_start);
        if (m) { pj = j0 + (uint32_t)__builtin_ctzll(m); break; }
    }
    // (the half of a piece's info written by the parse iteration whose records this walk reads)
    const uint32_t* __restrict__ pinfo = a.pinfo + (uint64_t)blk * a.max_spb * XZAMD_PINFO_WORDS + (SUB ? 0u : XZAMD_PINFO_WORDS / 2);

    Lz z;
    z.lc = a.lc; z.lp = a.lp; z.pb = SUB ? min(a.pb, 2u) : a.pb;     // (the snapshots are the PARSER's price model: its pb view)
    z.cnt_len = z.cnt_match = z.cnt_align = 0;
    z.lit = LITG ? reinterpret_cast<plit_t*>(litw) : nullptr;
    z.gp = nullptr;
    z.state = 0; z.rep0 = z.rep1 = z.rep2 = z.rep3 = 0;
    RC rc;
    rc.cpos = 0; rc.out = nullptr; rc.reset();
    rc.log = BND ? a.cb_log + (uint64_t)slot * a.model_slots_pad * XZAMD_LOG_WORDS : nullptr;
    rc.log_cap = a.log_cap ? a.log_cap : XZAMD_LOG_CAP;

    // how the span starts
    uint32_t hflags = 0;
    bool known = k == 0;                    // the model at the span start does not depend on the spans in front
    if (k != 0) {
        const uint32_t* pi = pinfo + (uint64_t)(pj - 1) * XZAMD_PINFO_WORDS;
        if (!SUB && uni(pi[5])) known = true;          // behind a stored piece: state reset
        else {
            const uint32_t w0 = uni(pi[0]);
            if (!(w0 & XZAMD_PI_STATE_OK)) hflags |= XZAMD_CB_BAD_START;
            z.state = w0 & 15u;
            z.rep0 = uni(pi[1]); z.rep1 = uni(pi[2]); z.rep2 = uni(pi[3]); z.rep3 = uni(pi[4]);
        }
    }
    if (known) hflags |= XZAMD_CB_KNOWN_START;
    uint32_t carry = 1;                     // MODE 0 / 3: 0 = round 5's reset (+ properties), 1 = carried, 2 = flat start, no properties
    if (!BND && k != 0) carry = uni(a.cb_carry[slot]);
    if (BND) {
        const uint32_t v0 = known ? (1024u | 1024u << 11) : (31u | 2017u << 11);
        for (uint32_t i = lane; i < nlds; i += 64) enc_pool[i] = v0;
        if (LITG) for (uint32_t i = lane; i < nlit; i += 64) lit_store(litw + i, v0);
    } else if (k != 0 && carry == 1) {
        const uint16_t* s16 = a.cb_start + (uint64_t)slot * a.model_slots_pad;
        const uint32_t* s32 = reinterpret_cast<const uint32_t*>(s16);
        for (uint32_t i = lane; i < model_words; i += 64) enc_pool[i] = s32[i];
        if (LITG) for (uint32_t i = lane; i < nlit; i += 64) lit_store(litw + i, (uint32_t)s16[P_LITERAL + i]);
    } else {
        for (uint32_t i = lane; i < model_words; i += 64) enc_pool[i] = 0x04000400u;
        if (LITG) for (uint32_t i = lane; i < nlit; i += 64) lit_store(litw + i, 1024u);
        z.state = 0; z.rep0 = z.rep1 = z.rep2 = z.rep3 = 0;
    }
    if (LITG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_sync();

    uint16_t* const tok0 = TOKM ? a.tok + XZAMD_TOK_BASE(span_start, slot) : nullptr;
    // Token budget of the span.  The buffer holds XZAMD_TOK_PER_BYTE tokens per input byte (+ 4096); a.tok_limit (tests only,
    // <= XZAMD_TOK_PER_BYTE) lowers the budget without changing the layout.  Data made of far three-byte matches needs more
    // (32 ... 41 tokens per 3 bytes): when the budget runs out the chunk is closed where it stands and the REST of the span
    // is stored as raw LZMA2 chunks (lzma2_encoder.c:110-131) -- valid output instead of a failed Stream -- and no later span
    // of the Block is carried (oracle: encode_block_syms).  The walks over iteration 1's records write no tokens: no budget.
    const uint64_t tok_cap = (uint64_t)(span_end - span_start) * (a.tok_limit ? a.tok_limit : XZAMD_TOK_PER_BYTE) + 4096u - 64u;
    rc.tok = tok0; rc.est = 0; rc.ptab = ptab;
    const uint32_t cbase = XZAMD_CHUNK_BASE(span_start, slot);
    const uint32_t ccap = XZAMD_CHUNK_CAP(span_end - span_start);
    uint32_t nchunks = 0;

    bool need_props = k == 0 || carry == 0, need_dict_reset = (span_start == block_start), need_state_reset = k == 0 || carry != 1;
    bool failed = false, tok_full = false;
    uint32_t f_pos = 0, f_back = 0, f_len = 0, f_d = 0;
    uint32_t cur = span_start;
    uint32_t p_end = uni(ptb[2 * pj + 1]);                           // end of the piece that holds cur
    const uint32_t* __restrict__ ptp = a.part_tab + (uint64_t)blk * a.max_spb;           // SUB: where the parsed part of a piece ends
    uint32_t seg_end = SUB ? uni(ptp[pj]) : p_end;   // end of what is walked of it (the seed piece has all its records)
    uint32_t raw_until = (!SUB && uni(pinfo[(uint64_t)pj * XZAMD_PINFO_WORDS + 5])) ? p_end : span_start;
    bool fresh_piece = true;                                          // cur is a piece start that has not been looked at
    SymRow R;
    R.base = span_start;
    symrow_load(a, span_start, R.l, R.d);
    symrow_load(a, span_start + 64, R.nl, R.nd);

    while (cur < span_end) {
        if (cur < raw_until) {
            // a stored piece, or the rest of a span that ran out of tokens: raw chunks of 64 KiB (lzma2_encoder.c:110-131)
            if (TOKM) {
                const uint32_t usize = min(raw_until - cur, 65536u);
                const uint32_t cidx = cbase + nchunks;
                if (nchunks + 1 >= ccap) { f_pos = cur - block_start; f_back = 0xFFFFFFF0u; failed = true; break; }
                uint8_t* const hdr = a.scratch + XZAMD_CHUNK_OUT(cur, cidx);
                for (uint32_t i = lane; i < usize; i += 64) hdr[3 + i] = in[cur + i];
                if (lane == 0) {
                    hdr[0] = need_dict_reset ? 1 : 2;
                    hdr[1] = (uint8_t)((usize - 1) >> 8);
                    hdr[2] = (uint8_t)(usize - 1);
                    xzamd_chunk c;
                    c.in_start = cur; c.usize = usize; c.tok_lo = 0; c.tok_hi = 0;
                    c.ntok = 0; c.csize = 3u + usize; c.flags = XZAMD_CH_RAW; c.pad_ = 0;
                    a.chunks[cidx] = c;
                }
                need_dict_reset = false;
                ++nchunks;
                cur += usize;
            } else
                cur = raw_until;
            need_state_reset = true;
            if (cur == p_end && cur < span_end) {                    // (a stored piece ends where the next piece starts)
                ++pj;
                p_end = uni(ptb[2 * pj + 1]);
                seg_end = p_end;
                if (uni(pinfo[(uint64_t)pj * XZAMD_PINFO_WORDS + 5])) raw_until = p_end;
            }
            continue;
        }
        if (need_state_reset && !(k != 0 && carry == 1 && cur == span_start)) {
            // lzma_lzma_encoder_reset (lzma_encoder.c:529-598); the flag stays set until a chunk header has announced it
            if (BND) { for (uint32_t i = lane; i < nlds; i += 64) enc_pool[i] = 1024u | 1024u << 11; }
            else { for (uint32_t i = lane; i < model_words; i += 64) enc_pool[i] = 0x04000400u; }
            if (LITG) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                for (uint32_t i = lane; i < nlit; i += 64) lit_store(litw + i, BND ? (1024u | 1024u << 11) : 1024u);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            wave_sync();
            z.state = 0; z.rep0 = z.rep1 = z.rep2 = z.rep3 = 0;
        }
        const uint32_t chunk_start = cur;
        uint16_t* const chunk_tok = rc.tok;
        rc.est = 0;
        bool piece_break = false;
        for (;;) {
            // the chunk rule of the two-phase coder (oracle: encode_block_syms): input limit of lzma2_encoder.c:167-181, and
            // the summed prices in place of the coded size
            if (TOKM && (cur - chunk_start >= (1u << 21) - MATCH_LEN_MAX || rc.est >= XZAMD_CHUNK_EST))
                break;
            if (cur >= span_end)
                break;
            if (cur >= seg_end) {
                // the next piece (SUB: behind the part of this one that iteration 1 parsed)
                cur = p_end;
                if (cur >= span_end) break;
                ++pj;
                p_end = uni(ptb[2 * pj + 1]);
                seg_end = SUB ? uni(ptp[pj]) : p_end;
                fresh_piece = true;
                if (!SUB && uni(pinfo[(uint64_t)pj * XZAMD_PINFO_WORDS + 5])) { raw_until = p_end; piece_break = true; break; }
            }
            if (MODE == 3 && fresh_piece && cur != block_start) {
                // snapshot: the price model piece pj starts iteration 2 from (its slot of a.prior / a.lit), state, rep distances
                const uint64_t ps_ = (uint64_t)blk * a.max_spb + pj;
                uint32_t* gp = a.prior + ps_ * XZAMD_PRIOR_WORDS;
                plit_t* gl = reinterpret_cast<plit_t*>(a.lit) + ps_ * (0x300ull << (a.lc + a.lp));
                for (uint32_t i = lane; i < P_LITERAL; i += 64) gp[i] = probs[i];
                if (LITG) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the literal updates so far
                    for (uint32_t i = lane; i < nlit; i += 64) gl[i] = (plit_t)lit_load(litw + i);
                } else
                    for (uint32_t i = lane; i < nlit; i += 64) gl[i] = (plit_t)probs[P_LITERAL + i];
                if (lane == 0) {
                    uint32_t* sr = a.snap_sr + ps_ * 8u;
                    sr[0] = z.state;                  // a parser's rep distances must be real ones: unknown -> 0, as after a reset
                    sr[1] = z.rep0 == XZAMD_REP_UNKNOWN ? 0u : z.rep0; sr[2] = z.rep1 == XZAMD_REP_UNKNOWN ? 0u : z.rep1;
                    sr[3] = z.rep2 == XZAMD_REP_UNKNOWN ? 0u : z.rep2; sr[4] = z.rep3 == XZAMD_REP_UNKNOWN ? 0u : z.rep3;
                }
            }
            fresh_piece = false;
            if (!SUB && (uint64_t)(rc.tok - tok0) + 64u > tok_cap) { tok_full = true; break; }
            uint32_t off = cur - R.base;
            if (off >= 64) {
                if (off < 128) {
                    R.l = R.nl; R.d = R.nd; R.base += 64; off -= 64;
                } else {
                    R.base = cur; off = 0;
                    symrow_load(a, cur, R.l, R.d);
                }
                symrow_load(a, R.base + 64, R.nl, R.nd);
            }
            uint32_t len = lane_of(R.l, off);
            const uint32_t d = lane_of(R.d, off);
            const uint32_t as_match = len >> 15;         // the parser's choice: a match, whatever the coder's rep distances are
            len &= 0x7FFFu;
            uint32_t back, l3 = 0;
            if (len == 0) {
                back = LITERAL; len = 1; l3 = d;
                if (z.state >= 7 && !(d >> 24))          // no match byte with the record (first symbol of a piece): fetch it
                    l3 = (d & 0xFFFFu) | (uni(in[cur - z.rep0 - 1]) << 16);
            } else if (as_match) {
                back = d + 4;
            } else if (len == 1) {
                if (d == z.rep0) back = 0;
                else { back = LITERAL; l3 = literal_bytes(in, cur, cur - block_start, z); }   // a short rep0 of another distance: code the byte
            } else if (d == z.rep0) back = 0;
            else if (d == z.rep1) back = 1;
            else if (d == z.rep2) back = 2;
            else if (d == z.rep3) back = 3;
            else back = d + 4;
            if (len > MATCH_LEN_MAX || cur + len > seg_end
                    || (back != LITERAL && back >= 4 && back - 4 >= cur - block_start)
                    || (TOKM && nchunks + 1 >= ccap)) {
                // internal consistency failure: report it and leave through the loop conditions (an early return from
                // inside the loops costs the compiler its proof that the coder state is wave-uniform)
                f_pos = cur - block_start; f_back = back; f_len = len; f_d = d;
                failed = true;
                cur = span_end;
                break;
            }
            encode_symbol_t<false, LITG, TOKM, false, BND>(rc, probs, z, cur - block_start, back, len, l3);
            cur += len;
        }
        if (failed) break;
        if (TOKM) {
            const uint32_t usize = cur - chunk_start;
            if (usize != 0) {
                const uint32_t ntok = (uint32_t)(rc.tok - chunk_tok);
                const uint32_t cidx = cbase + nchunks;
                uint32_t flags = 0u;
                if (need_props) flags |= XZAMD_CH_PROPS;
                if (need_dict_reset) flags |= XZAMD_CH_DICT_RESET;
                if (need_state_reset) flags |= XZAMD_CH_STATE_RESET;
                need_props = false; need_dict_reset = false; need_state_reset = false;
                if (lane == 0) {
                    const uint64_t tix = (uint64_t)(chunk_tok - a.tok);
                    xzamd_chunk c;
                    c.in_start = chunk_start; c.usize = usize; c.tok_lo = (uint32_t)tix; c.tok_hi = (uint32_t)(tix >> 32);
                    c.ntok = ntok; c.csize = 0u; c.flags = flags; c.pad_ = 0;
                    a.chunks[cidx] = c;
                }
                ++nchunks;
            }
        } else if (cur != chunk_start)
            need_state_reset = false;
        (void)piece_break;
        if (tok_full) raw_until = span_end;                // the rest of the span goes out raw
    }
    if (BND) {
        // what k_model_chain needs of the span: the bounds, and whether the span can hand its model on at all
        uint32_t* gb = a.cb_bnd + (uint64_t)slot * a.model_slots_pad;
        bool bad = false;
        if (LITG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (uint32_t i = lane; i < nprob; i += 64) {
            const uint32_t v = (LITG && i >= P_LITERAL) ? lit_load(gb + i) : enc_pool[i];
            if (!(LITG && i >= P_LITERAL)) gb[i] = v;
            bad = bad || ((v & 0x7FFu) != ((v >> 11) & 0x7FFu) && (v >> 22) >= rc.log_cap);
        }
        if (__builtin_amdgcn_ballot_w64(bad) != 0 || tok_full) hflags |= XZAMD_CB_BAD_END;
        if (lane == 0) a.cb_hdr[slot] = hflags;
    }
    if (failed && lane == 0 && a.err) {
        if (atomicCAS(a.err, 0u, 3u + (uint32_t)MODE * 16u) == 0u) {
            a.err[1] = slot; a.err[2] = f_pos; a.err[3] = f_back; a.err[4] = f_len;
            a.err[5] = f_d; a.err[6] = nchunks; a.err[7] = (uint32_t)(rc.tok - tok0);
        }
    }
}

// The model of every encode span at its start (oracle: the `failed` / lookback logic of encode_block_syms): one thread
// per probability and Block walks the Block's spans in order -- start value of span k + 1 = lo where the slot's bounds met
// in span k, else the slot's logged bits replayed on its start value in span k.  A span that cannot hand its model on
// (XZAMD_CB_BAD_END), or whose start state the piece in front does not name (XZAMD_CB_BAD_START), ends the carrying: every
// later span of the Block starts with round 5's reset.  cb_carry: 0 = reset + properties, 1 = carried, 2 = flat, no properties
// (behind a stored piece).
__global__ __launch_bounds__(256) void k_model_chain(xzamd_span_args a, uint32_t nblocks)
{
    const uint32_t blk = blockIdx.y;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (blk >= nblocks) return;
    const uint32_t nprob = P_LITERAL + (0x300u << (a.lc + a.lp));
    const uint32_t cnt = a.enc_cnt[blk];
    uint32_t p = 1024;
    bool ok = true;
    for (uint32_t k = 0; k < cnt; ++k) {
        const uint32_t slot = blk * a.max_esb + k;
        const uint32_t hf = a.cb_hdr[slot];
        if (k != 0 && (hf & XZAMD_CB_BAD_START)) ok = false;
        const bool known = (hf & XZAMD_CB_KNOWN_START) != 0;
        if (!ok || known) p = 1024;
        if (k != 0) {
            if (i == 0) a.cb_carry[slot] = !ok ? 0u : known ? 2u : 1u;
            if (i < a.model_slots_pad) a.cb_start[(uint64_t)slot * a.model_slots_pad + i] = (uint16_t)(i < nprob ? p : 1024u);
        }
        if (hf & XZAMD_CB_BAD_END) ok = false;
        if (i < nprob && ok) {
            const uint32_t v = a.cb_bnd[(uint64_t)slot * a.model_slots_pad + i];
            const uint32_t lo = v & 0x7FFu, hi = (v >> 11) & 0x7FFu, nb = v >> 22;
            if (lo == hi) p = lo;
            else {
                const uint32_t* lg = a.cb_log + ((uint64_t)slot * a.model_slots_pad + i) * XZAMD_LOG_WORDS;
                uint32_t w = 0;
                for (uint32_t t = 0; t < nb; ++t) {
                    if ((t & 31u) == 0) w = lg[t >> 5];
                    const uint32_t bit = (w >> (t & 31u)) & 1u;
                    p = bit ? p - (p >> 5) : p + ((2048u - p) >> 5);
                }
            }
        }
    }
}

// Range coder of the two-phase mode: one LANE per LZMA2 chunk (rangecoder/range_encoder.h:136-263 per lane).  A chunk's
// coder starts from the reset state, so the chunks of a batch -- thousands -- are independent given their tokens; the
// serial recurrence that costs a wavefront ~16 scalar instructions per decision when it runs one span costs a lane of
// this kernel one vector instruction slot shared with 63 other chunks.
__global__ __launch_bounds__(64) void k_rc_chunks(xzamd_span_args a, uint32_t nchunk_slots)
{
    const uint32_t idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= nchunk_slots) return;
    const xzamd_chunk c = a.chunks[idx];
    if (c.usize == 0 || (c.flags & XZAMD_CH_RAW)) return;
    const uint16_t* __restrict__ tk = a.tok + (((uint64_t)c.tok_hi << 32) | c.tok_lo);
    uint8_t* const hdr = a.scratch + XZAMD_CHUNK_OUT(c.in_start, idx);
    const uint32_t hl = (c.flags & XZAMD_CH_PROPS) ? 6u : 5u;
    uint8_t* const out = hdr + hl;
    uint32_t low_lo = 0, low_hi = 0, range = 0xFFFFFFFFu, cache = 0, cache_size = 1, cpos = 0;
    const uint32_t cap = c.usize + (c.usize >> 3) + 8u;           // never reached: the chunk was estimated to shrink
    auto shift_low = [&]() {
        if (low_lo < 0xFF000000u || low_hi != 0) {
            uint32_t b = cache + low_hi;
            for (uint32_t i = 0; i < cache_size; ++i) {
                if (cpos < cap) out[cpos] = (uint8_t)b;
                ++cpos;
                b = 0xFFu + low_hi;
            }
            cache_size = 0;
            cache = low_lo >> 24;
        }
        ++cache_size;
        low_lo <<= 8;
        low_hi = 0;
    };
    auto code = [&](uint32_t t) {
        if (range < (1u << 24)) { shift_low(); range <<= 8; }
        const uint32_t bit = (t >> 12) & 1u;
        uint32_t add;
        if (t & 0x8000u) {
            range >>= 1;
            add = bit ? range : 0u;
        } else {
            const uint32_t bound = (range >> 11) * (t & 0xFFFu);
            add = bit ? bound : 0u;
            range = bit ? range - bound : bound;
        }
        const uint32_t nl = low_lo + add;
        low_hi += nl < low_lo ? 1u : 0u;
        low_lo = nl;
    };
    // tokens eight at a time (one 16-byte load per lane), the next eight in flight while these are coded: a lane's token
    // stream is sequential, but a 2-byte load per decision would put the cache latency in front of every one of them
    auto load8 = [&](uint32_t i) -> uint4 {
        uint4 v;
        __builtin_memcpy(&v, tk + i, 16);        // (the token buffer has 64 spare entries behind the last token)
        return v;
    };
    uint4 cur = make_uint4(0, 0, 0, 0);
    if (c.ntok) cur = load8(0);
    for (uint32_t i = 0; i < c.ntok; i += 8) {
        uint4 nxt = cur;
        if (i + 8 < c.ntok) nxt = load8(i + 8);
        const uint32_t left = c.ntok - i;
        code(cur.x & 0xFFFFu);
        if (left > 1) code(cur.x >> 16);
        if (left > 2) code(cur.y & 0xFFFFu);
        if (left > 3) code(cur.y >> 16);
        if (left > 4) code(cur.z & 0xFFFFu);
        if (left > 5) code(cur.z >> 16);
        if (left > 6) code(cur.w & 0xFFFFu);
        if (left > 7) code(cur.w >> 16);
        cur = nxt;
    }
    if (range < (1u << 24)) { shift_low(); range <<= 8; }     // the queue loop normalizes before the first RC_FLUSH
    for (int i = 0; i < 5; ++i) shift_low();
    const uint32_t csize = cpos;
    uint32_t ctl;
    if (c.flags & XZAMD_CH_PROPS) ctl = (c.flags & XZAMD_CH_DICT_RESET) ? 0xE0u : 0xC0u;
    else ctl = (c.flags & XZAMD_CH_STATE_RESET) ? 0xA0u : 0x80u;
    hdr[0] = (uint8_t)(ctl + ((c.usize - 1) >> 16));
    hdr[1] = (uint8_t)((c.usize - 1) >> 8);
    hdr[2] = (uint8_t)(c.usize - 1);
    hdr[3] = (uint8_t)((csize - 1) >> 8);
    hdr[4] = (uint8_t)(csize - 1);
    if (c.flags & XZAMD_CH_PROPS) hdr[5] = (uint8_t)((a.pb * 5 + a.lp) * 9 + a.lc);
    a.chunks[idx].csize = hl + csize;
    if ((csize > 65536u || csize >= cap) && a.err) atomicCAS(a.err, 0u, 4u);      // the price sum was far off: cannot happen
}

// ------------------------------------------------------------------------------------------
// Batch match finders: one wavefront per run of FIND_RUN consecutive positions.  Because find and
// skip both insert (lz_encoder_mf.c:366-441), the matches of a position depend on the data only, so
// they are computed for every position of the batch ahead of the (serial) parser, which streams
// them.  Per position one 32-byte record: LIST_K entries sorted by length (length << 23 | distance-1
// when the dictionary is <= 8 MiB, else distance-1 with the lengths in a u16 side array), the > nice_len
// extension folded into the last one, and a trailer word
//     count | len2(longest) << 8 | len2(second longest) << 16
// where len2 = length of the rep0 run behind the byte that follows the match (the "match + literal +
// rep0" edge of the parser, lzma_encoder_optimum_normal.c:728-790).
// ------------------------------------------------------------------------------------------
constexpr uint32_t FIND_RUN = 256;

// bytes matched inside one 16-byte trip (16 = all)
__device__ __forceinline__ uint32_t match16(const uint4& a, const uint4& b)
{
    const uint32_t d0 = a.x ^ b.x, d1 = a.y ^ b.y, d2 = a.z ^ b.z, d3 = a.w ^ b.w;
    const uint32_t off = d0 ? 0u : d1 ? 4u : d2 ? 8u : 12u;
    const uint32_t d = d0 ? d0 : d1 ? d1 : d2 ? d2 : d3;
    return d ? off + ((uint32_t)__builtin_ctz(d) >> 3) : 16u;
}

// continue a compare whose first `len` bytes are known equal
__device__ __forceinline__ uint32_t lane_cmplen16_from(const uint8_t* __restrict__ in, uint32_t q, uint32_t x, uint32_t len,
        uint32_t lim)
{
    while (len + 16 <= lim) {
        uint4 a, b;
        __builtin_memcpy(&a, in + q + len, 16);
        __builtin_memcpy(&b, in + x + len, 16);
        const uint32_t m = match16(a, b);
        len += m;
        if (m < 16) return len;
    }
    while (len < lim && in[q + len] == in[x + len]) ++len;
    return len;
}

// inclusive prefix maximum inside each half (32 lanes) of the wavefront
__device__ __forceinline__ uint32_t prefix_max_half(uint32_t v)
{
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false));   // row_shr:1
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false));   // row_shr:2
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false));   // row_shr:4
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false));   // row_shr:8
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));   // row_bcast:15 -> rows 1, 3
    return v;
}

struct SnArgs {
    const uint8_t* __restrict__ in;
    const uint32_t* __restrict__ sa;        // slot -> position (Block-major: the slots of a Block are its positions' range)
    const uint32_t* __restrict__ sa_rank;   // position -> slot
    const uint32_t* __restrict__ prev2;
    const uint32_t* __restrict__ prev4;
    const uint32_t* __restrict__ prev8;
    const uint32_t* __restrict__ prev16;
    const uint32_t* __restrict__ prev24;    // nearest earlier position with the same 24 / 32 bytes
    const uint32_t* __restrict__ prev32;
    // Which runs a launch covers: 0 = all; 1 = the runs that touch the first XZAMD_SEED_LEN bytes of a Block (workgroup =
    // Block * SEED_RUNS + j); 2 = all the others.  The two-phase mode parses the seed pieces (k_parse_pieces phase 0)
    // underneath launch 2.
    uint32_t mode;
};
constexpr uint32_t SEED_RUNS = XZAMD_SEED_LEN / 256 + 1;
constexpr uint32_t SN_WMAX = 5;

// Suffix-neighbourhood finder (oracle: find_sn).  Both hot kernels of this path are bound by instruction
// issue, not by memory, so the finder gives a position only the lanes it can use: a wavefront works on FOUR
// positions at once, one per DPP row of 16 lanes, and everything "uniform per position" lives in vector
// registers (the scalar unit only runs the loop).  Lane roles inside a row, t = lane & 15, slot r = own slot:
//   t =  0..4   slots r-1 .. r-5   (left neighbours, nearest first)
//   t =  5..9   slots r+1 .. r+5   (right neighbours, nearest first)
//   t = 10 / 11 / 12   nearest previous position with equal hash2 / hash3 / hash4
//   t = 13 / 14        nearest previous position with the same 8 / 16 bytes (by-products of the sort rounds)
// A neighbour is eligible when it lies earlier in the same Block and inside the dictionary; it is a candidate
// when it is more recent than every eligible neighbour nearer on its side ("recency record": exactly the
// nodes BT4's descent would visit).  All candidates are compared with the text at x in parallel (16 bytes per
// trip) and filtered by the Pareto rule with all-pairs compares inside the row (15 DPP row rotations).
// Row r of the wavefront at x0 owns positions x0 + 64 r .. x0 + 64 r + 63.  The loop is software pipelined
// three deep: while position i is compared and filtered, the first 16 bytes of every candidate of i + 1 and
// the window of i + 2 are in flight.
constexpr uint32_t SN_NONE = 0xFFFFFFFFu;     // "no neighbour in this lane" (positions are < 2^31)
constexpr uint32_t ROW_RUN = FIND_RUN / 4;    // positions per row

template <int N>
__device__ __forceinline__ uint32_t row_ror(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + N, 0xF, 0xF, false);    // row_ror:N
}
template <int N>
__device__ __forceinline__ uint32_t row_shr(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + N, 0xF, 0xF, false);    // row_shr:N, 0 shifted in
}

// one all-pairs step: partner = the lane N places away (inside the row)
template <int N>
__device__ __forceinline__ void pareto_step(uint32_t dist, uint32_t Lok, uint32_t t, bool& dom, uint32_t& rank)
{
    const uint32_t dj = row_ror<N>(dist), Lj = row_ror<N>(Lok), tj = row_ror<N>(t);
    const bool before = Lj != 0 && (dj < dist || (dj == dist && tj < t));      // partner is a candidate and sorts before me
    dom = dom || (before && (Lj >= Lok || dj == dist));
    rank += before ? 1u : 0u;
}
template <int N>
__device__ __forceinline__ void count_step(uint32_t key, uint32_t t, uint32_t& idx)
{
    const uint32_t kj = row_ror<N>(key), tj = row_ror<N>(t);
    idx += (kj < key || (kj == key && tj < t)) ? 1u : 0u;
}

__global__ __launch_bounds__(64) void k_find_sn(xzamd_span_args a, SnArgs sn, uint16_t* __restrict__ mlen,
        uint32_t* __restrict__ mdist)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t t = lane & 15, row = lane >> 4;
    uint32_t run = blockIdx.x;
    if (sn.mode == 1) {
        const uint32_t b = blockIdx.x / SEED_RUNS, j = blockIdx.x - b * SEED_RUNS;
        const uint32_t bs = b * a.block_size;
        run = bs / FIND_RUN + j;
        if ((uint64_t)run * FIND_RUN >= (uint64_t)bs + XZAMD_SEED_LEN || (uint64_t)run * FIND_RUN >= a.n) return;
    } else if (sn.mode == 2) {
        // a run belongs to launch 1 when it starts inside the seed region of its Block or reaches into the next Block
        const uint32_t x0 = run * FIND_RUN;
        const uint32_t b = x0 / a.block_size, bs = b * a.block_size;
        const uint64_t be = (uint64_t)bs + a.block_size;
        if (x0 < bs + XZAMD_SEED_LEN || ((uint64_t)x0 + FIND_RUN > be && be < a.n)) return;
    }
    const uint32_t xr0 = run * FIND_RUN + row * ROW_RUN;                 // first position of this row
    const uint8_t* __restrict__ in = sn.in;
    const uint32_t W = a.sa_window;
    const uint32_t cyclic = a.dict_size + 1;
    const uint32_t nice = a.nice_len;
    const uint32_t n = a.n;
    // lane roles
    const bool left = t < 5, right = t >= 5 && t < 10;
    const uint32_t k = left ? t : t - 5;
    const bool win_lane = (left || right) && k < W;
    // hash2 / hash4 heads and the 8- / 16-byte left neighbours.  (A hash3 head, lane 11, was measured to be worth
    // nothing on text and 0.1 % on executables next to these: its sort and inversion are not built for this finder.)
    // Lanes 11 / 15: the nearest earlier position with the same 24 / 32 bytes (by-products of the suffix-order rounds):
    // between "same 16 bytes" and the suffix-order neighbours (which share the LONGEST prefixes, at any distance) these
    // are the near candidates of medium length BT4's descent finds (measured through the oracle: -0.6 points of size on
    // a SQLite file, -0.45 on C headers, -0.9 at 9e).
    const bool hash_lane = t >= 10;
    const uint32_t* __restrict__ hp = t == 10 ? sn.prev2 : t == 11 ? sn.prev24 : t == 12 ? sn.prev4 : t == 13 ? sn.prev8
            : t == 14 ? sn.prev16 : sn.prev32;
    const uint32_t hstride = 1u;
    const uint32_t minlen = t == 10 ? 2u : 4u;
    // masks for the prefix maximum inside a side (the right side must not look into the left one)
    const bool sh1 = t != 0 && t != 5, sh2 = (left && t >= 2) || (right && t >= 7), sh4 = (left && t >= 4) || (right && t >= 9);

    // geometry of the row's current position (per lane, equal inside a row): Block start / end.  The records are
    // span independent (a match may run to the Block end): the spans are cut from these lists afterwards
    // (k_span_est / k_span_cut) and the parser clamps what it reads to its span (round_lists).
    uint32_t g_bs, g_be;
    {
        const uint32_t p = xr0 < n ? xr0 : 0u;
        const uint32_t blk = p / a.block_size;
        g_bs = blk * a.block_size;
        g_be = min(n, g_bs + a.block_size);
    }
    auto geo_next = [&](uint32_t& bs, uint32_t& be, uint32_t p) {      // (bs, be) of p - 1 -> of p
        const bool nb = p >= be;
        bs = nb ? be : bs;
        be = nb ? min(n, be + a.block_size) : be;
    };
    auto clampp = [&](uint32_t p) -> uint32_t { return p < n ? p : n - 1; };
    // neighbour of this lane for position p with rank r inside Block [bs, be)
    auto window = [&](uint32_t r, uint32_t bs, uint32_t be, bool live) -> uint32_t {
        const int32_t slot = right ? (int32_t)(r + 1 + k) : (int32_t)r - 1 - (int32_t)k;
        const bool inb = live && win_lane && slot >= (int32_t)bs && slot < (int32_t)be;
        return inb ? sn.sa[slot] : SN_NONE;
    };
    auto candidate = [&](uint32_t p, uint32_t wq, uint32_t hw, uint32_t& q, bool& valid) {
        const bool elig = wq != SN_NONE && wq < p && p - wq < cyclic;
        const uint32_t v = elig ? wq + 1 : 0u;
        uint32_t pm = v;
        { const uint32_t o = row_shr<1>(pm); pm = max(pm, sh1 ? o : 0u); }
        { const uint32_t o = row_shr<2>(pm); pm = max(pm, sh2 ? o : 0u); }
        { const uint32_t o = row_shr<4>(pm); pm = max(pm, sh4 ? o : 0u); }
        const uint32_t o1 = row_shr<1>(pm);
        const uint32_t ex = sh1 ? o1 : 0u;
        valid = elig && v > ex;
        q = elig ? wq : 0u;
        // The six hash / prefix lanes often name the same position (the nearest earlier position with the same 8 bytes is
        // usually also the one with the same 16, 24, 32): the candidate stays in the lowest of those lanes only -- which is
        // the one the Pareto rule below would keep -- and the others do not fetch its bytes again.  (hw is 0 in the
        // window lanes; computed by every lane so that the row shifts see all their source lanes.)
        bool dup = false;
        { const uint32_t o = row_shr<1>(hw); dup = dup || (t >= 11 && o == hw); }
        { const uint32_t o = row_shr<2>(hw); dup = dup || (t >= 12 && o == hw); }
        { const uint32_t o = row_shr<3>(hw); dup = dup || (t >= 13 && o == hw); }
        { const uint32_t o = row_shr<4>(hw); dup = dup || (t >= 14 && o == hw); }
        { const uint32_t o = row_shr<5>(hw); dup = dup || (t >= 15 && o == hw); }
        if (hash_lane) {
            valid = hw != 0 && hw < cyclic && hw <= p && !dup;
            q = valid ? p - hw : 0u;
        }
    };
    auto load16 = [&](uint32_t off) -> uint4 {
        uint4 v;
        __builtin_memcpy(&v, in + off, 16);
        return v;
    };

    // ---- prologue: positions i = 0, 1, 2 of the row
    const uint32_t xend = min(n, xr0 + ROW_RUN);                          // this row's positions: [xr0, xend)
    uint32_t g1_bs = g_bs, g1_be = g_be;                                  // geometry of x + 1
    geo_next(g1_bs, g1_be, xr0 + 1);
    uint32_t g2_bs = g1_bs, g2_be = g1_be;                                // geometry of x + 2
    geo_next(g2_bs, g2_be, xr0 + 2);
    uint32_t rk1 = sn.sa_rank[clampp(xr0 + 1)];
    uint32_t rk2 = sn.sa_rank[clampp(xr0 + 2)];
    uint32_t hw1 = hash_lane ? hp[hstride * clampp(xr0 + 1)] : 0u;
    uint32_t w1 = window(rk1, g1_bs, g1_be, xr0 + 1 < xend);
    uint32_t q0; bool v0;
    {
        const uint32_t rk0 = sn.sa_rank[clampp(xr0)];
        const uint32_t hw0 = hash_lane ? hp[hstride * clampp(xr0)] : 0u;
        candidate(xr0, window(rk0, g_bs, g_be, xr0 < xend), hw0, q0, v0);
        if (!(xr0 < xend)) v0 = false;
    }
    bool pf0 = xr0 + 16 <= n;
    uint4 A0 = make_uint4(0, 0, 0, 0), B0 = A0;
    if (pf0) { A0 = load16(q0); B0 = load16(xr0); }

    uint32_t macc = 0;
    uint32_t i = 0;
    for (; i < ROW_RUN; ++i) {
        const uint32_t x = xr0 + i;
        if (__builtin_amdgcn_readfirstlane((int)(blockIdx.x * FIND_RUN + i)) >= (int)n) break;   // every row is past the end
        // stage 0: window of x + 2 (its rank arrived last iteration), rank of x + 3, hash word of x + 2
        const uint32_t w2 = window(rk2, g2_bs, g2_be, x + 2 < xend);
        const uint32_t rk3 = sn.sa_rank[clampp(x + 3)];
        const uint32_t hw2 = hash_lane ? hp[hstride * clampp(x + 2)] : 0u;
        // stage 1: candidates of x + 1 and their first 16 bytes
        uint32_t q1; bool v1;
        candidate(x + 1, w1, hw1, q1, v1);
        if (!(x + 1 < xend)) { v1 = false; q1 = 0; }
        const bool pf1 = x + 1 + 16 <= n;
        uint4 A1 = A0, B1 = B0;
        if (pf1) { A1 = load16(q1); B1 = load16(x + 1); }
        // stage 2: position x
        uint32_t mval = 0;                                  // this position's 16-bit summary for the span plan (top lane)
        if (x < xend) {
            const uint32_t q = q0;
            const bool valid = v0;
            const uint32_t avail = g_be - x;
            const uint32_t buf_avail = avail < MATCH_LEN_MAX ? avail : MATCH_LEN_MAX;
            const uint32_t len_limit = nice <= avail ? nice : avail;
            const bool mf_ok = nice <= avail || avail >= 4;    // "pending": nothing is reported (lz_encoder_mf.c:190-201)
            const uint64_t rec_base = (uint64_t)x * LIST_W;
            const uint32_t lim = (valid && mf_ok) ? len_limit : 0u;
            uint32_t L;
            if (pf0) {
                const uint32_t m = match16(A0, B0);
                L = m < lim ? m : lim;
                if (m == 16 && lim > 16) L = lane_cmplen16_from(in, q, x, 16, lim);
            } else {
                L = lane_cmplen16_from(in, q, x, 0, lim);
            }
            const bool ok = lim != 0 && L >= minlen;
            const uint32_t dist = x - q;                        // delta >= 1 on ok lanes
            const uint32_t Lok = ok ? L : 0u;
            // Pareto set inside the row: drop a candidate when another one sorts before it (closer, or the same
            // position in a lower lane) and is at least as long; rank = candidates sorting before me
            bool dom = false;
            uint32_t rank = 0;
            pareto_step<1>(dist, Lok, t, dom, rank); pareto_step<2>(dist, Lok, t, dom, rank); pareto_step<3>(dist, Lok, t, dom, rank);
            pareto_step<4>(dist, Lok, t, dom, rank); pareto_step<5>(dist, Lok, t, dom, rank); pareto_step<6>(dist, Lok, t, dom, rank);
            pareto_step<7>(dist, Lok, t, dom, rank); pareto_step<8>(dist, Lok, t, dom, rank); pareto_step<9>(dist, Lok, t, dom, rank);
            pareto_step<10>(dist, Lok, t, dom, rank); pareto_step<11>(dist, Lok, t, dom, rank); pareto_step<12>(dist, Lok, t, dom, rank);
            pareto_step<13>(dist, Lok, t, dom, rank); pareto_step<14>(dist, Lok, t, dom, rank); pareto_step<15>(dist, Lok, t, dom, rank);
            const bool keep = ok && !dom;
            // index among the kept entries in distance (= length) order, and their number
            const uint32_t key = keep ? rank : 0xFFu;
            uint32_t kidx = 0;
            count_step<1>(key, t, kidx); count_step<2>(key, t, kidx); count_step<3>(key, t, kidx); count_step<4>(key, t, kidx);
            count_step<5>(key, t, kidx); count_step<6>(key, t, kidx); count_step<7>(key, t, kidx); count_step<8>(key, t, kidx);
            count_step<9>(key, t, kidx); count_step<10>(key, t, kidx); count_step<11>(key, t, kidx); count_step<12>(key, t, kidx);
            count_step<13>(key, t, kidx); count_step<14>(key, t, kidx); count_step<15>(key, t, kidx);
            const uint64_t kmask = __ballot(keep);
            const uint32_t cnt = (uint32_t)__builtin_popcount((uint32_t)(kmask >> (lane & 48)) & 0xFFFFu);
            const bool top = keep && kidx + 1 == cnt, second = keep && kidx + 2 == cnt;
            uint32_t len_out = L;
            if (top && L == nice) len_out = lane_cmplen16_from(in, q, x, L, buf_avail);      // > nice_len extension
            // rep0 run behind the byte after the match, for the two longest entries
            uint32_t l2 = 0;
            if ((top || second) && len_out + 1 < avail) {
                const uint32_t lim2 = min(avail, len_out + 1 + LEN2_MAX);
                l2 = lane_cmplen16_from(in, q, x, len_out + 1, lim2) - (len_out + 1);
            }
            const uint32_t drop = cnt > LIST_K ? cnt - LIST_K : 0;
            if (keep && kidx >= drop) {
                const uint64_t o = rec_base + (kidx - drop);
                if (a.list_packed) {
                    mdist[o] = (len_out << 23) | (dist - 1);
                } else {
                    mlen[o] = (uint16_t)len_out;
                    mdist[o] = dist - 1;
                }
            }
            // trailer: count | len2(longest) << 8 | len2(second) << 16, written bytewise by the lanes that know
            uint8_t* tr = reinterpret_cast<uint8_t*>(mdist + rec_base + LIST_K);
            if (cnt == 0) {
                if (t == 0) mdist[rec_base + LIST_K] = 0;
            } else {
                if (top) {
                    *reinterpret_cast<uint16_t*>(tr) = (uint16_t)((cnt - drop) | (l2 << 8));
                    tr[3] = 0;
                    if (cnt == 1) tr[2] = 0;
                    // what the span plan's walk needs of this position (k_span_est), 2 bytes instead of the 32-byte record
                    mval = len_out | ((dist > 1 ? 32u - (uint32_t)__builtin_clz(dist - 1) : 0u) << 9);
                }
                if (second) tr[2] = (uint8_t)l2;
            }
        }
        // the summaries of 16 consecutive positions of a row are collected in its 16 lanes and stored together
        // (a 2-byte store per position from one lane per row costs as much as the whole 32-byte record)
        {
            uint32_t rv = mval;
            rv |= row_ror<1>(rv); rv |= row_ror<2>(rv); rv |= row_ror<4>(rv); rv |= row_ror<8>(rv);
            macc = t == (i & 15u) ? rv : macc;
            if ((i & 15u) == 15u) {
                const uint32_t px = xr0 + (i & ~15u) + t;
                if (px < xend) a.mtop[px] = (uint16_t)macc;
            }
        }
        // shift the pipeline
        q0 = q1; v0 = v1; pf0 = pf1; A0 = A1; B0 = B1;
        w1 = w2; hw1 = hw2;
        rk2 = rk3;
        g_bs = g1_bs; g_be = g1_be;
        g1_bs = g2_bs; g1_be = g2_be;
        geo_next(g2_bs, g2_be, x + 3);
    }
    if (i & 15u) {                                          // the loop ended inside a group of 16 (end of the batch)
        const uint32_t px = xr0 + (i & ~15u) + t;
        if (t < (i & 15u) && px < xend) a.mtop[px] = (uint16_t)macc;
    }
}

// ------------------------------------------------------------------------------------------
// Cost-balanced spans (oracle: est_chunk / plan_spans).  A wavefront needs one step per position the optimal
// parser visits, and positions covered by a match of nice_len bytes or more are not visited; a state reset costs a
// few hundred bytes of model learning whatever the data.  So spans are cut by estimated parser work instead of
// input bytes: the spans of a launch take about equally long whatever they hold, and highly compressible data
// gets the long spans its small output needs.
//   k_span_est   one thread per chunk of XZAMD_EST_CHUNK positions: a walk over the match lists.  A position whose
//                longest match reaches nice_len costs EST_LONG units and the walk jumps over the match, any other
//                position one unit; alongside, a greedy-parse estimate of the coded size in bits.
//   k_span_cut   one wavefront per Block: k = max(1, work of the Block / target) spans of equal estimated work.
// ------------------------------------------------------------------------------------------
constexpr uint32_t EST_LONG = 4;

__global__ __launch_bounds__(256) void k_span_est(xzamd_span_args a, uint32_t nblocks, uint32_t cpb,
        uint32_t* __restrict__ est, unsigned long long* __restrict__ totals)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nch = nblocks * cpb;
    if (t >= nch) return;
    const uint32_t b = t / cpb, c = t - b * cpb;
    const uint32_t bs = b * a.block_size;
    const uint32_t be = min(a.n, bs + a.block_size);
    const uint64_t c0_ = (uint64_t)bs + (uint64_t)c * XZAMD_EST_CHUNK;
    uint32_t w = 0, bits = 0;
    if (c0_ < be) {
        const uint32_t c0 = (uint32_t)c0_;
        const uint32_t c1 = be - c0 < XZAMD_EST_CHUNK ? be : c0 + XZAMD_EST_CHUNK;
        uint32_t x = c0, gnext = c0;
        uint32_t cb = 0xFFFFFFFFu;                       // eight summaries at a time (16 bytes) through registers
        uint4 buf = make_uint4(0, 0, 0, 0);
        while (x < c1) {
            const uint32_t base = x & ~7u;
            if (base != cb) { buf = *reinterpret_cast<const uint4*>(a.mtop + base); cb = base; }
            const uint32_t k = x & 7u;
            const uint32_t wv = k < 2 ? buf.x : k < 4 ? buf.y : k < 6 ? buf.z : buf.w;
            const uint32_t v = (k & 1u) ? wv >> 16 : wv & 0xFFFFu;
            const uint32_t len = v & 0x1FFu, bl = v >> 9;          // bl <= 7 <=> zero-based distance < 128
            if (x >= gnext) {
                if (len >= 3 || (len == 2 && bl <= 7)) {
                    bits += 14 + bl;
                    gnext = x + len;
                } else {
                    bits += 6;
                    gnext = x + 1;
                }
            }
            if (len >= a.nice_len) { w += EST_LONG; x += len; }
            else { w += 1; x += 1; }
        }
        atomicAdd(&totals[b], (unsigned long long)w);
        atomicAdd(&totals[nblocks], (unsigned long long)w);
    }
    est[t] = w;
    est[nch + t] = bits;
}

__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v)
{
    const uint32_t lane = threadIdx.x;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const uint32_t o = __shfl_up(v, s);
        if (lane >= (uint32_t)s) v += o;
    }
    return v;
}

__global__ __launch_bounds__(64) void k_span_cut(xzamd_span_args a, uint32_t nblocks, uint32_t cpb,
        const uint32_t* __restrict__ est, unsigned long long* __restrict__ totals, uint32_t* __restrict__ span_tab,
        uint32_t* __restrict__ span_cnt, uint32_t cost_min, uint32_t bits_min, uint32_t min_len,
        uint32_t* __restrict__ enc_tab, uint32_t* __restrict__ enc_cnt, uint32_t* __restrict__ span_key)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    const uint32_t nch = nblocks * cpb;
    const uint32_t bs = b * a.block_size;
    const uint32_t be = min(a.n, bs + a.block_size);
    const uint32_t m = (be - bs + XZAMD_EST_CHUNK - 1) / XZAMD_EST_CHUNK;      // chunks of this Block
    // Work target of a span: cost_min, whatever the batch or the GPU -- the plan (and with it the output) of a Block is
    // a function of the Block and the options alone.
    const unsigned long long T = cost_min;
    if (b == 0 && lane == 0) totals[nblocks + 1] = T;
    const uint32_t* wk = est + (uint64_t)b * cpb;
    const uint32_t* bt = est + nch + (uint64_t)b * cpb;
    // two-phase: the first XZAMD_SEED_LEN bytes are the seed piece and the plan covers the rest
    const bool two = a.enc_bits != 0;
    const uint32_t seed_chunks = two && be - bs > XZAMD_SEED_LEN ? XZAMD_SEED_LEN / XZAMD_EST_CHUNK : 0u;
    // spans of this Block: total / T of equal estimated work, but no more than its estimated coded size allows at
    // bits_min per span (highly compressible Blocks get fewer, longer spans)
    unsigned long long total = 0, total_bits = 0, all_bits = 0;
    for (uint32_t c0 = 0; c0 < m; c0 += 64) {
        const uint32_t c = c0 + lane;
        const uint32_t vb = c < m ? bt[c] : 0u, vw = c < m ? wk[c] : 0u;
        const bool planned = c >= seed_chunks;
        all_bits += lane_of(wave_incl_sum(vb), 63);
        total_bits += lane_of(wave_incl_sum(planned ? vb : 0u), 63);
        total += lane_of(wave_incl_sum(planned ? vw : 0u), 63);
    }
    unsigned long long k = total / T;
    if (bits_min) {
        // (round 5's growth of bits_min on highly compressible Blocks is gone: oracle plan_spans_ex)
        const unsigned long long kb = total_bits / (unsigned long long)bits_min;
        if (kb < k) k = kb;
    }
    if (k == 0) k = 1;
    const unsigned long long Tb = (total + k - 1) / k;
    // encode spans (two-phase): ke of about equal estimated coded size, each closed at a piece end
    unsigned long long ke = two ? all_bits / a.enc_bits : 1ull;
    if (ke > (be - bs) / XZAMD_ENC_MIN_LEN) ke = (be - bs) / XZAMD_ENC_MIN_LEN;
    if (ke == 0) ke = 1;
    const unsigned long long Eb = (all_bits + ke - 1) / ke;
    uint32_t* tab = span_tab + 2ull * b * a.max_spb;
    uint32_t* etab = two ? enc_tab + 2ull * b * a.max_esb : nullptr;
    uint32_t ns = 1, start = 0;                       // spans so far, first chunk of the open span
    uint32_t ne = 1, estart = 0;                      // encode spans so far, first chunk of the open one
    unsigned long long carry_w = 0;                   // estimated work of the open span in front of the window
    unsigned long long carry_b = 0;                   // estimated bits of the open encode span in front of the window
    if (lane == 0) { tab[0] = bs; if (two) etab[0] = bs; }
    for (uint32_t c0 = 0; c0 < m; c0 += 64) {
        const uint32_t c = c0 + lane;
        const uint32_t pw = wave_incl_sum(c < m ? wk[c] : 0u);
        const uint32_t pb = two ? wave_incl_sum(c < m ? bt[c] : 0u) : 0u;
        uint32_t subw = 0;                            // window sum up to the last cut inside the window
        uint32_t subb = 0;                            // the same for the bits, up to the last encode cut
        for (;;) {
            const unsigned long long accw = carry_w + (pw - subw);
            const unsigned long long len = (unsigned long long)(c + 1 - start) * XZAMD_EST_CHUNK;
            const bool cut = c + 1 < m && c >= start && ns < a.max_spb
                    && (len >= XZAMD_SPAN_MAX || (accw >= Tb && len >= min_len) || c + 1 == seed_chunks);
            const uint64_t mask = __builtin_amdgcn_ballot_w64(cut);
            if (!mask) break;
            const uint32_t L = (uint32_t)__builtin_ctzll(mask);
            start = c0 + L + 1;
            const unsigned long long closed = carry_w + (lane_of(pw, L) - subw);      // estimated work of the span just closed
            subw = lane_of(pw, L);
            carry_w = 0;
            const uint32_t p = bs + start * XZAMD_EST_CHUNK;
            if (lane == 0) {
                tab[2 * ns - 1] = p;                  // end of the span just closed
                tab[2 * ns] = p;
                // launch-order key: heaviest first (ascending sort of ~work); unused slots keep 0xFFFFFFFF = last
                span_key[(uint64_t)b * a.max_spb + ns - 1] = ~(uint32_t)min(closed ? closed : 1ull, 0xFFFFFFFEull);
            }
            ++ns;
            if (two) {
                const unsigned long long accb = carry_b + (lane_of(pb, L) - subb);
                if (accb >= Eb && (unsigned long long)(start - estart) * XZAMD_EST_CHUNK >= XZAMD_ENC_MIN_LEN && ne < a.max_esb) {
                    if (lane == 0) { etab[2 * ne - 1] = p; etab[2 * ne] = p; }
                    ++ne;
                    estart = start;
                    subb = lane_of(pb, L);
                    carry_b = 0;
                }
            }
        }
        carry_w += lane_of(pw, 63) - subw;
        carry_b += lane_of(pb, 63) - subb;
    }
    if (lane == 0) {
        tab[2 * ns - 1] = be;
        span_cnt[b] = ns;
        span_key[(uint64_t)b * a.max_spb + ns - 1] = ~(uint32_t)min(carry_w ? carry_w : 1ull, 0xFFFFFFFEull);
        if (two) { etab[2 * ne - 1] = be; enc_cnt[b] = ne; }
    }
}

// Where the first part of every piece ends (oracle: part_end_of): behind the 4 KiB chunk at which an eighth of the piece's
// estimated work has been seen, at least XZAMD_PART_MIN bytes; the seed piece's part is the whole of it.  One thread per piece slot.
__global__ __launch_bounds__(256) void k_part_ends(xzamd_span_args a, uint32_t nblocks, uint32_t cpb, const uint32_t* __restrict__ est,
        uint32_t* __restrict__ part_tab)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= nblocks * a.max_spb) return;
    const uint32_t blk = slot / a.max_spb, k = slot - blk * a.max_spb;
    if (k >= a.span_cnt[blk]) return;
    const uint32_t bs = blk * a.block_size;
    const uint32_t s0 = a.span_tab[2 * slot], pe = a.span_tab[2 * slot + 1];
    uint32_t end = pe;
    if (k != 0 && pe - s0 > XZAMD_PART_MIN) {
        const uint32_t* wk = est + (uint64_t)blk * cpb;
        const uint32_t c0 = (s0 - bs) / XZAMD_EST_CHUNK, c1 = (pe - bs + XZAMD_EST_CHUNK - 1) / XZAMD_EST_CHUNK;
        unsigned long long total = 0, acc = 0;
        for (uint32_t c = c0; c < c1; ++c) total += wk[c];
        const unsigned long long target = (total + 7) / 8;
        for (uint32_t c = c0; c < c1; ++c) {             // the chunk boundary nearest to where the eighth is reached
            const uint64_t b64 = (uint64_t)bs + (uint64_t)c * XZAMD_EST_CHUNK;
            if (acc + wk[c] / 2 >= target && b64 >= (uint64_t)s0 + XZAMD_PART_MIN) { end = (uint32_t)b64; break; }
            acc += wk[c];
        }
    }
    part_tab[slot] = end;
}

// Exact HC3/HC4 finder in list form (optimal parser over the reference's match finder; test
// configuration).  Same record layout, len2 = 0.
__global__ __launch_bounds__(64) void k_find_exact(xzamd_span_args a, uint16_t* __restrict__ mlen,
        uint32_t* __restrict__ mdist)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t x0 = blockIdx.x * FIND_RUN;
    if (x0 >= a.n) return;
    const uint32_t x1 = min(a.n, x0 + FIND_RUN);
    Env e;
    e.in = a.in; e.rank = a.rank; e.sorted_pos = a.sorted_pos; e.prev2 = a.prev2; e.prev3 = a.prev3;
    e.nice = a.nice_len; e.depth = a.depth; e.hb = a.hash_bytes; e.cyclic = a.dict_size + 1;
    e.block_end = 0; e.n_last = a.n - 1;
    e.mlen = nullptr; e.mdist = nullptr; e.packed = a.list_packed;
    Pre P;
    P.valid = false; P.pos = 0; P.ent = 0;
    P.a.rk = P.a.d2 = P.a.d3 = 0; P.an = P.a;
    uint32_t span_end = 0;
    for (uint32_t x = x0; x < x1; ++x) {
        if (x >= span_end) {
            const uint32_t blk = x / a.block_size;
            const uint32_t block_start = blk * a.block_size;
            const uint32_t block_end = min(a.n, block_start + a.block_size);
            const uint64_t k = (x - block_start) / a.span_size;
            const uint64_t se = (uint64_t)block_start + (k + 1) * a.span_size;
            span_end = se < block_end ? (uint32_t)se : block_end;
            e.block_end = block_end;
        }
        Round R;
        do_round<false>(e, P, x, span_end, 0, 0, 0, 0, R);
        const uint64_t lt = (1ull << lane) - 1;
        const bool rec = (R.mask >> lane) & 1;
        const uint32_t idx = (uint32_t)__builtin_popcountll(R.mask & lt);
        const uint32_t cnt = (uint32_t)__builtin_popcountll(R.mask);
        const uint32_t drop = cnt > LIST_K ? cnt - LIST_K : 0;
        const uint64_t rec_base = (uint64_t)x * LIST_W;
        if (rec && idx >= drop) {
            const uint32_t len = idx + 1 == cnt ? R.longest : R.L;
            const uint64_t o = rec_base + (idx - drop);
            if (e.packed) {
                mdist[o] = (len << 23) | R.D;
            } else {
                mlen[o] = (uint16_t)len;
                mdist[o] = R.D;
            }
        }
        if (lane == 0) mdist[rec_base + LIST_K] = cnt - drop;
    }
}

// ------------------------------------------------------------------------------------------
// x86 BCJ encoder (simple/x86.c:26-118), one Block = one fresh filter (x86.c:121-136), start offset 0.
// The filter is a sequential state machine, but (1) every decision reads ORIGINAL bytes only -- a
// converted CALL/JMP skips its own four operand bytes, nothing re-reads a patched byte -- and (2) its
// state (prev_mask, prev_pos) is void at any position preceded by five bytes without an E8/E9: the
// next opcode then sees offset > 5 and clears prev_mask whatever came before, and no conversion can
// straddle such a position.  So a chunk owner starts at the first such synchronisation point of its
// chunk and runs to the first one at or after the chunk end (= where the next owner starts): exact,
// chunk-parallel, and sequential only on input without synchronisation points.
// `out` already holds a copy of `in`; only converted operands are written.
// ------------------------------------------------------------------------------------------
constexpr uint32_t BCJ_CHUNK = 2048;

__device__ __forceinline__ bool x86_is_op(uint32_t b) { return (b & 0xFEu) == 0xE8u; }
__device__ __forceinline__ bool x86_ms(uint32_t b) { return b == 0u || b == 0xFFu; }

__global__ __launch_bounds__(256) void k_x86_bcj(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t n,
        uint32_t block_size, uint32_t chunks_per_block, uint32_t nchunks)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunks) return;
    const uint32_t blk = t / chunks_per_block, k = t - blk * chunks_per_block;
    const uint32_t bs = blk * block_size;
    if (bs >= n) return;
    const uint32_t size = min(n - bs, block_size);
    if (size < 5) return;
    const uint32_t limit = size - 5;                 // last position the filter examines
    const uint32_t s = k * BCJ_CHUNK;
    if (s > limit) return;
    const uint32_t e = s + BCJ_CHUNK;                // may exceed size; only compared
    const uint8_t* __restrict__ b = in + bs;
    uint8_t* __restrict__ o = out + bs;
    uint32_t pos = 0, run = 0;                       // run = non-opcode bytes immediately before pos
    if (k != 0) {
        bool found = false;
        for (uint32_t q = s - 5; q <= limit && q < e; ++q) {
            if (q >= s && run >= 5) { pos = q; found = true; break; }
            run = x86_is_op(b[q]) ? 0u : run + 1;
        }
        if (!found) return;                           // the previous owner runs through this chunk
    }
    uint32_t prev_mask = 0, prev_pos = pos - 6;       // "long ago" (x86.c:133: -5 at the Block start acts the same)
    while (pos <= limit) {
        if (pos >= e && run >= 5) break;              // next owner's start
        const uint32_t c = b[pos];
        if (!x86_is_op(c)) { ++pos; ++run; continue; }
        const uint32_t offset = pos - prev_pos;
        prev_pos = pos;
        if (offset > 5) prev_mask = 0;
        else for (uint32_t i = 0; i < offset; ++i) prev_mask = (prev_mask & 0x77u) << 1;
        uint32_t b4 = b[pos + 4];
        if (x86_ms(b4) && (prev_mask >> 1) <= 4 && (prev_mask >> 1) != 3) {
            const uint32_t b1 = b[pos + 1], b2 = b[pos + 2], b3 = b[pos + 3];
            uint32_t src = (b4 << 24) | (b3 << 16) | (b2 << 8) | b1;
            uint32_t dest;
            for (;;) {
                dest = src + (pos + 5);
                if (prev_mask == 0) break;
                const uint32_t pm = prev_mask >> 1;
                const uint32_t i = pm == 0 ? 0u : pm == 1 ? 1u : pm <= 3 ? 2u : 3u;    // MASK_TO_BIT_NUMBER
                const uint32_t bb = (dest >> (24 - i * 8)) & 0xFFu;
                if (!x86_ms(bb)) break;
                src = dest ^ ((1u << (32 - i * 8)) - 1);
            }
            o[pos + 4] = (uint8_t)(~(((dest >> 24) & 1) - 1));
            o[pos + 3] = (uint8_t)(dest >> 16);
            o[pos + 2] = (uint8_t)(dest >> 8);
            o[pos + 1] = (uint8_t)dest;
            run = x86_is_op(b1) ? 0u : 1u;
            run = x86_is_op(b2) ? 0u : run + 1;
            run = x86_is_op(b3) ? 0u : run + 1;
            run = x86_is_op(b4) ? 0u : run + 1;
            pos += 5;
            prev_mask = 0;
        } else {
            ++pos;
            run = 0;
            prev_mask |= 1;
            if (x86_ms(b4)) prev_mask |= 0x10;
        }
    }
}

// ------------------------------------------------------------------------------------------
// ARM64 BCJ encoder (simple/arm64.c:20-105) and delta encoder (delta/delta_encoder.c:20-45), one fresh
// filter per Block, start offset 0.  Both are stateless given the ORIGINAL bytes (ARM64: every aligned
// 4-byte instruction on its own, pc = offset inside the Block; delta: byte minus the byte `dist` before it
// in the Block, zero history), so they are plain data-parallel maps -- one thread per instruction / byte.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_arm64_bcj(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t n,
        uint32_t block_size, uint32_t nblocks)
{
    const uint32_t spb = block_size / 4;                      // instruction slots per full Block
    if (spb == 0) return;
    const uint64_t total = (uint64_t)spb * nblocks;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const uint32_t b = (uint32_t)(t / spb), k = (uint32_t)(t - (uint64_t)b * spb);
        const uint32_t bs = b * block_size;
        if (bs >= n) continue;
        const uint32_t len = min(n - bs, block_size) & ~3u;   // arm64.c:26: the tail (size & 3) stays as it is
        const uint32_t pc = k * 4;
        if (pc + 4 > len) continue;
        const uint32_t g = bs + pc;
        uint32_t instr;
        __builtin_memcpy(&instr, in + g, 4);
        if ((instr >> 26) == 0x25) {
            instr = 0x94000000u | ((instr + (pc >> 2)) & 0x03FFFFFFu);
            __builtin_memcpy(out + g, &instr, 4);
        } else if ((instr & 0x9F000000u) == 0x90000000u) {
            const uint32_t src = ((instr >> 29) & 3) | ((instr >> 3) & 0x001FFFFCu);
            if ((src + 0x00020000u) & 0x001C0000u) continue;
            instr &= 0x9000001Fu;
            const uint32_t dest = src + (pc >> 12);
            instr |= (dest & 3) << 29;
            instr |= (dest & 0x0003FFFCu) << 3;
            instr |= (0u - (dest & 0x00020000u)) & 0x00E00000u;
            __builtin_memcpy(out + g, &instr, 4);
        }
    }
}

// ARM / PowerPC / SPARC (simple/arm.c, powerpc.c, sparc.c: one 4-byte instruction per slot), ARM-Thumb
// (simple/armthumb.c: 2-byte slots, a BL pair is 4 bytes) and IA-64 (simple/ia64.c: 16-byte bundles of three
// 41-bit slots).  pc = offset inside the Block (start offset 0); the tail the reference leaves unfiltered
// (size & 3, size & 15, the last < 4 bytes) stays as it is.  Every slot converts on its own: in ARM-Thumb the
// reference skips the halfword behind a converted pair, but that halfword can never start a pair itself (its
// second byte would have to be 0xF0..0xF7 and 0xF8..0xFF at once), so the slots are independent there too.
__global__ __launch_bounds__(256) void k_bcj_simple(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t n,
        uint32_t block_size, uint32_t nblocks, uint32_t kind)
{
    const uint32_t unit = kind == 8 ? 2u : kind == 6 ? 16u : 4u;
    const uint32_t spb = block_size / unit;                   // slots per full Block
    if (spb == 0) return;
    const uint64_t total = (uint64_t)spb * nblocks;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const uint32_t b = (uint32_t)(t / spb), k = (uint32_t)(t - (uint64_t)b * spb);
        const uint32_t bs = b * block_size;
        if (bs >= n) continue;
        const uint32_t blen = min(n - bs, block_size);
        const uint32_t pc = k * unit;
        const uint8_t* p = in + bs + pc;
        uint8_t* q = out + bs + pc;
        if (kind == 8) {                                      // ARM-Thumb BL pair
            if (blen < 4 || pc > blen - 4) continue;
            if ((p[1] & 0xF8u) != 0xF0u || (p[3] & 0xF8u) != 0xF8u) continue;
            uint32_t src = ((uint32_t)(p[1] & 7u) << 19) | ((uint32_t)p[0] << 11) | ((uint32_t)(p[3] & 7u) << 8) | p[2];
            src <<= 1;
            const uint32_t dest = (pc + 4 + src) >> 1;
            q[1] = (uint8_t)(0xF0u | ((dest >> 19) & 7u));
            q[0] = (uint8_t)(dest >> 11);
            q[3] = (uint8_t)(0xF8u | ((dest >> 8) & 7u));
            q[2] = (uint8_t)dest;
        } else if (kind == 6) {                               // IA-64 bundle
            if (pc + 16 > (blen & ~15u)) continue;
            uint8_t bun[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) bun[i] = p[i];
            const uint32_t tmpl = bun[0] & 0x1Fu;
            // branch slots by template (ia64.c BRANCH_TABLE): 16,17: 4  18,19: 6  22,23: 7  24,25,28,29: 4
            const uint32_t mask = (tmpl == 16 || tmpl == 17 || tmpl == 24 || tmpl == 25 || tmpl == 28 || tmpl == 29) ? 4u
                    : (tmpl == 18 || tmpl == 19) ? 6u : (tmpl == 22 || tmpl == 23) ? 7u : 0u;
            bool changed = false;
            uint32_t bit_pos = 5;
            for (uint32_t slot = 0; slot < 3; ++slot, bit_pos += 41) {
                if (((mask >> slot) & 1u) == 0) continue;
                const uint32_t byte_pos = bit_pos >> 3, bit_res = bit_pos & 7u;
                uint64_t instruction = 0;
                for (uint32_t j = 0; j < 6; ++j) instruction += (uint64_t)bun[j + byte_pos] << (8 * j);
                uint64_t norm = instruction >> bit_res;
                if (((norm >> 37) & 0xFu) != 0x5u || ((norm >> 9) & 0x7u) != 0) continue;
                uint32_t src = (uint32_t)((norm >> 13) & 0xFFFFFu);
                src |= (uint32_t)((norm >> 36) & 1u) << 20;
                src <<= 4;
                const uint32_t dest = (pc + src) >> 4;
                norm &= ~((uint64_t)0x8FFFFF << 13);
                norm |= (uint64_t)(dest & 0xFFFFFu) << 13;
                norm |= (uint64_t)(dest & 0x100000u) << (36 - 20);
                instruction &= (1u << bit_res) - 1;
                instruction |= norm << bit_res;
                for (uint32_t j = 0; j < 6; ++j) bun[j + byte_pos] = (uint8_t)(instruction >> (8 * j));
                changed = true;
            }
            if (changed) {
#pragma unroll
                for (int i = 0; i < 16; ++i) q[i] = bun[i];
            }
        } else {
            if (pc + 4 > (blen & ~3u)) continue;
            if (kind == 7) {                                  // ARM BL (little endian, condition "always")
                if (p[3] != 0xEBu) continue;
                uint32_t src = ((uint32_t)p[2] << 16) | ((uint32_t)p[1] << 8) | p[0];
                src <<= 2;
                const uint32_t dest = (pc + 8 + src) >> 2;
                q[2] = (uint8_t)(dest >> 16); q[1] = (uint8_t)(dest >> 8); q[0] = (uint8_t)dest;
            } else if (kind == 5) {                           // PowerPC b/bl with AA = 0, LK = 1 (big endian)
                if ((p[0] >> 2) != 0x12u || (p[3] & 3u) != 1u) continue;
                const uint32_t src = ((uint32_t)(p[0] & 3u) << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (p[3] & ~3u);
                const uint32_t dest = pc + src;
                q[0] = (uint8_t)(0x48u | ((dest >> 24) & 3u));
                q[1] = (uint8_t)(dest >> 16);
                q[2] = (uint8_t)(dest >> 8);
                q[3] = (uint8_t)((p[3] & 3u) | (dest & 0xFFu));
            } else {                                          // SPARC call (big endian)
                if (!((p[0] == 0x40u && (p[1] & 0xC0u) == 0x00u) || (p[0] == 0x7Fu && (p[1] & 0xC0u) == 0xC0u))) continue;
                uint32_t src = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
                src <<= 2;
                uint32_t dest = (pc + src) >> 2;
                dest = (((0u - ((dest >> 22) & 1u)) << 22) & 0x3FFFFFFFu) | (dest & 0x3FFFFFu) | 0x40000000u;
                q[0] = (uint8_t)(dest >> 24); q[1] = (uint8_t)(dest >> 16); q[2] = (uint8_t)(dest >> 8); q[3] = (uint8_t)dest;
            }
        }
    }
}

// RISC-V (simple/riscv.c:352-609, encoder).  The reference walks the Block in 2-byte steps; what it finds at an
// examined position decides how far it jumps: 2 (nothing), 4 (a converted JAL, or an AUIPC with rd x0/x2 that is
// not the special form), 6 (an AUIPC that has no partner), 8 (a converted AUIPC pair in either direction).  All
// of that is read from bytes no earlier conversion has touched, so step(i) is a function of the input, and a
// position is examined unless an examined position 2, 4 or 6 bytes before it jumps over it.  Hence the
// synchronisation rule used to cut the walk into chunks (the x86 kernel above does the same with its own
// rule): a position whose three predecessors cannot reach over it whatever their state -- step(i-2) <= 2,
// step(i-4) <= 4, step(i-6) <= 6 -- is examined by every walk.  Each chunk's owner starts at the first such
// position inside its chunk and stops at the first one behind its chunk.
__device__ __forceinline__ uint32_t rv_rd32(const uint8_t* p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
// not a pair: rd of the AUIPC != rs1 of the second instruction, or its two lowest opcode bits are not 11
__device__ __forceinline__ bool rv_not_pair(uint32_t auipc, uint32_t inst2) { return (((auipc << 8) ^ inst2) & 0xF8003u) != 3u; }
// the special form the encoder itself produces: rd = x2, bits 13:12 = 11, and a "rs1" that is neither x0 nor x2
__device__ __forceinline__ bool rv_special(uint32_t auipc) { return (auipc & 0x3FFFu) == 0x3117u && ((auipc >> 27) & 0x1Du) != 0; }
__device__ __forceinline__ uint32_t rv_step(const uint8_t* b, uint32_t i, uint32_t limit)
{
    if (i > limit) return 2;
    const uint32_t b0 = b[i];
    if (b0 == 0xEFu) return (b[i + 1] & 0x0Du) ? 2u : 4u;
    if ((b0 & 0x7Fu) != 0x17u) return 2;
    const uint32_t inst = rv_rd32(b + i);
    if (inst & 0xE80u) return rv_not_pair(inst, rv_rd32(b + i + 4)) ? 6u : 8u;
    return rv_special(inst) ? 8u : 4u;
}
__device__ __forceinline__ bool rv_sync(const uint8_t* b, uint32_t i, uint32_t limit)
{
    return (i < 2 || rv_step(b, i - 2, limit) <= 2) && (i < 4 || rv_step(b, i - 4, limit) <= 4) && (i < 6 || rv_step(b, i - 6, limit) <= 6);
}

__global__ __launch_bounds__(256) void k_riscv_bcj(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t n,
        uint32_t block_size, uint32_t chunks_per_block, uint32_t nchunks)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunks) return;
    const uint32_t blk = t / chunks_per_block, k = t - blk * chunks_per_block;
    const uint32_t bs = blk * block_size;
    if (bs >= n) return;
    const uint32_t size = min(n - bs, block_size);
    if (size < 8) return;
    const uint32_t limit = size - 8;                 // last position the filter examines (riscv.c:364-372)
    const uint32_t s = k * BCJ_CHUNK;                // BCJ_CHUNK is even
    if (s > limit) return;
    const uint32_t e = s + BCJ_CHUNK;
    const uint8_t* __restrict__ b = in + bs;
    uint8_t* __restrict__ o = out + bs;
    uint32_t pos = 0;
    if (k != 0) {
        bool found = false;
        for (uint32_t q = s; q <= limit && q < e; q += 2)
            if (rv_sync(b, q, limit)) { pos = q; found = true; break; }
        if (!found) return;                          // the previous owner walks through this chunk
    }
    while (pos <= limit) {
        if (pos >= e && rv_sync(b, pos, limit)) break;      // the next owner's start
        const uint32_t b0 = b[pos];
        if (b0 == 0xEFu) {
            // JAL with rd = x1 / x5: pc-relative 20-bit immediate -> absolute, stored big endian (riscv.c:379-438)
            const uint32_t b1 = b[pos + 1];
            if (b1 & 0x0Du) { pos += 2; continue; }
            const uint32_t b2 = b[pos + 2], b3 = b[pos + 3];
            uint32_t addr = ((b1 & 0xF0u) << 8) | ((b2 & 0x0Fu) << 16) | ((b2 & 0x10u) << 7) | ((b2 & 0xE0u) >> 4)
                    | ((b3 & 0x7Fu) << 4) | ((b3 & 0x80u) << 13);
            addr += pos;
            o[pos + 1] = (uint8_t)((b1 & 0x0Fu) | ((addr >> 13) & 0xF0u));
            o[pos + 2] = (uint8_t)(addr >> 9);
            o[pos + 3] = (uint8_t)(addr >> 1);
            pos += 4;
        } else if ((b0 & 0x7Fu) == 0x17u) {
            uint32_t inst = rv_rd32(b + pos);
            if (inst & 0xE80u) {
                // AUIPC with rd other than x0 / x2 (riscv.c:440-551)
                const uint32_t inst2 = rv_rd32(b + pos + 4);
                if (rv_not_pair(inst, inst2)) { pos += 6; continue; }
                uint32_t addr = inst & 0xFFFFF000u;
                addr += (inst2 >> 20) - ((inst2 >> 19) & 0x1000u);
                addr += pos;
                inst = 0x17u | (2u << 7) | (inst2 << 12);
                o[pos] = (uint8_t)inst; o[pos + 1] = (uint8_t)(inst >> 8); o[pos + 2] = (uint8_t)(inst >> 16); o[pos + 3] = (uint8_t)(inst >> 24);
                o[pos + 4] = (uint8_t)(addr >> 24); o[pos + 5] = (uint8_t)(addr >> 16); o[pos + 6] = (uint8_t)(addr >> 8); o[pos + 7] = (uint8_t)addr;
            } else {
                // AUIPC with rd x0 / x2: only the special form is (un)converted (riscv.c:552-602)
                if (!rv_special(inst)) { pos += 4; continue; }
                const uint32_t fake_rs1 = inst >> 27;
                const uint32_t fake_addr = rv_rd32(b + pos + 4);
                const uint32_t fake_inst2 = (inst >> 12) | (fake_addr << 20);
                inst = 0x17u | (fake_rs1 << 7) | (fake_addr & 0xFFFFF000u);
                o[pos] = (uint8_t)inst; o[pos + 1] = (uint8_t)(inst >> 8); o[pos + 2] = (uint8_t)(inst >> 16); o[pos + 3] = (uint8_t)(inst >> 24);
                o[pos + 4] = (uint8_t)fake_inst2; o[pos + 5] = (uint8_t)(fake_inst2 >> 8); o[pos + 6] = (uint8_t)(fake_inst2 >> 16);
                o[pos + 7] = (uint8_t)(fake_inst2 >> 24);
            }
            pos += 8;
        } else {
            pos += 2;
        }
    }
}

__global__ __launch_bounds__(256) void k_delta(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t n,
        uint32_t block_size, uint32_t dist)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < n; g += stride) {
        const uint32_t bs = (g / block_size) * block_size;
        const uint8_t prev = g - bs >= dist ? in[g - dist] : (uint8_t)0;
        out[g] = (uint8_t)(in[g] - prev);
    }
}

// ------------------------------------------------------------------------------------------
// SHA-256 Block check (check/sha256.c:120-189, FIPS 180-4): a serial hash per Block, so one THREAD per
// Block (Blocks are the parallelism; the default Check, CRC64, stays the fast path).  32 bytes per Block.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rotr32(uint32_t x, uint32_t r) { return (x >> r) | (x << (32 - r)); }

__global__ __launch_bounds__(64) void k_sha256_blocks(const uint8_t* __restrict__ in, uint32_t n, uint32_t block_size,
        uint32_t nblocks, uint8_t* __restrict__ out)
{
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
        0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
        0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
        0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
        0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
        0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2 };
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const uint32_t bs = b * block_size;
    const uint32_t len = min(n, bs + block_size) - bs;
    uint32_t h[8] = { 0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19 };
    const uint32_t nchunks = (len + 9 + 63) / 64;               // message + 0x80 + 64-bit length
    const uint32_t nfull = len / 64;                            // chunks that are message bytes only
    // A SHA-256 is one serial chain of 64-byte compressions, so the parallelism is the Blocks: one LANE per Block (a
    // wavefront hashes 64 Blocks in lockstep).  Message words come in 16-byte loads, the schedule lives in a 16-word
    // ring in registers, the 64 rounds are unrolled.
    for (uint32_t c = 0; c < nchunks; ++c) {
        uint32_t w[16];
        if (c < nfull) {
            const uint8_t* p = in + bs + (uint64_t)c * 64;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint4 v;
                __builtin_memcpy(&v, p + 16 * q, 16);
                w[4 * q + 0] = __builtin_bswap32(v.x); w[4 * q + 1] = __builtin_bswap32(v.y);
                w[4 * q + 2] = __builtin_bswap32(v.z); w[4 * q + 3] = __builtin_bswap32(v.w);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                uint32_t v = 0;
                for (int k = 0; k < 4; ++k) {
                    const uint32_t o = c * 64 + i * 4 + k;
                    uint32_t byte = 0;
                    if (o < len) byte = in[bs + o];
                    else if (o == len) byte = 0x80;
                    else if (c + 1 == nchunks && i >= 14) {
                        const uint64_t bits = (uint64_t)len * 8;
                        byte = (uint32_t)(bits >> (8 * (7 - ((i - 14) * 4 + k)))) & 0xFF;
                    }
                    v = (v << 8) | byte;
                }
                w[i] = v;
            }
        }
        uint32_t a = h[0], bb = h[1], cc = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            if (i >= 16) {
                const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
                const uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
                const uint32_t s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
                w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
            }
            const uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
            const uint32_t ch = (e & f) ^ (~e & g);
            const uint32_t t1 = hh + S1 + ch + K[i] + w[i & 15];
            const uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
            const uint32_t mj = (a & bb) ^ (a & cc) ^ (bb & cc);
            const uint32_t t2 = S0 + mj;
            hh = g; g = f; f = e; e = d + t1; d = cc; cc = bb; bb = a; a = t1 + t2;
        }
        h[0] += a; h[1] += bb; h[2] += cc; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    for (int i = 0; i < 8; ++i) {
        out[b * 32 + i * 4 + 0] = (uint8_t)(h[i] >> 24);
        out[b * 32 + i * 4 + 1] = (uint8_t)(h[i] >> 16);
        out[b * 32 + i * 4 + 2] = (uint8_t)(h[i] >> 8);
        out[b * 32 + i * 4 + 3] = (uint8_t)h[i];
    }
}

// ------------------------------------------------------------------------------------------
// CRC64 (check/crc64_fast.c; ECMA-182 reflected, poly 0xC96C5795D7870F42)
// ------------------------------------------------------------------------------------------
// Both Block checks of the device path share the code: T = uint64_t is CRC64 (ECMA-182 reflected,
// check/crc64_fast.c), T = uint32_t is CRC32 (IEEE reflected, check/crc32_fast.c).
template <typename T> struct CrcP;
template <> struct CrcP<uint64_t> { static constexpr uint64_t POLY = 0xC96C5795D7870F42ull; static constexpr uint64_t TOP = 1ull << 63; };
template <> struct CrcP<uint32_t> { static constexpr uint32_t POLY = 0xEDB88320u; static constexpr uint32_t TOP = 1u << 31; };

// product of two residues in the reflected representation (MSB = x^0)
template <typename T>
__device__ __forceinline__ T gf_mul(T a, T b)
{
    T r = 0;
    for (int i = 0; i < (int)(8 * sizeof(T)); ++i) {
        if (a & CrcP<T>::TOP) r ^= b;
        a <<= 1;
        b = (T)((b >> 1) ^ ((b & 1) ? CrcP<T>::POLY : (T)0));
    }
    return r;
}

template <typename T>
__device__ __forceinline__ T gf_xpow8(uint64_t nbytes)
{
    T base = (T)(CrcP<T>::TOP >> 8);     // x^8
    T acc = CrcP<T>::TOP;                // 1
    while (nbytes) {
        if (nbytes & 1) acc = gf_mul<T>(acc, base);
        base = gf_mul<T>(base, base);
        nbytes >>= 1;
    }
    return acc;
}

// Standard CRC (init ~0, final ~) of each strip of `strip` bytes; strips never straddle Blocks.
template <typename T>
__global__ __launch_bounds__(256) void k_crc_strips(const uint8_t* __restrict__ in, uint32_t n,
        uint32_t block_size, uint32_t strip, uint32_t strips_per_block, uint32_t nstrips,
        T* __restrict__ out)
{
    __shared__ T tab[256];
    {
        T r = (T)threadIdx.x;
        for (int k = 0; k < 8; ++k) r = (T)((r >> 1) ^ ((r & 1) ? CrcP<T>::POLY : (T)0));
        tab[threadIdx.x] = r;
    }
    __syncthreads();
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nstrips) return;
    const uint32_t b = s / strips_per_block;
    const uint32_t bstart = b * block_size;
    const uint32_t bend = min(n, bstart + block_size);
    const uint32_t beg = bstart + (s - b * strips_per_block) * strip;
    T crc = (T)~(T)0;
    if (beg < bend) {
        const uint32_t end = min(bend, beg + strip);
        for (uint32_t i = beg; i < end; ++i)
            crc = (T)(tab[(crc ^ in[i]) & 0xFF] ^ (crc >> 8));
    }
    out[s] = (T)~crc;
}

// One wave per Block: fold the strip CRCs left to right: crc(A||B) = crc(A)*x^(8|B|) ^ crc(B).
// The Block's check is written as a uint64_t (CRC32 zero-extended).
template <typename T>
__global__ __launch_bounds__(64) void k_crc_fold(const T* __restrict__ strips, uint32_t n,
        uint32_t block_size, uint32_t strip, uint32_t strips_per_block, uint64_t* __restrict__ block_crc)
{
    const uint32_t b = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const uint32_t bstart = b * block_size;
    const uint32_t bend = min(n, bstart + block_size);
    const uint32_t blen = bend - bstart;
    const uint32_t ns = (blen + strip - 1) / strip;          // strips actually used
    const uint32_t per = (ns + 63) / 64;
    const uint32_t s0 = min(ns, lane * per), s1 = min(ns, s0 + per);
    const T xs = gf_xpow8<T>(strip);
    // lane-local fold over its contiguous strips
    T acc = 0;              // crc of the empty string is 0 and is the identity of the fold
    uint64_t bytes = 0;
    for (uint32_t s = s0; s < s1; ++s) {
        const uint32_t len = min(strip, blen - s * strip);
        const T c = strips[(uint64_t)b * strips_per_block + s];
        acc = (T)(gf_mul<T>(acc, len == strip ? xs : gf_xpow8<T>(len)) ^ c);
        bytes += len;
    }
    // sequential combine across lanes (64 steps, once per Block)
    T total = 0;
    for (uint32_t l = 0; l < 64; ++l) {
        const T cl = (T)__shfl((unsigned long long)acc, l);
        const uint64_t bl = __shfl((unsigned long long)bytes, l);
        if (bl) total = (T)(gf_mul<T>(total, gf_xpow8<T>(bl)) ^ cl);
    }
    if (lane == 0) block_crc[b] = (uint64_t)total;
}

// ------------------------------------------------------------------------------------------
// Assembly: gather span outputs (and small literal pieces prepared by the host: headers,
// end markers, padding, checks, index, footer) into the final stream buffer.
// One workgroup per copy segment.
// out[key[i]] = v[i] for a permutation `key` of 0..n-1: radix passes over the key bits above INV_LOW, then the LDS
// step.  keys_cur / vals_cur hold the pairs, the *_alt buffers are scratch; all four are clobbered.
// u32 values: out may be one of the value buffers.  u64 values: out_lo / out_hi must not overlap them.
template <typename V>
static hipError_t invert_perm(uint32_t* keys_cur, uint32_t* keys_alt, V* vals_cur, V* vals_alt, uint32_t n,
        uint32_t* out_lo, uint32_t* out_hi, void* tmp, size_t tmp_bytes, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    uint32_t bits = 1;
    while (bits < 32 && (1ull << bits) < n) ++bits;
    rocprim::double_buffer<uint32_t> kb(keys_cur, keys_alt);
    rocprim::double_buffer<V> vb(vals_cur, vals_alt);
    if (bits > INV_LOW) {
        size_t need = 0;
        hipError_t e = rocprim::radix_sort_pairs(nullptr, need, kb, vb, (size_t)n, INV_LOW, bits, st);
        if (e != hipSuccess) return e;
        if (need > tmp_bytes) return hipErrorOutOfMemory;
        e = rocprim::radix_sort_pairs(tmp, tmp_bytes, kb, vb, (size_t)n, INV_LOW, bits, st);
        if (e != hipSuccess) return e;
    }
    const uint32_t nb = (n + (1u << INV_LOW) - 1) >> INV_LOW;
    if constexpr (sizeof(V) == 4)
        hipLaunchKernelGGL(k_inv_low_u32, dim3(nb), dim3(1024), 0, st, kb.current(),
                reinterpret_cast<const uint32_t*>(vb.current()), n, out_lo);
    else
        hipLaunchKernelGGL(k_inv_low_u64, dim3(nb), dim3(1024), 0, st, kb.current(),
                reinterpret_cast<const uint64_t*>(vb.current()), n, out_lo, out_hi);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_assemble(const xzamd_copy_seg* __restrict__ segs, uint32_t nsegs,
        const uint8_t* __restrict__ scratch, const uint8_t* __restrict__ lits, const uint8_t* __restrict__ in,
        uint8_t* __restrict__ out)
{
    const uint32_t s = blockIdx.x;
    if (s >= nsegs) return;
    const xzamd_copy_seg sg = segs[s];
    const uint8_t* src = sg.kind == 0 ? scratch + sg.src : (sg.kind == 1 ? lits + sg.src : in + sg.src);
    uint8_t* dst = out + sg.dst;
    uint64_t len = sg.len;
    // byte-granular head to 16-byte destination alignment, then 16-byte body when the source
    // happens to be co-aligned, else dword/bytes.
    const uint32_t t = threadIdx.x;
    const uint64_t head = min<uint64_t>(len, (16 - ((uintptr_t)dst & 15)) & 15);
    if (t < head) dst[t] = src[t];
    src += head; dst += head; len -= head;
    if ((((uintptr_t)src) & 15) == 0) {
        const uint64_t nv = len / 16;
        const uint4* s4 = reinterpret_cast<const uint4*>(src);
        uint4* d4 = reinterpret_cast<uint4*>(dst);
        for (uint64_t i = t; i < nv; i += 256) d4[i] = s4[i];
        const uint64_t done = nv * 16;
        for (uint64_t i = done + t; i < len; i += 256) dst[i] = src[i];
    } else {
        const uint64_t nv = len / 16;
        uint4* d4 = reinterpret_cast<uint4*>(dst);
        for (uint64_t i = t; i < nv; i += 256) {
            uint4 v;
            __builtin_memcpy(&v, src + i * 16, 16);
            d4[i] = v;
        }
        const uint64_t done = nv * 16;
        for (uint64_t i = done + t; i < len; i += 256) dst[i] = src[i];
    }
}

inline uint32_t grid_for(uint64_t n, uint32_t threads, uint32_t cap)
{
    uint64_t g = (n + threads - 1) / threads;
    if (g > cap) g = cap;
    if (g == 0) g = 1;
    return (uint32_t)g;
}

} // namespace

// ==========================================================================================
// extern "C" launch wrappers (internal ABI between the plain-C host layer and the kernels)
// ==========================================================================================
extern "C" {

int xzk_sort_temp_bytes(uint32_t n, uint32_t end_bit, uint64_t* bytes)
{
    size_t sz = 0;
    rocprim::double_buffer<uint32_t> k(nullptr, nullptr), v(nullptr, nullptr);
    hipError_t e = rocprim::radix_sort_pairs(nullptr, sz, k, v, (size_t)n, 0u, end_bit, (hipStream_t)0);
    *bytes = sz;
    return (int)e;
}

// temporary storage the suffix-order build needs (largest of its three primitives)
int xzk_sa_temp_bytes(uint32_t n, uint64_t* bytes)
{
    size_t best = 0, sz = 0;
    {
        rocprim::double_buffer<uint64_t> k(nullptr, nullptr);
        rocprim::double_buffer<uint32_t> v(nullptr, nullptr);
        hipError_t e = rocprim::radix_sort_pairs(nullptr, sz, k, v, (size_t)n, 0u, 64u, (hipStream_t)0);
        if (e != hipSuccess) return (int)e;
        best = sz > best ? sz : best;
    }
    {
        rocprim::double_buffer<uint32_t> k(nullptr, nullptr);
        rocprim::double_buffer<uint64_t> v(nullptr, nullptr);
        hipError_t e = rocprim::radix_sort_pairs(nullptr, sz, k, v, (size_t)n, 0u, 32u, (hipStream_t)0);
        if (e != hipSuccess) return (int)e;
        best = sz > best ? sz : best;
    }
    {
        hipError_t e = rocprim::inclusive_scan(nullptr, sz, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n,
                rocprim::maximum<uint32_t>(), (hipStream_t)0);
        if (e != hipSuccess) return (int)e;
        best = sz > best ? sz : best;
    }
    *bytes = best;
    return 0;
}

// Builds the match-finder structure of a batch.
//   exact finder (sa == NULL):   rank / sorted_pos (main chain), prev2, prev3
//   suffix-neighbourhood finder: prev2, prev4, the suffix order sa / sa_rank and its by-products
//                                rp8 / rp16: per position (rank of the round, distance to the nearest earlier
//                                position with the same 8 / 16 bytes)
// keys_a/keys_b/vals_a/vals_b: n u32 each; key64_a/key64_b: n u64 each (sa != NULL only).
int xzk_build_chains(const uint8_t* d_in, uint32_t n, uint32_t block_size, uint32_t nblocks,
        uint32_t hash_bytes, uint32_t hash_mask, uint32_t hash_bits, uint32_t sa_depth,
        uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b,
        void* sort_tmp, uint64_t sort_tmp_bytes,
        uint32_t* rank, uint32_t* sorted_pos, uint32_t* prev2, uint32_t* prev3,
        uint32_t* prev4, uint64_t* rp8, uint64_t* rp16, uint64_t* key64_a, uint64_t* key64_b,
        uint32_t* sa, uint32_t* sa_rank, uint32_t* prev24, uint32_t* prev32, void* stream_)
{
    hipStream_t st = (hipStream_t)stream_;
    const uint32_t g = grid_for(n, 256, 256 * 16);
    uint32_t bb = 0;
    while ((1u << bb) < nblocks + 1) ++bb;
    size_t tb = sort_tmp_bytes;
    const uint32_t which_list[3] = { 2u, 3u, 0u };
    // hash2 heads without a sort when the segment tables fit the scratch (always, except for tiny Blocks)
    const uint32_t h2_spb = (block_size + H2_SEG - 1) / H2_SEG;
    const bool h2_direct = hash_bytes >= 2 && nblocks != 0 && (uint64_t)nblocks * h2_spb * 1024ull <= (uint64_t)n;
    if (h2_direct) {
        const uint32_t nseg = nblocks * h2_spb;
        hipLaunchKernelGGL(k_h2_last, dim3(nseg), dim3(64), 0, st, d_in, n, block_size, h2_spb, hash_bytes, keys_a);
        hipLaunchKernelGGL(k_h2_scan, dim3(nblocks * 4), dim3(256), 0, st, keys_a, h2_spb);
        hipLaunchKernelGGL(k_h2_prev, dim3(nseg), dim3(64), 0, st, d_in, n, block_size, h2_spb, hash_bytes, keys_a, prev2);
    }
    for (int w = 0; w < 3; ++w) {
        const uint32_t which = which_list[w];
        if (which == 2 && h2_direct) continue;
        if (which == 3 && (hash_bytes != 4 || sa != nullptr)) continue;      // the suffix-neighbourhood finder has no hash3 head
        const uint32_t kbits = which == 2 ? 10u : (which == 3 ? 16u : hash_bits);
        hipLaunchKernelGGL(k_hash_keys, dim3(g), dim3(256), 0, st, d_in, n, block_size, nblocks, hash_bytes,
                hash_mask, hash_bits, which, keys_a, vals_a);
        rocprim::double_buffer<uint32_t> kb(keys_a, keys_b);
        rocprim::double_buffer<uint32_t> vb(vals_a, vals_b);
        size_t need = 0;
        hipError_t e = rocprim::radix_sort_pairs(nullptr, need, kb, vb, (size_t)n, 0u, kbits + bb, st);
        if (e != hipSuccess) return (int)e;
        if (need > tb) return (int)hipErrorOutOfMemory;
        e = rocprim::radix_sort_pairs(sort_tmp, tb, kb, vb, (size_t)n, 0u, kbits + bb, st);
        if (e != hipSuccess) return (int)e;
        uint32_t* const target = which == 2 ? prev2 : which == 3 ? prev3 : sa != nullptr ? prev4 : nullptr;
        if (target != nullptr) {
            // distances in sorted order, then back to position order by a sort on the position
            hipLaunchKernelGGL(k_link_prev_seq, dim3(g), dim3(256), 0, st, kb.current(), vb.current(), n, target);
            e = invert_perm<uint32_t>(vb.current(), vb.alternate(), target, kb.current(), n, target, nullptr, sort_tmp, tb, st);
            if (e != hipSuccess) return (int)e;
        } else {
            hipLaunchKernelGGL(k_link_main, dim3(g), dim3(256), 0, st, kb.current(), vb.current(), n, sorted_pos, rank);
        }
    }
    if (sa == nullptr) return (int)hipGetLastError();

    // ---- suffix order ----
    if (prev24 != nullptr && hipMemsetAsync(prev24, 0, (size_t)n * 4, st) != hipSuccess) return (int)hipErrorUnknown;
    if (prev32 != nullptr && hipMemsetAsync(prev32, 0, (size_t)n * 4, st) != hipSuccess) return (int)hipErrorUnknown;
    uint32_t* grp = keys_a;                       // group-start scan buffer (the hash sorts above are done with it)
    hipError_t e;
    size_t need = 0;
    // round 0: chunk sort
    hipLaunchKernelGGL(k_sa_chunk_keys, dim3(g), dim3(256), 0, st, d_in, n, block_size, key64_a, vals_a);
    uint32_t* pos;
    uint32_t* pos_alt;
    if (nblocks > 1 && nblocks <= 256) {
        // Few, large Blocks (the normal case): one sort per Block.  The positions start out in Block order, so a
        // sort that never mixes Blocks needs no sort by Block number afterwards, and no rocprim call exceeds the
        // 2^30 elements above which it splits every pass into two launches.
        int in_alt = -1;
        for (uint32_t b = 0; b < nblocks; ++b) {
            const size_t off = (size_t)b * block_size;
            if (off >= n) break;
            const size_t cnt = (size_t)n - off < block_size ? (size_t)n - off : block_size;
            rocprim::double_buffer<uint64_t> k(key64_a + off, key64_b + off);
            rocprim::double_buffer<uint32_t> v(vals_a + off, vals_b + off);
            e = rocprim::radix_sort_pairs(nullptr, need, k, v, cnt, 0u, 64u, st);
            if (e != hipSuccess) return (int)e;
            if (need > tb) return (int)hipErrorOutOfMemory;
            e = rocprim::radix_sort_pairs(sort_tmp, tb, k, v, cnt, 0u, 64u, st);
            if (e != hipSuccess) return (int)e;
            const int alt = k.current() == key64_b + off;
            if (in_alt < 0) in_alt = alt;
            else if (alt != in_alt) {
                // a Block of another size class took another route through rocprim: bring it to the common side
                e = hipMemcpyAsync((in_alt ? key64_b : key64_a) + off, k.current(), cnt * 8, hipMemcpyDeviceToDevice, st);
                if (e == hipSuccess)
                    e = hipMemcpyAsync((in_alt ? vals_b : vals_a) + off, v.current(), cnt * 4, hipMemcpyDeviceToDevice, st);
                if (e != hipSuccess) return (int)e;
            }
        }
        pos = in_alt ? vals_b : vals_a;
        pos_alt = in_alt ? vals_a : vals_b;
        hipLaunchKernelGGL(k_sa_flags64, dim3(g), dim3(256), 0, st, in_alt ? key64_b : key64_a, n, block_size, grp);
        e = rocprim::inclusive_scan(nullptr, need, grp, grp, (size_t)n, rocprim::maximum<uint32_t>(), st);
        if (e != hipSuccess) return (int)e;
        if (need > tb) return (int)hipErrorOutOfMemory;
        e = rocprim::inclusive_scan(sort_tmp, tb, grp, grp, (size_t)n, rocprim::maximum<uint32_t>(), st);
        if (e != hipSuccess) return (int)e;
    } else {
        rocprim::double_buffer<uint64_t> k64(key64_a, key64_b);
        rocprim::double_buffer<uint32_t> pv(vals_a, vals_b);
        e = rocprim::radix_sort_pairs(nullptr, need, k64, pv, (size_t)n, 0u, 64u, st);
        if (e != hipSuccess) return (int)e;
        if (need > tb) return (int)hipErrorOutOfMemory;
        e = rocprim::radix_sort_pairs(sort_tmp, tb, k64, pv, (size_t)n, 0u, 64u, st);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k_sa_flags64, dim3(g), dim3(256), 0, st, k64.current(), n, 0u, grp);
        e = rocprim::inclusive_scan(nullptr, need, grp, grp, (size_t)n, rocprim::maximum<uint32_t>(), st);
        if (e != hipSuccess) return (int)e;
        if (need > tb) return (int)hipErrorOutOfMemory;
        e = rocprim::inclusive_scan(sort_tmp, tb, grp, grp, (size_t)n, rocprim::maximum<uint32_t>(), st);
        if (e != hipSuccess) return (int)e;
        pos = pv.current();
        pos_alt = pv.alternate();
        if (nblocks > 1) {
            // many small Blocks: one sort of everything, then a stable sort by Block number; values = (position,
            // chunk group), the chunk keys are dead now
            uint32_t* bk_a = keys_b;
            uint32_t* bk_b = pos_alt;
            uint64_t* bv_a = k64.alternate();
            uint64_t* bv_b = k64.current();
            hipLaunchKernelGGL(k_sa_block_keys, dim3(g), dim3(256), 0, st, pos, grp, n, block_size, bk_a, bv_a);
            rocprim::double_buffer<uint32_t> bk(bk_a, bk_b);
            rocprim::double_buffer<uint64_t> bv(bv_a, bv_b);
            e = rocprim::radix_sort_pairs(nullptr, need, bk, bv, (size_t)n, 0u, bb, st);
            if (e != hipSuccess) return (int)e;
            if (need > tb) return (int)hipErrorOutOfMemory;
            e = rocprim::radix_sort_pairs(sort_tmp, tb, bk, bv, (size_t)n, 0u, bb, st);
            if (e != hipSuccess) return (int)e;
            // positions go back to `pos` (vals buffer that held them before; its content is dead)
            hipLaunchKernelGGL(k_sa_block_unpack, dim3(g), dim3(256), 0, st, bk.current(), bv.current(), n, pos, grp);
            // pos_alt may have been used as a key buffer: both vals buffers are free for reuse below except `pos`
            e = rocprim::inclusive_scan(sort_tmp, tb, grp, grp, (size_t)n, rocprim::maximum<uint32_t>(), st);
            if (e != hipSuccess) return (int)e;
        }
    }
    // doubling rounds
    uint32_t sbits = 1, fbits = 1;                    // bits of a Block-relative rank (<= block_size), of a rank (<= n)
    while (sbits < 32 && (1ull << sbits) <= (uint64_t)min(block_size, n)) ++sbits;
    while (fbits < 32 && (1ull << fbits) <= (uint64_t)n) ++fbits;
    if (sa_depth < 32) sa_depth = 32;
    // round h = 8: every slot takes part (text: 71 % of the positions still share their 8 bytes with another one)
    {
        uint32_t* const rk32 = reinterpret_cast<uint32_t*>(rp8);     // rank at rk32[0..n), left-neighbour distance at rk32[n..2n)
        hipLaunchKernelGGL(k_sa_rank_seq, dim3(g), dim3(256), 0, st, pos, grp, n, reinterpret_cast<uint2*>(key64_a));
        e = invert_perm<uint64_t>(pos, pos_alt, key64_a, key64_b, n, rk32, rk32 + n, sort_tmp, tb, st);
        if (e != hipSuccess) return (int)e;
        // keys in position order (values = iota): both `pos` buffers are free again
        hipLaunchKernelGGL(k_sa_pair_keys_pos, dim3(g), dim3(256), 0, st, rk32, n, block_size, 8u, sbits, key64_a, pos);
        rocprim::double_buffer<uint64_t> kk(key64_a, key64_b);
        rocprim::double_buffer<uint32_t> vv(pos, pos_alt);
        e = rocprim::radix_sort_pairs(sort_tmp, tb, kk, vv, (size_t)n, 0u, sbits + fbits, st);
        if (e != hipSuccess) return (int)e;
        pos = vv.current();
        pos_alt = vv.alternate();
        hipLaunchKernelGGL(k_sa_flags64, dim3(g), dim3(256), 0, st, kk.current(), n, 0u, grp);
        e = rocprim::inclusive_scan(sort_tmp, tb, grp, grp, (size_t)n, rocprim::maximum<uint32_t>(), st);
        if (e != hipSuccess) return (int)e;
    }
    // rounds h = 16, 32, ...: by-position rank in rkpos (kept up to date by the compact rounds), slots in pos, groups in grp
    uint32_t* const rkpos = reinterpret_cast<uint32_t*>(rp16);
    uint32_t* const idx = keys_b;                     // free since round 0
    uint32_t* const d_count = reinterpret_cast<uint32_t*>(key64_b);
    bool rank_valid = false;
    const char* const cenv = getenv("XZAMD_SA_COMPACT");          // 0: every round orders all slots (measurement knob)
    const bool compact_on = !(cenv && *cenv == '0');
    for (uint32_t h = 16; 2 * h <= sa_depth; h *= 2) {
        const bool more = 4 * h <= sa_depth;
        if (h == 16) {
            // (rank, distance to the left neighbour inside the 16-byte group) of every slot, brought to position order;
            // the slot order itself stays (the inversion works on a copy of it)
            hipLaunchKernelGGL(k_sa_rank_seq, dim3(g), dim3(256), 0, st, pos, grp, n, reinterpret_cast<uint2*>(key64_a));
            e = hipMemcpyAsync(idx, pos, (size_t)n * 4, hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) return (int)e;
            e = invert_perm<uint64_t>(idx, sa, key64_a, key64_b, n, rkpos, rkpos + n, sort_tmp, tb, st);
            if (e != hipSuccess) return (int)e;
            rank_valid = true;
        } else if (!rank_valid) {
            uint32_t* const ra = reinterpret_cast<uint32_t*>(key64_a);
            uint32_t* const rb = reinterpret_cast<uint32_t*>(key64_b);
            hipLaunchKernelGGL(k_sa_rank_only_seq, dim3(g), dim3(256), 0, st, grp, n, ra);
            e = hipMemcpyAsync(idx, pos, (size_t)n * 4, hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) return (int)e;
            e = invert_perm<uint32_t>(idx, sa, ra, rb, n, rkpos, nullptr, sort_tmp, tb, st);
            if (e != hipSuccess) return (int)e;
            rank_valid = true;
        }
        uint32_t m = n;
        if (compact_on && n >= 2) {
            // how many slots are still undecided?  (One word back to the host: the sort below is sized by it.)
            hipLaunchKernelGGL(k_sa_unres, dim3(g), dim3(256), 0, st, grp, n, idx);
            e = rocprim::exclusive_scan(nullptr, need, idx, idx, 0u, (size_t)n, rocprim::plus<uint32_t>(), st);
            if (e != hipSuccess) return (int)e;
            if (need > tb) return (int)hipErrorOutOfMemory;
            e = rocprim::exclusive_scan(sort_tmp, tb, idx, idx, 0u, (size_t)n, rocprim::plus<uint32_t>(), st);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(k_sa_count, dim3(1), dim3(1), 0, st, idx, grp, n, d_count);
            e = hipMemcpyAsync(&m, d_count, 4, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) return (int)e;
            if (m == 0) break;                        // every suffix is distinguished: the order is final
        }
        if (h == 16 && prev24 != nullptr && compact_on && n >= 2) {
            // prev24 (the nearest earlier position with the same 24 bytes): the members of the 16-byte groups once more,
            // ordered by (group, rank after 8 bytes of p + 16) -- an extra sort of the m undecided slots only; the key,
            // value and slot buffers of the compact round below are free until it runs
            uint32_t* const cslot = pos_alt;
            hipLaunchKernelGGL(k_sa_compact, dim3(g), dim3(256), 0, st, pos, grp, idx, reinterpret_cast<const uint32_t*>(rp8), n,
                    block_size, 16u, sbits, key64_a, sa, cslot);
            rocprim::double_buffer<uint64_t> kk(key64_a, key64_b);
            rocprim::double_buffer<uint32_t> vv(sa, sa_rank);
            e = rocprim::radix_sort_pairs(nullptr, need, kk, vv, (size_t)m, 0u, sbits + fbits, st);
            if (e != hipSuccess) return (int)e;
            if (need > tb) return (int)hipErrorOutOfMemory;
            e = rocprim::radix_sort_pairs(sort_tmp, tb, kk, vv, (size_t)m, 0u, sbits + fbits, st);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(k_sa_prev_scatter, dim3(grid_for(m, 256, 256 * 16)), dim3(256), 0, st, kk.current(), vv.current(), m, prev24);
        }
        if (compact_on && n >= 2 && (uint64_t)m * 10 <= (uint64_t)n * 6) {
            uint32_t* const cslot = pos_alt;
            const uint32_t gm = grid_for(m, 256, 256 * 16);
            hipLaunchKernelGGL(k_sa_compact, dim3(g), dim3(256), 0, st, pos, grp, idx, rkpos, n, block_size, h, sbits,
                    key64_a, sa, cslot);
            rocprim::double_buffer<uint64_t> kk(key64_a, key64_b);
            rocprim::double_buffer<uint32_t> vv(sa, sa_rank);
            e = rocprim::radix_sort_pairs(sort_tmp, tb, kk, vv, (size_t)m, 0u, sbits + fbits, st);
            if (e != hipSuccess) return (int)e;
            if (h == 16 && prev32 != nullptr)         // by-product: the 32-byte groups' left neighbours
                hipLaunchKernelGGL(k_sa_prev_scatter, dim3(gm), dim3(256), 0, st, kk.current(), vv.current(), m, prev32);
            hipLaunchKernelGGL(k_sa_newgrp, dim3(gm), dim3(256), 0, st, kk.current(), cslot, m, idx);
            e = rocprim::inclusive_scan(sort_tmp, tb, idx, idx, (size_t)m, rocprim::maximum<uint32_t>(), st);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(k_sa_writeback, dim3(gm), dim3(256), 0, st, vv.current(), cslot, idx, m, pos, grp, rkpos);
        } else {
            // most slots are undecided (highly repetitive data): the full round, keys made in position order
            hipLaunchKernelGGL(k_sa_pair_keys_pos, dim3(g), dim3(256), 0, st, rkpos, n, block_size, h, sbits, key64_a, pos);
            rocprim::double_buffer<uint64_t> kk(key64_a, key64_b);
            rocprim::double_buffer<uint32_t> vv(pos, pos_alt);
            e = rocprim::radix_sort_pairs(sort_tmp, tb, kk, vv, (size_t)n, 0u, sbits + fbits, st);
            if (e != hipSuccess) return (int)e;
            pos = vv.current();
            pos_alt = vv.alternate();
            if (h == 16 && prev32 != nullptr)
                hipLaunchKernelGGL(k_sa_prev_scatter, dim3(g), dim3(256), 0, st, kk.current(), pos, n, prev32);
            if (more) {            // another round follows: its ranks need the groups of this order
                hipLaunchKernelGGL(k_sa_flags64, dim3(g), dim3(256), 0, st, kk.current(), n, 0u, grp);
                e = rocprim::inclusive_scan(sort_tmp, tb, grp, grp, (size_t)n, rocprim::maximum<uint32_t>(), st);
                if (e != hipSuccess) return (int)e;
            }
            rank_valid = false;
        }
    }
    // sa = slot order; sa_rank = its inverse (grp = keys_a is dead: scratch of the inversion)
    hipLaunchKernelGGL(k_sa_final_seq, dim3(g), dim3(256), 0, st, pos, n, sa, sa_rank);
    e = invert_perm<uint32_t>(pos, pos_alt, sa_rank, keys_a, n, sa_rank, nullptr, sort_tmp, tb, st);
    if (e != hipSuccess) return (int)e;
    return (int)hipGetLastError();
}

int xzk_find_matches(const xzamd_span_args* a, const uint32_t* sa, const uint32_t* sa_rank, const uint32_t* prev4,
        const uint64_t* rp8, const uint64_t* rp16, const uint32_t* prev24, const uint32_t* prev32,
        uint16_t* mlen, uint32_t* mdist, int part, void* stream_)
{
    hipStream_t st = (hipStream_t)stream_;
    const uint32_t runs = (a->n + FIND_RUN - 1) / FIND_RUN;
    if (runs == 0) return 0;
    if (a->sa_window) {
        if (!sa || !sa_rank || !prev4 || !rp8 || !rp16 || !prev24 || !prev32 || !a->mtop || a->sa_window > SN_WMAX)
            return (int)hipErrorInvalidValue;
        if (part != 0 && a->block_size < XZAMD_SEED_LEN + 2 * FIND_RUN) return (int)hipErrorInvalidValue;
        SnArgs sn;
        sn.in = a->in; sn.sa = sa; sn.sa_rank = sa_rank; sn.prev2 = a->prev2; sn.prev4 = prev4;
        sn.prev8 = reinterpret_cast<const uint32_t*>(rp8) + a->n;     // second array of the round's (rank, distance) pair
        sn.prev16 = reinterpret_cast<const uint32_t*>(rp16) + a->n;
        sn.prev24 = prev24; sn.prev32 = prev32;
        sn.mode = (uint32_t)part;
        const uint32_t nblocks = (a->n + a->block_size - 1) / a->block_size;
        hipLaunchKernelGGL(k_find_sn, dim3(part == 1 ? nblocks * SEED_RUNS : runs), dim3(64), 0, st, *a, sn, mlen, mdist);
    } else {
        hipLaunchKernelGGL(k_find_exact, dim3(runs), dim3(64), 0, st, *a, mlen, mdist);
    }
    return (int)hipGetLastError();
}

int xzk_span_plan(const xzamd_span_args* a, uint32_t nblocks, uint32_t* est, unsigned long long* totals,
        uint32_t* span_tab, uint32_t* span_cnt, uint32_t cost_min, uint32_t bits_min, uint32_t min_len,
        uint32_t* enc_tab, uint32_t* enc_cnt,
        uint32_t* order_bufs, void* sort_tmp, uint64_t sort_tmp_bytes, uint32_t** order_out, void* stream_)
{
    hipStream_t st = (hipStream_t)stream_;
    if (order_out) *order_out = nullptr;
    if (nblocks == 0 || a->n == 0) return 0;
    if (!a->mtop || a->max_spb == 0 || cost_min == 0 || !order_bufs || !order_out) return (int)hipErrorInvalidValue;
    if (a->enc_bits && (!enc_tab || !enc_cnt || a->max_esb == 0 || min_len < XZAMD_SEED_LEN)) return (int)hipErrorInvalidValue;
    const uint32_t cpb = (a->block_size + XZAMD_EST_CHUNK - 1) / XZAMD_EST_CHUNK;
    const uint32_t nslots = nblocks * a->max_spb;
    uint32_t* key_a = order_bufs;
    uint32_t* key_b = order_bufs + nslots;
    uint32_t* val_a = order_bufs + 2ull * nslots;
    uint32_t* val_b = order_bufs + 3ull * nslots;
    hipError_t e = hipMemsetAsync(totals, 0, (size_t)(nblocks + 2) * sizeof(unsigned long long), st);
    if (e == hipSuccess) e = hipMemsetAsync(key_a, 0xFF, (size_t)nslots * 4, st);
    if (e != hipSuccess) return (int)e;
    const uint64_t nch = (uint64_t)nblocks * cpb;
    hipLaunchKernelGGL(k_span_est, dim3((uint32_t)((nch + 255) / 256)), dim3(256), 0, st, *a, nblocks, cpb, est, totals);
    hipLaunchKernelGGL(k_span_cut, dim3(nblocks), dim3(64), 0, st, *a, nblocks, cpb, est, totals, span_tab, span_cnt,
            cost_min, bits_min, min_len, enc_tab, enc_cnt, key_a);
    if (a->part_tab) {
        xzamd_span_args a2 = *a;
        a2.span_tab = span_tab; a2.span_cnt = span_cnt;
        hipLaunchKernelGGL(k_part_ends, dim3((nslots + 255) / 256), dim3(256), 0, st, a2, nblocks, cpb, est, a->part_tab);
    }
    // launch order: span slots by estimated work, heaviest first (a launch then ends with its short spans instead of
    // waiting for a heavy one that happened to start late); the order does not change a byte of the output
    hipLaunchKernelGGL(k_iota, dim3(grid_for(nslots, 256, 4096)), dim3(256), 0, st, val_a, nslots);
    rocprim::double_buffer<uint32_t> kb(key_a, key_b);
    rocprim::double_buffer<uint32_t> vb(val_a, val_b);
    size_t need = 0;
    e = rocprim::radix_sort_pairs(nullptr, need, kb, vb, (size_t)nslots, 0u, 32u, st);
    if (e != hipSuccess) return (int)e;
    if (need > sort_tmp_bytes) return (int)hipErrorOutOfMemory;
    size_t tb = (size_t)sort_tmp_bytes;
    e = rocprim::radix_sort_pairs(sort_tmp, tb, kb, vb, (size_t)nslots, 0u, 32u, st);
    if (e != hipSuccess) return (int)e;
    *order_out = vb.current();
    return (int)hipGetLastError();
}

// waves = 0: one wavefront per span; else a persistent launch of min(waves, nspans) wavefronts that pull
// span numbers from *counter (must be zero at launch).
int xzk_span_encode(const xzamd_span_args* a, uint32_t nspans, uint32_t waves, uint32_t* counter, void* stream_)
{
    hipStream_t st = (hipStream_t)stream_;
    if (nspans == 0) return 0;
    if (!a->span_tab || !a->span_cnt || a->max_spb == 0) return (int)hipErrorInvalidValue;
    const bool persist = waves != 0 && counter != nullptr && waves < nspans;
    const uint32_t grid = persist ? waves : nspans;
    uint32_t* cnt = persist ? counter : nullptr;
    if (a->parser) {
        if ((!a->mlen && !a->list_packed) || !a->mdist) return (int)hipErrorInvalidValue;
        // (the single-phase optimal-parser kernel: explicit spans with the parser -- a test configuration, one window size)
        hipLaunchKernelGGL((k_span_encode_t<2, true>), dim3(grid), dim3(64), 0, st, *a, nspans, cnt);
    } else {
        if (a->sa_window) return (int)hipErrorInvalidValue;      // the fast parser runs on the exact finder only
        hipLaunchKernelGGL((k_span_encode_t<0, false>), dim3(grid), dim3(64), 0, st, *a, nspans, cnt);
    }
    return (int)hipGetLastError();
}

// Two-phase mode, phase 1 of the pipeline: the parse pieces.  phase 0 = the seed piece of every Block, 1 = the others.
int xzk_parse_pieces(const xzamd_span_args* a, uint32_t nblocks, int phase, uint32_t waves, uint32_t* counter, void* stream_)
{
    hipStream_t st = (hipStream_t)stream_;
    if (nblocks == 0) return 0;
    if ((phase != 0 && (!a->span_tab || !a->span_cnt)) || a->max_spb == 0 || !a->sym_len || !a->sym_dist || !a->prior || !a->lit
            || (!a->mlen && !a->list_packed) || !a->mdist || !a->parser)
        return (int)hipErrorInvalidValue;
    const uint32_t nitems = phase == 0 ? nblocks : nblocks * a->max_spb;
    const bool persist = phase != 0 && waves != 0 && counter != nullptr && waves < nitems;
    const uint32_t grid = persist ? waves : nitems;
    uint32_t* cnt = persist ? counter : nullptr;
    // (specialised for the list format: a launch-time constant the compiler cannot see)
    if (a->list_packed) hipLaunchKernelGGL((k_parse_pieces<WMAX_STD, true>), dim3(grid), dim3(64), 0, st, *a, nitems, phase, cnt);
    else hipLaunchKernelGGL((k_parse_pieces<WMAX_STD, false>), dim3(grid), dim3(64), 0, st, *a, nitems, phase, cnt);
    return (int)hipGetLastError();
}

// Two-phase mode, phase 2: the model pass (one wavefront per encode-span slot, the whole model in LDS) and the range
// coder (one lane per chunk slot).
// XZAMD_WALK_LITG=0 / 1 (measurement knob): the literal coders of the model walks in LDS / in global memory
static bool walk_litg()
{
    const char* e = getenv("XZAMD_WALK_LITG");
    return e ? *e == '1' : XZAMD_WALK_LITG_DEFAULT;
}
static uint32_t walk_lds(const xzamd_span_args* a)
{
    const uint32_t n = walk_litg() ? (uint32_t)P_LITERAL : P_LITERAL + (0x300u << (a->lc + a->lp));
    return ((((n + 1) / 2) * 4 + 128) + 15) & ~15u;
}

// the bounds walk and the chain: what every encode span starts from (sub: over the records of parse iteration 1)
static int carried_starts(const xzamd_span_args* a, uint32_t nblocks, bool sub, hipStream_t st)
{
    if (!a->cb_bnd || !a->cb_log || !a->cb_hdr || !a->cb_start || !a->cb_carry || !a->pinfo || !a->span_tab || !a->span_cnt
            || a->model_slots_pad == 0)
        return (int)hipErrorInvalidValue;
    const uint32_t nslots = nblocks * a->max_esb;
    const uint32_t nprob = P_LITERAL + (0x300u << (a->lc + a->lp));
    hipError_t e = hipMemsetAsync(a->cb_log, 0, (size_t)nslots * a->model_slots_pad * XZAMD_LOG_WORDS * 4, st);
    if (e != hipSuccess) return (int)e;
    const bool litg = walk_litg();
    const uint32_t lds = ((litg ? (uint32_t)P_LITERAL : nprob) * 4 + 15) & ~15u;
    if (sub) {
        if (litg) hipLaunchKernelGGL((k_model_walk<2, true>), dim3(nslots), dim3(64), lds, st, *a, nslots);
        else hipLaunchKernelGGL((k_model_walk<2, false>), dim3(nslots), dim3(64), lds, st, *a, nslots);
    } else {
        if (litg) hipLaunchKernelGGL((k_model_walk<1, true>), dim3(nslots), dim3(64), lds, st, *a, nslots);
        else hipLaunchKernelGGL((k_model_walk<1, false>), dim3(nslots), dim3(64), lds, st, *a, nslots);
    }
    hipLaunchKernelGGL(k_model_chain, dim3((a->model_slots_pad + 255) / 256, nblocks), dim3(256), 0, st, *a, nblocks);
    return (int)hipGetLastError();
}

int xzk_encode_syms(const xzamd_span_args* a, uint32_t nblocks, void* stream_)
{
    hipStream_t st = (hipStream_t)stream_;
    if (nblocks == 0) return 0;
    if (!a->enc_tab || !a->enc_cnt || a->max_esb == 0 || !a->sym_len || !a->sym_dist || !a->scratch || !a->tok || !a->chunks)
        return (int)hipErrorInvalidValue;
    const uint32_t nslots = nblocks * a->max_esb;
    const uint32_t nch = XZAMD_CHUNK_SLOTS(a->n, nslots);
    int r = carried_starts(a, nblocks, false, st);
    if (r) return r;
    hipError_t e = hipMemsetAsync(a->chunks, 0, (size_t)nch * sizeof(xzamd_chunk), st);
    if (e != hipSuccess) return (int)e;
    const uint32_t lds = walk_lds(a);
    if (walk_litg()) hipLaunchKernelGGL((k_model_walk<0, true>), dim3(nslots), dim3(64), lds, st, *a, nslots);
    else hipLaunchKernelGGL((k_model_walk<0, false>), dim3(nslots), dim3(64), lds, st, *a, nslots);
    hipLaunchKernelGGL(k_rc_chunks, dim3((nch + 63) / 64), dim3(64), 0, st, *a, nch);
    return (int)hipGetLastError();
}

int xzk_model_snapshots(const xzamd_span_args* a, uint32_t nblocks, void* stream_)
{
    hipStream_t st = (hipStream_t)stream_;
    if (nblocks == 0) return 0;
    if (!a->enc_tab || !a->enc_cnt || a->max_esb == 0 || !a->sym_len || !a->sym_dist || !a->prior || !a->lit || !a->snap_sr)
        return (int)hipErrorInvalidValue;
    const uint32_t nslots = nblocks * a->max_esb;
    int r = carried_starts(a, nblocks, true, st);
    if (r) return r;
    const uint32_t lds = walk_lds(a);
    if (walk_litg()) hipLaunchKernelGGL((k_model_walk<3, true>), dim3(nslots), dim3(64), lds, st, *a, nslots);
    else hipLaunchKernelGGL((k_model_walk<3, false>), dim3(nslots), dim3(64), lds, st, *a, nslots);
    return (int)hipGetLastError();
}

// wavefronts of the span kernel one CU holds at once (what the launch geometry of the span plan is sized for)
int xzk_span_occupancy(int parser, uint32_t nice_len, int* waves_per_cu)
{
    int nb = 0;
    hipError_t e = !parser ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_span_encode_t<0, false>, 64, 0)
            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_span_encode_t<2, true>, 64, 0);
    *waves_per_cu = nb;
    return (int)e;
}

int xzk_x86_bcj(const uint8_t* d_in, uint8_t* d_out, uint32_t n, uint32_t block_size, uint32_t nblocks, void* stream_)
{
    hipStream_t st = (hipStream_t)stream_;
    int e = (int)hipMemcpyAsync(d_out, d_in, n, hipMemcpyDeviceToDevice, st);
    if (e) return e;
    const uint32_t cpb = (block_size + BCJ_CHUNK - 1) / BCJ_CHUNK;
    const uint64_t nch = (uint64_t)cpb * nblocks;
    if (nch == 0 || nch > 0xFFFFFFFFull) return nch ? (int)hipErrorInvalidValue : 0;
    hipLaunchKernelGGL(k_x86_bcj, dim3((uint32_t)((nch + 255) / 256)), dim3(256), 0, st, d_in, d_out, n, block_size, cpb,
            (uint32_t)nch);
    return (int)hipGetLastError();
}

// prefilter kind: 0x0A = ARM64 BCJ, 0x0B = RISC-V BCJ, 5 / 6 / 7 / 8 / 9 = PowerPC / IA-64 / ARM / ARM-Thumb / SPARC BCJ, 3 = delta (dist 1..256):
// d_out = filtered copy of d_in
int xzk_prefilter(const uint8_t* d_in, uint8_t* d_out, uint32_t n, uint32_t block_size, uint32_t nblocks, uint32_t kind, uint32_t dist,
        void* stream_)
{
    hipStream_t st = (hipStream_t)stream_;
    if (n == 0) return 0;
    if (kind == 0x0A) {
        int e = (int)hipMemcpyAsync(d_out, d_in, n, hipMemcpyDeviceToDevice, st);
        if (e) return e;
        hipLaunchKernelGGL(k_arm64_bcj, dim3(grid_for((uint64_t)n / 4 + 1, 256, 65536)), dim3(256), 0, st, d_in, d_out, n, block_size, nblocks);
    } else if (kind == 0x0B) {
        int e = (int)hipMemcpyAsync(d_out, d_in, n, hipMemcpyDeviceToDevice, st);
        if (e) return e;
        const uint32_t cpb = (block_size + BCJ_CHUNK - 1) / BCJ_CHUNK;
        const uint64_t nch = (uint64_t)cpb * nblocks;
        if (nch > 0xFFFFFFFFull) return (int)hipErrorInvalidValue;
        hipLaunchKernelGGL(k_riscv_bcj, dim3((uint32_t)((nch + 255) / 256)), dim3(256), 0, st, d_in, d_out, n, block_size, cpb,
                (uint32_t)nch);
    } else if (kind >= 5 && kind <= 9) {
        int e = (int)hipMemcpyAsync(d_out, d_in, n, hipMemcpyDeviceToDevice, st);
        if (e) return e;
        hipLaunchKernelGGL(k_bcj_simple, dim3(grid_for((uint64_t)n / 2 + 1, 256, 65536)), dim3(256), 0, st, d_in, d_out, n, block_size,
                nblocks, kind);
    } else if (kind == 3) {
        hipLaunchKernelGGL(k_delta, dim3(grid_for(n, 256, 65536)), dim3(256), 0, st, d_in, d_out, n, block_size, dist);
    } else {
        return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}

int xzk_sha256_blocks(const uint8_t* d_in, uint32_t n, uint32_t block_size, uint32_t nblocks, uint8_t* d_out32, void* stream_)
{
    if (nblocks == 0) return 0;
    hipLaunchKernelGGL(k_sha256_blocks, dim3((nblocks + 63) / 64), dim3(64), 0, (hipStream_t)stream_, d_in, n, block_size, nblocks, d_out32);
    return (int)hipGetLastError();
}

int xzk_crc_blocks(const uint8_t* d_in, uint32_t n, uint32_t block_size, uint32_t nblocks,
        uint32_t strip, int crc32, uint64_t* d_strip_crc, uint64_t* d_block_crc, void* stream_)
{
    hipStream_t st = (hipStream_t)stream_;
    const uint32_t spb = (block_size + strip - 1) / strip;
    const uint32_t ns = spb * nblocks;
    if (crc32) {
        uint32_t* strips = reinterpret_cast<uint32_t*>(d_strip_crc);
        hipLaunchKernelGGL((k_crc_strips<uint32_t>), dim3((ns + 255) / 256), dim3(256), 0, st, d_in, n, block_size, strip,
                spb, ns, strips);
        hipLaunchKernelGGL((k_crc_fold<uint32_t>), dim3(nblocks), dim3(64), 0, st, strips, n, block_size, strip, spb,
                d_block_crc);
    } else {
        hipLaunchKernelGGL((k_crc_strips<uint64_t>), dim3((ns + 255) / 256), dim3(256), 0, st, d_in, n, block_size, strip,
                spb, ns, d_strip_crc);
        hipLaunchKernelGGL((k_crc_fold<uint64_t>), dim3(nblocks), dim3(64), 0, st, d_strip_crc, n, block_size, strip, spb,
                d_block_crc);
    }
    return (int)hipGetLastError();
}

int xzk_assemble(const xzamd_copy_seg* d_segs, uint32_t nsegs, const uint8_t* d_scratch, const uint8_t* d_lits,
        const uint8_t* d_in, uint8_t* d_out, void* stream_)
{
    hipStream_t st = (hipStream_t)stream_;
    if (nsegs == 0) return 0;
    hipLaunchKernelGGL(k_assemble, dim3(nsegs), dim3(256), 0, st, d_segs, nsegs, d_scratch, d_lits, d_in, d_out);
    return (int)hipGetLastError();
}

// --- thin runtime shims so the plain-C host layer needs no HIP headers ---------------------
int xzk_malloc(void** p, uint64_t bytes) { return (int)hipMalloc(p, bytes); }
int xzk_free(void* p) { return (int)hipFree(p); }
int xzk_host_alloc(void** p, uint64_t bytes) { return (int)hipHostMalloc(p, bytes, hipHostMallocDefault); }
int xzk_host_free(void* p) { return (int)hipHostFree(p); }
int xzk_h2d(void* d, const void* h, uint64_t bytes, void* st) { return (int)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, (hipStream_t)st); }
int xzk_d2h(void* h, const void* d, uint64_t bytes, void* st) { return (int)hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, (hipStream_t)st); }
int xzk_memset(void* d, int v, uint64_t bytes, void* st) { return (int)hipMemsetAsync(d, v, bytes, (hipStream_t)st); }
int xzk_sync(void* st) { return (int)hipStreamSynchronize((hipStream_t)st); }
int xzk_set_device(int dev) { return (int)hipSetDevice(dev); }
int xzk_get_device(int* dev) { return (int)hipGetDevice(dev); }
int xzk_device_count(int* n) { return (int)hipGetDeviceCount(n); }
int xzk_cu_count(int dev, int* cus) { return (int)hipDeviceGetAttribute(cus, hipDeviceAttributeMultiprocessorCount, dev); }
int xzk_stream_create(void** st) { return (int)hipStreamCreateWithFlags((hipStream_t*)st, hipStreamNonBlocking); }
// lowest-priority stream of the device (work on it only fills slots the caller's stream leaves free)
int xzk_stream_create_low(void** st)
{
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
    return (int)hipStreamCreateWithPriority((hipStream_t*)st, hipStreamNonBlocking, least);
}
int xzk_stream_wait_event(void* st, void* ev) { return (int)hipStreamWaitEvent((hipStream_t)st, (hipEvent_t)ev, 0); }
int xzk_stream_destroy(void* st) { return (int)hipStreamDestroy((hipStream_t)st); }
int xzk_event_create(void** ev) { return (int)hipEventCreate((hipEvent_t*)ev); }
int xzk_event_destroy(void* ev) { return (int)hipEventDestroy((hipEvent_t)ev); }
int xzk_event_record(void* ev, void* st) { return (int)hipEventRecord((hipEvent_t)ev, (hipStream_t)st); }
int xzk_event_elapsed_ms(void* a, void* b, float* ms) { return (int)hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b); }
int xzk_event_query(void* ev) { return (int)hipEventQuery((hipEvent_t)ev); }
const char* xzk_error_string(int e) { return hipGetErrorString((hipError_t)e); }
int xzk_mem_info(uint64_t* free_b, uint64_t* total_b)
{
    size_t f = 0, t = 0;
    hipError_t e = hipMemGetInfo(&f, &t);
    *free_b = f; *total_b = t;
    return (int)e;
}

} // extern "C"
