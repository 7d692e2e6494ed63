// Micro-benchmark: rocprim onesweep configurations for the key / value widths of the structure build.
// build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/microbench/sort_bench.hip -o gpurun_out/sort_bench
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <vector>

__global__ void fill(uint64_t* k64, uint32_t* k32, uint32_t* v32, uint64_t* v64, uint32_t n)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint64_t x = (uint64_t)i * 0x9E3779B97F4A7C15ull;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        k64[i] = x; k32[i] = (uint32_t)(x >> 7); v32[i] = i; v64[i] = x;
    }
}

template <class Config, class K, class V>
static void run(const char* name, K* ka, K* kb, V* va, V* vb, uint32_t n, unsigned bits, void* tmp, size_t tmp_cap)
{
    rocprim::double_buffer<K> k(ka, kb);
    rocprim::double_buffer<V> v(va, vb);
    size_t need = 0;
    if (rocprim::radix_sort_pairs<Config>(nullptr, need, k, v, (size_t)n, 0u, bits, (hipStream_t)0) != hipSuccess || need > tmp_cap) {
        printf("%-40s temp %zu > %zu or error\n", name, need, tmp_cap);
        return;
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 3; ++it) {
        rocprim::double_buffer<K> k2(ka, kb);
        rocprim::double_buffer<V> v2(va, vb);
        hipEventRecord(e0, 0);
        hipError_t e = rocprim::radix_sort_pairs<Config>(tmp, need, k2, v2, (size_t)n, 0u, bits, (hipStream_t)0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        if (e != hipSuccess) { printf("%-40s error %d\n", name, (int)e); return; }
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const unsigned passes = (bits + 7) / 8;
    const double bytes = 2.0 * (sizeof(K) + sizeof(V)) * n * passes;
    printf("%-40s K%zu V%zu bits %2u: %8.2f ms  (%.2f ms/pass, %.2f TB/s)\n", name, sizeof(K), sizeof(V), bits, best, best / passes,
            bytes / best / 1e9);
    fflush(stdout);
}

using rocprim::kernel_config;
using rocprim::radix_sort_config;
using rocprim::radix_sort_onesweep_config;
using rocprim::default_config;
constexpr auto MATCH = rocprim::block_radix_rank_algorithm::match;
constexpr auto BASIC = rocprim::block_radix_rank_algorithm::basic;
constexpr auto MEMO = rocprim::block_radix_rank_algorithm::basic_memoize;
template <unsigned BS, unsigned IPT, unsigned RB, rocprim::block_radix_rank_algorithm A>
using OS = radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<BS, IPT>, kernel_config<BS, IPT>, RB, A>, 0>;

int main(int argc, char** argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)atoll(argv[1]) : 1434451968u;
    uint64_t *k64a, *k64b, *v64a, *v64b; uint32_t *k32a, *k32b, *v32a, *v32b; void* tmp;
    const size_t tmp_cap = (size_t)2 << 30;
    hipMalloc(&k64a, 8ull * n); hipMalloc(&k64b, 8ull * n); hipMalloc(&v64a, 8ull * n); hipMalloc(&v64b, 8ull * n);
    hipMalloc(&k32a, 4ull * n); hipMalloc(&k32b, 4ull * n); hipMalloc(&v32a, 4ull * n); hipMalloc(&v32b, 4ull * n);
    hipMalloc(&tmp, tmp_cap);
    fill<<<4096, 256>>>(k64a, k32a, v32a, v64a, n);
    hipDeviceSynchronize();
#define ALL(NAME, CFG) \
    run<CFG>(NAME, k32a, k32b, v32a, v32b, n, 32, tmp, tmp_cap); \
    run<CFG>(NAME, k64a, k64b, v32a, v32b, n, 64, tmp, tmp_cap); \
    run<CFG>(NAME, k32a, k32b, v64a, v64b, n, 32, tmp, tmp_cap);
    ALL("default", default_config)
    { using C = OS<1024, 4, 8, MATCH>; ALL("1024x4 r8 match", C) }
    { using C = OS<512, 8, 8, MATCH>; ALL("512x8 r8 match", C) }
    { using C = OS<512, 12, 8, MATCH>; ALL("512x12 r8 match", C) }
    { using C = OS<256, 16, 8, MATCH>; ALL("256x16 r8 match", C) }
    { using C = OS<256, 12, 8, MATCH>; ALL("256x12 r8 match", C) }
    { using C = OS<512, 16, 8, MATCH>; ALL("512x16 r8 match", C) }
    { using C = OS<256, 24, 8, MATCH>; ALL("256x24 r8 match", C) }
    { using C = OS<1024, 12, 8, MATCH>; ALL("1024x12 r8 match", C) }
    { using C = OS<512, 8, 7, MATCH>; ALL("512x8 r7 match", C) }
    return 0;
}
