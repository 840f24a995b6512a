// Micro-benchmark (gfx950): rocPRIM radix_sort_pairs of n (u32 key, i32 value) pairs over `bits` key bits, with the default
// configuration (merge sort up to 2^20 items) and with the merge-sort limit lowered (onesweep), as sort_by_cell uses it.
// build: hipcc --offload-arch=gfx950 -O2 sort_bench.hip -o sort_bench
#include <cstring>
#include <cstdio>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <vector>
#include <random>

template <class Config>
static float run(unsigned *ka, unsigned *kb, int *va, int *vb, size_t n, unsigned bits, hipStream_t st)
{
    size_t tmp = 0;
    rocprim::radix_sort_pairs<Config>(nullptr, tmp, ka, kb, va, vb, n, 0u, bits, st);
    void *d_tmp = nullptr;
    hipMalloc(&d_tmp, tmp + 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int k = 0; k < 3; k++) rocprim::radix_sort_pairs<Config>(d_tmp, tmp, ka, kb, va, vb, n, 0u, bits, st);
    hipEventRecord(e0, st);
    const int reps = 20;
    for (int k = 0; k < reps; k++) rocprim::radix_sort_pairs<Config>(d_tmp, tmp, ka, kb, va, vb, n, 0u, bits, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(d_tmp);
    return 1e3f * ms / reps;
}

int main()
{
    hipStream_t st;
    hipStreamCreate(&st);
    using Low = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 32768>;
    using rocprim::kernel_config;
    constexpr auto kMatch = rocprim::block_radix_rank_algorithm::match;
    // onesweep with smaller blocks (more of them: a million items are 163 blocks of 512 x 12) and other digit widths
    using A = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::radix_sort_onesweep_config<kernel_config<256, 12>, kernel_config<256, 12>, 8, kMatch>, 32768>;
    using B = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::radix_sort_onesweep_config<kernel_config<256, 8>, kernel_config<256, 8>, 8, kMatch>, 32768>;
    using C = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::radix_sort_onesweep_config<kernel_config<256, 6>, kernel_config<256, 6>, 7, kMatch>, 32768>;
    using D = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::radix_sort_onesweep_config<kernel_config<512, 4>, kernel_config<512, 4>, 8, kMatch>, 32768>;
    std::printf("%10s %5s %12s %12s %12s %12s %12s %12s\n", "n", "bits", "default us", "onesweep", "256x12", "256x8", "256x6 r7", "512x4");
    for (size_t n : {100000ul, 200000ul, 500000ul, 1000000ul})
        for (unsigned bits : {16u, 21u, 24u}) {
            std::vector<unsigned> h(n);
            std::mt19937 rng(7);
            for (auto &k : h) k = rng() & ((1u << bits) - 1u);
            unsigned *ka, *kb;
            int *va, *vb;
            hipMalloc(&ka, n * 4), hipMalloc(&kb, n * 4), hipMalloc(&va, n * 4), hipMalloc(&vb, n * 4);
            hipMemcpy(ka, h.data(), n * 4, hipMemcpyHostToDevice);
            const float a = run<rocprim::default_config>(ka, kb, va, vb, n, bits, st);
            const float b = run<Low>(ka, kb, va, vb, n, bits, st);
            const float c = run<A>(ka, kb, va, vb, n, bits, st);
            const float d = run<B>(ka, kb, va, vb, n, bits, st);
            const float e = run<C>(ka, kb, va, vb, n, bits, st);
            const float f = run<D>(ka, kb, va, vb, n, bits, st);
            std::printf("%10zu %5u %12.1f %12.1f %12.1f %12.1f %12.1f %12.1f\n", n, bits, a, b, c, d, e, f);
            hipFree(ka), hipFree(kb), hipFree(va), hipFree(vb);
        }
    return 0;
}
