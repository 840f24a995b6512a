// Micro-experiment: does a kernel launched with hipExtAnyOrderLaunch start before the previous kernel of the same stream
// has finished (no barrier bit on its packet) on this GPU?  Kernel A busy-waits ~wait_us, then raises a flag; kernel B
// records at its start whether the flag was already up.  Also times K back-to-back dependent pairs both ways.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <chrono>

__global__ void slow_kernel(unsigned *flag, long long cycles)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void probe_kernel(unsigned *flag, unsigned *seen, int wait)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        seen[0] = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wait) {
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(8);
            seen[1] = 1;
        }
    }
}
__global__ void tiny_kernel(unsigned *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[2] += 1; }

int main()
{
    unsigned *d, h[4];
    hipMalloc(&d, 64);
    hipStream_t s;
    hipStreamCreate(&s);
    for (int any = 0; any < 2; any++) {
        hipMemsetAsync(d, 0, 64, s);
        hipStreamSynchronize(s);
        slow_kernel<<<1, 64, 0, s>>>(d, 100 * 100);  // wall_clock64 ticks at 100 MHz: ~100 us
        if (any) hipExtLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d, d + 1, 1);
        else probe_kernel<<<1, 64, 0, s>>>(d, d + 1, 1);
        hipStreamSynchronize(s);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%s launch: flag seen at the probe's start = %u (0: it started before the slow kernel finished)\n", any ? "any-order" : "ordinary", h[1]);
    }
    // gap between dependent tiny kernels, both ways
    for (int any = 0; any < 2; any++) {
        hipStreamSynchronize(s);
        const int K = 2000;
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < K; k++) {
            if (any) hipExtLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d);
            else tiny_kernel<<<1, 64, 0, s>>>(d);
        }
        hipStreamSynchronize(s);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("%s: %d tiny kernels back to back: %.2f us each\n", any ? "any-order" : "ordinary", K, us / K);
    }
    return 0;
}
