// Micro-experiment (gfx950): what does a launch cost whose workgroups all leave at once?  4035 workgroups of 256 threads
// that read one flag and return, by static LDS size, register count and number of dependent loads before the return.
// (nn_fast_kernel<..., VERLET> in a steady iteration is such a launch: every workgroup's rows were answered by
// nn_verify_kernel.)
// build: hipcc --offload-arch=gfx950 -O2 idle_launch.hip -o idle_launch
#include <hip/hip_runtime.h>
#include <cstdio>

template <int LDS, int REGS, int CHAIN>
__global__ __launch_bounds__(256) void idle_kernel(const unsigned *flag, const unsigned *chain, float *sink)
{
    __shared__ float s[LDS / 4 + 1];
    unsigned v = flag[0];
    if (CHAIN >= 1) v += chain[(blockIdx.x + v) & 4095];           // a dependent (vector) load
    if (CHAIN >= 2) v += chain[(threadIdx.x + v) & 4095];
    if (CHAIN >= 3) v += chain[(threadIdx.x * 7 + v) & 4095];
    if (v == 0) return;
    // never reached at run time (flag and chain hold zeros); keeps LDS and registers allocated
    float acc[REGS];
    for (int k = 0; k < REGS; k++) acc[k] = (float)(threadIdx.x + k) * (float)v;
    s[threadIdx.x % (LDS / 4 + 1)] = acc[0];
    __syncthreads();
    for (int r = 0; r < 8; r++)
        for (int k = 0; k < REGS; k++) acc[k] = acc[k] * s[(threadIdx.x + k + r) % (LDS / 4 + 1)] + acc[(k + 1) % REGS];
    float t = 0;
    for (int k = 0; k < REGS; k++) t += acc[k];
    sink[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int LDS, int REGS, int CHAIN>
static void run(const char *what, int grid, const unsigned *flag, const unsigned *chain, float *sink)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int k = 0; k < 20; k++) idle_kernel<LDS, REGS, CHAIN><<<grid, 256>>>(flag, chain, sink);
    hipEventRecord(e0);
    for (int k = 0; k < 200; k++) idle_kernel<LDS, REGS, CHAIN><<<grid, 256>>>(flag, chain, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipFuncAttributes a;
    hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&idle_kernel<LDS, REGS, CHAIN>));
    std::printf("%-44s grid %5d  LDS %6d B  VGPR %3d : %7.2f us per launch\n", what, grid, (int)a.sharedSizeBytes, a.numRegs, ms * 1e3f / 200);
}

int main()
{
    unsigned *flag, *chain;
    float *sink;
    hipMalloc(&flag, 4), hipMalloc(&chain, 4096 * 4), hipMalloc(&sink, 4035 * 256 * 4 + 1024);
    hipMemset(flag, 0, 4), hipMemset(chain, 0, 4096 * 4);
    run<64, 8, 0>("no LDS, few registers, one load", 4035, flag, chain, sink);
    run<64, 8, 0>("  same, 1280 workgroups", 1280, flag, chain, sink);
    run<16000, 8, 0>("16 KB LDS", 4035, flag, chain, sink);
    run<31792, 8, 0>("31.8 KB LDS", 4035, flag, chain, sink);
    run<31792, 72, 0>("31.8 KB LDS, ~80 registers", 4035, flag, chain, sink);
    run<31792, 72, 1>("  + one dependent vector load", 4035, flag, chain, sink);
    run<31792, 72, 3>("  + three dependent vector loads", 4035, flag, chain, sink);
    run<64, 72, 3>("no LDS, ~80 registers, three dependent loads", 4035, flag, chain, sink);
    run<64, 120, 0>("no LDS, ~128 registers", 4035, flag, chain, sink);
    return 0;
}
