// Micro-experiment (gfx950): does a hipGraph shorten the boundary between two DEPENDENT kernels?  A chain of K tiny kernels
// (each adds one to a word the next one reads) timed three ways with HIP events around the device-side execution: launched
// one by one on a stream (the host far ahead of the device), captured from that stream into a graph, and built as a graph of
// K kernel nodes (every chain enqueued behind a kernel that holds the stream, so that only the device's side shows).  Also a chain that alternates a 256-workgroup kernel with a one-workgroup kernel, as an iteration of the
// registration loop does (association, then fold and solve).
// build: hipcc --offload-arch=gfx950 -O2 graph_chain.hip -o graph_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void tiny_kernel(unsigned *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void wide_kernel(unsigned *p, unsigned *sink)
{
    const unsigned v = p[0];
    if (v == 0xFFFFFFFFu) sink[blockIdx.x] = v;  // (never: keeps the read)
}

__global__ void slow_kernel(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// the chain is enqueued behind a kernel that holds the stream for ~4 ms: the device never waits for the host's launches
static float timed(hipStream_t s, void (*body)(hipStream_t, void *), void *arg)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    slow_kernel<<<1, 64, 0, s>>>(400000);  // wall_clock64 ticks at 100 MHz
    hipEventRecord(e0, s);
    body(s, arg);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0), hipEventDestroy(e1);
    return ms * 1e3f;
}

struct Ctx {
    unsigned *d, *sink;
    int K;
    bool mixed;
    hipGraphExec_t exec;
};
static void launch_chain(hipStream_t s, void *arg)
{
    Ctx *c = static_cast<Ctx *>(arg);
    for (int k = 0; k < c->K; k++) {
        if (c->mixed && (k & 1) == 0) wide_kernel<<<1024, 256, 0, s>>>(c->d, c->sink);
        else tiny_kernel<<<1, 64, 0, s>>>(c->d);
    }
}
static void launch_graph(hipStream_t s, void *arg) { hipGraphLaunch(static_cast<Ctx *>(arg)->exec, s); }

int main()
{
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    Ctx c{};
    hipMalloc(&c.d, 64), hipMalloc(&c.sink, 4096 * 4);
    hipMemset(c.d, 0, 64);
    c.K = 400;
    for (int mixed = 0; mixed < 2; mixed++) {
        c.mixed = mixed != 0;
        // (1) stream launches
        timed(s, launch_chain, &c);
        float best_stream = 1e30f;
        for (int r = 0; r < 5; r++) best_stream = std::min(best_stream, timed(s, launch_chain, &c));
        // (2) the same launches captured into a graph
        hipGraph_t g;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        launch_chain(s, &c);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&c.exec, g, nullptr, nullptr, 0);
        timed(s, launch_graph, &c);
        float best_graph = 1e30f;
        for (int r = 0; r < 5; r++) best_graph = std::min(best_graph, timed(s, launch_graph, &c));
        hipGraphExecDestroy(c.exec);
        hipGraphDestroy(g);
        std::printf("%s chain of %d dependent kernels: stream launches %.2f us per kernel, captured graph %.2f us per kernel\n",
                    mixed ? "wide/tiny" : "tiny", c.K, best_stream / c.K, best_graph / c.K);
    }
    return 0;
}
