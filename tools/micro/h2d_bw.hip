// H2D copy of a 16 MB cloud: pageable hipMemcpy (what ppcr_set_target / ppcr_set_source do today) against a pinned source and
// against staging through pinned chunks (memcpy on the calling thread + hipMemcpyAsync, double-buffered).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t bytes = 16u << 20;
    char *d = nullptr, *pinned = nullptr, *stage = nullptr;
    hipMalloc((void **)&d, bytes);
    hipHostMalloc((void **)&pinned, bytes, hipHostMallocDefault);
    std::vector<char> pageable(bytes, 1);
    memset(pinned, 1, bytes);
    hipStream_t s;
    hipStreamCreate(&s);
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    std::vector<double> a, b;
    for (int r = 0; r < 21; r++) {
        double t = now();
        hipMemcpy(d, pageable.data(), bytes, hipMemcpyHostToDevice);
        a.push_back(now() - t);
        t = now();
        hipMemcpy(d, pinned, bytes, hipMemcpyHostToDevice);
        b.push_back(now() - t);
    }
    printf("pageable hipMemcpy %.3f ms (%.1f GB/s)   pinned hipMemcpy %.3f ms (%.1f GB/s)\n", 1e3 * med(a), bytes / med(a) / 1e9, 1e3 * med(b), bytes / med(b) / 1e9);
    for (size_t chunk : {size_t(1) << 20, size_t(2) << 20, size_t(4) << 20}) {
        hipHostMalloc((void **)&stage, 2 * chunk, hipHostMallocDefault);
        hipEvent_t ev[2];
        hipEventCreateWithFlags(&ev[0], hipEventDisableTiming);
        hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
        std::vector<double> c;
        for (int r = 0; r < 21; r++) {
            const double t = now();
            int k = 0;
            for (size_t off = 0; off < bytes; off += chunk, k ^= 1) {
                if (off >= 2 * chunk) hipEventSynchronize(ev[k]);
                memcpy(stage + k * chunk, pageable.data() + off, chunk);
                hipMemcpyAsync(d + off, stage + k * chunk, chunk, hipMemcpyHostToDevice, s);
                hipEventRecord(ev[k], s);
            }
            hipStreamSynchronize(s);
            c.push_back(now() - t);
        }
        printf("staged through 2 x %zu MB pinned chunks: %.3f ms (%.1f GB/s)\n", chunk >> 20, 1e3 * med(c), bytes / med(c) / 1e9);
        hipHostFree(stage);
    }
    // host memcpy alone
    std::vector<char> dst(bytes);
    std::vector<double> m;
    for (int r = 0; r < 21; r++) {
        const double t = now();
        memcpy(dst.data(), pageable.data(), bytes);
        m.push_back(now() - t);
    }
    printf("host memcpy of 16 MB: %.3f ms (%.1f GB/s)\n", 1e3 * med(m), bytes / med(m) / 1e9);
    return 0;
}
