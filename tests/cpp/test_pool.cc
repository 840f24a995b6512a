// The sub-allocator the handles' device buffers are cut from (csrc/ppcr_pool.hpp), run on the CPU with a host stand-in
// for the driver: randomised alloc / free traffic from several threads, checked for overlap, alignment, coalescing,
// trimming and out-of-memory behaviour.  Built by tests/test_sanitizers.py under ASan + UBSan and under TSan.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "ppcr_pool.hpp"

namespace {

std::atomic<long long> g_driver_bytes{0};
std::atomic<long long> g_driver_limit{1ll << 40};
std::atomic<int> g_driver_calls{0};
std::mutex g_sizes_mu;
std::unordered_map<void *, size_t> g_sizes;

struct FakeDriver {
    static hipError_t malloc(void **p, size_t bytes)
    {
        g_driver_calls++;
        if (g_driver_bytes.load() + (long long)bytes > g_driver_limit.load()) return hipErrorOutOfMemory;
        *p = std::aligned_alloc(4096, (bytes + 4095) / 4096 * 4096);
        if (!*p) return hipErrorOutOfMemory;
        g_driver_bytes += (long long)bytes;
        std::lock_guard<std::mutex> lk(g_sizes_mu);
        g_sizes[*p] = bytes;
        return hipSuccess;
    }
    static void free(void *p)
    {
        std::lock_guard<std::mutex> lk(g_sizes_mu);
        g_driver_bytes -= (long long)g_sizes[p];
        g_sizes.erase(p);
        std::free(p);
    }
    static void forget_error() {}
};
#ifdef POOL_TEST_SMALL_SLABS  // (TSan: resetting the shadow of 64 MB - 1 GB slabs dominates the run; same logic on 1 - 16 MB)
using Pool = ppcr::BasicDevicePool<FakeDriver, (1ull << 20), (16ull << 20)>;
#else
using Pool = ppcr::BasicDevicePool<FakeDriver>;
#endif
constexpr size_t kUnit = Pool::kFirstSlab / 64;  // 1 MB with the library's slabs

int g_failed = 0;
#define CHECK(cond)                                                      \
    do {                                                                 \
        if (!(cond)) {                                                   \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            g_failed++;                                                  \
        }                                                                \
    } while (0)

struct Block {
    unsigned char *p;
    size_t bytes;
    unsigned char tag;
};

// every live block keeps its own byte pattern: an overlap between two blocks shows up as a wrong byte (and an
// out-of-slab block as a sanitizer report)
void churn(Pool &pool, unsigned seed, int rounds, std::atomic<int> *failures)
{
    std::mt19937 rng(seed);
    std::vector<Block> live;
    auto verify = [&](const Block &b) {
        for (size_t k = 0; k < b.bytes; k += 61)
            if (b.p[k] != b.tag) return false;
        return b.bytes == 0 || b.p[b.bytes - 1] == b.tag;
    };
    for (int r = 0; r < rounds; r++) {
        const bool grow = live.size() < 8 || (rng() % 100 < 55 && live.size() < 60);
        if (grow) {
            // the sizes of a handle: many tiny control blocks, some per-point arrays, a few large ones
            const unsigned kind = rng() % 10;
            size_t bytes = kind < 5 ? 8 + rng() % 4096 : kind < 9 ? kUnit / 16 + rng() % (4 * kUnit) : 16 * kUnit + rng() % (48 * kUnit);
            void *p = nullptr;
            if (pool.alloc(bytes, &p) != hipSuccess || !p) {
                (*failures)++;
                continue;
            }
            if ((uintptr_t)p % Pool::kAlign) (*failures)++;
            Block b{static_cast<unsigned char *>(p), bytes, (unsigned char)(1 + rng() % 250)};
            for (size_t k = 0; k < b.bytes; k += 61) b.p[k] = b.tag;  // (sampled: the whole block under TSan costs minutes)
            if (b.bytes) b.p[b.bytes - 1] = b.tag;
            live.push_back(b);
        } else {
            const size_t k = rng() % live.size();
            if (!verify(live[k])) (*failures)++;
            pool.free(live[k].p);
            live[k] = live.back();
            live.pop_back();
        }
    }
    for (auto &b : live) {
        if (!verify(b)) (*failures)++;
        pool.free(b.p);
    }
}

}  // namespace

int main(int argc, char **argv)
{
    const double scale = argc > 1 ? std::atof(argv[1]) : 1.0;  // (TSan runs a shorter version)

    {
        // one thread: everything given back coalesces into whole slabs again; trimming returns them to the driver
        Pool pool;
        std::atomic<int> failures{0};
        churn(pool, 1, (int)(4000 * scale), &failures);
        CHECK(failures.load() == 0);
        const ppcr::PoolStats st = pool.stats();
        CHECK(st.in_use_bytes == 0);
        CHECK(st.reserved_bytes == (uint64_t)g_driver_bytes.load());
        CHECK(st.driver_allocs >= 1 && st.driver_allocs <= 12);  // slabs, not buffers: thousands of blocks were served
        CHECK(st.block_allocs > (uint64_t)(1500 * scale));
        size_t whole = 0;
        CHECK(pool.slab_count() >= 1);
        whole = pool.largest_free_range();
        CHECK(whole >= Pool::kFirstSlab);  // (a slab free from end to end again)
        const size_t released = pool.trim();
        CHECK(released == st.reserved_bytes && g_driver_bytes.load() == 0 && pool.slab_count() == 0);
        // zero-size and tiny requests get distinct, aligned blocks
        void *a = nullptr, *b = nullptr;
        CHECK(pool.alloc(0, &a) == hipSuccess && pool.alloc(1, &b) == hipSuccess && a && b && a != b);
        pool.free(a);
        pool.free(b);
        pool.free(nullptr);
        pool.trim();
    }
    if (scale >= 1.0) {
        // a request larger than any slab of the series gets a slab of its own; geometric growth keeps the driver calls few
        Pool pool;
        void *big = nullptr;
        CHECK(pool.alloc(3ull << 30, &big) == hipSuccess && big);
        CHECK(pool.stats().reserved_bytes >= (3ull << 30));
        pool.free(big);
        CHECK(pool.stats().in_use_bytes == 0);
        // (more than PPCR_POOL_KEEP_MB idle: the free above has already handed it back)
        CHECK(pool.stats().reserved_bytes <= Pool::keep_idle_bytes());
        pool.trim();
        CHECK(g_driver_bytes.load() == 0);
    }
    {
        // a full device: idle slabs are trimmed and the request retried at its exact size; a request that cannot be met
        // fails cleanly and leaves the pool usable
        Pool pool;
        void *a = nullptr, *b = nullptr, *c = nullptr;
        CHECK(pool.alloc(40 * kUnit, &a) == hipSuccess);  // 64 MB slab
        pool.free(a);                                       // ... idle now
        g_driver_limit = (long long)(100 * kUnit);
        CHECK(pool.alloc(90 * kUnit, &b) == hipSuccess && b);  // needs the idle slab's bytes back
        CHECK(pool.alloc(90 * kUnit, &c) == hipErrorOutOfMemory && c == nullptr);
        CHECK(pool.alloc(1024, &c) == hipSuccess);          // (fits the big block's slab tail or fails over to... the limit)
        pool.free(b);
        pool.free(c);
        pool.trim();
        g_driver_limit = 1ll << 40;
        CHECK(g_driver_bytes.load() == 0);
    }
    {
        // several threads on one pool (ppcr_batch_run's preparing threads and the callers' own)
        Pool pool;
        std::atomic<int> failures{0};
        std::vector<std::thread> th;
        for (unsigned t = 0; t < 4; t++) th.emplace_back(churn, std::ref(pool), 100 + t, (int)(1500 * scale), &failures);
        for (auto &x : th) x.join();
        CHECK(failures.load() == 0);
        CHECK(pool.stats().in_use_bytes == 0);
        pool.trim();
        CHECK(g_driver_bytes.load() == 0);
    }
    std::printf("pool test: %d failed\n", g_failed);
    return g_failed ? 1 : 0;
}
