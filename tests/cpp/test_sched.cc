// The threads of ppcr_batch_run (csrc/ppcr_batch_sched.hpp: sched::run_device_share, the hand-over queues, the first-error
// latch) with a host stand-in for the handles and registrations — no HIP anywhere — so that they run under ThreadSanitizer
// and AddressSanitizer (tests/test_sanitizers.py).  Checked: every pair is prepared, run and retired exactly once, on a
// handle nobody else holds at that moment; no more handles than lanes + 2 per preparing thread; every handle is released
// exactly once; an error in any operation stops the share, abandons what is in flight and is the error reported.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "ppcr_batch_sched.hpp"

using namespace ppcr::sched;

static std::atomic<int> g_failed{0};
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            g_failed++;                                                    \
        }                                                                  \
    } while (0)

struct FakeHandle {
    std::atomic<int> holder{0};  // 0 nobody, 1 a preparing thread, 2 the running side
    int64_t prepared_for = -1;   // (plain: whoever holds the handle owns it)
    int released = 0;
};
struct FakeJob {
    FakeHandle *h;
    int64_t k;
    int steps_left;
    bool abandoned = false;
};

enum FailAt { kNowhere, kAcquire, kPrepare, kStart, kAdvance };

struct FakeOps {
    using Handle = FakeHandle *;
    using Job = FakeJob;
    int64_t mine;
    FailAt fail_at;
    int64_t fail_k;
    std::vector<std::unique_ptr<FakeHandle>> all;  // (calling thread only)
    std::vector<std::atomic<int>> prepared, retired;
    std::vector<long> result;
    std::atomic<int> in_flight{0}, max_in_flight{0}, abandoned{0};
    std::mt19937 rng_run{7};  // (calling thread only)
    FakeOps(int64_t n, FailAt f, int64_t fk) : mine(n), fail_at(f), fail_k(fk), prepared((size_t)n), retired((size_t)n), result((size_t)n, -1)
    {
        for (auto &a : prepared) a.store(0);
        for (auto &a : retired) a.store(0);
    }
    Handle acquire(int *rc, std::string *err)
    {
        if (fail_at == kAcquire && (int64_t)all.size() == fail_k) {
            *rc = -7;
            *err = "acquire failed";
            return nullptr;
        }
        all.emplace_back(new FakeHandle);
        return all.back().get();
    }
    int prepare(Handle h, int64_t k, std::string *err)
    {
        int expect = 0;
        CHECK(h->holder.compare_exchange_strong(expect, 1));  // nobody else holds it
        std::this_thread::sleep_for(std::chrono::microseconds(20 + (k * 37) % 200));
        h->prepared_for = k;
        prepared[(size_t)k]++;
        h->holder.store(0);
        if (fail_at == kPrepare && k == fail_k) {
            *err = "prepare failed at pair " + std::to_string(k);
            return -3;
        }
        return 0;
    }
    std::unique_ptr<Job> start(Handle h, int64_t k, int *rc, std::string *err)
    {
        int expect = 0;
        CHECK(h->holder.compare_exchange_strong(expect, 2));
        CHECK(h->prepared_for == k);
        if (fail_at == kStart && k == fail_k) {
            *rc = -4;
            *err = "start failed";
            h->holder.store(0);
            return nullptr;
        }
        const int now = ++in_flight;
        int seen = max_in_flight.load();
        while (now > seen && !max_in_flight.compare_exchange_weak(seen, now)) {
        }
        *rc = 0;
        return std::unique_ptr<Job>(new FakeJob{h, k, 3 + (int)(rng_run() % 17)});
    }
    int advance(Job &j, int64_t k, bool *progressed, std::string *err)
    {
        CHECK(j.k == k && j.h->holder.load() == 2);
        *progressed = (rng_run() % 3) != 0;
        if (*progressed) j.steps_left--;
        if (fail_at == kAdvance && k == fail_k && j.steps_left < 2) {
            *err = "advance failed";
            return -5;
        }
        return 0;
    }
    bool finished(const Job &j) const { return j.steps_left <= 0; }
    Handle handle_of(Job &j) const { return j.h; }
    void retire(Job &j, int64_t k)
    {
        result[(size_t)k] = 1000 + (long)k;
        retired[(size_t)k]++;
        in_flight--;
        j.h->holder.store(0);
    }
    void abandon(Job &j)
    {
        j.abandoned = true;
        abandoned++;
        j.h->holder.store(0);
    }
    void release(Handle h, bool) { h->released++; }
};

static void one_share(int64_t mine, int lanes, FailAt fail_at, int64_t fail_k)
{
    FakeOps ops(mine, fail_at, fail_k);
    FirstError first;
    run_device_share(ops, mine, lanes, first);
    const int l = (int)std::max<int64_t>(1, std::min<int64_t>(lanes, mine));
    const int n_preparing = (l >= 3 && mine >= 8) ? 2 : 1;
    CHECK((int64_t)ops.all.size() <= std::min<int64_t>(mine, l + 2 * n_preparing));
    for (auto &h : ops.all) CHECK(h->released == 1);
    CHECK(ops.max_in_flight.load() <= l);
    if (fail_at == kNowhere) {
        CHECK(!first.failed());
        for (int64_t k = 0; k < mine; k++) {
            CHECK(ops.prepared[(size_t)k].load() == 1);
            CHECK(ops.retired[(size_t)k].load() == 1);
            CHECK(ops.result[(size_t)k] == 1000 + k);
        }
        CHECK(ops.abandoned.load() == 0);
    } else {
        CHECK(first.failed());
        const int want = fail_at == kAcquire ? -7 : fail_at == kPrepare ? -3 : fail_at == kStart ? -4 : -5;
        CHECK(first.rc == want);
        for (int64_t k = 0; k < mine; k++) {
            CHECK(ops.prepared[(size_t)k].load() <= 1);
            CHECK(ops.retired[(size_t)k].load() <= 1);
        }
    }
}

int main()
{
    const int64_t sizes[] = {0, 1, 2, 5, 7, 8, 40, 64};
    const int lanes[] = {1, 2, 3, 4, 8};
    for (int64_t mine : sizes)
        for (int l : lanes) one_share(mine, l, kNowhere, 0);
    for (int64_t mine : {5, 40})
        for (int l : {1, 4})
            for (FailAt f : {kAcquire, kPrepare, kStart, kAdvance})
                for (int64_t fk : {(int64_t)0, (int64_t)1, mine / 2, mine - 1}) one_share(mine, l, f, f == kAcquire ? std::min<int64_t>(fk, 1) : fk);
    // several devices' shares side by side with one error latch, as ppcr_batch_run runs them: an error on one device stops all
    for (int round = 0; round < 6; round++) {
        FirstError first;
        std::vector<std::unique_ptr<FakeOps>> ops;
        for (int d = 0; d < 4; d++) ops.emplace_back(new FakeOps(24, (round % 2 == 1 && d == 2) ? kAdvance : kNowhere, 11));
        std::vector<std::thread> pool;
        for (int d = 0; d < 4; d++) pool.emplace_back([&, d] { run_device_share(*ops[(size_t)d], 24, 3, first); });
        for (auto &t : pool) t.join();
        CHECK(first.failed() == (round % 2 == 1));
        for (auto &o : ops) {
            for (auto &h : o->all) CHECK(h->released == 1);
            if (round % 2 == 0)
                for (int64_t k = 0; k < 24; k++) CHECK(o->retired[(size_t)k].load() == 1);
        }
    }
    std::printf("sched test: %d failed\n", g_failed.load());
    return g_failed.load() == 0 ? 0 : 1;
}
