// The threads of ppcr_batch_run, free of HIP: the hand-over queues, the first-error latch and one device's share of a
// batch as a template over the operations it performs on handles and registrations.  ppcr_hip_batch.inc instantiates it
// with the library's handles (BatchOps there); tests/cpp/test_sched.cc with a host stand-in, under ThreadSanitizer — the
// scheduler's threads themselves, which no GPU-side tool can watch.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace ppcr {
namespace sched {

struct FirstError {
    std::mutex mu;
    int rc = 0;
    std::string text;
    void set(int code, const std::string &msg)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (rc == 0) {
            rc = code;
            text = msg;
        }
    }
    bool failed()
    {
        std::lock_guard<std::mutex> lk(mu);
        return rc != 0;
    }
};

// hand-over point between the two sides a device's share of a batch runs on
template <class T>
struct HandOver {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<T> items;
    bool closed = false;
    void push(const T &v)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            items.push_back(v);
        }
        cv.notify_one();
    }
    void close()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            closed = true;
        }
        cv.notify_all();
    }
    // 1: *out taken; 0: nothing there now (wait = false only); -1: closed and empty
    int pop(T *out, bool wait)
    {
        std::unique_lock<std::mutex> lk(mu);
        if (wait) cv.wait(lk, [&] { return closed || !items.empty(); });
        if (items.empty()) return closed ? -1 : 0;
        *out = items.front();
        items.pop_front();
        return 1;
    }
};

// One device's share of a batch — its `mine` pairs, numbered 0 .. mine - 1 — on TWO kinds of threads.  The preparing
// threads (one, or two from three registrations in flight on) do all the WAITING of a registration — uploads, grid
// build, levels, source sort — on handles nobody is iterating on; the calling thread only enqueues and polls, up to
// `lanes` registrations in flight.  Handles circulate: idle -> (prepare) -> ready -> (run) -> idle.
//
// Ops (all called with a handle no other thread holds at that moment):
//   using Handle = ...; using Job = ...;
//   Handle acquire(int *rc, std::string *err)        a pooled or fresh handle, nullptr on failure        [calling thread]
//   int prepare(Handle, int64_t k, std::string *err) everything before the first iteration of pair k     [preparing threads]
//   std::unique_ptr<Job> start(Handle, int64_t k, int *rc, std::string *err)                             [calling thread]
//   int advance(Job &, int64_t k, bool *progressed, std::string *err)   what can be done without waiting [calling thread]
//   bool finished(const Job &);  Handle handle_of(Job &)
//   void retire(Job &, int64_t k)                    the registration is over: hand its totals out       [calling thread]
//   void abandon(Job &)                              after a failure: leave the handle in a defined state
//   void release(Handle, bool ok)                    back to the pool, or destroyed                      [calling thread]
template <class Ops>
void run_device_share(Ops &ops, int64_t mine, int lanes_per_device, FirstError &first)
{
    using Handle = typename Ops::Handle;
    using Job = typename Ops::Job;
    // (fewer pairs than devices, or a device listed twice: a share may be empty — and with no handle to pass around the
    //  two sides below would wait for each other forever)
    if (mine <= 0) return;
    const int lanes = (int)std::max<int64_t>(1, std::min<int64_t>(lanes_per_device, mine));
    std::vector<Handle> handles;  // in flight, being prepared, or waiting on either side
    HandOver<Handle> idle;
    struct Prepared {
        Handle h;
        int64_t k;
    };
    HandOver<Prepared> ready;
    // (two preparing threads from three registrations in flight on: pairs of 100k points are prepared in 0.4 ms and
    //  iterated on in 0.25 ms)
    const int n_preparing = (lanes >= 3 && mine >= 8) ? 2 : 1;
    for (int64_t h = 0; h < std::min<int64_t>(mine, (int64_t)lanes + 2 * n_preparing); h++) {
        std::string err;
        int rc = 0;
        Handle c = ops.acquire(&rc, &err);
        if (!c) {
            first.set(rc != 0 ? rc : -1, err);
            break;
        }
        handles.push_back(c);
        idle.push(c);
    }
    if (!first.failed()) {
        std::atomic<int64_t> next_mine{0};
        std::atomic<int> still_preparing{n_preparing};
        auto prepare = [&]() {
            for (;;) {
                if (first.failed()) break;
                Handle c{};
                if (idle.pop(&c, true) != 1) break;  // (closed: the other side met an error, or every pair is taken)
                const int64_t k = next_mine.fetch_add(1);
                if (k >= mine) {
                    idle.push(c);
                    break;
                }
                std::string err;
                const int rc = ops.prepare(c, k, &err);
                if (rc != 0) {
                    first.set(rc, err);
                    break;
                }
                ready.push(Prepared{c, k});
            }
            if (still_preparing.fetch_sub(1) == 1) ready.close();
        };
        std::vector<std::thread> preparing;
        for (int t = 0; t < n_preparing; t++) preparing.emplace_back(prepare);
        struct Running {
            std::unique_ptr<Job> job;
            int64_t k;
        };
        std::vector<Running> window;
        bool more = true;
        while ((more || !window.empty()) && !first.failed()) {
            while (more && (int)window.size() < lanes) {
                Prepared pp{Handle{}, 0};
                const int got = ready.pop(&pp, window.empty());
                if (got < 0) more = false;
                if (got <= 0) break;
                int rc = 0;
                std::string err;
                std::unique_ptr<Job> job = ops.start(pp.h, pp.k, &rc, &err);
                if (rc != 0 || !job) {
                    first.set(rc != 0 ? rc : -1, err);
                    break;
                }
                window.push_back(Running{std::move(job), pp.k});
            }
            bool any = false;
            for (size_t w = 0; w < window.size() && !first.failed();) {
                Running &r = window[w];
                bool progressed = false;
                std::string err;
                const int rc = ops.advance(*r.job, r.k, &progressed, &err);
                if (rc != 0) {
                    first.set(rc, err);
                    break;
                }
                any = any || progressed;
                if (ops.finished(*r.job)) {
                    // the totals are on the host; what is still queued on the handle's stream the next upload waits for
                    ops.retire(*r.job, r.k);
                    idle.push(ops.handle_of(*r.job));
                    window.erase(window.begin() + (long)w);
                } else {
                    w++;
                }
            }
            if (!any && !window.empty()) std::this_thread::yield();
        }
        if (first.failed())
            for (auto &r : window) ops.abandon(*r.job);
        idle.close();
        for (auto &th : preparing) th.join();
    }
    // handles that worked go back to the pool (their buffers stay allocated for the next batch); after a failure none of
    // them is trusted again
    const bool ok = !first.failed();
    for (Handle c : handles) ops.release(c, ok);
}

}  // namespace sched
}  // namespace ppcr
