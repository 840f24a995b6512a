// Device and pinned-host memory of the handles: a few large driver allocations, sub-allocated on the host side.
//
// Why.  A handle owns ~50 device buffers and 6 pinned ones.  One hipMalloc / hipHostMalloc each means ~50 driver calls
// (KFD ioctls: allocate + map into the GPU page tables) on the path of the FIRST registration of every fresh handle —
// ProbPointCloudRegistration::align() on a new object, every pair of a batch on a cold pool — and hipFree synchronises
// the whole device.  Their cost is not ours to control: a few microseconds each on one box, 100+ on another (the round-4
// driver measured +3 ms per fresh align() on boxes where the same binaries of the round before had not: nothing in the
// library had changed).  With 288 GB of HBM per GPU the right shape is the opposite: take memory from the driver in
// slabs that grow geometrically (64 MB, 128 MB, ... 1 GB), carve the buffers out of them with a first-fit free list, and
// keep what a destroyed handle gives back for the next one.  A 1M <-> 1M registration (~0.3 GB) is then 3 driver calls on
// a cold process and none on a warm one.
//
// Rules the callers keep (DevBuf, ppcr_destroy):
//  * a block goes back to the pool only when no work that may touch it is in flight (ppcr_destroy synchronises the
//    handle's streams first; DevBuf's grow path synchronises the device, as the hipFree it replaces did implicitly);
//  * blocks come back with whatever the previous owner left in them (as hipMalloc'd memory does): every buffer whose
//    contents matter before its first write is cleared by its owner when (re)allocated.
//
// Host code only; one pool per device, process-wide, behind a mutex (allocation is never on a steady-state path).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace ppcr {

struct PoolStats {
    uint64_t reserved_bytes = 0;   // held from the driver
    uint64_t in_use_bytes = 0;     // handed out to buffers
    uint64_t driver_allocs = 0;    // hipMalloc calls made so far
    uint64_t driver_frees = 0;     // hipFree calls made so far
    uint64_t block_allocs = 0;     // sub-allocations served so far
};

// (Driver: where slabs come from — hipMalloc / hipFree in the library, a host stand-in in tests/cpp/test_pool.cc, which
//  runs the sub-allocator under ASan + UBSan on the CPU)
struct HipDriver {
    static hipError_t malloc(void **p, size_t bytes) { return hipMalloc(p, bytes); }
    static void free(void *p) { (void)hipFree(p); }
    static void forget_error() { (void)hipGetLastError(); }
};

template <class Driver, size_t FirstSlabBytes = (64ull << 20), size_t MaxSlabBytes = (1ull << 30)>
class BasicDevicePool {
public:
    static constexpr size_t kAlign = 512;                  // every block: address and size multiples of this
    static constexpr size_t kFirstSlab = FirstSlabBytes;   // slabs grow 64 MB, 128 MB, ... kMaxSlab
    static constexpr size_t kMaxSlab = MaxSlabBytes;
    static constexpr size_t kGranule = FirstSlabBytes / 32;  // a slab of its own is sized in these (2 MB)

    // (device already current: the callers have run hipSetDevice)
    hipError_t alloc(size_t bytes, void **out)
    {
        *out = nullptr;
        const size_t want = round_up(std::max<size_t>(bytes, 1), kAlign);
        std::lock_guard<std::mutex> lk(mu_);
        bool trimmed = false;
        for (;;) {
            for (auto &s : slabs_)
                if (void *p = carve(*s, want)) {
                    *out = p;
                    return hipSuccess;
                }
            // no slab has room: one more from the driver — the next size of the geometric series, or the request itself
            // when that is larger (a buffer of its own: 2 MB granules)
            size_t slab_bytes = std::min(kMaxSlab, kFirstSlab << std::min<size_t>(slabs_.size(), 8));
            if (want > slab_bytes) slab_bytes = round_up(want, kGranule);
            void *base = nullptr;
            hipError_t e = Driver::malloc(&base, slab_bytes);
            if (e != hipSuccess && slab_bytes > round_up(want, kGranule)) {
                Driver::forget_error();
                slab_bytes = round_up(want, kGranule);  // (a device nearly full: exactly what is asked for)
                e = Driver::malloc(&base, slab_bytes);
            }
            if (e != hipSuccess) {
                Driver::forget_error();
                if (!trimmed && trim_locked() > 0) {  // give the driver back what nobody uses, then once more
                    trimmed = true;
                    continue;
                }
                return e;
            }
            stats_.driver_allocs++;
            stats_.reserved_bytes += slab_bytes;
            std::unique_ptr<Slab> s(new Slab);
            s->base = static_cast<char *>(base);
            s->size = slab_bytes;
            s->free_by_offset[0] = slab_bytes;
            slabs_.push_back(std::move(s));  // (the next turn of the loop carves from it)
        }
    }

    void free(void *p)
    {
        if (!p) return;
        std::lock_guard<std::mutex> lk(mu_);
        auto it = live_.find(p);
        if (it == live_.end()) return;  // (not ours: never happens; nothing sensible to do)
        Slab &s = *it->second.slab;
        size_t off = (size_t)(static_cast<char *>(p) - s.base), len = it->second.bytes;
        live_.erase(it);
        stats_.in_use_bytes -= len;
        s.live_blocks--;
        // merge with the free neighbours
        auto next = s.free_by_offset.lower_bound(off);
        if (next != s.free_by_offset.end() && off + len == next->first) {
            len += next->second;
            next = s.free_by_offset.erase(next);
        }
        if (next != s.free_by_offset.begin()) {
            auto prev = std::prev(next);
            if (prev->first + prev->second == off) {
                off = prev->first;
                len += prev->second;
                s.free_by_offset.erase(prev);
            }
        }
        s.free_by_offset[off] = len;
        // keep at most kKeepIdle bytes of wholly unused slabs (288 GB of HBM: a few GB parked cost nothing, and the
        // next handle starts without a driver call)
        if (s.live_blocks == 0 && idle_bytes_locked() > keep_idle_bytes()) trim_locked(keep_idle_bytes());
    }

    // hand every slab nobody uses back to the driver (hipFree synchronises the device); returns the bytes released
    size_t trim()
    {
        std::lock_guard<std::mutex> lk(mu_);
        return trim_locked();
    }

    PoolStats stats()
    {
        std::lock_guard<std::mutex> lk(mu_);
        return stats_;
    }
    // (tests) bytes of the largest free range over all slabs, and the number of slabs
    size_t largest_free_range()
    {
        std::lock_guard<std::mutex> lk(mu_);
        size_t best = 0;
        for (auto &s : slabs_)
            for (auto &f : s->free_by_offset) best = std::max(best, f.second);
        return best;
    }
    size_t slab_count()
    {
        std::lock_guard<std::mutex> lk(mu_);
        return slabs_.size();
    }

private:
    struct Slab {
        char *base = nullptr;
        size_t size = 0;
        size_t live_blocks = 0;
        std::map<size_t, size_t> free_by_offset;  // offset -> length
    };
    struct Live {
        Slab *slab;
        size_t bytes;
    };

    static size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
public:
    // bytes of wholly unused slabs kept for the next handle (PPCR_POOL_KEEP_MB, default 4096)
    static size_t keep_idle_bytes()
    {
        static const size_t keep = [] {
            const char *e = std::getenv("PPCR_POOL_KEEP_MB");
            return (size_t)(e && *e ? std::strtoull(e, nullptr, 10) : 4096ull) << 20;
        }();
        return keep;
    }

private:
    void *carve(Slab &s, size_t want)
    {
        for (auto it = s.free_by_offset.begin(); it != s.free_by_offset.end(); ++it) {
            if (it->second < want) continue;
            const size_t off = it->first, len = it->second;
            s.free_by_offset.erase(it);
            if (len > want) s.free_by_offset[off + want] = len - want;
            s.live_blocks++;
            void *p = s.base + off;
            live_[p] = Live{&s, want};
            stats_.in_use_bytes += want;
            stats_.block_allocs++;
            return p;
        }
        return nullptr;
    }
    size_t idle_bytes_locked() const
    {
        size_t idle = 0;
        for (auto &s : slabs_)
            if (s->live_blocks == 0) idle += s->size;
        return idle;
    }
    // release unused slabs, largest first, until at most `keep` bytes of them remain
    size_t trim_locked(size_t keep = 0)
    {
        size_t released = 0, idle = idle_bytes_locked();
        while (idle > keep) {
            size_t pick = slabs_.size();
            for (size_t k = 0; k < slabs_.size(); k++)
                if (slabs_[k]->live_blocks == 0 && (pick == slabs_.size() || slabs_[k]->size > slabs_[pick]->size)) pick = k;
            if (pick == slabs_.size()) break;
            Driver::free(slabs_[pick]->base);
            stats_.driver_frees++;
            stats_.reserved_bytes -= slabs_[pick]->size;
            released += slabs_[pick]->size;
            idle -= slabs_[pick]->size;
            slabs_.erase(slabs_.begin() + (long)pick);
        }
        return released;
    }

    std::mutex mu_;
    std::vector<std::unique_ptr<Slab>> slabs_;
    std::unordered_map<void *, Live> live_;
    PoolStats stats_;
};

using DevicePool = BasicDevicePool<HipDriver>;

// one pool per device id; the pools live as long as the process (their slabs go with it: no destructor runs after the
// HIP runtime has shut down)
inline DevicePool &device_pool(int device)
{
    static std::mutex mu;
    static std::vector<DevicePool *> *pools = new std::vector<DevicePool *>();
    std::lock_guard<std::mutex> lk(mu);
    if (device < 0) device = 0;
    if ((size_t)device >= pools->size()) pools->resize((size_t)device + 1, nullptr);
    if (!(*pools)[(size_t)device]) (*pools)[(size_t)device] = new DevicePool();
    return *(*pools)[(size_t)device];
}

// The pinned, device-mapped host memory of a handle (mailbox ring, report ring, small read-back areas): ONE
// hipHostMalloc per handle instead of six, and the block of a destroyed handle serves the next one of that device.
class PinnedPool {
public:
    hipError_t take(size_t bytes, void **out)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (size_t k = 0; k < idle_.size(); k++)
                if (idle_[k].bytes >= bytes) {
                    *out = idle_[k].p;
                    sizes_[*out] = idle_[k].bytes;
                    idle_.erase(idle_.begin() + (long)k);
                    return hipSuccess;
                }
        }
        const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocMapped | hipHostMallocCoherent);
        if (e == hipSuccess) {
            std::lock_guard<std::mutex> lk(mu_);
            sizes_[*out] = bytes;
        }
        return e;
    }
    void give(void *p)
    {
        if (!p) return;
        std::lock_guard<std::mutex> lk(mu_);
        auto it = sizes_.find(p);
        if (it == sizes_.end()) return;
        if (idle_.size() < 64) idle_.push_back(Block{p, it->second});
        else (void)hipHostFree(p);
        sizes_.erase(it);
    }
    void trim()
    {
        std::lock_guard<std::mutex> lk(mu_);
        for (auto &b : idle_) (void)hipHostFree(b.p);
        idle_.clear();
    }

private:
    struct Block {
        void *p;
        size_t bytes;
    };
    std::mutex mu_;
    std::vector<Block> idle_;
    std::unordered_map<void *, size_t> sizes_;
};

inline PinnedPool &pinned_pool(int device)
{
    static std::mutex mu;
    static std::vector<PinnedPool *> *pools = new std::vector<PinnedPool *>();
    std::lock_guard<std::mutex> lk(mu);
    if (device < 0) device = 0;
    if ((size_t)device >= pools->size()) pools->resize((size_t)device + 1, nullptr);
    if (!(*pools)[(size_t)device]) (*pools)[(size_t)device] = new PinnedPool();
    return *(*pools)[(size_t)device];
}

}  // namespace ppcr
