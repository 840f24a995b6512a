// The ONE collective of the multi-GPU deployment, native: one process per GPU, every rank registers its own pairs
// (ppcr_batch_run / ppcr_align_many on its device) and the final transforms are all-gathered over RCCL (xGMI inside a
// node) — ppcr_comm_* / ppcr_gather_transforms of include/ppcr.h.  The Python front end does the same through
// torch.distributed (batch.py); this is the entry point below Python (the command line's --rank / --world).
//
// RCCL is bound at run time (dlopen of librccl.so.1): a single-GPU user of libppcr_hip.so needs no RCCL at all, and a
// process that already carries one (PyTorch-ROCm bundles its own copy under the same soname) shares it.
#include "ppcr.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

namespace {

thread_local std::string g_comm_error;

int comm_fail(int code, const std::string &msg)
{
    g_comm_error = msg;
    return code;
}

// the handful of RCCL entry points used, with their rccl.h signatures (ncclResult_t / ncclDataType_t are ints there;
// ncclDouble = 8; ncclUniqueId is 128 opaque bytes passed BY VALUE to ncclCommInitRank)
struct UniqueId {
    char internal[PPCR_COMM_ID_BYTES];
};
typedef void *Comm;
struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, Comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string why;
};
constexpr int kNcclDouble = 8;

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            r.why = std::string("librccl.so not found: ") + dlerror();
            return;
        }
        auto sym = [&](const char *n) {
            void *p = dlsym(r.handle, n);
            if (!p && r.why.empty()) r.why = std::string("librccl.so lacks ") + n;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return r;
}

int rccl_ready()
{
    const Rccl &r = rccl();
    if (!r.why.empty()) return comm_fail(PPCR_ERR_STATE, r.why);
    return PPCR_OK;
}

std::string nccl_text(int rc)
{
    const Rccl &r = rccl();
    return r.GetErrorString ? r.GetErrorString(rc) : ("ncclResult " + std::to_string(rc));
}

}  // namespace

struct ppcr_comm {
    int device = 0, rank = 0, world = 1;
    Comm comm = nullptr;
    hipStream_t stream = nullptr;
    double *d_send = nullptr, *d_recv = nullptr;
    size_t cap_send = 0, cap_recv = 0;  // doubles
};

extern "C" {

const char *ppcr_comm_last_error(void) { return g_comm_error.c_str(); }

int ppcr_comm_get_id(unsigned char id[PPCR_COMM_ID_BYTES])
{
    if (!id) return comm_fail(PPCR_ERR_INVALID, "null id");
    if (int rc = rccl_ready()) return rc;
    UniqueId u;
    const int rc = rccl().GetUniqueId(&u);
    if (rc != 0) return comm_fail(PPCR_ERR_HIP, "ncclGetUniqueId: " + nccl_text(rc));
    std::memcpy(id, u.internal, PPCR_COMM_ID_BYTES);
    return PPCR_OK;
}

int ppcr_comm_create(int device_id, int rank, int world, const unsigned char id[PPCR_COMM_ID_BYTES], ppcr_comm **out)
{
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return comm_fail(PPCR_ERR_INVALID, "ppcr_comm_create: bad argument");
    *out = nullptr;
    if (int rc = rccl_ready()) return rc;
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) {
        (void)hipGetLastError();
        return comm_fail(PPCR_ERR_NODEVICE, "no HIP device visible");
    }
    if (device_id < 0 || device_id >= visible) return comm_fail(PPCR_ERR_INVALID, "ppcr_comm_create: device id out of range");
    if (hipSetDevice(device_id) != hipSuccess) return comm_fail(PPCR_ERR_HIP, "hipSetDevice failed");
    ppcr_comm *c = new ppcr_comm;
    c->device = device_id, c->rank = rank, c->world = world;
    UniqueId u;
    std::memcpy(u.internal, id, PPCR_COMM_ID_BYTES);
    const int rc = rccl().CommInitRank(&c->comm, world, u, rank);  // (collective: returns once every rank has called it)
    if (rc != 0) {
        delete c;
        return comm_fail(PPCR_ERR_HIP, "ncclCommInitRank: " + nccl_text(rc));
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        rccl().CommDestroy(c->comm);
        delete c;
        return comm_fail(PPCR_ERR_HIP, "hipStreamCreate failed");
    }
    *out = c;
    return PPCR_OK;
}

int ppcr_comm_destroy(ppcr_comm *c)
{
    if (!c) return PPCR_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) rccl().CommDestroy(c->comm);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return PPCR_OK;
}

int ppcr_gather_transforms(ppcr_comm *c, const double *T_local, int64_t n_pairs, double *T_all)
{
    if (!c || n_pairs < 0 || (n_pairs > 0 && !T_all)) return comm_fail(PPCR_ERR_INVALID, "ppcr_gather_transforms: bad argument");
    if (n_pairs == 0) return PPCR_OK;
    if (hipSetDevice(c->device) != hipSuccess) return comm_fail(PPCR_ERR_HIP, "hipSetDevice failed");
    // pair p lives on rank p % world (the deal of ppcr_batch_run and of batch.py): rank r holds ceil((n - r) / world)
    // pairs, in ascending p; every rank sends `per` slots of 12 doubles (the short ranks pad with NaN)
    const int64_t world = c->world, per = (n_pairs + world - 1) / world;
    const int64_t mine = (n_pairs - c->rank + world - 1) / world;
    if (mine > 0 && !T_local) return comm_fail(PPCR_ERR_INVALID, "ppcr_gather_transforms: null T_local");
    const size_t n_send = (size_t)per * 12, n_recv = n_send * (size_t)world;
    auto grow = [](double *&p, size_t &cap, size_t want) {
        if (want <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr, cap = 0;
        const hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), want * sizeof(double));
        if (e == hipSuccess) cap = want;
        return e;
    };
    if (grow(c->d_send, c->cap_send, n_send) != hipSuccess || grow(c->d_recv, c->cap_recv, n_recv) != hipSuccess)
        return comm_fail(PPCR_ERR_NOMEM, "ppcr_gather_transforms: hipMalloc failed");
    std::vector<double> stage(n_send, std::numeric_limits<double>::quiet_NaN());
    if (mine > 0) std::memcpy(stage.data(), T_local, (size_t)mine * 12 * sizeof(double));
    if (hipMemcpyAsync(c->d_send, stage.data(), n_send * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess)
        return comm_fail(PPCR_ERR_HIP, "upload of the local transforms failed");
    const int rc = rccl().AllGather(c->d_send, c->d_recv, n_send, kNcclDouble, c->comm, c->stream);
    if (rc != 0) return comm_fail(PPCR_ERR_HIP, "ncclAllGather: " + nccl_text(rc));
    std::vector<double> all(n_recv);
    if (hipMemcpyAsync(all.data(), c->d_recv, n_recv * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
        return comm_fail(PPCR_ERR_HIP, "download of the gathered transforms failed");
    for (int64_t p = 0; p < n_pairs; p++)
        std::memcpy(T_all + (size_t)p * 12, all.data() + ((size_t)(p % world) * (size_t)per + (size_t)(p / world)) * 12, 12 * sizeof(double));
    return PPCR_OK;
}

}  // extern "C"
