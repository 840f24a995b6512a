// C ABI (include/ppcr.h) over the gfx950 kernels of ppcr_kernels.hip.h and (K1) ppcr_nn_tile.hip.
// One ppcr_ctx = one device + one HIP stream + the device-resident state of one source/target
// pair.  There is no CPU fallback anywhere in this file: without a GPU ppcr_create() fails with
// PPCR_ERR_NODEVICE and nothing else can be called.
#include "ppcr.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <numeric>
#include <string>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "ppcr_host_math.hpp"
#include "ppcr_kernels.hip.h"
#include "ppcr_nn_tile_launch.hip.h"

using namespace ppcr;
using namespace ppcr::dev;

namespace {

thread_local std::string g_create_error;

enum KernelId {
    K_REPACK = 0,
    K_BBOX,
    K_CELL_KEY,
    K_RADIX_SORT,
    K_GATHER,
    K_CELL_START,
    K_NN_TOPM,
    K_NN_CLEANUP,
    K_NN_COUNT,
    K_NN_SCAN,
    K_NN_FILL,
    K_NN_SELECT,
    K_NN_COMPACT,
    K_COUNT_SUM,
    K_WEIGHTS,
    K_ACCUMULATE,
    K_REDUCE,
    K_TRANSFORM,
    K_INNER,
    K_TRACK,
    K_NN_WIDE,
    K_NUM
};
const char *const kKernelNames[K_NUM] = {
    "repack_kernel",   "bbox_kernel",    "cell_key_kernel",   "radix_sort",       "gather_points_kernel",
    "cell_start_kernel", "nn_fast_kernel", "nn_tile_cleanup_kernel", "nn_count_kernel",  "nn_scan",          "nn_fill_kernel",
    "nn_select_kernel", "csr_compact_kernel", "ell_count_sum_kernel", "weights_kernel", "accumulate_kernel",
    "reduce_partials_kernel", "transform_kernel", "inner_steps_kernel", "track_kernel", "nn_wide_kernel"};

constexpr int kMailboxRing = 4;    // mailbox / report slots; at most kMaxAhead iterations are ever in flight
constexpr int kMaxAhead = 3;       // outer iterations ppcr_align may have enqueued beyond the last one the host has seen
constexpr int kEllMaxWidth = 32;   // widest register-list NN variant / widest ELL association
constexpr int kAccumMaxBlocks = 1024;

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 8 + 64;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct ProfRec {
    int id;
    hipEvent_t start, stop;
};

}  // namespace

struct ppcr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // parameters
    double radius = 1.0;
    int max_nb = 20;
    double dof = 5.0;
    int dim = 3;
    int opt_sort_source = 1;
    int opt_stamps = 0;
    DevBuf<unsigned long long> d_stamps;
    size_t stamps_wgs = 0;  // workgroups of the last launch that wrote stamps (its grid, idle split slots included)

    // clouds
    int64_t ns = 0, nt = 0;
    bool have_src = false, have_tgt = false;
    DevBuf<unsigned char> staging;
    DevBuf<float4> tgt_raw, tgt_sorted, src, src_alt;
    bool grid_valid = false;
    double grid_radius = -1;
    int grid_max_nb = -1;
    // the grid build in two halves (grid_begin / grid_finish): with a host buffer coming in, the first half is enqueued on
    // aux_stream at the end of ppcr_set_target and runs while the caller's ppcr_set_source copies over PCIe
    hipStream_t aux_stream = nullptr;   // the device's shared second stream (device_aux_stream), not owned
    hipEvent_t aux_done = nullptr;      // recorded behind this handle's early build
    int opt_eager_grid = 1;
    bool params_set = false;         // ppcr_set_params has been called (the early build does not start on the defaults)
    bool grid_pending = false;       // grid_begin ran on aux_stream for (pending_radius, pending_max_nb); grid_finish has not yet
    double pending_radius = -1;
    int pending_max_nb = -1;
    double grid_search = 1.0;        // the build in progress: its search radius, which attempt it is, whether its occupancy
    int grid_attempt = 0;            // histogram is on its way to h_occupancy
    bool grid_bounded = false, grid_occ_inflight = false;
    unsigned long long *h_occupancy = nullptr;  // pinned, kOccBins words
    float *h_bbox = nullptr;                    // pinned, kBboxBlocks partial boxes of the target (bbox_launch / bbox_fold)
    int bbox_blocks = 0;
    // two-pass radius search (see nn_wide_kernel): the grid is built for search_radius = radius / reach <= radius
    double search_radius = 1.0;
    int reach = 1;
    int opt_two_pass = 1;
    int opt_fuse_max_handed_over = 4;  // more workgroups than this handed over last time: K23 is not folded into K1
    int opt_first_pass_fill = 22;  // tenths: the first-pass sphere should hold this many times max_neighbours points
    int opt_first_pass_occ = 110;  // tenths of a point per first-pass cell (see target_occupancy)
    DevBuf<unsigned long long> d_occupancy;
    DevBuf<int> d_short;         // second pass: [0], [1] the number of listed rows (ping-pong pair, see ovf_parity), [2] last count seen, [3 ..] the list
    bool src_sorted = false;
    GridDesc grid{};
    DevBuf<int> cell_start;
    // multi-level search (GridLevel): the levels OTHER than the base (whose grid / tgt_sorted / cell_start are the members
    // above), the table nn_fast_kernel<..., MULTI> reads (ascending r2_cap) and which of its entries is the base
    struct ExtraLevel {
        GridDesc g{};
        DevBuf<float4> tgt;
        DevBuf<int> cell_start, to_base;
        double radius = 0;
    };
    std::vector<ExtraLevel> extra_levels;
    DevBuf<GridLevel> d_levels;
    DevBuf<int> level_inv;
    DevBuf<unsigned> level_dbg;        // diagnostic counters of the multi-level search (option "level_stats")
    int opt_level_stats = 0;
    DevBuf<unsigned short> level_cap;  // per 256-query block of the (sorted) source: cap | floor << 4 | split << 8 (UnansweredRows::level_cap)
    bool level_cap_clean = false;      // ... all 0xFF for the current source order
    int n_levels = 1, base_level = 0, finest_extra = -1;  // finest_extra: index into extra_levels of the finest level (-1: the base is)
    float level_r2_cap[kMaxLevels] = {};
    int opt_levels = -1;               // -1 automatic (non-uniform clouds / radii that hold far more than max_neighbours), 0 one level
    double origin[3] = {0, 0, 0};
    bool origin_valid = false;

    // sort scratch
    DevBuf<unsigned> keys_a, keys_b;
    DevBuf<int> vals_a, vals_b;
    DevBuf<unsigned char> cub_tmp;
    DevBuf<float> bbox_part;

    // association
    enum { ASSOC_NONE, ASSOC_ELL, ASSOC_CSR } assoc = ASSOC_NONE;
    int ell_width = 0;
    DevBuf<int> nbr, cnt, row_ptr;
    DevBuf<unsigned char> split_flag;  // per 256-query block: scanned in two halves (its halo outgrew the steady-state capacity)
    DevBuf<int> split_list;            // the split blocks (<= kMaxSplit), in ascending id
    DevBuf<unsigned> split_state;      // [0] registrations so far, [1] blocks in the list, [2] registrations the list was last rebuilt for
    bool split_clean = false;          // the three buffers above are zeroed for the current source order
    DevBuf<int> ovf_list;        // blocks nn_fast_kernel handed over to nn_tile_cleanup_kernel
    DevBuf<unsigned> ovf_state;  // two list counters used alternately (ovf_parity): the idle one is cleared by the fast kernel
    int ovf_parity = 0;
    unsigned ovf_last = ~0u;     // blocks handed over by the most recent association whose count reached the host (~0: unknown)
    // ... and the count the NEXT association's choices go by.  Inside the device-paced loop that is the count of the
    // iteration kMaxAhead back — always consumed by the time an iteration is enqueued, whatever the scheduler's timing —
    // so that two runs of one registration take the same path (see AlignJob::enqueue); elsewhere the latest.
    unsigned ovf_decide = ~0u;
    bool ovf_decide_pinned = false;
    DevBuf<unsigned> dm2;    // per (sorted) source row: float d2 bits of its m-th neighbour in the last tiled K1
    bool dm2_valid = false;  // dm2 matches the current source order / target / radius / max_neighbours
    int opt_temporal = 1;
    DevBuf<int> gen_counts, gen_row_ptr, gen_pos;
    DevBuf<unsigned long long> gen_keys;
    DevBuf<unsigned long long> d_total;
    int64_t nnz = -1;

    // reductions
    DevBuf<double> partials, d_sums;
    DevBuf<unsigned> d_ticket;
    bool move_pending = false;        // a source move that the next tiled K1 will apply in its prologue
    double pending_T[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    double *h_sums = nullptr;          // pinned
    HostMailbox *h_mbox = nullptr;     // pinned + device-mapped ring of kMailboxRing slots written by the GPU
    HostMailbox *d_mbox = nullptr;     // device-side alias of h_mbox
    unsigned mbox_seq = 0;             // sequence number of the last fold-and-solve launch (slot = seq % ring)
    DevBuf<Pose> d_pose;               // transform solved by the last reduce_solve_kernel (next K1's move)
    bool move_on_device = false;       // the pending source move is *d_pose (host copy not read back yet)
    int opt_mailbox = 1;               // 1: deliver the moments through the mailbox and spin (default)
    // device-paced inner loop (ppcr_align with inner_steps > 1): see LoopState / inner_steps_kernel
    DevBuf<LoopState> d_loop;
    DevBuf<unsigned> d_inner_ctl;      // [0..1] step_done (one 64-bit word), [2 ..] completion flags of inner_steps_kernel's K23 workgroups
    int opt_fold_stamps = 0;           // diagnostic: the solve lane leaves wall-clock stamps (ppcr_debug_get_fold_stamps)
    DevBuf<unsigned long long> d_fold_dbg;
    int opt_inner_dev_steps = 3;       // IRLS steps 2.. the device may take on its own per outer iteration (<= kMaxDevSteps)
    // what inner_steps_kernel reads from device memory instead of taking it as kernel arguments, and the host's copy
    DevBuf<InnerConst> d_inner_const;
    InnerConst h_inner_const{};
    bool inner_const_valid = false;
    // per-iteration reports of the device-paced loop (track_kernel)
    HostReport *h_report = nullptr;    // pinned + device-mapped ring of kMailboxRing slots
    HostReport *d_report = nullptr;
    DevBuf<double> track_part;
    DevBuf<unsigned> track_ticket;
    unsigned long long *h_total = nullptr;  // pinned

    // weights export scratch
    DevBuf<double> d_w, d_s;

    // host caches for the export / import paths (never touched by the hot loop)
    bool csr_cache_valid = false;
    std::vector<int> h_csr_row_ptr, h_csr_col;
    std::vector<size_t> h_csr_slot;

    // profiling
    bool prof_on = false;
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[K_NUM] = {0};
    int64_t prof_n[K_NUM] = {0};

    double dbg_host[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // diagnostics of the last ppcr_align (ppcr_debug_get_host_times)
    // reporting clouds (the step after each iteration: cc:110-129): a full-resolution companion that follows every
    // move of the source, the ground truth and the previous-iteration snapshot, all in the caller's index order
    DevBuf<float4> companion, ground_truth, previous;
    int64_t n_companion = 0, n_ground_truth = 0, n_previous = 0;
    bool have_companion = false, have_ground_truth = false, have_previous = false;
    DevBuf<double> mse_part;
    int opt_short_lists = 1;
    int opt_k1_halves = 0;       // steady-state K1 scans every block as two half-blocks: 0 never (default), 1 always, -1 small clouds
    int opt_fuse_k23 = 1;        // ppcr_align's one-step iterations fold K23 into the steady-state K1
    int opt_merge_fold = 1;      // ... and the fold-and-solve step rides in the cleanup launch
    bool assoc_folded = false;   // the last association's launches included the fold-and-solve step
    bool assoc_fused = false;    // the last association also left the partial moments of the pose it was given
    int fused_slots = 0;         // ... in this many partial vectors
    int opt_run_ahead = 1;       // ppcr_align keeps the device one iteration ahead of the host when the rule allows
    int opt_defer_moves = 0;     // ppcr_apply_transform leaves the move to the next association's prologue (as ppcr_iterate does)
    int opt_brick_xshift = 0;    // log2 of the source bricks' x extent in cells (0: 4x4 yz columns walked along x)
    int opt_grid_xf = 4;         // x slices per grid cell (GridDesc::xr)
    float tgt_lo[3] = {0, 0, 0}, tgt_hi[3] = {0, 0, 0};
    // which copy of the target the association's positions index: 0 caller order, 1 grid-sorted
    int assoc_space = 0;
    const float4 *tgt_space(int sp) const { return sp == 1 ? tgt_sorted.p : tgt_raw.p; }
    const float4 *tgt_cur() const { return tgt_space(assoc_space); }
};

namespace {

int fail(ppcr_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg;
    else g_create_error = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(ctx, e_ == hipErrorOutOfMemory ? PPCR_ERR_NOMEM : PPCR_ERR_HIP,            \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                        \
    } while (0)

#define PPCR_TRY(expr)                  \
    do {                                \
        int rc_ = (expr);               \
        if (rc_ != PPCR_OK) return rc_; \
    } while (0)

inline int nblocks(int64_t n, int block = kBlock) { return (int)((n + block - 1) / block); }

struct ProfScope {
    ppcr_ctx *c;
    ProfRec r{};
    bool active = false;
    ProfScope(ppcr_ctx *ctx, int id) : c(ctx) { begin(id); }
    ~ProfScope() { end(); }
    void begin(int id)
    {
        if (!c->prof_on) return;
        for (int k = 0; k < 2; k++) {
            hipEvent_t ev;
            if (!c->prof_pool.empty()) {
                ev = c->prof_pool.back();
                c->prof_pool.pop_back();
            } else if (hipEventCreate(&ev) != hipSuccess) {
                return;
            }
            (k == 0 ? r.start : r.stop) = ev;
        }
        r.id = id;
        active = true;
        (void)hipEventRecord(r.start, c->stream);
    }
    void end()
    {
        if (!active) return;
        (void)hipEventRecord(r.stop, c->stream);
        c->prof_recs.push_back(r);
        active = false;
    }
    // close the running record here and time what follows under another id (two launches inside one call)
    void split(int id)
    {
        if (!active) return;
        end();
        begin(id);
    }
};

int check_launch(ppcr_ctx *c, const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, PPCR_ERR_HIP, std::string(what) + " launch: " + hipGetErrorString(e));
    return PPCR_OK;
}

void invalidate_association(ppcr_ctx *c)
{
    c->assoc = ppcr_ctx::ASSOC_NONE;
    c->nnz = -1;
    c->csr_cache_valid = false;
}

Model make_model(const ppcr_ctx *c)
{
    Model m;
    m.is_normal = !(c->dof < std::numeric_limits<double>::infinity());
    m.v = c->dof;
    m.texp = -(c->dof + c->dim) / 2.0;
    m.vpd = c->dof + c->dim;
    m.vpd_int = 0;
    if (!m.is_normal && m.vpd == std::floor(m.vpd) && m.vpd >= 1 && m.vpd <= 64) m.vpd_int = (int)m.vpd;
    return m;
}

Pose make_pose(const ppcr_ctx *c, const Mat3 &R, const double t[3])
{
    Pose P;
    for (int a = 0; a < 3; a++) {
        for (int b = 0; b < 3; b++) P.R[3 * a + b] = R.m[a][b];
        P.t[a] = t[a];
        P.c[a] = c->origin[a];
    }
    return P;
}

// upload (host or device pointer) + repack into float4 {x,y,z,original index}
int bbox_launch(ppcr_ctx *c, const float4 *pts, int n);
// with_bbox: the cloud's bounding box rides along (bbox_launch: folded by the caller after return, the stream is idle then)
int upload_cloud(ppcr_ctx *c, const void *ptr, bool on_device, int64_t n, int64_t stride, DevBuf<float4> &dst, bool with_bbox = false)
{
    if (n < 0 || n > (int64_t)INT32_MAX - 1024) return fail(c, PPCR_ERR_INVALID, "cloud size out of range");
    if (stride < 12 || (stride % 4) != 0) return fail(c, PPCR_ERR_INVALID, "stride_bytes must be a multiple of 4 and >= 12");
    if (n > 0 && ptr == nullptr) return fail(c, PPCR_ERR_INVALID, "null cloud pointer");
    HIP_TRY(c, dst.reserve((size_t)std::max<int64_t>(n, 1)));
    if (with_bbox) c->bbox_blocks = 0;
    if (n == 0) return PPCR_OK;
    const unsigned char *raw = static_cast<const unsigned char *>(ptr);
    if (!on_device) {
        // last point may be packed: copy exactly (n-1)*stride + 12 bytes
        const size_t bytes = (size_t)(n - 1) * (size_t)stride + 12;
        HIP_TRY(c, c->staging.reserve(bytes));
        HIP_TRY(c, hipMemcpyAsync(c->staging.p, ptr, bytes, hipMemcpyHostToDevice, c->stream));
        raw = c->staging.p;
    }
    {
        ProfScope ps(c, K_REPACK);
        repack_kernel<<<nblocks(n), kBlock, 0, c->stream>>>(raw, n, stride, dst.p);
    }
    PPCR_TRY(check_launch(c, "repack_kernel"));
    if (with_bbox) PPCR_TRY(bbox_launch(c, dst.p, (int)n));
    // the caller's host buffer may be freed after return
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PPCR_OK;
}

// sort `n` points of `in` by grid cell into `out` (stable: ties keep ascending original index)
// order: 0 the grid's cells (x fastest), 1 bricks (see brick_key_kernel), 2 the Hilbert curve over g's whole cells
int sort_by_cell(ppcr_ctx *c, const GridDesc &g, const float4 *in, int n, float4 *out, DevBuf<int> *cell_start_out,
                 int order = 0)
{
    const bool brick_order = order == 1, hilbert = order == 2;
    const bool want_cell_start = cell_start_out != nullptr;
    HIP_TRY(c, c->keys_a.reserve((size_t)n + 1));
    HIP_TRY(c, c->keys_b.reserve((size_t)n + 1));
    HIP_TRY(c, c->vals_a.reserve((size_t)n + 1));
    HIP_TRY(c, c->vals_b.reserve((size_t)n + 1));
    if (n > 0) {
        int end_bit_override = 0;
        {
            ProfScope ps(c, K_CELL_KEY);
            int hbits = 1, hshift = 0;
            if (hilbert) {
                const int most = std::max(std::max(g.n[0] >> g.xr_shift, g.n[1]), g.n[2]);
                while ((1 << hbits) < most) hbits++;
                if (hbits > 10) hshift = hbits - 10, hbits = 10;  // 30-bit keys: coarser curve cells on very large grids
            }
            if (hilbert)
                hilbert_key_kernel<<<nblocks(n), kBlock, 0, c->stream>>>(in, n, g, c->keys_a.p, c->vals_a.p, hbits, hshift);
            else if (brick_order)
                brick_key_kernel<<<nblocks(n), kBlock, 0, c->stream>>>(in, n, g, c->keys_a.p, c->vals_a.p, c->opt_brick_xshift);
            else
                cell_key_kernel<<<nblocks(n), kBlock, 0, c->stream>>>(in, n, g, c->keys_a.p, c->vals_a.p);
            if (hilbert) end_bit_override = 3 * hbits;
        }
        PPCR_TRY(check_launch(c, "cell_key_kernel"));
        long long nkeys = g.ncells;
        if (brick_order)
            nkeys = (64ll << c->opt_brick_xshift) * (((g.n[0] >> g.xr_shift) + (1 << c->opt_brick_xshift) - 1) >> c->opt_brick_xshift) *
                    brick_count(g.n[1]) * brick_count(g.n[2]);
        int end_bit = 1;
        while (end_bit < 32 && (1ll << end_bit) < nkeys) end_bit++;
        if (end_bit_override) end_bit = end_bit_override;
        // rocPRIM's radix sort, configured for clouds of 10^5..10^6 points (tools/micro/sort_bench.hip; (u32, i32) pairs, 21
        // key bits).  By default it merge-sorts up to 2^20 items — 159 us for a million, 60 us for 200k — and its onesweep
        // passes work in blocks of 6144 items (102 / 91 us).  Onesweep from 96k items on, in blocks of 2048 (a pass is bound
        // by the chain of its blocks' look-backs and by how few blocks 200k items are): 98 us for a million, 66 for 500k,
        // 49 for 200k, 43 for 100k.  Stable either way (ties keep ascending original index).
        using SortConfig = rocprim::radix_sort_config<
            rocprim::default_config, rocprim::default_config,
            rocprim::radix_sort_onesweep_config<rocprim::kernel_config<512, 4>, rocprim::kernel_config<512, 4>, 8, rocprim::block_radix_rank_algorithm::match>,
            98304>;
        size_t tmp_bytes = 0;
        HIP_TRY(c, rocprim::radix_sort_pairs<SortConfig>(nullptr, tmp_bytes, c->keys_a.p, c->keys_b.p, c->vals_a.p, c->vals_b.p, (size_t)n, 0u,
                                                         (unsigned)end_bit, c->stream));
        HIP_TRY(c, c->cub_tmp.reserve(tmp_bytes + 16));
        {
            ProfScope ps(c, K_RADIX_SORT);
            HIP_TRY(c, rocprim::radix_sort_pairs<SortConfig>(c->cub_tmp.p, tmp_bytes, c->keys_a.p, c->keys_b.p, c->vals_a.p, c->vals_b.p,
                                                             (size_t)n, 0u, (unsigned)end_bit, c->stream));
        }
        {
            ProfScope ps(c, K_GATHER);
            gather_points_kernel<<<nblocks(n), kBlock, 0, c->stream>>>(in, c->vals_b.p, n, out);
        }
        PPCR_TRY(check_launch(c, "gather_points_kernel"));
    }
    if (want_cell_start) {
        HIP_TRY(c, cell_start_out->reserve((size_t)g.ncells + 1));
        {
            ProfScope ps(c, K_CELL_START);
            cell_start_kernel<<<nblocks((int64_t)n + 1), kBlock, 0, c->stream>>>(c->keys_b.p, n, g.ncells, cell_start_out->p);
        }
        PPCR_TRY(check_launch(c, "cell_start_kernel"));
    }
    return PPCR_OK;
}

// bounding box of the finite points of a float4 cloud: per-block partial boxes on their way to pinned host memory
// (bbox_launch), folded once the stream has been synchronised (bbox_fold)
constexpr int kBboxBlocks = 1024;
int bbox_launch(ppcr_ctx *c, const float4 *pts, int n)
{
    c->bbox_blocks = 0;
    if (n <= 0) return PPCR_OK;
    const int nb = std::min(kBboxBlocks, nblocks(n));
    HIP_TRY(c, c->bbox_part.reserve((size_t)kBboxBlocks * 6));
    {
        ProfScope ps(c, K_BBOX);
        bbox_kernel<<<nb, kBlock, 0, c->stream>>>(pts, n, c->bbox_part.p);
    }
    PPCR_TRY(check_launch(c, "bbox_kernel"));
    HIP_TRY(c, hipMemcpyAsync(c->h_bbox, c->bbox_part.p, (size_t)nb * 6 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    c->bbox_blocks = nb;
    return PPCR_OK;
}
void bbox_fold(const ppcr_ctx *c, float lo[3], float hi[3])
{
    for (int a = 0; a < 3; a++) {
        lo[a] = INFINITY;
        hi[a] = -INFINITY;
    }
    for (int b = 0; b < c->bbox_blocks; b++)
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], c->h_bbox[(size_t)b * 6 + a]);
            hi[a] = std::max(hi[a], c->h_bbox[(size_t)b * 6 + 3 + a]);
        }
    for (int a = 0; a < 3; a++)
        if (!(lo[a] <= hi[a])) lo[a] = hi[a] = 0;  // no finite coordinate at all (or no point)
}

constexpr int kMaxReach = 8;  // the second pass's stencil is (2 reach + 1)^2 rows

// Cell occupancy the first pass of a two-pass search aims at (points per cell of edge r', where the typical point is): the
// first-pass sphere holds 4.19 q points, `fill` x max_neighbours of them (option "first_pass_fill", default 2.2) answer
// nearly every row in the first pass; but the fuller the cells, the more 256-query blocks' halos (36-49 rows of ~4 cells)
// outgrow the LDS tile and leave their rows to the second pass as well (`cap`, option "first_pass_occupancy", default 11
// points per cell).
double target_occupancy(int max_nb, double cap, double fill) { return std::min(fill * (double)max_nb / 4.19, cap); }

// First-pass search radius for a measured / estimated occupancy `per_cell` of cells of edge `radius`: the radius itself
// when such cells are not too full, else the radius at which a cell holds target_occupancy() points (never less than
// radius / kMaxReach).  The second pass then reaches ceil(radius / r') cells.
double choose_search_radius(double radius, double per_cell, int max_nb, double cap, double fill)
{
    if (!(per_cell > 0) || !std::isfinite(per_cell)) return radius;
    const double q = target_occupancy(max_nb, cap, fill);
    if (per_cell <= std::max(q, cap)) return radius;
    return std::max(radius * std::cbrt(q / per_cell), radius / kMaxReach);
}

// uniform grid over [lo, hi] whose 27-cell stencil covers a search of `cell_radius`
void make_grid_desc(int n, const float lo[3], const float hi[3], double cell_radius, int xf, GridDesc &g)
{
    float amax = 0;
    for (int a = 0; a < 3; a++) amax = std::max(amax, std::max(std::fabs(lo[a]), std::fabs(hi[a])));
    // cell edge slightly above the radius: float rounding of the cell index can then never push an
    // in-radius target outside the query's 27-cell stencil
    float h = (float)cell_radius * 1.001f + 16.0f * FLT_EPSILON * amax;
    // table bound: ~4 cells per target point, and small enough that brick-order keys fit 32 bits
    const double max_cells = std::min(4.0 * (double)n + 4096.0, 67108864.0);
    double ext[3];
    for (int a = 0; a < 3; a++) ext[a] = (double)hi[a] - (double)lo[a];
    for (;;) {
        double nc = xf;
        for (int a = 0; a < 3; a++) nc *= std::floor(ext[a] / h) + 1;
        if (nc <= max_cells) break;
        if (xf > 1) xf >>= 1;  // a table that large: give up the x refinement before growing the cells
        else h *= 1.26f;
    }
    int64_t ncells = 1;
    for (int a = 0; a < 3; a++) {
        g.org[a] = lo[a];
        g.n[a] = ((int)std::floor(ext[a] / h) + 1) * (a == 0 ? xf : 1);
        ncells *= g.n[a];
    }
    g.inv_h = 1.0f / h;
    g.inv_hx = (float)xf * g.inv_h;  // exact (power of two)
    g.h = h;
    double emax = 0;
    for (int a = 0; a < 3; a++) emax = std::max(emax, ext[a]);
    g.eps = 32.0f * FLT_EPSILON * (float)(amax + emax + h);
    g.xr = xf;
    g.xr_shift = 0;
    while ((1 << g.xr_shift) < xf) g.xr_shift++;
    g.ncells = (int)ncells;
}

// occupancy of the grid just built as the typical point sees it (cell_occupancy_kernel): the median over the points of
// the count of the cell they live in, less one (a point of a uniform cloud of q per cell sits in a cell of q + 1)
int grid_occupancy_launch(ppcr_ctx *c)
{
    if (c->nt <= 0) return PPCR_OK;
    HIP_TRY(c, c->d_occupancy.reserve(kOccBins));
    HIP_TRY(c, hipMemsetAsync(c->d_occupancy.p, 0, kOccBins * sizeof(unsigned long long), c->stream));
    cell_occupancy_kernel<<<std::min(1024, nblocks(c->grid.ncells)), kBlock, 0, c->stream>>>(c->cell_start.p, c->grid, c->d_occupancy.p);
    PPCR_TRY(check_launch(c, "cell_occupancy_kernel"));
    HIP_TRY(c, hipMemcpyAsync(c->h_occupancy, c->d_occupancy.p, kOccBins * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    return PPCR_OK;
}
// (the stream grid_occupancy_launch ran on has been synchronised)
void grid_occupancy_read(const ppcr_ctx *c, double *occ, double *occ_p99)
{
    *occ = 0;
    if (occ_p99) *occ_p99 = 0;
    if (c->nt <= 0) return;
    const unsigned long long *hist = c->h_occupancy;
    unsigned long long total = 0, run = 0;
    for (int b = 0; b < kOccBins; b++) total += hist[b];
    bool have_median = false;
    for (int b = 0; b < kOccBins; b++) {
        run += hist[b];
        if (!have_median && 2 * run >= total) {
            *occ = std::max(occ_bin_value(b) - 1.0, 0.0);
            have_median = true;
        }
        if (100 * run >= 99 * total) {  // the cell 99 % of the points do not exceed (bins of 16 above 255, saturating at ~4300)
            if (occ_p99) *occ_p99 = std::max(occ_bin_value(b) - 1.0, 0.0);
            break;
        }
    }
}
int grid_occupancy(ppcr_ctx *c, double *occ, double *occ_p99 = nullptr)
{
    PPCR_TRY(grid_occupancy_launch(c));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    grid_occupancy_read(c, occ, occ_p99);
    return PPCR_OK;
}

// The levels of a multi-level search around the base grid just built (GridLevel): COARSER ones — cell edges doubling up
// to the full radius — when the radius reaches beyond the base cells (rows of sparse regions, whose m-th neighbour lies
// farther out than the first-pass radius, are then answered by the tiled kernel on a coarser level instead of one row per
// wave), FINER ones — edges shrinking by sqrt(2), at most six — when 1 % of the points sit in cells far fuller than the first pass
// aims at (dense blobs, the near field of a scan: blocks there pick a level whose halo fits the LDS tile).  A uniform
// cloud searched with a radius of a few points' spacing keeps its single level, and with it every kernel it ran before.
void release_levels(ppcr_ctx *c)
{
    for (auto &lv : c->extra_levels) {
        lv.tgt.release();
        lv.cell_start.release();
        lv.to_base.release();
    }
    c->extra_levels.clear();
    c->n_levels = 1, c->base_level = 0, c->finest_extra = -1;
}
// (a new target or radius: the levels' buffers stay with the handle — freeing and allocating three per level cost a scan
//  of 200k points 5-7 ms per pair, most of its set-up; they are released with the handle)
void reset_levels(ppcr_ctx *c) { c->n_levels = 1, c->base_level = 0, c->finest_extra = -1; }
int build_levels(ppcr_ctx *c, bool bounded, const double *known_p99 = nullptr)
{
    reset_levels(c);
    const int n = (int)c->nt;
    if (!bounded || c->opt_two_pass != 1 || c->opt_levels == 0 || n <= 0) return PPCR_OK;
    std::vector<double> finer, coarser;
    {
        double occ = 0, occ99 = known_p99 ? *known_p99 : 0.0;
        if (!known_p99) PPCR_TRY(grid_occupancy(c, &occ, &occ99));
        const double cap = 0.1 * c->opt_first_pass_occ, q_want = std::max(target_occupancy(c->max_nb, cap, 0.1 * c->opt_first_pass_fill), 1.0);
        // (an edge shorter by sqrt(2) divides a cell's count by 2.83 in a volume but only by 2 on a SURFACE — and scans are
        //  surfaces: one finer level per factor 2 of the 99th percentile above the aim, at most six.  That is where the
        //  blocks of a scan's near field are, and a query there tests 9 h^2 rho candidates for the pi R^2 rho it needs; the
        //  coarser levels, thinly populated, stay a factor two apart)
        if (occ99 > 4.0 * q_want)  // (the dense tail that switches the levels on at all)
            for (int k = 1; k <= 6 && occ99 > std::pow(2.0, k) * q_want; k++) finer.push_back(c->search_radius / std::pow(1.41421356, k));
    }
    for (double s = c->search_radius; s < c->radius * (1.0 - 1e-9) && (int)(finer.size() + coarser.size()) < kMaxLevels - 1;) {
        s = std::min(2.0 * s, c->radius);
        coarser.push_back(s);
    }
    if (!coarser.empty()) coarser.back() = c->radius;  // the last level covers the full radius
    // Several levels only for clouds WITH a dense tail.  A uniform cloud whose radius reaches beyond the first-pass cells
    // leaves a fraction of a per cent of its rows (the cloud's edge) to nn_wide_kernel, and the multi-level kernel's
    // longer prologue (feedback byte, level table, a barrier) costs it more than those rows do: measured 7.6 k against
    // 9.7 k it/s at the command line's defaults on the uniform 200k cloud.
    if (finer.empty()) return PPCR_OK;
    std::vector<double> radii;  // ascending, the base in between
    for (auto it = finer.rbegin(); it != finer.rend(); ++it) radii.push_back(*it);
    const int base_at = (int)radii.size();
    radii.push_back(c->search_radius);
    for (double s : coarser) radii.push_back(s);
    const float r2_full = (float)(c->radius * c->radius);
    HIP_TRY(c, c->level_inv.reserve((size_t)n));
    level_inverse_kernel<<<nblocks(n), kBlock, 0, c->stream>>>(c->tgt_sorted.p, n, c->level_inv.p);
    PPCR_TRY(check_launch(c, "level_inverse_kernel"));
    std::vector<GridLevel> table(radii.size());
    if (c->extra_levels.size() < radii.size() - 1) c->extra_levels.resize(radii.size() - 1);
    size_t e = 0;
    for (size_t l = 0; l < radii.size(); l++) {
        GridLevel &t = table[l];
        std::memset(&t, 0, sizeof(t));
        if ((int)l == base_at) {
            t.g = c->grid, t.tgt = c->tgt_sorted.p, t.cell_start = c->cell_start.p, t.to_base = nullptr;
            t.r2_cap = std::min((float)(c->search_radius * c->search_radius), r2_full);  // = the first-pass r2 of associate_impl
            continue;
        }
        ppcr_ctx::ExtraLevel &lv = c->extra_levels[e];
        lv.radius = radii[l];
        make_grid_desc(n, c->tgt_lo, c->tgt_hi, lv.radius, c->opt_grid_xf, lv.g);
        HIP_TRY(c, lv.tgt.reserve((size_t)n));
        PPCR_TRY(sort_by_cell(c, lv.g, c->tgt_raw.p, n, lv.tgt.p, &lv.cell_start));
        HIP_TRY(c, lv.to_base.reserve((size_t)n));
        level_to_base_kernel<<<nblocks(n), kBlock, 0, c->stream>>>(lv.tgt.p, n, c->level_inv.p, lv.to_base.p);
        PPCR_TRY(check_launch(c, "level_to_base_kernel"));
        t.g = lv.g, t.tgt = lv.tgt.p, t.cell_start = lv.cell_start.p, t.to_base = lv.to_base.p;
        t.r2_cap = (l + 1 == radii.size()) ? r2_full : std::min((float)(lv.radius * lv.radius), r2_full);
        if (l == 0 && base_at > 0) c->finest_extra = (int)e;
        e++;
    }
    HIP_TRY(c, c->d_levels.reserve(table.size()));
    HIP_TRY(c, hipMemcpyAsync(c->d_levels.p, table.data(), table.size() * sizeof(GridLevel), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // (`table` leaves scope)
    c->n_levels = (int)radii.size();
    c->base_level = base_at;
    for (size_t l = 0; l < table.size(); l++) c->level_r2_cap[l] = table[l].r2_cap;
    return PPCR_OK;
}

// K0: bounding box -> cell edge -> cell-sorted target + cell_start.
// The cell edge follows the SEARCH radius of the first pass, radius / reach (reach = 1: the radius itself).  A bounded
// search whose radius holds far more than max_neighbours points is split in two passes (nn_wide_kernel): reach is chosen so
// that a cell holds at most ~11 points where the points are (option "first_pass_occupancy": the halo of a 256-query block
// must fit the LDS tile) while the first-pass radius still holds ~2.2 max_neighbours of them (option "first_pass_fill")
// where it can.  The estimate starts from the bounding box and
// is corrected with the occupancy measured on the grid it produced (dense blobs in a sparse box): at most three builds,
// once per (target, radius, max_neighbours).
// one build of the base grid for c->grid_search; unless it is the last one allowed, its occupancy histogram is sent on its
// way to the host behind it.  Nothing here waits for the device.
int grid_build_attempt(ppcr_ctx *c)
{
    const int n = (int)c->nt;
    c->search_radius = c->grid_search;
    c->reach = c->grid_search < c->radius ? std::min(kMaxReach, (int)std::ceil(c->radius / c->grid_search - 1e-9)) : 1;
    make_grid_desc(n, c->tgt_lo, c->tgt_hi, c->search_radius, c->opt_grid_xf, c->grid);
    PPCR_TRY(sort_by_cell(c, c->grid, c->tgt_raw.p, n, c->tgt_sorted.p, &c->cell_start));
    c->grid_occ_inflight = false;
    if (!(c->grid_bounded && c->opt_two_pass == 1) || c->grid_attempt == 2 || c->grid_search <= c->radius / kMaxReach) return PPCR_OK;
    PPCR_TRY(grid_occupancy_launch(c));
    c->grid_occ_inflight = true;
    return PPCR_OK;
}

// first half: the estimate of the first-pass radius from the bounding box, the first build enqueued on c->stream
int grid_begin(ppcr_ctx *c)
{
    invalidate_association(c);
    c->dm2_valid = false;
    const int n = (int)c->nt;  // (tgt_lo / tgt_hi: folded by ppcr_set_target, the box came with the upload)
    HIP_TRY(c, c->tgt_sorted.reserve((size_t)std::max(n, 1)));
    const bool bounded = c->max_nb > 0 && (int64_t)c->max_nb < c->nt && c->max_nb <= kEllMaxWidth;
    double search = c->radius;
    if (bounded && c->opt_two_pass && n > 0) {
        // points per cell of edge `radius` if the cloud filled its bounding box evenly (flat clouds: a slab one radius thick)
        double vol = 1;
        for (int a = 0; a < 3; a++) vol *= std::max((double)c->tgt_hi[a] - (double)c->tgt_lo[a], c->radius);
        const double per_cell = (double)n * c->radius * c->radius * c->radius / vol;
        search = c->opt_two_pass >= 2 ? c->radius / c->opt_two_pass : choose_search_radius(c->radius, per_cell, c->max_nb, 0.1 * c->opt_first_pass_occ, 0.1 * c->opt_first_pass_fill);
    }
    c->grid_bounded = bounded;
    c->grid_search = search;
    c->grid_attempt = 0;
    return grid_build_attempt(c);
}

// second half: read the occupancy the build produced, rebuild while it is far from the aim, then the levels
int grid_finish(ppcr_ctx *c)
{
    for (int a = 0; a < 3; a++) c->origin[a] = 0.5 * ((double)c->tgt_lo[a] + (double)c->tgt_hi[a]);
    c->origin_valid = true;
    bool have_occ = false;
    double occ_p99 = 0;
    while (c->grid_occ_inflight) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        double q_here = 0;  // points per cell (edge ~search) where the dense part of the cloud lives
        grid_occupancy_read(c, &q_here, &occ_p99);
        c->grid_occ_inflight = false;
        have_occ = true;  // (of the grid as it stands: build_levels need not measure it again)
        const double cap = 0.1 * c->opt_first_pass_occ, q_want = target_occupancy(c->max_nb, cap, 0.1 * c->opt_first_pass_fill);
        if (q_here <= 1.5 * std::max(q_want, cap)) break;  // close enough: keep this grid
        // the points live in fuller cells than the bounding box suggested (a cloud that does not fill its box)
        c->grid_search = std::max(c->grid_search * std::cbrt(q_want / q_here), c->radius / kMaxReach);
        c->grid_attempt++;
        have_occ = false;
        PPCR_TRY(grid_build_attempt(c));
    }
    PPCR_TRY(build_levels(c, c->grid_bounded, have_occ ? &occ_p99 : nullptr));
    c->grid_valid = true;
    c->grid_radius = c->radius;
    c->grid_max_nb = c->max_nb;
    c->src_sorted = false;  // re-sort against the new grid at the next associate()
    return PPCR_OK;
}

// a first half in flight on aux_stream: wait for it (ppcr_set_target again, an option that changes the build, destroy)
int grid_settle(ppcr_ctx *c)
{
    if (!c->grid_pending) return PPCR_OK;
    c->grid_pending = false;
    c->grid_occ_inflight = false;
    HIP_TRY(c, hipEventSynchronize(c->aux_done));
    return PPCR_OK;
}

int ensure_grid(ppcr_ctx *c)
{
    if (!c->have_tgt) return fail(c, PPCR_ERR_STATE, "target cloud not set");
    if (c->grid_valid && c->grid_radius == c->radius && c->grid_max_nb == c->max_nb) return PPCR_OK;
    if (!(c->radius > 0) || !std::isfinite(c->radius)) return fail(c, PPCR_ERR_INVALID, "radius must be positive and finite");
    if (c->grid_pending) {
        const bool usable = c->pending_radius == c->radius && c->pending_max_nb == c->max_nb;
        const bool occ = c->grid_occ_inflight;
        PPCR_TRY(grid_settle(c));
        if (usable) {
            c->grid_occ_inflight = occ;
            return grid_finish(c);
        }
    }
    PPCR_TRY(grid_begin(c));
    return grid_finish(c);
}

// One second stream per DEVICE, shared by its handles and kept for the life of the process: a second stream per handle
// cost a batch a quarter of its rate (64 handles x 250k points on one GPU: 31.6 k it/s against 42.0 k, the extra streams
// idle) — the runtime spreads streams over a few hardware queues.
int device_aux_stream(ppcr_ctx *c, hipStream_t *out)
{
    static std::mutex mu;
    static std::vector<hipStream_t> streams;
    std::lock_guard<std::mutex> lock(mu);
    if ((int)streams.size() <= c->device) streams.resize((size_t)c->device + 1, nullptr);
    if (!streams[(size_t)c->device]) HIP_TRY(c, hipStreamCreateWithFlags(&streams[(size_t)c->device], hipStreamNonBlocking));
    *out = streams[(size_t)c->device];
    return PPCR_OK;
}

// ppcr_set_target's tail: the first half of the grid build, on aux_stream, when the search is already configured
int grid_begin_early(ppcr_ctx *c)
{
    // (max_neighbours 0: the handle of an exact-association caller — ppcr_set_association, the weight updater — which never searches)
    if (!c->opt_eager_grid || !c->params_set || c->nt <= 0 || c->max_nb <= 0 || !(c->radius > 0) || !std::isfinite(c->radius)) return PPCR_OK;
    if (!c->aux_stream) PPCR_TRY(device_aux_stream(c, &c->aux_stream));
    if (!c->aux_done) HIP_TRY(c, hipEventCreateWithFlags(&c->aux_done, hipEventDisableTiming));
    hipStream_t main_stream = c->stream;
    c->stream = c->aux_stream;
    const int rc = grid_begin(c);
    c->stream = main_stream;
    const hipError_t e = hipEventRecord(c->aux_done, c->aux_stream);
    if (rc != PPCR_OK || e != hipSuccess) {
        (void)hipStreamSynchronize(c->aux_stream);
        c->grid_occ_inflight = false;
        if (rc != PPCR_OK) return rc;
        HIP_TRY(c, e);
    }
    c->grid_pending = true;
    c->pending_radius = c->radius;
    c->pending_max_nb = c->max_nb;
    return PPCR_OK;
}

int ensure_source_sorted(ppcr_ctx *c)
{
    if (!c->have_src) return fail(c, PPCR_ERR_STATE, "source cloud not set");
    if (c->src_sorted || !c->opt_sort_source || c->ns == 0) return PPCR_OK;
    invalidate_association(c);
    HIP_TRY(c, c->src_alt.reserve((size_t)c->ns));
    // (multi-level searches: the Hilbert curve over the finest level's cells, see hilbert_key_kernel)
    // — only when there ARE finer levels (dense parts, whose blocks must be compact at a fine scale): the curve's blocks have
    // larger bounding boxes than bricks (K1 58 against 51 us on a uniform cloud), which coarser levels alone do not need
    if (c->n_levels > 1 && c->finest_extra >= 0 && c->opt_sort_source == 1)
        PPCR_TRY(sort_by_cell(c, c->extra_levels[(size_t)c->finest_extra].g, c->src.p, (int)c->ns, c->src_alt.p, nullptr, 2));
    else
        PPCR_TRY(sort_by_cell(c, c->grid, c->src.p, (int)c->ns, c->src_alt.p, nullptr, c->opt_sort_source == 1 ? 1 : 0));
    std::swap(c->src, c->src_alt);
    c->src_sorted = true;
    c->dm2_valid = false;  // row order changed
    c->split_clean = false;
    c->level_cap_clean = false;
    return PPCR_OK;
}

constexpr int kAccumRows = 1;     // rows per lane of accumulate_ell_kernel
constexpr int kAccumBlock = 256;  // threads per block of accumulate_ell_kernel (fewer partial vectors to fold)

// which form of K23 serves a model: the two models the reference's CLI reaches by default are compiled in (Gaussian -u;
// t with dof 5, dim 3: v + dim = 8), any other dof takes the run-time form of the same arithmetic.  The one-pass form
// (likelihoods relative to s = 0 instead of the row's smallest s) only while that ratio stays far from underflow for
// every s the association can hold (s < radius^2 up to the float rounding of d2 and the f64 re-evaluation at another
// pose; the factor 4 on the radius covers residuals at a pose other than the one the association was made at).
struct K23Form {
    int tm;        // 0 Gaussian, 8 t with v + dim = 8, -3 t with another integer v + dim (one-pass only), -1 run-time t model
    bool onepass;
};
K23Form k23_form(const ppcr_ctx *c, const Model &md)
{
    const double s_max = 16.0 * c->radius * c->radius;
    if (md.is_normal) return K23Form{0, 0.5 * s_max < 600.0};
    // (v / (v + s))^((v + d) / 2) > 1e-250
    const bool safe = 0.5 * md.vpd * std::log10((md.v + s_max) / md.v) < 250.0;
    if (md.vpd_int == 8) return K23Form{8, safe};
    if (md.vpd_int != 0 && safe) return K23Form{-3, true};  // -d 3, -d 10, ...: integer power by squaring
    return K23Form{-1, safe};
}

template <int W>
void launch_accumulate_ell(ppcr_ctx *c, int nb, const Pose &P, const Model &md, const LoopState *loop_st)
{
#define PPCR_K23(TMc, ONEc)                                                                                        \
    accumulate_ell_kernel<W, kAccumRows, kAccumBlock, TMc, ONEc><<<nb, kAccumBlock, 0, c->stream>>>(        \
        c->nbr.p, c->cnt.p, c->src.p, c->tgt_cur(), (int)c->ns, P, md, c->partials.p, c->ell_width, loop_st)
    const K23Form f = k23_form(c, md);
    if (f.tm == 0) {
        if (f.onepass) PPCR_K23(0, true);
        else PPCR_K23(0, false);
    } else if (f.tm == 8) {
        if (f.onepass) PPCR_K23(8, true);
        else PPCR_K23(8, false);
    } else if (f.tm == -3) {
        PPCR_K23(-3, true);
    } else {
        if (f.onepass) PPCR_K23(-1, true);
        else PPCR_K23(-1, false);
    }
#undef PPCR_K23
}

int flush_pending_move(ppcr_ctx *c);

// the K1 translation unit compiled for the narrowest list width that holds max_neighbours
void dispatch_tile(TileLaunch &tl, int m)
{
    if (m <= 4) launch_tile_m4(tl);
    else if (m <= 5) launch_tile_m5(tl);
    else if (m <= 8) launch_tile_m8(tl);
    else if (m <= 10) launch_tile_m10(tl);
    else if (m <= 16) launch_tile_m16(tl);
    else if (m <= 20) launch_tile_m20(tl);
    else launch_tile_m32(tl);
}

// Small clouds leave most of the chip idle with one workgroup per 256 queries (100k points: 391 workgroups on 1280 resident
// slots) and an iteration lasts as long as ONE workgroup's dependent chain.  Scanning every block as two half-blocks
// (SplitTable::all_halves; option k1_halves) was built to shorten that chain and MEASURED NEUTRAL TO NEGATIVE: K1 26.3 us
// with halves against 25.9 us whole at 100k, 23.3 / 22.7 at 50k, 47 / 38 at 250k (docs/experiments.md, round 4) — the
// chain is the workgroup's dependent memory round trips and LDS latencies, not its scan work.  Off by default.
bool k1_all_halves(const ppcr_ctx *c)
{
    if (c->opt_k1_halves >= 0) return c->opt_k1_halves != 0;
    return nblocks(std::max<int64_t>(c->ns, 1), 256) <= 640;
}
// workgroups of a steady-state K1 launch = slots of the partial sums it leaves when K23 is folded in
int k1_steady_slots(const ppcr_ctx *c) { return steady_grid(nblocks(std::max<int64_t>(c->ns, 1), 256), k1_all_halves(c)); }

// K1 (or the generic count/scan/fill path for unbounded searches and max_neighbours > 32).
// fuse_R / fuse_t (nullable): the pose the first IRLS half-step will be evaluated at; when given, the steady-state K1
// also produces that step's partial moments (c->assoc_fused, c->fused_slots) and the caller skips the K23 launch.
struct StepTicket {
    unsigned seq = 0;  // mailbox sequence number (mailbox path)
    int nb = 0;        // partial vectors to fold (copy path)
};
int prepare_fold(ppcr_ctx *c, int nslots, StepTicket &tk, FoldSolve &fs, const LoopCtl *loop);
// merge_tk (with a fuse pose): the fold-and-solve step may ride in the cleanup launch; c->assoc_folded tells whether it did
// loop (device-paced align loop only): the launches step aside while LoopState::abort is up, the fast kernel opens a new
// outer iteration in the loop state, a merged fold decides about the inner loop
int associate_impl(ppcr_ctx *c, const Mat3 *fuse_R = nullptr, const double *fuse_t = nullptr, StepTicket *merge_tk = nullptr,
                   const LoopCtl *loop = nullptr)
{
    PPCR_TRY(ensure_grid(c));
    if (!c->src_sorted) PPCR_TRY(flush_pending_move(c));  // the one-time spatial sort reads the source
    PPCR_TRY(ensure_source_sorted(c));
    invalidate_association(c);
    const int ns = (int)c->ns;
    const float r2_full = (float)(c->radius * c->radius);  // PCL: static_cast<float>(radius * radius)
    // first pass of a two-pass search: a smaller radius on a grid built for it (see ensure_grid); one pass: the radius
    const float r2 = c->reach > 1 ? std::min((float)(c->search_radius * c->search_radius), r2_full) : r2_full;
    const bool unbounded = (c->max_nb <= 0 || (int64_t)c->max_nb >= c->nt);
    const bool tiled = !unbounded && c->max_nb <= kEllMaxWidth && ns > 0;
    PendingMove pm;
    std::memset(&pm, 0, sizeof(pm));
    if (c->move_on_device && tiled) {
        // the previous iteration's transform is still on its way to the host: K1 reads it from device memory
        pm.enabled = 2;
        pm.dev = c->d_pose.p;
        c->move_on_device = false;
    } else if (c->move_pending && tiled) {
        // the deferred source move rides in this kernel's prologue
        pm.enabled = 1;
        for (int a = 0; a < 3; a++) {
            for (int b = 0; b < 3; b++) pm.P.R[3 * a + b] = c->pending_T[4 * a + b];
            pm.P.t[a] = c->pending_T[4 * a + 3];
            pm.P.c[a] = 0;
        }
        c->move_pending = false;
    } else {
        PPCR_TRY(flush_pending_move(c));
    }
    if (!unbounded && c->max_nb <= kEllMaxWidth) {
        const int m = c->max_nb;
        HIP_TRY(c, c->nbr.reserve((size_t)m * (size_t)std::max(ns, 1)));
        HIP_TRY(c, c->cnt.reserve((size_t)std::max(ns, 1)));
        HIP_TRY(c, c->dm2.reserve((size_t)std::max(ns, 1)));
        HIP_TRY(c, c->ovf_list.reserve((size_t)std::max(nblocks(std::max(ns, 1), 256) + kMaxSplit, steady_grid(nblocks(std::max(ns, 1), 256), true))));
        if (!c->split_clean) {
            const size_t nbk = (size_t)nblocks(std::max(ns, 1), 256);
            HIP_TRY(c, c->split_flag.reserve(nbk));
            HIP_TRY(c, c->split_list.reserve(kMaxSplit));
            HIP_TRY(c, c->split_state.reserve(3));
            HIP_TRY(c, hipMemsetAsync(c->split_flag.p, 0, nbk, c->stream));
            HIP_TRY(c, hipMemsetAsync(c->split_state.p, 0, 3 * sizeof(unsigned), c->stream));
            c->split_clean = true;
        }
        if (!c->ovf_state.p) {
            HIP_TRY(c, c->ovf_state.reserve(2));
            HIP_TRY(c, hipMemsetAsync(c->ovf_state.p, 0, 2 * sizeof(unsigned), c->stream));
        }
        // dm2 is only trusted when the source moved by nothing but the deferred rigid move applied in this
        // very kernel since the association that wrote it
        if (!tiled) c->dm2_valid = false;
        c->assoc_fused = false;
        if (ns > 0) {
            // K23 folded in: only the two compiled-in models in their one-pass form (see launch_accumulate_ell)
            FusedMoments fm;
            const FusedMoments *fuse = nullptr;
            int fuse_tm = -2;
            // (two passes: the rows are final only after the second; a cloud whose associations keep handing workgroups over
            //  — dense parts whose halos outgrow the LDS tile — leaves those rows to nn_wide_kernel, which does not fold)
            const unsigned ovf_known = c->ovf_decide_pinned ? c->ovf_decide : c->ovf_last;
            const bool many_handed_over = ovf_known != ~0u && ovf_known > (unsigned)c->opt_fuse_max_handed_over;
            if (fuse_R && c->opt_fuse_k23 && c->nt > 0 && c->reach == 1 && c->n_levels <= 1 && !many_handed_over) {
                const Model md = make_model(c);
                const K23Form form = k23_form(c, md);
                if (form.onepass && form.tm != -1) {  // the three forms compiled into K1: Gaussian, v + dim = 8, integer v + dim
                    const int slots = k1_steady_slots(c);
                    HIP_TRY(c, c->partials.reserve((size_t)slots * kNSums));
                    fm.P = make_pose(c, *fuse_R, fuse_t);
                    fm.md = md;
                    fuse_tm = form.tm;
                    fm.partials = c->partials.p;
                    fm.nslots = slots;
                    fuse = &fm;
                }
            }
            bool fused = false, merged = false;
            // the fold-and-solve step in the same launch as the cleanup: only worth preparing when K23 will be folded
            // in, i.e. for the steady-state variant (dm2 valid, short lists, m <= 12)
            FoldSolve fs;
            const FoldSolve *fold = nullptr;
            // (also when several handles share the GPU: before the source-order fix of round 3 the merged launch lost there,
            //  24.0 k it/s against 28.2 k with the small fold kernel for 64 x 250k, 8 in flight; now 41.5 k against 41.0 k,
            //  and 1.33 k against 0.88 k pairs/s end to end with four pairs in flight)
            // (m <= 10: the widths that HAVE a steady-state variant are 4, 5, 8 and 10 — launch_tile<M> with M <= 12)
            const bool steady_next = fuse && c->opt_merge_fold && c->opt_mailbox && c->opt_temporal && c->dm2_valid &&
                                     c->opt_short_lists && m <= 10 && !c->opt_stamps;
            if (merge_tk && steady_next) {
                PPCR_TRY(prepare_fold(c, fm.nslots, *merge_tk, fs, loop));
                fold = &fs;
            }
            ProfScope ps(c, K_NN_TOPM);
            c->ovf_parity ^= 1;
            TileLaunch tl{};
            tl.stream = c->stream;
            tl.src = c->src.p, tl.ns = (int)c->ns, tl.tgt = c->tgt_sorted.p, tl.cell_start = c->cell_start.p, tl.grid = c->grid;
            tl.r2 = r2, tl.m = m;
            tl.reach = c->reach, tl.r2_full = r2_full;
            tl.levels = c->n_levels > 1 ? c->d_levels.p : nullptr, tl.n_levels = c->n_levels, tl.base_level = c->base_level;
            if (c->n_levels > 1) {
                const size_t nbk = (size_t)nblocks(std::max(ns, 1), 256);
                if (!c->level_cap_clean || c->level_cap.cap < nbk) {
                    HIP_TRY(c, c->level_cap.reserve(nbk));
                    HIP_TRY(c, hipMemsetD16Async(reinterpret_cast<hipDeviceptr_t>(c->level_cap.p), 0x000F, nbk, c->stream));  // cap 15, floor 0, whole
                    c->level_cap_clean = true;
                }
                tl.level_cap = c->level_cap.p;
                for (int l = 0; l < kMaxLevels; l++) tl.r2_cap[l] = c->level_r2_cap[l];
                tl.level_dbg = (c->opt_level_stats && c->level_dbg.p) ? c->level_dbg.p : nullptr;
            }
            // who redoes the rows of handed-over workgroups: the cleanup role of the second launch when K23 is folded in (it
            // folds K23 for them as well, but walks a dense neighbourhood one candidate per lane at a time), nn_wide_kernel
            // otherwise — every launch that cannot fold, every two-pass search (an idle nn_wide_kernel costs what an idle
            // cleanup launch costs, and the choice does not depend on when a hand-over count reaches the host)
            // (K23 can only be folded in by the steady-state variant: widths up to 10, a valid cut-off, short lists)
            const bool may_fuse = fuse != nullptr && m <= 10 && c->opt_temporal && c->dm2_valid && c->opt_short_lists && !c->opt_stamps;
            if (c->reach > 1 || c->n_levels > 1 || !may_fuse) {
                if (c->d_short.cap < (size_t)ns + 3) {
                    HIP_TRY(c, hipStreamSynchronize(c->stream));
                    HIP_TRY(c, c->d_short.reserve((size_t)ns + 3));
                    HIP_TRY(c, hipMemsetAsync(c->d_short.p, 0, 3 * sizeof(int), c->stream));
                }
                // (the list's counters alternate with the hand-over counters: ovf_parity, rolled back with recalled trains)
                tl.short_count = reinterpret_cast<unsigned *>(c->d_short.p) + c->ovf_parity, tl.short_list = c->d_short.p + 3;
                tl.short_seen = reinterpret_cast<unsigned *>(c->d_short.p) + 2;
            }
            if (c->d_short.p) tl.short_next = reinterpret_cast<unsigned *>(c->d_short.p) + (c->ovf_parity ^ 1);
            tl.nbr = c->nbr.p, tl.cnt = c->cnt.p, tl.dm2 = c->dm2.p;
            tl.dm2_in = (c->opt_temporal && c->dm2_valid) ? 1 : 0;
            tl.short_lists = c->opt_short_lists;
            tl.all_halves = k1_all_halves(c) ? 1 : 0;
            if (c->opt_stamps) {
                // sized from the grid this launch really has (the steady-state variant adds kMaxSplit workgroups in front):
                // 8 words per wave, then one word per lane (its sorted run lengths)
                // (a multi-level search launches two workgroups per block; the reader sums the records of exactly this grid)
                const size_t nwg = c->n_levels > 1 ? (size_t)steady_grid(nblocks(ns, 256), true) : (size_t)std::max(nblocks(ns, 256) + kMaxSplit, k1_steady_slots(c));
                const size_t nst = (nwg * (kBlock / 64) + 64) * 8 + nwg * 256;
                if (c->d_stamps.cap < nst) {
                    HIP_TRY(c, hipStreamSynchronize(c->stream));
                    HIP_TRY(c, c->d_stamps.reserve(nst));
                }
                HIP_TRY(c, hipMemsetAsync(c->d_stamps.p, 0, nst * sizeof(unsigned long long), c->stream));
                c->stamps_wgs = nwg;
            }
            tl.stamps = c->opt_stamps ? c->d_stamps.p : nullptr;
            tl.ovf_list = c->ovf_list.p;
            tl.ovf_now = c->ovf_state.p + c->ovf_parity, tl.ovf_next = c->ovf_state.p + (c->ovf_parity ^ 1);
            tl.quiet = c->ovf_last == 0;
            tl.split_flag = c->split_flag.p, tl.split_list = c->split_list.p, tl.split_state = c->split_state.p;
            tl.pm = pm, tl.fuse = fuse, tl.fold = fold;
            tl.fuse_tm = fuse_tm;
            tl.loop_st = loop ? loop->st : nullptr;
            tl.between = [](void *scope) { static_cast<ProfScope *>(scope)->split(K_NN_CLEANUP); };
            tl.between2 = [](void *scope) { static_cast<ProfScope *>(scope)->split(K_NN_WIDE); };
            tl.between_arg = &ps;
            dispatch_tile(tl, m);
            fused = tl.fused, merged = tl.merged;
            c->assoc_fused = fused;
            c->assoc_folded = merged;
            c->fused_slots = fused ? fm.nslots : 0;
            if (fold && !merged) {
                // (cannot happen while steady_next mirrors launch_tile's dispatch; were it to, the sequence number
                //  drawn for the fold is handed back and the caller launches the fold on its own)
                c->mbox_seq--;
            }
        }
        c->assoc_space = 1;
        PPCR_TRY(check_launch(c, "nn_fast_kernel"));
        c->dm2_valid = tiled;
        c->assoc = ppcr_ctx::ASSOC_ELL;
        c->ell_width = m;
        return PPCR_OK;
    }
    // generic path: count -> scan -> fill [-> select + compact]
    c->assoc_space = 1;
    HIP_TRY(c, c->gen_counts.reserve((size_t)ns + 1));
    HIP_TRY(c, c->gen_row_ptr.reserve((size_t)ns + 1));
    HIP_TRY(c, hipMemsetAsync(c->gen_counts.p, 0, sizeof(int) * ((size_t)ns + 1), c->stream));
    if (ns > 0) {
        ProfScope ps(c, K_NN_COUNT);
        nn_count_kernel<<<nblocks(ns), kBlock, 0, c->stream>>>(c->src.p, ns, c->tgt_sorted.p, c->cell_start.p,
                                                               c->grid, r2, c->gen_counts.p);
    }
    PPCR_TRY(check_launch(c, "nn_count_kernel"));
    auto scan = [&](int *in, int *out) -> int {
        size_t tmp_bytes = 0;
        HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, in, out, ns + 1, c->stream));
        HIP_TRY(c, c->cub_tmp.reserve(tmp_bytes + 16));
        ProfScope ps(c, K_NN_SCAN);
        HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(c->cub_tmp.p, tmp_bytes, in, out, ns + 1, c->stream));
        return PPCR_OK;
    };
    // guard int32 CSR offsets: total in-radius pairs must fit
    HIP_TRY(c, c->d_total.reserve(1));
    HIP_TRY(c, hipMemsetAsync(c->d_total.p, 0, sizeof(unsigned long long), c->stream));
    if (ns > 0) {
        ProfScope ps(c, K_COUNT_SUM);
        ell_count_sum_kernel<<<std::min(1024, nblocks(ns)), kBlock, 0, c->stream>>>(c->gen_counts.p, ns, c->d_total.p);
    }
    HIP_TRY(c, hipMemcpyAsync(c->h_total, c->d_total.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const unsigned long long total_all = *c->h_total;
    if (total_all > (unsigned long long)INT32_MAX - 1024)
        return fail(c, PPCR_ERR_INVALID, "association too large for 32-bit CSR offsets (reduce radius or set max_neighbours)");
    PPCR_TRY(scan(c->gen_counts.p, c->gen_row_ptr.p));
    HIP_TRY(c, c->gen_keys.reserve((size_t)std::max<unsigned long long>(total_all, 1)));
    HIP_TRY(c, c->gen_pos.reserve((size_t)std::max<unsigned long long>(total_all, 1)));
    if (ns > 0) {
        ProfScope ps(c, K_NN_FILL);
        nn_fill_kernel<<<nblocks(ns), kBlock, 0, c->stream>>>(c->src.p, ns, c->tgt_sorted.p, c->cell_start.p, c->grid,
                                                              r2, c->gen_row_ptr.p, c->gen_keys.p, c->gen_pos.p);
    }
    PPCR_TRY(check_launch(c, "nn_fill_kernel"));
    if (unbounded) {
        std::swap(c->row_ptr, c->gen_row_ptr);
        std::swap(c->nbr, c->gen_pos);
        c->nnz = (int64_t)total_all;
    } else {
        HIP_TRY(c, hipMemsetAsync(c->gen_counts.p, 0, sizeof(int) * ((size_t)ns + 1), c->stream));
        if (ns > 0) {
            ProfScope ps(c, K_NN_SELECT);
            nn_select_kernel<<<nblocks(ns), kBlock, 0, c->stream>>>(ns, c->gen_row_ptr.p, c->gen_keys.p, c->gen_pos.p,
                                                                    c->max_nb, c->gen_counts.p);
        }
        PPCR_TRY(check_launch(c, "nn_select_kernel"));
        HIP_TRY(c, c->row_ptr.reserve((size_t)ns + 1));
        PPCR_TRY(scan(c->gen_counts.p, c->row_ptr.p));
        HIP_TRY(c, c->nbr.reserve((size_t)std::max<unsigned long long>(total_all, 1)));
        if (ns > 0) {
            ProfScope ps(c, K_NN_COMPACT);
            csr_compact_kernel<<<nblocks(ns), kBlock, 0, c->stream>>>(ns, c->gen_row_ptr.p, c->gen_pos.p, c->row_ptr.p,
                                                                      c->nbr.p);
        }
        PPCR_TRY(check_launch(c, "csr_compact_kernel"));
        c->nnz = -1;
    }
    c->assoc = ppcr_ctx::ASSOC_CSR;
    return PPCR_OK;
}

int ensure_nnz(ppcr_ctx *c)
{
    if (c->assoc == ppcr_ctx::ASSOC_NONE) return fail(c, PPCR_ERR_STATE, "no association (call ppcr_associate or ppcr_set_association)");
    if (c->nnz >= 0) return PPCR_OK;
    const int ns = (int)c->ns;
    if (c->assoc == ppcr_ctx::ASSOC_CSR) {
        int last = 0;
        HIP_TRY(c, hipMemcpyAsync(&last, c->row_ptr.p + ns, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->nnz = last;
        return PPCR_OK;
    }
    HIP_TRY(c, c->d_total.reserve(1));
    HIP_TRY(c, hipMemsetAsync(c->d_total.p, 0, sizeof(unsigned long long), c->stream));
    if (ns > 0) {
        ProfScope ps(c, K_COUNT_SUM);
        ell_count_sum_kernel<<<std::min(1024, nblocks(ns)), kBlock, 0, c->stream>>>(c->cnt.p, ns, c->d_total.p);
    }
    PPCR_TRY(check_launch(c, "ell_count_sum_kernel"));
    HIP_TRY(c, hipMemcpyAsync(c->h_total, c->d_total.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->nnz = (int64_t)*c->h_total;
    return PPCR_OK;
}

// download the w lane (original index) of a float4 cloud
int download_order(ppcr_ctx *c, const float4 *pts, int64_t n, std::vector<int> &order)
{
    std::vector<float4> tmp((size_t)n);
    if (n > 0) {
        HIP_TRY(c, hipMemcpyAsync(tmp.data(), pts, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    order.resize((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        int v;
        std::memcpy(&v, &tmp[(size_t)i].w, sizeof(int));
        order[(size_t)i] = v;
    }
    return PPCR_OK;
}

// Build the CSR view (original row order, ascending original columns) of a device-made association
int build_csr_cache(ppcr_ctx *c)
{
    if (c->csr_cache_valid) return PPCR_OK;
    PPCR_TRY(ensure_nnz(c));
    const int64_t ns = c->ns;
    std::vector<int> src_order, tgt_order;
    PPCR_TRY(download_order(c, c->src.p, ns, src_order));
    PPCR_TRY(download_order(c, c->tgt_cur(), c->nt, tgt_order));
    std::vector<int> h_nbr, h_cnt, h_rp;
    if (c->assoc == ppcr_ctx::ASSOC_ELL) {
        h_nbr.resize((size_t)c->ell_width * (size_t)ns);
        h_cnt.resize((size_t)ns);
        if (ns > 0) {
            HIP_TRY(c, hipMemcpyAsync(h_nbr.data(), c->nbr.p, h_nbr.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipMemcpyAsync(h_cnt.data(), c->cnt.p, h_cnt.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
    } else {
        h_rp.resize((size_t)ns + 1);
        h_nbr.resize((size_t)c->nnz);
        HIP_TRY(c, hipMemcpyAsync(h_rp.data(), c->row_ptr.p, h_rp.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        if (c->nnz > 0)
            HIP_TRY(c, hipMemcpyAsync(h_nbr.data(), c->nbr.p, h_nbr.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    auto row_count = [&](int64_t r) { return c->assoc == ppcr_ctx::ASSOC_ELL ? h_cnt[(size_t)r] : h_rp[(size_t)r + 1] - h_rp[(size_t)r]; };
    auto slot_of = [&](int64_t r, int k) -> size_t {
        return c->assoc == ppcr_ctx::ASSOC_ELL ? (size_t)k * (size_t)ns + (size_t)r : (size_t)h_rp[(size_t)r] + (size_t)k;
    };
    c->h_csr_row_ptr.assign((size_t)ns + 1, 0);
    for (int64_t r = 0; r < ns; r++) c->h_csr_row_ptr[(size_t)src_order[(size_t)r] + 1] = row_count(r);
    for (int64_t i = 0; i < ns; i++) c->h_csr_row_ptr[(size_t)i + 1] += c->h_csr_row_ptr[(size_t)i];
    const size_t nnz = (size_t)c->h_csr_row_ptr[(size_t)ns];
    c->h_csr_col.resize(nnz);
    c->h_csr_slot.resize(nnz);
    std::vector<std::pair<int, size_t>> row;
    for (int64_t r = 0; r < ns; r++) {
        const int n = row_count(r);
        row.clear();
        for (int k = 0; k < n; k++) {
            const size_t sl = slot_of(r, k);
            row.emplace_back(tgt_order[(size_t)h_nbr[sl]], sl);
        }
        std::sort(row.begin(), row.end());
        const size_t base = (size_t)c->h_csr_row_ptr[(size_t)src_order[(size_t)r]];
        for (int k = 0; k < n; k++) {
            c->h_csr_col[base + k] = row[(size_t)k].first;
            c->h_csr_slot[base + k] = row[(size_t)k].second;
        }
    }
    c->csr_cache_valid = true;
    return PPCR_OK;
}

// One IRLS half-step on the device: K23 (weights + moments at the pose R, t) -> fold -> closed-form solve.
struct StepResult {
    double sums[PPCR_NSUMS];
    double T[12];      // minimiser for these moments
    double cost;       // 0.5 * sum w |y - R x - t|^2 at T (Ceres' convention)
    bool degenerate;   // no weight mass: T = identity
    // device-paced inner loop (MailboxStatus other than kStepResult)
    unsigned status = kStepResult;
    int steps = 0;          // IRLS steps the device took in this outer iteration
    double cost_init = 0;   // 0.5 * sum w s at the pose the outer iteration started from
};
// buffers and arguments of one fold-and-solve step; draws the step's mailbox sequence number
int prepare_fold(ppcr_ctx *c, int nslots, StepTicket &tk, FoldSolve &fs, const LoopCtl *loop)
{
    HIP_TRY(c, c->d_sums.reserve(kNSums));
    HIP_TRY(c, c->d_pose.reserve(1));
    if (!c->d_ticket.p) {
        HIP_TRY(c, c->d_ticket.reserve(2));
        HIP_TRY(c, hipMemsetAsync(c->d_ticket.p, 0, 2 * sizeof(unsigned), c->stream));
    }
    tk.seq = ++c->mbox_seq;
    tk.nb = nslots;
    fs.partials = c->partials.p;
    fs.nslots = nslots;
    fs.sums = c->d_sums.p;
    fs.origin = make_double3(c->origin[0], c->origin[1], c->origin[2]);
    fs.pose_out = c->d_pose.p;
    fs.mbox = c->d_mbox + (tk.seq % kMailboxRing);
    fs.ticket = c->d_ticket.p;
    fs.seq = tk.seq;
    fs.handed_over = c->ovf_state.p ? c->ovf_state.p + c->ovf_parity : nullptr;
    fs.split_list = c->split_clean ? c->split_list.p : nullptr;
    fs.split_flag = c->split_clean ? c->split_flag.p : nullptr;
    fs.split_total = c->split_clean ? c->split_state.p : nullptr;
    fs.split_visible = c->split_clean ? c->split_state.p + 1 : nullptr;
    fs.split_rebuilt = c->split_clean ? c->split_state.p + 2 : nullptr;
    fs.split_nblocks = nblocks(std::max((int)c->ns, 1), 256);
    fs.dbg = c->opt_fold_stamps ? c->d_fold_dbg.p : nullptr;
    if (loop) {
        fs.loop = *loop;
    } else {
        fs.loop.st = nullptr;  // host-paced: every step is published, the host decides
        fs.loop.f_tol = 0.0;
        fs.loop.max_steps = 0;
        fs.loop.first = fs.loop.last_dev = 0;
    }
    return PPCR_OK;
}

// enqueue K23 + fold (+ solve); nothing here waits for the device.  use_fused: the association just made already
// produced the partial moments for this pose (associate_impl with a fuse pose): only the fold and the solve remain.
int launch_step(ppcr_ctx *c, const Mat3 &R, const double t[3], StepTicket &tk, bool use_fused = false,
                const LoopCtl *loop = nullptr)
{
    if (c->assoc == ppcr_ctx::ASSOC_NONE) return fail(c, PPCR_ERR_STATE, "no association (call ppcr_associate or ppcr_set_association)");
    if (!c->origin_valid) {
        // set_association path without a grid: origin = 0 is fine for the exact-association API,
        // but prefer the target bounding-box centre when a grid has been built
        c->origin[0] = c->origin[1] = c->origin[2] = 0;
    }
    const int ns = (int)c->ns;
    const Pose P = make_pose(c, R, t);
    const Model md = make_model(c);
    const bool ell_rows = c->assoc == ppcr_ctx::ASSOC_ELL && c->nt > 0;
    // the ELL kernel covers every row exactly once (kAccumRows rows per lane); the generic one grid-strides
    const int nb = use_fused ? c->fused_slots
                             : (ell_rows ? std::max(1, nblocks(ns, kAccumBlock * kAccumRows)) : std::max(1, std::min(kAccumMaxBlocks, nblocks(ns))));
    HIP_TRY(c, c->partials.reserve((size_t)nb * kNSums));
    HIP_TRY(c, c->d_sums.reserve(kNSums));
    const LoopState *loop_st = loop ? loop->st : nullptr;
    if (!use_fused) {
        ProfScope ps(c, K_ACCUMULATE);
        if (c->assoc == ppcr_ctx::ASSOC_ELL && c->nt > 0) {
            const int w = c->ell_width;
            if (w <= 4) launch_accumulate_ell<4>(c, nb, P, md, loop_st);
            else if (w <= 8) launch_accumulate_ell<8>(c, nb, P, md, loop_st);
            else if (w <= 10) launch_accumulate_ell<10>(c, nb, P, md, loop_st);
            else if (w <= 16) launch_accumulate_ell<16>(c, nb, P, md, loop_st);
            else if (w <= 20) launch_accumulate_ell<20>(c, nb, P, md, loop_st);
            else launch_accumulate_ell<32>(c, nb, P, md, loop_st);
        } else if (c->assoc == ppcr_ctx::ASSOC_ELL) {
            EllAssoc a{c->nbr.p, c->cnt.p, ns};
            accumulate_kernel<EllAssoc><<<nb, kBlock, 0, c->stream>>>(a, c->src.p, c->tgt_cur(), ns, P, md, c->partials.p);
        } else {
            CsrAssoc a{c->nbr.p, c->row_ptr.p};
            accumulate_kernel<CsrAssoc><<<nb, kBlock, 0, c->stream>>>(a, c->src.p, c->tgt_cur(), ns, P, md, c->partials.p);
        }
    }
    PPCR_TRY(check_launch(c, "accumulate_kernel"));
    tk.nb = nb;
    if (c->opt_mailbox) {
        // fold + solve on the device; moments, transform and cost arrive in the host mailbox ring
        FoldSolve fs;
        PPCR_TRY(prepare_fold(c, nb, tk, fs, loop));
        const auto tl0 = std::chrono::steady_clock::now();
        {
            ProfScope ps(c, K_REDUCE);
            reduce_solve_kernel<<<kNSums, kBlock, 0, c->stream>>>(fs);
        }
        PPCR_TRY(check_launch(c, "reduce_solve_kernel"));
        c->dbg_host[6] = std::max(c->dbg_host[6], std::chrono::duration<double>(std::chrono::steady_clock::now() - tl0).count());
        return PPCR_OK;
    }
    {
        ProfScope ps(c, K_REDUCE);
        reduce_partials_kernel<<<kNSums, kBlock, 0, c->stream>>>(c->partials.p, nb, c->d_sums.p);
    }
    return check_launch(c, "reduce_partials_kernel");
}

// has that step's result reached the host mailbox yet?  (mailbox path only; never blocks)
bool step_arrived(const ppcr_ctx *c, const StepTicket &tk)
{
    const HostMailbox *mb = c->h_mbox + (tk.seq % kMailboxRing);
    return __atomic_load_n(&mb->seq, __ATOMIC_ACQUIRE) == tk.seq;
}

// wait for that step and fetch its result
int collect_step(ppcr_ctx *c, const StepTicket &tk, StepResult &out)
{
    if (c->opt_mailbox) {
        const HostMailbox *mb = c->h_mbox + (tk.seq % kMailboxRing);
        volatile const unsigned *flag = &mb->seq;
        bool arrived = false;
        const auto tw0 = std::chrono::steady_clock::now();
        for (long spin = 0; spin < 200000000L; spin++) {  // ~ seconds; a fault on the device ends up below
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == tk.seq) {
                arrived = true;
                break;
            }
            // a short wait is spun through (an iteration lasts ~0.1 ms); past that the core is offered to whoever
            // else wants it, so many handles driven from more threads than cores do not starve each other
            if (spin > 4096) std::this_thread::yield();
            // (a device fault must not take seconds to surface: look at the stream every 2^16 polls, a few milliseconds)
            if ((spin & 0xFFFF) == 0xFFFF && hipStreamQuery(c->stream) != hipErrorNotReady) {
                arrived = __atomic_load_n(flag, __ATOMIC_ACQUIRE) == tk.seq;
                break;
            }
        }
        if (!arrived) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != tk.seq) return fail(c, PPCR_ERR_HIP, "moment mailbox never arrived");
        }
        for (int j = 0; j < kNSums; j++) out.sums[j] = mb->sums[j];
        for (int j = 0; j < 12; j++) out.T[j] = mb->T[j];
        out.cost = mb->cost;
        out.degenerate = mb->degenerate != 0;
        out.status = mb->status;
        out.steps = mb->steps;
        out.cost_init = mb->cost_init;
        if (out.status != kLaunchSkipped) {
            c->ovf_last = mb->handed_over;
            c->dbg_host[7] += (double)mb->handed_over;  // workgroups the associations of this ppcr_align handed to the cleanup kernel
        }
        const double w = std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count();
        c->dbg_host[0] += 1;
        c->dbg_host[1] += w;
        c->dbg_host[2] = std::max(c->dbg_host[2], w);
        return PPCR_OK;
    }
    // copy path: moments back, solve on the host (same source as the device solve)
    HIP_TRY(c, hipMemcpyAsync(c->h_sums, c->d_sums.p, sizeof(double) * kNSums, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::memcpy(out.sums, c->h_sums, sizeof(double) * kNSums);
    const RigidSolve rs = solve_rigid_from_moments(out.sums, c->origin);
    pack_T(rs.R, rs.t, out.T);
    out.degenerate = rs.degenerate;
    out.cost = rs.degenerate ? 0.5 * out.sums[16] : cost_from_moments(out.sums, c->origin, rs.R, rs.t);
    return PPCR_OK;
}

// buffers of the device-paced loop (once per handle)
int ensure_loop_state(ppcr_ctx *c)
{
    if (!c->d_loop.p) {
        HIP_TRY(c, c->d_loop.reserve(1));
        HIP_TRY(c, hipMemsetAsync(c->d_loop.p, 0, sizeof(LoopState), c->stream));
    }
    {
        // sized for the largest inner-step launch; words stamped with sequence numbers, so cleared only when (re)allocated
        const size_t want = 2 + (size_t)kMaxDevSteps * kInnerMaxG;
        if (c->d_inner_ctl.cap < want) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            HIP_TRY(c, c->d_inner_ctl.reserve(want));
            HIP_TRY(c, hipMemsetAsync(c->d_inner_ctl.p, 0, c->d_inner_ctl.cap * sizeof(unsigned), c->stream));
        }
    }
    if (!c->d_ticket.p) {
        HIP_TRY(c, c->d_ticket.reserve(2));
        HIP_TRY(c, hipMemsetAsync(c->d_ticket.p, 0, 2 * sizeof(unsigned), c->stream));
    }
    HIP_TRY(c, c->d_pose.reserve(1));
    return PPCR_OK;
}

// The later IRLS steps of the current outer iteration as ONE launch the device walks through on its own
// (inner_steps_kernel): step 1 was enqueued with loop control by the caller; its mailbox slot (tk.seq) is the one the
// iteration's last step publishes in.
template <int W>
void launch_inner_w(ppcr_ctx *c, const InnerArgs &a, const K23Form &f, int grid)
{
#define PPCR_INNER(TMc, ONEc) inner_steps_kernel<W, TMc, ONEc><<<grid, kBlock, 0, c->stream>>>(a)
    if (f.tm == 0) {
        if (f.onepass) PPCR_INNER(0, true);
        else PPCR_INNER(0, false);
    } else if (f.tm == 8) {
        if (f.onepass) PPCR_INNER(8, true);
        else PPCR_INNER(8, false);
    } else if (f.tm == -3) {
        PPCR_INNER(-3, true);
    } else {
        if (f.onepass) PPCR_INNER(-1, true);
        else PPCR_INNER(-1, false);
    }
#undef PPCR_INNER
}
int launch_inner(ppcr_ctx *c, const StepTicket &tk, const LoopCtl &loop, int n_dev_steps)
{
    if (n_dev_steps <= 0) return PPCR_OK;
    if (c->assoc != ppcr_ctx::ASSOC_ELL || c->nt <= 0) return fail(c, PPCR_ERR_STATE, "internal: device-paced inner steps need an ELL association");
    const int ns = (int)c->ns;
    const int ntiles = nblocks(ns, kBlock);
    // about five K23 workgroups per CU at a time, every one with the same number of tiles (+-1): the launch carries
    // n_dev_steps * (G + 19) workgroups that all have to start and look at the loop state even when step 1 ended the
    // loop (the common case), so G is not simply ntiles (3 x 3926 idle workgroups cost ~15 us per iteration at 1M)
    // (and one step's G + 19 workgroups must be resident TOGETHER — five per CU, 1280 — or the step takes two rounds:
    //  1303 + 19 of them cost 57 us per step, 977 + 19 cost 48)
    const int per_wg = std::max(1, (ntiles + 1199) / 1200);
    int G = std::max(1, std::min((ntiles + per_wg - 1) / per_wg, kInnerMaxG));
    if (per_wg > 1) G = std::min((G + 7) & ~7, kInnerMaxG);  // (a multiple of eight: the kernel's XCD-aware tile map)
    HIP_TRY(c, c->partials.reserve((size_t)std::max(G, k1_steady_slots(c)) * kNSums));
    InnerArgs a;
    std::memset(&a, 0, sizeof(a));
    a.nbr = c->nbr.p, a.cnt = c->cnt.p, a.src = c->src.p, a.tgt = c->tgt_cur();
    a.ns = ns, a.width = c->ell_width;
    a.md = make_model(c);
    InnerConst ic;
    std::memset(&ic, 0, sizeof(ic));
    StepTicket same = tk;
    const unsigned seq_keep = c->mbox_seq;
    PPCR_TRY(prepare_fold(c, G, same, ic.fs, &loop));  // (draws a sequence number: handed back — the steps share tk.seq)
    c->mbox_seq = seq_keep;
    ic.fs.seq = 0, ic.fs.mbox = nullptr, ic.fs.handed_over = nullptr;  // per launch
    ic.fs.loop.first = 0, ic.fs.loop.last_dev = 0;                     // per step
    ic.flags = c->d_inner_ctl.p + 2, ic.step_done = reinterpret_cast<unsigned long long *>(c->d_inner_ctl.p);
    ic.mbox_ring = c->d_mbox, ic.mbox_slots = kMailboxRing;
    ic.ovf_state = c->ovf_state.p;
    ic.G = G;
    HIP_TRY(c, c->d_inner_const.reserve(1));
    if (!c->inner_const_valid || std::memcmp(&ic, &c->h_inner_const, sizeof(ic)) != 0) {
        // (once per change of buffers or loop control: launches already in flight read the old contents)
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->h_inner_const = ic;
        HIP_TRY(c, hipMemcpy(c->d_inner_const.p, &c->h_inner_const, sizeof(ic), hipMemcpyHostToDevice));
        c->inner_const_valid = true;
    }
    a.ic = c->d_inner_const.p;
    a.seq = tk.seq;
    a.ovf_index = c->ovf_parity;
    a.n_steps = n_dev_steps;
    const K23Form f = k23_form(c, a.md);
    const int grid = n_dev_steps * (G + kInnerFoldSlots);
    {
        ProfScope ps(c, K_INNER);
        const int w = c->ell_width;
        if (w <= 10) launch_inner_w<10>(c, a, f, grid);
        else if (w <= 16) launch_inner_w<16>(c, a, f, grid);
        else if (w <= 20) launch_inner_w<20>(c, a, f, grid);  // (the command line's default width)
        else launch_inner_w<32>(c, a, f, grid);
    }
    return check_launch(c, "inner_steps_kernel");
}

// What follows an outer iteration inside the device-paced loop: the companion cloud moves (in place) by the transform in
// *d_pose, and the per-iteration reports are formed (track_kernel).  report_flags: PPCR_REPORT_*; the means arrive in
// the report ring under sequence number `seq`.
int launch_track(ppcr_ctx *c, unsigned seq, int report_flags)
{
    const bool want_truth = (report_flags & PPCR_REPORT_TRUTH) != 0, want_moved = (report_flags & PPCR_REPORT_MOVED) != 0;
    if (!c->have_companion && !want_truth && !want_moved) return PPCR_OK;
    TrackArgs a;
    std::memset(&a, 0, sizeof(a));
    if (c->have_companion) {
        a.cloud = c->companion.p, a.n = (int)c->n_companion, a.sorted_source = 0, a.write_back = 1;
    } else {
        a.cloud = c->src.p, a.n = (int)c->ns, a.sorted_source = 1, a.write_back = 0;
    }
    if (want_truth) {
        if (!c->have_ground_truth) return fail(c, PPCR_ERR_STATE, "ground truth cloud not set");
        if ((int64_t)a.n != c->n_ground_truth) return fail(c, PPCR_ERR_INVALID, "ground truth and source clouds differ in size");
        a.truth = c->ground_truth.p;
    }
    a.want_moved = want_moved ? 1 : 0;
    a.pose = c->d_pose.p;
    const int nb = std::max(1, std::min(1024, nblocks(a.n)));  // the grid of mean_distance(): same block sums
    HIP_TRY(c, c->track_part.reserve((size_t)2 * nb));
    if (!c->track_ticket.p) {
        HIP_TRY(c, c->track_ticket.reserve(1));
        HIP_TRY(c, hipMemsetAsync(c->track_ticket.p, 0, sizeof(unsigned), c->stream));
    }
    a.part = c->track_part.p;
    a.ticket = c->track_ticket.p;
    a.out = (want_truth || want_moved) ? c->d_report + (seq % kMailboxRing) : nullptr;
    a.seq = seq;
    a.st = c->d_loop.p;
    {
        ProfScope ps(c, K_TRACK);
        track_kernel<<<nb, kBlock, 0, c->stream>>>(a);
    }
    return check_launch(c, "track_kernel");
}

// wait for the report of iteration `seq` (it trails the iteration's mailbox slot by one small kernel)
int collect_report(ppcr_ctx *c, unsigned seq, double *mse_truth, double *moved)
{
    const HostReport *r = c->h_report + (seq % kMailboxRing);
    bool arrived = false;
    for (long spin = 0; spin < 200000000L; spin++) {
        if (__atomic_load_n(&r->seq, __ATOMIC_ACQUIRE) == seq) {
            arrived = true;
            break;
        }
        if (spin > 4096) std::this_thread::yield();
        if ((spin & 0xFFFF) == 0xFFFF && hipStreamQuery(c->stream) != hipErrorNotReady) {
            arrived = __atomic_load_n(&r->seq, __ATOMIC_ACQUIRE) == seq;
            break;
        }
    }
    if (!arrived) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (__atomic_load_n(&r->seq, __ATOMIC_ACQUIRE) != seq) return fail(c, PPCR_ERR_HIP, "iteration report never arrived");
    }
    *mse_truth = r->mse_truth;
    *moved = r->moved;
    return PPCR_OK;
}

int run_accumulate(ppcr_ctx *c, const Mat3 &R, const double t[3], double sums[PPCR_NSUMS])
{
    PPCR_TRY(flush_pending_move(c));
    StepTicket tk;
    StepResult res;
    PPCR_TRY(launch_step(c, R, t, tk));
    PPCR_TRY(collect_step(c, tk, res));
    std::memcpy(sums, res.sums, sizeof(res.sums));
    return PPCR_OK;
}

int apply_transform_now(ppcr_ctx *c, const double T[12]);

// make the device copy of the source current (a move deferred to the next tiled K1 is applied now)
int flush_pending_move(ppcr_ctx *c)
{
    if (c->move_on_device) return fail(c, PPCR_ERR_STATE, "internal: a device-resident move is pending outside the align loop");
    if (!c->move_pending) return PPCR_OK;
    c->move_pending = false;
    return apply_transform_now(c, c->pending_T);
}

// move the source; when `defer` the move rides along with the next tiled association instead of its own launch
int apply_transform_impl(ppcr_ctx *c, const double T[12], bool defer = false, bool move_companion = true)
{
    if (!c->have_src) return fail(c, PPCR_ERR_STATE, "source cloud not set");
    PPCR_TRY(flush_pending_move(c));
    if (move_companion && c->have_companion && c->n_companion > 0) {
        // the full-resolution copy moves with the same f64 -> f32 arithmetic, at once (it is off the hot path)
        Pose P;
        for (int a = 0; a < 3; a++) {
            for (int b = 0; b < 3; b++) P.R[3 * a + b] = T[4 * a + b];
            P.t[a] = T[4 * a + 3];
            P.c[a] = 0;
        }
        ProfScope ps(c, K_TRANSFORM);
        transform_kernel<<<nblocks(c->n_companion), kBlock, 0, c->stream>>>(c->companion.p, (int)c->n_companion, P);
    }
    if (defer) {
        std::memcpy(c->pending_T, T, sizeof(c->pending_T));
        c->move_pending = true;
        return PPCR_OK;
    }
    return apply_transform_now(c, T);
}

int apply_transform_now(ppcr_ctx *c, const double T[12])
{
    c->dm2_valid = false;  // the source moved outside a tiled K1: the temporal cut-off starts over
    Pose P;
    for (int a = 0; a < 3; a++) {
        for (int b = 0; b < 3; b++) P.R[3 * a + b] = T[4 * a + b];
        P.t[a] = T[4 * a + 3];
        P.c[a] = 0;
    }
    if (c->ns > 0) {
        ProfScope ps(c, K_TRANSFORM);
        transform_kernel<<<nblocks(c->ns), kBlock, 0, c->stream>>>(c->src.p, (int)c->ns, P);
    }
    return check_launch(c, "transform_kernel");
}

// IRLS on the current association (see ppcr_solve in ppcr.h): every half-step is K23 -> fold -> solve on the device;
// the host only compares costs
int solve_impl(ppcr_ctx *c, const double q0[4], const double t0[3], int max_steps, double f_tol, double T_out[12],
               double cost_out[2], int *steps_out)
{
    const double qn = q0[0] * q0[0] + q0[1] * q0[1] + q0[2] * q0[2] + q0[3] * q0[3];
    if (!(qn > 0) || !std::isfinite(qn)) return fail(c, PPCR_ERR_INVALID, "initial rotation quaternion has zero or non-finite norm");
    PPCR_TRY(flush_pending_move(c));
    Mat3 R = quat_to_rot(q0);
    Vec3 t{{t0[0], t0[1], t0[2]}};
    StepTicket tk;
    StepResult res;
    PPCR_TRY(launch_step(c, R, t.v, tk));
    PPCR_TRY(collect_step(c, tk, res));
    double cost_old = 0.5 * res.sums[16];
    cost_out[0] = cost_out[1] = cost_old;
    int steps = 0;
    if (max_steps < 1) max_steps = 1;
    for (;;) {
        const double fc = res.degenerate ? cost_old : res.cost;
        steps++;
        for (int a = 0; a < 3; a++) {
            for (int b = 0; b < 3; b++) R.m[a][b] = res.T[4 * a + b];
            t[a] = res.T[4 * a + 3];
        }
        cost_out[1] = fc;
        if (res.degenerate || steps >= max_steps) break;
        // a decrease below the rounding floor of the moment-based cost (eps * (Sxx + Syy)) is no decrease
        if ((cost_old - fc) <= std::max(f_tol * cost_old, 1e-14 * 0.5 * (res.sums[17] + res.sums[18]))) break;
        PPCR_TRY(launch_step(c, R, t.v, tk));
        PPCR_TRY(collect_step(c, tk, res));
        cost_old = 0.5 * res.sums[16];
    }
    // transformation(): normalised quaternion -> rotation (..._iteration.hpp:59-67); R is already
    // orthonormal to rounding so the round trip through a quaternion is not needed here
    pack_T(R, t, T_out);
    if (steps_out) *steps_out = steps;
    return PPCR_OK;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int ppcr_abi_version(void) { return PPCR_ABI_VERSION; }

int ppcr_device_count(int *count)
{
    if (!count) return PPCR_ERR_INVALID;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return PPCR_OK;
}

const char *ppcr_last_error(const ppcr_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int ppcr_create(int device_id, ppcr_ctx **out)
{
    if (!out) return fail(nullptr, PPCR_ERR_INVALID, "out is null");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(nullptr, PPCR_ERR_NODEVICE, "no HIP device visible: this library has no CPU fallback");
    }
    if (device_id < 0 || device_id >= n) return fail(nullptr, PPCR_ERR_INVALID, "device_id out of range");
    HIP_TRY(nullptr, hipSetDevice(device_id));
    ppcr_ctx *c = new (std::nothrow) ppcr_ctx();
    if (!c) return fail(nullptr, PPCR_ERR_NOMEM, "out of host memory");
    c->device = device_id;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&c->h_occupancy), sizeof(unsigned long long) * kOccBins, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&c->h_bbox), sizeof(float) * 6 * kBboxBlocks, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&c->h_sums), sizeof(double) * kNSums, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&c->h_total), sizeof(unsigned long long), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&c->h_mbox), sizeof(HostMailbox) * kMailboxRing, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) {
        std::memset(c->h_mbox, 0, sizeof(HostMailbox) * kMailboxRing);
        e = hipHostGetDevicePointer(reinterpret_cast<void **>(&c->d_mbox), c->h_mbox, 0);
    }
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&c->h_report), sizeof(HostReport) * kMailboxRing, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) {
        std::memset(c->h_report, 0, sizeof(HostReport) * kMailboxRing);
        e = hipHostGetDevicePointer(reinterpret_cast<void **>(&c->d_report), c->h_report, 0);
    }
    if (e != hipSuccess) {
        std::string msg = std::string("context setup: ") + hipGetErrorString(e);
        ppcr_destroy(c);
        return fail(nullptr, PPCR_ERR_HIP, msg);
    }
    *out = c;
    return PPCR_OK;
}

int ppcr_destroy(ppcr_ctx *c)
{
    if (!c) return PPCR_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->grid_pending) (void)hipEventSynchronize(c->aux_done);
    if (c->aux_done) (void)hipEventDestroy(c->aux_done);
    for (auto &r : c->prof_recs) {
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    for (auto ev : c->prof_pool) (void)hipEventDestroy(ev);
    release_levels(c);
    c->d_levels.release();
    c->level_inv.release();
    c->level_cap.release();
    c->staging.release();
    c->tgt_raw.release();
    c->tgt_sorted.release();
    c->src.release();
    c->src_alt.release();
    c->cell_start.release();
    c->companion.release();
    c->ground_truth.release();
    c->previous.release();
    c->keys_a.release();
    c->keys_b.release();
    c->vals_a.release();
    c->vals_b.release();
    c->cub_tmp.release();
    c->bbox_part.release();
    c->nbr.release();
    c->dm2.release();
    c->ovf_list.release();
    c->split_flag.release();
    c->split_list.release();
    c->split_state.release();
    c->ovf_state.release();
    c->cnt.release();
    c->row_ptr.release();
    c->gen_counts.release();
    c->gen_row_ptr.release();
    c->gen_pos.release();
    c->gen_keys.release();
    c->d_total.release();
    c->d_stamps.release();
    c->partials.release();
    c->d_sums.release();
    c->d_ticket.release();
    c->d_pose.release();
    c->d_w.release();
    c->d_s.release();
    c->mse_part.release();
    c->d_loop.release();
    c->d_fold_dbg.release();
    c->d_occupancy.release();
    c->d_short.release();
    c->d_inner_const.release();
    c->d_inner_ctl.release();
    c->track_part.release();
    c->track_ticket.release();
    if (c->h_sums) (void)hipHostFree(c->h_sums);
    if (c->h_total) (void)hipHostFree(c->h_total);
    if (c->h_occupancy) (void)hipHostFree(c->h_occupancy);
    if (c->h_bbox) (void)hipHostFree(c->h_bbox);
    if (c->h_mbox) (void)hipHostFree(c->h_mbox);
    if (c->h_report) (void)hipHostFree(c->h_report);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return PPCR_OK;
}

#define CTX_ENTER(c)                                  \
    if (!(c)) return PPCR_ERR_INVALID;                \
    HIP_TRY(c, hipSetDevice((c)->device))

int ppcr_set_params(ppcr_ctx *c, double radius, int max_neighbours, double dof, int dim)
{
    CTX_ENTER(c);
    if (!(radius > 0) || !std::isfinite(radius)) return fail(c, PPCR_ERR_INVALID, "radius must be positive and finite");
    if (!(dof > 0)) return fail(c, PPCR_ERR_INVALID, "dof must be > 0 (probabilistic_weights.hpp:34 asserts v > 0)");
    if (dim <= 0) return fail(c, PPCR_ERR_INVALID, "dim must be > 0 (probabilistic_weights.hpp:33)");
    if (radius != c->radius || max_neighbours != c->max_nb) {
        invalidate_association(c);
        c->dm2_valid = false;
    }
    c->radius = radius;
    c->max_nb = max_neighbours;
    c->dof = dof;
    c->dim = dim;
    c->params_set = true;
    return PPCR_OK;
}

int ppcr_set_option(ppcr_ctx *c, const char *key, int value)
{
    CTX_ENTER(c);
    if (!key) return fail(c, PPCR_ERR_INVALID, "null option key");
    PPCR_TRY(grid_settle(c));  // an early grid build was begun under the options as they were: it is dropped
    if (std::strcmp(key, "eager_grid") == 0) {  // ppcr_set_target starts the grid build (on a second stream) when the search is configured
        c->opt_eager_grid = value ? 1 : 0;
        return PPCR_OK;
    }
    if (std::strcmp(key, "sort_source") == 0) {
        if (c->have_src && c->src_sorted && !value)
            return fail(c, PPCR_ERR_STATE, "sort_source can only be switched off before the source has been sorted");
        c->opt_sort_source = value;  // 0 = keep caller order, 1 = brick/snake order (default), 2 = x-fastest cell order
        return PPCR_OK;
    }
    if (std::strcmp(key, "grid_xf") == 0) {
        if (value != 1 && value != 2 && value != 4 && value != 8) return fail(c, PPCR_ERR_INVALID, "grid_xf must be 1, 2, 4 or 8");
        if (c->have_tgt) return fail(c, PPCR_ERR_STATE, "grid_xf must be set before the target cloud");
        c->opt_grid_xf = value;
        return PPCR_OK;
    }
    if (std::strcmp(key, "brick_x") == 0) {
        if (value != 1 && value != 2 && value != 4) return fail(c, PPCR_ERR_INVALID, "brick_x must be 1, 2 or 4");
        c->opt_brick_xshift = value == 1 ? 0 : (value == 2 ? 1 : 2);
        c->src_sorted = false;
        return PPCR_OK;
    }
    if (std::strcmp(key, "merge_fold") == 0) {
        c->opt_merge_fold = value ? 1 : 0;
        return PPCR_OK;
    }
    if (std::strcmp(key, "fuse_k23") == 0) {
        c->opt_fuse_k23 = value ? 1 : 0;
        return PPCR_OK;
    }
    if (std::strcmp(key, "run_ahead") == 0) {
        c->opt_run_ahead = value ? 1 : 0;
        return PPCR_OK;
    }
    if (std::strcmp(key, "short_lists") == 0) {
        c->opt_short_lists = value ? 1 : 0;
        return PPCR_OK;
    }
    if (std::strcmp(key, "debug_mbox_seq") == 0) {  // TEST HOOK: continue the handle's sequence numbers from `value` (nothing in flight)
        if (value < 0) return fail(c, PPCR_ERR_INVALID, "debug_mbox_seq must be >= 0");
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->mbox_seq = (unsigned)value;
        return PPCR_OK;
    }
    if (std::strcmp(key, "level_stats") == 0) {  // diagnostic: cumulative per-level counters (ppcr_debug_get_levels); setting it clears them
        HIP_TRY(c, c->level_dbg.reserve((size_t)kMaxLevels * kLevelDbgWords));
        HIP_TRY(c, hipMemsetAsync(c->level_dbg.p, 0, (size_t)kMaxLevels * kLevelDbgWords * sizeof(unsigned), c->stream));
        c->opt_level_stats = value ? 1 : 0;
        return PPCR_OK;
    }
    if (std::strcmp(key, "levels") == 0) {
        if (value < -1 || value > 1) return fail(c, PPCR_ERR_INVALID, "levels must be -1 (automatic), 0 (one level) or 1 (same as -1)");
        c->opt_levels = value;
        c->grid_valid = false;
        return PPCR_OK;
    }
    if (std::strcmp(key, "k1_halves") == 0) {
        if (value < -1 || value > 1) return fail(c, PPCR_ERR_INVALID, "k1_halves must be -1 (automatic), 0 or 1");
        c->opt_k1_halves = value;
        return PPCR_OK;
    }
    if (std::strcmp(key, "defer_moves") == 0) {
        c->opt_defer_moves = value ? 1 : 0;
        return PPCR_OK;
    }
    if (std::strcmp(key, "two_pass") == 0) {
        if (value < 0 || value > kMaxReach) return fail(c, PPCR_ERR_INVALID, "two_pass must be 0 (off), 1 (automatic) or a reach of 2..8");
        c->opt_two_pass = value;
        c->grid_valid = false;
        return PPCR_OK;
    }
    if (std::strcmp(key, "fuse_max_handed_over") == 0) {
        if (value < 0) return fail(c, PPCR_ERR_INVALID, "fuse_max_handed_over must be >= 0");
        c->opt_fuse_max_handed_over = value;
        return PPCR_OK;
    }
    if (std::strcmp(key, "first_pass_fill") == 0) {
        if (value < 10 || value > 100) return fail(c, PPCR_ERR_INVALID, "first_pass_fill is in tenths of max_neighbours, 10..100");
        c->opt_first_pass_fill = value;
        c->grid_valid = false;
        return PPCR_OK;
    }
    if (std::strcmp(key, "first_pass_occupancy") == 0) {
        if (value < 10 || value > 640) return fail(c, PPCR_ERR_INVALID, "first_pass_occupancy is in tenths of a point per cell, 10..640");
        c->opt_first_pass_occ = value;
        c->grid_valid = false;
        return PPCR_OK;
    }
    if (std::strcmp(key, "fold_stamps") == 0) {
        HIP_TRY(c, c->d_fold_dbg.reserve(8));
        HIP_TRY(c, hipMemsetAsync(c->d_fold_dbg.p, 0, 8 * sizeof(unsigned long long), c->stream));
        c->opt_fold_stamps = value ? 1 : 0;
        return PPCR_OK;
    }
    if (std::strcmp(key, "inner_dev_steps") == 0) {
        if (value < 0 || value > kMaxDevSteps) return fail(c, PPCR_ERR_INVALID, "inner_dev_steps must be in [0, 8]");
        c->opt_inner_dev_steps = value;
        return PPCR_OK;
    }
    if (std::strcmp(key, "mailbox") == 0) {
        c->opt_mailbox = value ? 1 : 0;
        return PPCR_OK;
    }
    if (std::strcmp(key, "temporal") == 0) {  // 1: start each query's cut-off from its previous m-th distance (default)
        c->opt_temporal = value ? 1 : 0;
        return PPCR_OK;
    }
    if (std::strcmp(key, "stamps") == 0) {  // diagnostic: per-phase cycle totals of nn_fast_kernel (buffer: see associate_impl)
        c->opt_stamps = value;
        return PPCR_OK;
    }
    return fail(c, PPCR_ERR_INVALID, std::string("unknown option: ") + key);
}

static int set_target_common(ppcr_ctx *c, const void *p, bool dev, int64_t n, int64_t stride)
{
    PPCR_TRY(grid_settle(c));  // (a build of the previous target still reading tgt_raw)
    PPCR_TRY(upload_cloud(c, p, dev, n, stride, c->tgt_raw, true));
    bbox_fold(c, c->tgt_lo, c->tgt_hi);
    c->nt = n;
    c->have_tgt = true;
    c->grid_valid = false;
    c->origin_valid = false;
    c->dm2_valid = false;
    c->assoc_space = 0;
    invalidate_association(c);
    // (upload_cloud left the main stream idle: nothing earlier still reads the buffers the build writes)
    return grid_begin_early(c);
}

// Sequence numbers (mailbox slots, completion flags, LoopState::finished, the report ring) grow by one per fold for the
// life of a handle and are compared for equality or order; long before the 32-bit counter could wrap — at a new
// source, i.e. between registrations, when nothing is in flight — everything stamped with them starts again from zero.
static int restart_sequence_numbers(ppcr_ctx *c)
{
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->mbox_seq = 0;
    if (c->h_mbox)
        for (int k = 0; k < kMailboxRing; k++) c->h_mbox[k].seq = 0;
    if (c->h_report)
        for (int k = 0; k < kMailboxRing; k++) c->h_report[k].seq = 0;
    if (c->d_inner_ctl.p) HIP_TRY(c, hipMemsetAsync(c->d_inner_ctl.p, 0, c->d_inner_ctl.cap * sizeof(unsigned), c->stream));
    if (c->d_loop.p) HIP_TRY(c, hipMemsetAsync(c->d_loop.p, 0, sizeof(LoopState), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PPCR_OK;
}

static int set_source_common(ppcr_ctx *c, const void *p, bool dev, int64_t n, int64_t stride)
{
    if (c->mbox_seq > (1u << 30)) PPCR_TRY(restart_sequence_numbers(c));
    c->ovf_last = ~0u;
    c->split_clean = false;
    c->level_cap_clean = false;
    c->move_pending = false;  // a deferred move of the previous source dies with it
    c->move_on_device = false;
    c->dm2_valid = false;
    PPCR_TRY(upload_cloud(c, p, dev, n, stride, c->src));
    c->ns = n;
    c->have_src = true;
    c->src_sorted = false;
    if (!c->have_companion) c->have_previous = false;  // the snapshot belonged to the previous source
    invalidate_association(c);
    return PPCR_OK;
}

int ppcr_set_target(ppcr_ctx *c, const float *xyz, int64_t n, int64_t stride_bytes)
{
    CTX_ENTER(c);
    return set_target_common(c, xyz, false, n, stride_bytes);
}
int ppcr_set_source(ppcr_ctx *c, const float *xyz, int64_t n, int64_t stride_bytes)
{
    CTX_ENTER(c);
    return set_source_common(c, xyz, false, n, stride_bytes);
}
int ppcr_set_target_device(ppcr_ctx *c, const void *d_xyz, int64_t n, int64_t stride_bytes)
{
    CTX_ENTER(c);
    return set_target_common(c, d_xyz, true, n, stride_bytes);
}
int ppcr_set_source_device(ppcr_ctx *c, const void *d_xyz, int64_t n, int64_t stride_bytes)
{
    CTX_ENTER(c);
    return set_source_common(c, d_xyz, true, n, stride_bytes);
}

int ppcr_associate(ppcr_ctx *c)
{
    CTX_ENTER(c);
    if (!c->have_src || !c->have_tgt) return fail(c, PPCR_ERR_STATE, "set source and target before ppcr_associate");
    return associate_impl(c);
}

int ppcr_association_size(ppcr_ctx *c, int64_t *n_rows, int64_t *nnz)
{
    CTX_ENTER(c);
    PPCR_TRY(ensure_nnz(c));
    if (n_rows) *n_rows = c->ns;
    if (nnz) *nnz = c->nnz;
    return PPCR_OK;
}

int ppcr_get_association(ppcr_ctx *c, int32_t *row_ptr, int32_t *col, float *d2)
{
    CTX_ENTER(c);
    PPCR_TRY(flush_pending_move(c));
    PPCR_TRY(build_csr_cache(c));
    const size_t ns = (size_t)c->ns, nnz = c->h_csr_col.size();
    if (row_ptr) std::memcpy(row_ptr, c->h_csr_row_ptr.data(), sizeof(int) * (ns + 1));
    if (col && nnz) std::memcpy(col, c->h_csr_col.data(), sizeof(int) * nnz);
    if (d2 && nnz) {
        // recomputed from the CURRENT source positions with the kernel's exact float op order
        std::vector<float4> hs(ns), ht((size_t)c->nt);
        HIP_TRY(c, hipMemcpyAsync(hs.data(), c->src.p, sizeof(float4) * ns, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(ht.data(), c->tgt_cur(), sizeof(float4) * (size_t)c->nt, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        std::vector<float4> so(ns), to((size_t)c->nt);
        for (size_t r = 0; r < ns; r++) { int o; std::memcpy(&o, &hs[r].w, 4); so[(size_t)o] = hs[r]; }
        for (size_t r = 0; r < (size_t)c->nt; r++) { int o; std::memcpy(&o, &ht[r].w, 4); to[(size_t)o] = ht[r]; }
        for (size_t i = 0; i < ns; i++)
            for (int k = c->h_csr_row_ptr[i]; k < c->h_csr_row_ptr[i + 1]; k++) {
                const float4 a = so[i], b = to[(size_t)c->h_csr_col[(size_t)k]];
                volatile float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
                volatile float r = dx * dx;
                volatile float yy = dy * dy;
                r = r + yy;
                volatile float zz = dz * dz;
                r = r + zz;
                d2[k] = r;
            }
    }
    return PPCR_OK;
}

int ppcr_set_association(ppcr_ctx *c, const int32_t *row_ptr, const int32_t *col, int64_t n_rows)
{
    CTX_ENTER(c);
    if (!c->have_src || !c->have_tgt) return fail(c, PPCR_ERR_STATE, "set source and target before ppcr_set_association");
    if (n_rows != c->ns) return fail(c, PPCR_ERR_INVALID, "n_rows must equal the source size");
    if (!row_ptr) return fail(c, PPCR_ERR_INVALID, "null row_ptr");
    const int64_t ns = c->ns;
    if (row_ptr[0] != 0) return fail(c, PPCR_ERR_INVALID, "row_ptr[0] must be 0");
    int maxlen = 0;
    for (int64_t i = 0; i < ns; i++) {
        const int len = row_ptr[i + 1] - row_ptr[i];
        if (len < 0) return fail(c, PPCR_ERR_INVALID, "row_ptr must be non-decreasing");
        maxlen = std::max(maxlen, len);
    }
    const int64_t nnz = row_ptr[ns];
    if (nnz > 0 && !col) return fail(c, PPCR_ERR_INVALID, "null col");
    for (int64_t k = 0; k < nnz; k++)
        if (col[k] < 0 || col[k] >= c->nt) return fail(c, PPCR_ERR_INVALID, "column index out of range");
    invalidate_association(c);
    c->assoc_space = c->grid_valid ? 1 : 0;
    std::vector<int> src_order, tgt_order;
    PPCR_TRY(download_order(c, c->src.p, ns, src_order));
    PPCR_TRY(download_order(c, c->tgt_cur(), c->nt, tgt_order));
    std::vector<int> src_pos((size_t)ns), tgt_pos((size_t)c->nt);
    for (int64_t r = 0; r < ns; r++) src_pos[(size_t)src_order[(size_t)r]] = (int)r;
    for (int64_t r = 0; r < c->nt; r++) tgt_pos[(size_t)tgt_order[(size_t)r]] = (int)r;
    c->h_csr_row_ptr.assign(row_ptr, row_ptr + ns + 1);
    c->h_csr_col.assign(col, col + nnz);
    c->h_csr_slot.resize((size_t)nnz);
    if (maxlen <= kEllMaxWidth) {
        const int w = std::max(maxlen, 1);
        std::vector<int> h_nbr((size_t)w * (size_t)std::max<int64_t>(ns, 1), -1), h_cnt((size_t)std::max<int64_t>(ns, 1), 0);
        for (int64_t i = 0; i < ns; i++) {
            const int r = src_pos[(size_t)i];
            h_cnt[(size_t)r] = row_ptr[i + 1] - row_ptr[i];
            for (int k = row_ptr[i]; k < row_ptr[i + 1]; k++) {
                const size_t sl = (size_t)(k - row_ptr[i]) * (size_t)ns + (size_t)r;
                h_nbr[sl] = tgt_pos[(size_t)col[k]];
                c->h_csr_slot[(size_t)k] = sl;
            }
        }
        HIP_TRY(c, c->nbr.reserve(h_nbr.size()));
        HIP_TRY(c, c->cnt.reserve(h_cnt.size()));
        HIP_TRY(c, hipMemcpyAsync(c->nbr.p, h_nbr.data(), h_nbr.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->cnt.p, h_cnt.data(), h_cnt.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->assoc = ppcr_ctx::ASSOC_ELL;
        c->ell_width = w;
    } else {
        std::vector<int> h_rp((size_t)ns + 1, 0), h_nbr((size_t)std::max<int64_t>(nnz, 1));
        for (int64_t i = 0; i < ns; i++) h_rp[(size_t)src_pos[(size_t)i] + 1] = row_ptr[i + 1] - row_ptr[i];
        for (int64_t r = 0; r < ns; r++) h_rp[(size_t)r + 1] += h_rp[(size_t)r];
        for (int64_t i = 0; i < ns; i++) {
            const int r = src_pos[(size_t)i];
            for (int k = row_ptr[i]; k < row_ptr[i + 1]; k++) {
                const size_t sl = (size_t)h_rp[(size_t)r] + (size_t)(k - row_ptr[i]);
                h_nbr[sl] = tgt_pos[(size_t)col[k]];
                c->h_csr_slot[(size_t)k] = sl;
            }
        }
        HIP_TRY(c, c->row_ptr.reserve(h_rp.size()));
        HIP_TRY(c, c->nbr.reserve(h_nbr.size()));
        HIP_TRY(c, hipMemcpyAsync(c->row_ptr.p, h_rp.data(), h_rp.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->nbr.p, h_nbr.data(), h_nbr.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->assoc = ppcr_ctx::ASSOC_CSR;
    }
    c->nnz = nnz;
    c->csr_cache_valid = true;
    return PPCR_OK;
}

int ppcr_weights(ppcr_ctx *c, const double q[4], const double t[3], double *w_out, double *s_out)
{
    CTX_ENTER(c);
    if (!q || !t) return fail(c, PPCR_ERR_INVALID, "null pose");
    PPCR_TRY(flush_pending_move(c));
    PPCR_TRY(build_csr_cache(c));
    const int ns = (int)c->ns;
    const size_t nnz = c->h_csr_col.size();
    if (nnz == 0) return PPCR_OK;
    const size_t slots = (c->assoc == ppcr_ctx::ASSOC_ELL) ? (size_t)c->ell_width * (size_t)ns : (size_t)c->nnz;
    HIP_TRY(c, c->d_w.reserve(slots));
    HIP_TRY(c, c->d_s.reserve(slots));
    const Mat3 R = quat_to_rot(q);
    const Pose P = make_pose(c, R, t);
    const Model md = make_model(c);
    {
        ProfScope ps(c, K_WEIGHTS);
        if (c->assoc == ppcr_ctx::ASSOC_ELL) {
            EllAssoc a{c->nbr.p, c->cnt.p, ns};
            weights_kernel<EllAssoc><<<nblocks(ns), kBlock, 0, c->stream>>>(a, c->src.p, c->tgt_cur(), ns, P, md, c->d_w.p, c->d_s.p);
        } else {
            CsrAssoc a{c->nbr.p, c->row_ptr.p};
            weights_kernel<CsrAssoc><<<nblocks(ns), kBlock, 0, c->stream>>>(a, c->src.p, c->tgt_cur(), ns, P, md, c->d_w.p, c->d_s.p);
        }
    }
    PPCR_TRY(check_launch(c, "weights_kernel"));
    std::vector<double> hw(slots), hs(slots);
    HIP_TRY(c, hipMemcpyAsync(hw.data(), c->d_w.p, slots * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(hs.data(), c->d_s.p, slots * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (size_t k = 0; k < nnz; k++) {
        if (w_out) w_out[k] = hw[c->h_csr_slot[k]];
        if (s_out) s_out[k] = hs[c->h_csr_slot[k]];
    }
    return PPCR_OK;
}

int ppcr_accumulate(ppcr_ctx *c, const double q[4], const double t[3], double sums[PPCR_NSUMS])
{
    CTX_ENTER(c);
    if (!q || !t || !sums) return fail(c, PPCR_ERR_INVALID, "null argument");
    return run_accumulate(c, quat_to_rot(q), t, sums);
}

int ppcr_get_origin(ppcr_ctx *c, double o[3])
{
    CTX_ENTER(c);
    if (!o) return fail(c, PPCR_ERR_INVALID, "null argument");
    for (int a = 0; a < 3; a++) o[a] = c->origin_valid ? c->origin[a] : 0.0;
    return PPCR_OK;
}

int ppcr_solve_moments(const double sums[PPCR_NSUMS], const double origin[3], double R[9], double t[3])
{
    if (!sums || !origin || !R || !t) return PPCR_ERR_INVALID;
    const RigidSolve rs = solve_rigid_from_moments(sums, origin);
    for (int a = 0; a < 3; a++) {
        for (int b = 0; b < 3; b++) R[3 * a + b] = rs.R.m[a][b];
        t[a] = rs.t[a];
    }
    return rs.degenerate ? 1 : PPCR_OK;
}

double ppcr_cost_from_moments(const double sums[PPCR_NSUMS], const double origin[3], const double R[9], const double t[3])
{
    Mat3 Rm;
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) Rm.m[a][b] = R[3 * a + b];
    return cost_from_moments(sums, origin, Rm, Vec3{{t[0], t[1], t[2]}});
}

int ppcr_update_weights(int device_id, const int32_t *row_ptr, int64_t n_rows, const double *sq_errors, double dof,
                        int dim, double *w_out)
{
    if (!row_ptr || n_rows < 0) return fail(nullptr, PPCR_ERR_INVALID, "bad row_ptr / n_rows");
    if (!(dof > 0) || dim <= 0) return fail(nullptr, PPCR_ERR_INVALID, "dof and dim must be > 0 (probabilistic_weights.hpp:33-34)");
    const int64_t nnz = row_ptr[n_rows];
    if (nnz == 0) return PPCR_OK;
    if (!sq_errors || !w_out) return fail(nullptr, PPCR_ERR_INVALID, "null errors / output");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(nullptr, PPCR_ERR_NODEVICE, "no HIP device visible: this library has no CPU fallback");
    }
    if (device_id < 0 || device_id >= n) return fail(nullptr, PPCR_ERR_INVALID, "device_id out of range");
    HIP_TRY(nullptr, hipSetDevice(device_id));
    int *d_rp = nullptr;
    double *d_s = nullptr, *d_w = nullptr;
    auto cleanup = [&]() {
        if (d_rp) (void)hipFree(d_rp);
        if (d_s) (void)hipFree(d_s);
        if (d_w) (void)hipFree(d_w);
    };
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_rp), sizeof(int) * (size_t)(n_rows + 1));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_s), sizeof(double) * (size_t)nnz);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_w), sizeof(double) * (size_t)nnz);
    if (e == hipSuccess) e = hipMemcpy(d_rp, row_ptr, sizeof(int) * (size_t)(n_rows + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_s, sq_errors, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        ppcr_ctx tmp;  // only used to derive the model constants
        tmp.dof = dof;
        tmp.dim = dim;
        const Model md = make_model(&tmp);
        weights_from_errors_kernel<<<nblocks(n_rows), kBlock>>>(d_rp, n_rows, d_s, md, d_w);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(w_out, d_w, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return fail(nullptr, PPCR_ERR_HIP, std::string("ppcr_update_weights: ") + hipGetErrorString(e));
    return PPCR_OK;
}

int ppcr_solve(ppcr_ctx *c, const double q0[4], const double t0[3], int max_steps, double f_tol, double T_out[12],
               double cost_out[2], int *steps_out)
{
    CTX_ENTER(c);
    if (!q0 || !t0 || !T_out || !cost_out) return fail(c, PPCR_ERR_INVALID, "null argument");
    return solve_impl(c, q0, t0, max_steps, f_tol, T_out, cost_out, steps_out);
}

int ppcr_apply_transform(ppcr_ctx *c, const double T[12])
{
    CTX_ENTER(c);
    if (!T) return fail(c, PPCR_ERR_INVALID, "null transform");
    return apply_transform_impl(c, T, c->opt_defer_moves != 0);
}

int ppcr_iterate(ppcr_ctx *c, const double q0[4], const double t0[3], int inner_steps, double f_tol, double T_out[12],
                 double cost_out[2], int *steps_out)
{
    CTX_ENTER(c);
    if (!q0 || !t0 || !T_out || !cost_out) return fail(c, PPCR_ERR_INVALID, "null argument");
    if (!c->have_src || !c->have_tgt) return fail(c, PPCR_ERR_STATE, "set source and target before ppcr_iterate");
    {
        const double qn = q0[0] * q0[0] + q0[1] * q0[1] + q0[2] * q0[2] + q0[3] * q0[3];
        if (!(qn > 0) || !std::isfinite(qn)) return fail(c, PPCR_ERR_INVALID, "initial rotation quaternion has zero or non-finite norm");
    }
    PPCR_TRY(associate_impl(c));
    PPCR_TRY(solve_impl(c, q0, t0, inner_steps, f_tol, T_out, cost_out, steps_out));
    return apply_transform_impl(c, T_out, /*defer=*/true);
}

int ppcr_stop_rule_check(ppcr_stop_rule *rule, int n_iter, double cost_drop_thresh, double n_cost_drop_it)
{
    if (!rule) return PPCR_ERR_INVALID;
    if (rule->iteration == n_iter) return PPCR_STOP_MAX_ITERATIONS;  // cc:140
    if (!(rule->cost_drop < cost_drop_thresh)) {                     // also taken by a NaN drop (cc:154-156)
        rule->idle = 0;
        return PPCR_CONTINUE;
    }
    if ((double)rule->idle > n_cost_drop_it) return PPCR_STOP_COST_DROP;  // cc:146
    rule->idle += 1;
    return PPCR_CONTINUE;
}

}  // extern "C"

// ---- reporting clouds: helpers shared by the align loop and the report entry points ----------------------------------------------------
namespace {

// the cloud the reference reports on: the full-resolution companion when one is set, else the source itself
struct Tracked {
    const float4 *p;
    int64_t n;
    int sorted;  // 1: the handle's sorted source (w lane = caller's index)
};
int tracked_cloud(ppcr_ctx *c, Tracked &t)
{
    if (c->have_companion) {
        t = Tracked{c->companion.p, c->n_companion, 0};
        return PPCR_OK;
    }
    if (!c->have_src) return fail(c, PPCR_ERR_STATE, "source cloud not set");
    PPCR_TRY(flush_pending_move(c));
    t = Tracked{c->src.p, c->ns, 1};
    return PPCR_OK;
}

int mean_distance(ppcr_ctx *c, const Tracked &t, const float4 *other, double *out)
{
    if (t.n == 0) {
        *out = std::numeric_limits<double>::quiet_NaN();  // 0 / 0, as the reference's loop would produce
        return PPCR_OK;
    }
    const int nb = std::min(1024, nblocks(t.n));
    HIP_TRY(c, c->mse_part.reserve((size_t)nb));
    mean_distance_kernel<<<nb, kBlock, 0, c->stream>>>(t.p, (int)t.n, other, t.sorted, c->mse_part.p);
    PPCR_TRY(check_launch(c, "mean_distance_kernel"));
    std::vector<double> h((size_t)nb);
    HIP_TRY(c, hipMemcpyAsync(h.data(), c->mse_part.p, sizeof(double) * (size_t)nb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double sum = 0;
    for (double v : h) sum += v;
    *out = sum / (double)t.n;
    return PPCR_OK;
}

int snapshot_tracked(ppcr_ctx *c, const Tracked &t)
{
    HIP_TRY(c, c->previous.reserve((size_t)std::max<int64_t>(t.n, 1)));
    if (t.n > 0) snapshot_kernel<<<nblocks(t.n), kBlock, 0, c->stream>>>(t.p, (int)t.n, t.sorted, c->previous.p);
    PPCR_TRY(check_launch(c, "snapshot_kernel"));
    c->n_previous = t.n;
    c->have_previous = true;
    return PPCR_OK;
}

}  // namespace

namespace {

// One align() call as a resumable state machine, so that ONE host thread can keep several handles busy (each handle
// has its own stream; the device paces itself through the fold-and-solve lanes and PendingMove::dev).
//
// Pipelined mode (every bounded search on the tiled kernels): the device runs one iteration AHEAD of the host.  An
// outer iteration is a fixed train of launches — association (K23 of the first IRLS step folded in), fold-and-solve,
// the later IRLS steps of the inner loop as one device-walked launch (inner_steps_kernel; only when inner_steps > 1),
// the companion move / per-iteration reports (track_kernel; only when there is something to move or report) — and
// the lane that solves a step decides on the device whether the inner loop is over (LoopCtl: the test of solve_impl).
// Iteration k + 1 is enqueued before the host has seen iteration k — but only when hasConverged() cannot stop in
// between whatever the cost of iteration k turns out to be: the cap is not reached and the idle count is within the
// patience.  The rule therefore stays exact: nothing speculative is ever enqueued.
// Should an inner loop need more steps than were enqueued for the device (option inner_dev_steps), the device raises
// LoopState::abort, everything enqueued behind steps aside untouched, and the host finishes that iteration one step
// at a time (take_over) before the train continues.
struct AlignJob {
    ppcr_ctx *c = nullptr;
    int n_iter = 0, inner_steps = 1;
    double thresh = 0, patience = 0, f_tol = 1e-5;
    double q0[4] = {1, 0, 0, 0}, t0[3] = {0, 0, 0};
    double *history = nullptr, *costs = nullptr;
    int32_t *steps = nullptr;
    int report_flags = 0;                   // PPCR_REPORT_*
    ppcr_iteration_fn on_iteration = nullptr;
    void *user = nullptr;
    ppcr_stop_rule rule = {0, 0, 0.0};  // hasConverged(), shared with the C++ class (ppcr.h)
    double Tcum[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    double T_last[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    Mat3 R0 = Mat3::identity();
    StepTicket in_flight[kMailboxRing];
    int enq = 0, done = 0;
    bool pipelined = false, finished = false;
    unsigned handed_over_of[kMailboxRing] = {~0u, ~0u, ~0u, ~0u};  // per iteration in flight / just consumed (index % ring)
    static_assert(kMailboxRing == 4, "initialiser above");
    int max_steps = 1, n_dev_steps = 0;
    LoopCtl loop{};
    Pose handback{};  // take_over: the pose the host solved, on its way back to the device

    int validate()
    {
        if (!c) return PPCR_ERR_INVALID;
        HIP_TRY(c, hipSetDevice(c->device));
        // n_iter < 0 means "no iteration cap" in the reference (cc:140 never fires): legal, but then the per-iteration
        // arrays cannot be sized by the caller
        if (n_iter < 0 && (history || costs || steps))
            return fail(c, PPCR_ERR_INVALID, "ppcr_align: n_iter < 0 (no iteration cap) needs history, costs and steps to be NULL");
        if (n_iter < 0 && !(thresh > 0))
            return fail(c, PPCR_ERR_INVALID, "ppcr_align: n_iter < 0 (no iteration cap) needs cost_drop_thresh > 0, or the loop never ends");
        if (!c->have_src || !c->have_tgt) return fail(c, PPCR_ERR_STATE, "set source and target before ppcr_align");
        const double qn = q0[0] * q0[0] + q0[1] * q0[1] + q0[2] * q0[2] + q0[3] * q0[3];
        if (!(qn > 0) || !std::isfinite(qn)) return fail(c, PPCR_ERR_INVALID, "initial rotation quaternion has zero or non-finite norm");
        if ((report_flags & PPCR_REPORT_TRUTH) && !c->have_ground_truth) return fail(c, PPCR_ERR_STATE, "ground truth cloud not set");
        if (report_flags & PPCR_REPORT_TRUTH) {
            const int64_t n_tracked = c->have_companion ? c->n_companion : c->ns;
            if (n_tracked != c->n_ground_truth) return fail(c, PPCR_ERR_INVALID, "ground truth and source clouds differ in size");
        }
        R0 = quat_to_rot(q0);
        for (double &v : c->dbg_host) v = 0;
        const bool unbounded = (c->max_nb <= 0 || (int64_t)c->max_nb >= c->nt);
        pipelined = c->opt_mailbox && c->opt_run_ahead && !unbounded && c->max_nb <= kEllMaxWidth && c->ns > 0;
        max_steps = std::max(inner_steps, 1);
        n_dev_steps = (pipelined && c->nt > 0) ? std::min(max_steps - 1, c->opt_inner_dev_steps) : 0;
        if (pipelined) {
            PPCR_TRY(ensure_loop_state(c));
            loop.st = c->d_loop.p;
            loop.f_tol = f_tol;
            loop.max_steps = max_steps;
            loop.first = 1;
            loop.last_dev = n_dev_steps == 0 ? 1 : 0;
        }
        return PPCR_OK;
    }

    // book one finished outer iteration; mse_truth / moved: NaN unless asked for
    void record(const double Tk[12], const double cost[2], int st, double mse_truth, double moved)
    {
        compose(Tk, Tcum, Tcum);  // T_cum <- T_k * T_cum (cc:101-107)
        const int it = rule.iteration;
        if (history) std::memcpy(history + (size_t)it * 12, Tcum, sizeof(Tcum));
        if (costs) {
            costs[2 * it] = cost[0];
            costs[2 * it + 1] = cost[1];
        }
        if (steps) steps[it] = st;
        if (on_iteration) {
            ppcr_iteration_info info;
            info.iteration = it;
            info.inner_steps = st;
            info.cost[0] = cost[0], info.cost[1] = cost[1];
            std::memcpy(info.T_step, Tk, sizeof(info.T_step));
            std::memcpy(info.T_cum, Tcum, sizeof(info.T_cum));
            info.mse_truth = mse_truth;
            info.moved = moved;
            on_iteration(user, &info);
        }
        rule.cost_drop = (cost[0] - cost[1]) / cost[0];  // cc:119
        rule.iteration++;                                // cc:130
    }

    int enqueue()
    {
        StepTicket &tk = in_flight[enq % kMailboxRing];
        // (choices that go by hand-over counts use the count of the iteration kMaxAhead back: consumed for sure, however
        //  far ahead of the results this thread happens to be — the same path in every run)
        c->ovf_decide = enq >= kMaxAhead ? handed_over_of[(enq - kMaxAhead) % kMailboxRing] : ~0u;
        c->ovf_decide_pinned = true;
        // moves the source by the previous iteration's transform in its prologue and (steady state) leaves this
        // iteration's partial moments at (q0, t0) behind: K23 folded in
        const int rc_assoc = associate_impl(c, &R0, t0, &tk, &loop);
        c->ovf_decide_pinned = false;
        PPCR_TRY(rc_assoc);
        if (!c->assoc_folded) PPCR_TRY(launch_step(c, R0, t0, tk, c->assoc_fused, &loop));
        PPCR_TRY(launch_inner(c, tk, loop, n_dev_steps));
        c->move_on_device = true;     // ... and this iteration's transform is the next pending move
        PPCR_TRY(launch_track(c, tk.seq, report_flags));
        enq++;
        return PPCR_OK;
    }

    // LoopState::abort is up: the trains of the iterations behind iteration `done` stepped aside.  Wait for their
    // (empty) mailbox slots and take them back, last first.
    int recall_later_trains()
    {
        while (enq > done + 1) {
            StepResult skipped;
            PPCR_TRY(collect_step(c, in_flight[(enq - 1) % kMailboxRing], skipped));
            if (skipped.status != kLaunchSkipped) return fail(c, PPCR_ERR_STATE, "internal: a launch ran past an aborted iteration");
            enq--;
            c->ovf_parity ^= 1;  // its association had claimed the other counter of the pair
        }
        return PPCR_OK;
    }

    // The device could not finish iteration `done` by itself (kIterationPending): finish its inner loop one IRLS step at
    // a time, as solve_impl does, hand the pose back to the device and redo what follows an iteration.
    int take_over(StepResult &res)
    {
        const unsigned seq = in_flight[done % kMailboxRing].seq;
        PPCR_TRY(recall_later_trains());
        HIP_TRY(c, hipMemsetAsync(&c->d_loop.p->abort, 0, sizeof(unsigned), c->stream));
        c->move_on_device = false;
        Mat3 R;
        Vec3 t;
        int st = res.steps;
        for (;;) {
            for (int a = 0; a < 3; a++) {
                for (int b = 0; b < 3; b++) R.m[a][b] = res.T[4 * a + b];
                t[a] = res.T[4 * a + 3];
            }
            StepTicket tk;
            const double c_init = res.cost_init;
            PPCR_TRY(launch_step(c, R, t.v, tk));
            PPCR_TRY(collect_step(c, tk, res));
            res.cost_init = c_init;
            st++;
            const double cost_old = 0.5 * res.sums[16];
            if (res.degenerate) res.cost = cost_old;
            if (res.degenerate || st >= max_steps) break;
            if ((cost_old - res.cost) <= std::max(f_tol * cost_old, 1e-14 * 0.5 * (res.sums[17] + res.sums[18]))) break;
        }
        res.steps = st;
        res.status = kIterationDone;
        for (int a = 0; a < 3; a++) {
            for (int b = 0; b < 3; b++) handback.R[3 * a + b] = res.T[4 * a + b];
            handback.t[a] = res.T[4 * a + 3];
            handback.c[a] = 0.0;
        }
        HIP_TRY(c, hipMemcpyAsync(c->d_pose.p, &handback, sizeof(Pose), hipMemcpyHostToDevice, c->stream));
        c->move_on_device = true;
        return launch_track(c, seq, report_flags);
    }

    int consume()
    {
        StepResult res;
        PPCR_TRY(collect_step(c, in_flight[done % kMailboxRing], res));
        const unsigned seq = in_flight[done % kMailboxRing].seq;
        if (res.status == kIterationPending) PPCR_TRY(take_over(res));
        if (res.status != kIterationDone) return fail(c, PPCR_ERR_STATE, "internal: unexpected mailbox status in the align loop");
        const double cost[2] = {res.cost_init, res.cost};
        double mse_truth = std::numeric_limits<double>::quiet_NaN(), moved = mse_truth;
        if (report_flags) PPCR_TRY(collect_report(c, seq, &mse_truth, &moved));
        std::memcpy(T_last, res.T, sizeof(T_last));
        record(res.T, cost, res.steps, mse_truth, moved);
        handed_over_of[done % kMailboxRing] = c->ovf_last;  // (collect_step has just read it from this iteration's mailbox)
        done++;
        // an iteration enqueued ahead was let through on the strength of the idle count: replay its check now
        if (enq > done && ppcr_stop_rule_check(&rule, n_iter, thresh, patience) != PPCR_CONTINUE)
            return fail(c, PPCR_ERR_STATE, "internal: run-ahead broke the stopping rule");
        return PPCR_OK;
    }

    // Pipelined mode: do whatever can be done without waiting; with may_block, wait for the oldest iteration in flight
    // when nothing else is possible.  *progressed tells a scheduler whether to come back soon.
    int advance(bool may_block, bool *progressed)
    {
        if (progressed) *progressed = false;
        if (finished) return PPCR_OK;
        HIP_TRY(c, hipSetDevice(c->device));
        if (enq == done) {  // nothing in flight: the ordinary check
            if (ppcr_stop_rule_check(&rule, n_iter, thresh, patience) != PPCR_CONTINUE) {
                finished = true;
                if (done > 0) {  // the last transform becomes an ordinary host-side pending move
                    c->move_on_device = false;
                    // (the companion has followed every iteration already: track_kernel)
                    PPCR_TRY(apply_transform_impl(c, T_last, /*defer=*/true, /*move_companion=*/false));
                }
                if (progressed) *progressed = true;
                return PPCR_OK;
            }
            PPCR_TRY(enqueue());
            if (progressed) *progressed = true;
        }
        // Iterations done .. enq - 1 are in flight (a = enq - done of them).  One more may join them when the checks that
        // will be replayed as they arrive cannot stop the loop whatever their costs turn out to be: the cap is not hit
        // (the iteration count will be rule.iteration + a at the check in question), and the idle count — already
        // updated by the check that let iteration `done` through, at worst one higher after every further check — stays
        // within the patience.  Up to kMaxAhead in flight: the host's launch jitter no longer reaches the device.
        while (enq - done < kMaxAhead) {
            const int a = enq - done;
            if (!((rule.iteration + a != n_iter) && !((double)(rule.idle + a - 1) > patience))) break;
            PPCR_TRY(enqueue());
            if (progressed) *progressed = true;
        }
        if (may_block || step_arrived(c, in_flight[done % kMailboxRing])) {
            PPCR_TRY(consume());
            if (progressed) *progressed = true;
        }
        return PPCR_OK;
    }

    // host-paced loop (unbounded or very wide searches, options mailbox = 0 / run_ahead = 0): one blocking step at a time
    int run_host_paced()
    {
        const double nan = std::numeric_limits<double>::quiet_NaN();
        Tracked tr{nullptr, 0, 0};
        if (report_flags & PPCR_REPORT_MOVED) {
            PPCR_TRY(tracked_cloud(c, tr));
            PPCR_TRY(snapshot_tracked(c, tr));
        }
        while (ppcr_stop_rule_check(&rule, n_iter, thresh, patience) == PPCR_CONTINUE) {
            double Tk[12], cost[2];
            int st = 0;
            PPCR_TRY(associate_impl(c));
            PPCR_TRY(solve_impl(c, q0, t0, inner_steps, f_tol, Tk, cost, &st));
            PPCR_TRY(apply_transform_impl(c, Tk, /*defer=*/true));  // rides in the next iteration's K1 prologue
            double mse_truth = nan, moved = nan;
            if (report_flags) {
                PPCR_TRY(tracked_cloud(c, tr));
                if (report_flags & PPCR_REPORT_TRUTH) PPCR_TRY(mean_distance(c, tr, c->ground_truth.p, &mse_truth));
                if (report_flags & PPCR_REPORT_MOVED) {
                    PPCR_TRY(mean_distance(c, tr, c->previous.p, &moved));
                    PPCR_TRY(snapshot_tracked(c, tr));
                }
            }
            record(Tk, cost, st, mse_truth, moved);
        }
        finished = true;
        return PPCR_OK;
    }

    // the whole call on this thread
    int run()
    {
        if (!pipelined) return run_host_paced();
        while (!finished) PPCR_TRY(advance(true, nullptr));
        return PPCR_OK;
    }

    // After a failure: leave the handle in a defined state (nothing in flight, no device-resident move, no abort flag).
    void abandon()
    {
        if (!c) return;
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        if (c->d_loop.p) (void)hipMemsetAsync(c->d_loop.p, 0, sizeof(LoopState), c->stream);
        if (c->move_on_device) {
            // the transform the device solved last moves the source the ordinary way at its next use
            Pose P;
            if (hipMemcpy(&P, c->d_pose.p, sizeof(Pose), hipMemcpyDeviceToHost) == hipSuccess) {
                for (int a = 0; a < 3; a++) {
                    for (int b = 0; b < 3; b++) c->pending_T[4 * a + b] = P.R[3 * a + b];
                    c->pending_T[4 * a + 3] = P.t[a];
                }
                c->move_pending = true;
            }
            c->move_on_device = false;
        }
        c->dm2_valid = false;
        (void)hipStreamSynchronize(c->stream);
        finished = true;
    }

    // Hand the totals over.  The last move stays pending, as after ppcr_iterate: whoever reads the source next (or the
    // association's distances, weights, reports ...) applies it first; a following ppcr_align / ppcr_iterate takes it
    // along in its first K1 and keeps the temporal cut-off — two calls of n and m iterations cost what one call of
    // n + m does.
    int finish(double *T_final, int *n_done)
    {
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (report_flags & PPCR_REPORT_MOVED) c->have_previous = false;  // the loop kept no snapshot of its own
        if (T_final) std::memcpy(T_final, Tcum, sizeof(Tcum));  // identity when no iteration ran
        if (n_done) *n_done = rule.iteration;
        return PPCR_OK;
    }
};

AlignJob make_job(ppcr_ctx *c, int n_iter, double cost_drop_thresh, double n_cost_drop_it, const double q0[4], const double t0[3],
                  int inner_steps, double f_tol, double *history, double *costs, int32_t *steps)
{
    AlignJob j;
    j.c = c;
    j.n_iter = n_iter;
    j.thresh = cost_drop_thresh;
    j.patience = n_cost_drop_it;
    j.inner_steps = inner_steps;
    j.f_tol = f_tol;
    std::memcpy(j.q0, q0, sizeof(j.q0));
    std::memcpy(j.t0, t0, sizeof(j.t0));
    j.history = history;
    j.costs = costs;
    j.steps = steps;
    return j;
}

// align() with the per-iteration outputs optional and the last cumulative transform returned separately
int align_impl(ppcr_ctx *c, int n_iter, double cost_drop_thresh, double n_cost_drop_it, const double q0[4],
               const double t0[3], int inner_steps, double f_tol, double *history, double *costs, int32_t *steps,
               double *T_final, int *n_done, ppcr_stop_rule *rule_io = nullptr, int report_flags = 0,
               ppcr_iteration_fn fn = nullptr, void *user = nullptr)
{
    if (!c) return PPCR_ERR_INVALID;
    if (!q0 || !t0) return fail(c, PPCR_ERR_INVALID, "null argument");
    if (report_flags & ~(PPCR_REPORT_TRUTH | PPCR_REPORT_MOVED)) return fail(c, PPCR_ERR_INVALID, "unknown report flag");
    AlignJob job = make_job(c, n_iter, cost_drop_thresh, n_cost_drop_it, q0, t0, inner_steps, f_tol, history, costs, steps);
    job.report_flags = fn ? report_flags : 0;
    job.on_iteration = fn;
    job.user = user;
    if (rule_io) job.rule = *rule_io;
    PPCR_TRY(job.validate());
    const int rc = job.run();
    if (rc != PPCR_OK) {
        const std::string msg = c->err;
        job.abandon();
        c->err = msg;
        return rc;
    }
    if (rule_io) *rule_io = job.rule;
    return job.finish(T_final, n_done);
}

}  // namespace

extern "C" {

int ppcr_align(ppcr_ctx *c, int n_iter, double cost_drop_thresh, double n_cost_drop_it, const double q0[4],
               const double t0[3], int inner_steps, double f_tol, double *history, double *costs, int32_t *steps,
               int *n_done)
{
    return align_impl(c, n_iter, cost_drop_thresh, n_cost_drop_it, q0, t0, inner_steps, f_tol, history, costs, steps,
                      nullptr, n_done);
}

int ppcr_align_report(ppcr_ctx *c, int n_iter, double cost_drop_thresh, double n_cost_drop_it, const double q0[4],
                      const double t0[3], int inner_steps, double f_tol, ppcr_stop_rule *rule_io, int report_flags,
                      ppcr_iteration_fn on_iteration, void *user, double T_final[12], int *n_done)
{
    if (c && n_iter < 0 && !(cost_drop_thresh > 0))
        return fail(c, PPCR_ERR_INVALID, "ppcr_align_report: n_iter < 0 (no iteration cap) needs cost_drop_thresh > 0, or the loop never ends");
    return align_impl(c, n_iter, cost_drop_thresh, n_cost_drop_it, q0, t0, inner_steps, f_tol, nullptr, nullptr, nullptr,
                      T_final, n_done, rule_io, report_flags, on_iteration, user);
}

int ppcr_get_source(ppcr_ctx *c, float *xyz, int64_t stride_bytes)
{
    CTX_ENTER(c);
    if (!c->have_src) return fail(c, PPCR_ERR_STATE, "source cloud not set");
    if (stride_bytes < 12 || stride_bytes % 4) return fail(c, PPCR_ERR_INVALID, "stride_bytes must be a multiple of 4 and >= 12");
    PPCR_TRY(flush_pending_move(c));
    if (c->ns == 0) return PPCR_OK;
    if (!xyz) return fail(c, PPCR_ERR_INVALID, "null output");
    std::vector<float4> h((size_t)c->ns);
    HIP_TRY(c, hipMemcpyAsync(h.data(), c->src.p, sizeof(float4) * (size_t)c->ns, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    unsigned char *out = reinterpret_cast<unsigned char *>(xyz);
    for (int64_t r = 0; r < c->ns; r++) {
        int o;
        std::memcpy(&o, &h[(size_t)r].w, 4);
        float *p = reinterpret_cast<float *>(out + (size_t)o * (size_t)stride_bytes);
        p[0] = h[(size_t)r].x;
        p[1] = h[(size_t)r].y;
        p[2] = h[(size_t)r].z;
    }
    return PPCR_OK;
}

// diagnostic (tools/exp_stamps.py): out[8] = per-phase cycle totals over all waves of nn_fast_kernel
int ppcr_debug_get_stamps(ppcr_ctx *c, unsigned long long out[8])
{
    CTX_ENTER(c);
    if (!c->d_stamps.p) return fail(c, PPCR_ERR_STATE, "stamps not enabled");
    const size_t nst = c->stamps_wgs * (kBlock / 64) * 8;  // every workgroup of the launch (idle split slots wrote zeros)
    std::vector<unsigned long long> h(nst);
    HIP_TRY(c, hipMemcpyAsync(h.data(), c->d_stamps.p, nst * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < 8; k++) out[k] = 0;
    for (size_t i = 0; i < nst; i++) out[i % 8] += h[i];
    return PPCR_OK;
}

// diagnostic: wall-clock stamps (100 MHz) of the last fold-and-solve: {entry of the solving block, folded, ticket drawn,
// sums read back, solved, published}
int ppcr_debug_get_fold_stamps(ppcr_ctx *c, unsigned long long out[8])
{
    CTX_ENTER(c);
    if (!c->d_fold_dbg.p) return fail(c, PPCR_ERR_STATE, "fold_stamps not enabled");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(out, c->d_fold_dbg.p, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return PPCR_OK;
}

// diagnostic: the raw per-wave stamp records (8 x u64 per wave) of the last STAMPS launch
int ppcr_debug_get_stamps_raw(ppcr_ctx *c, unsigned long long *out, size_t n)
{
    CTX_ENTER(c);
    if (!c->d_stamps.p) return fail(c, PPCR_ERR_STATE, "stamps not enabled");
    if (n > c->d_stamps.cap) return fail(c, PPCR_ERR_INVALID, "more records than the stamp buffer holds");
    HIP_TRY(c, hipMemcpyAsync(out, c->d_stamps.p, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PPCR_OK;
}

// diagnostic: host-side figures of the last ppcr_align: {mailbox waits, total wait (s), max wait, -, -, -, max reduce
// launch call, workgroups its associations handed over to the cleanup kernel (summed over the iterations)}
int ppcr_debug_get_host_times(ppcr_ctx *c, double out[8])
{
    CTX_ENTER(c);
    for (int k = 0; k < 8; k++) out[k] = c->dbg_host[k];
    return PPCR_OK;
}

// diagnostic: {reach (cells per radius: 1 = one-pass search), first-pass search radius} of the grid in use
int ppcr_debug_get_search(ppcr_ctx *c, double out[2])
{
    CTX_ENTER(c);
    out[0] = c->grid_valid ? (double)c->reach : 0.0;
    out[1] = c->grid_valid ? c->search_radius : 0.0;
    return PPCR_OK;
}

// diagnostic: the multi-level search's cumulative counters since option "level_stats" was set — out[0] = levels, out[1] = base
// level, then per level: cell radius * 1000, and kLevelDbgWords counters {blocks, handed over (shape), handed over (size), short
// rows listed, staged candidates, rows}
int ppcr_debug_get_levels(ppcr_ctx *c, unsigned *out, int capacity)
{
    CTX_ENTER(c);
    if (!out || capacity < 2 + kMaxLevels * (1 + kLevelDbgWords)) return fail(c, PPCR_ERR_INVALID, "ppcr_debug_get_levels: buffer too small");
    std::memset(out, 0, (size_t)capacity * sizeof(unsigned));
    out[0] = (unsigned)c->n_levels, out[1] = (unsigned)c->base_level;
    std::vector<unsigned> raw((size_t)kMaxLevels * kLevelDbgWords, 0u);
    if (c->level_dbg.p) {
        HIP_TRY(c, hipMemcpyAsync(raw.data(), c->level_dbg.p, raw.size() * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    size_t e = 0;
    for (int l = 0; l < c->n_levels && l < kMaxLevels; l++) {
        const double radius = (l == c->base_level) ? c->search_radius : c->extra_levels[e++].radius;
        unsigned *row = out + 2 + (size_t)l * (1 + kLevelDbgWords);
        row[0] = (unsigned)std::lround(radius * 1000.0);
        for (int k = 0; k < kLevelDbgWords; k++) row[1 + k] = raw[(size_t)l * kLevelDbgWords + (size_t)k];
    }
    return PPCR_OK;
}

// diagnostic: rows the first pass of the most recent two-pass association left short (0 for a one-pass search)
int ppcr_debug_get_short_rows(ppcr_ctx *c, unsigned *out)
{
    CTX_ENTER(c);
    *out = 0;
    if (c->d_short.p) {
        HIP_TRY(c, hipMemcpyAsync(out, c->d_short.p + 2, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return PPCR_OK;
}

int ppcr_synchronize(ppcr_ctx *c)
{
    CTX_ENTER(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->grid_pending) HIP_TRY(c, hipEventSynchronize(c->aux_done));
    return PPCR_OK;
}

int ppcr_profile_enable(ppcr_ctx *c, int enable)
{
    CTX_ENTER(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->grid_pending) HIP_TRY(c, hipEventSynchronize(c->aux_done));  // (events of an early grid build)
    for (auto &r : c->prof_recs) {
        c->prof_pool.push_back(r.start);
        c->prof_pool.push_back(r.stop);
    }
    c->prof_recs.clear();
    for (int k = 0; k < K_NUM; k++) {
        c->prof_ms[k] = 0;
        c->prof_n[k] = 0;
    }
    c->prof_on = enable != 0;
    return PPCR_OK;
}

int ppcr_profile_get(ppcr_ctx *c, ppcr_kernel_stat *out, int capacity, int *n_out)
{
    CTX_ENTER(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->grid_pending) HIP_TRY(c, hipEventSynchronize(c->aux_done));
    for (auto &r : c->prof_recs) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
            c->prof_ms[r.id] += ms;
            c->prof_n[r.id] += 1;
        }
        c->prof_pool.push_back(r.start);
        c->prof_pool.push_back(r.stop);
    }
    c->prof_recs.clear();
    int n = 0;
    for (int k = 0; k < K_NUM; k++) {
        if (c->prof_n[k] == 0) continue;
        if (out && n < capacity) {
            std::memset(&out[n], 0, sizeof(out[n]));
            std::snprintf(out[n].name, sizeof(out[n].name), "%s", kKernelNames[k]);
            out[n].launches = c->prof_n[k];
            out[n].total_ms = c->prof_ms[k];
        }
        n++;
    }
    if (n_out) *n_out = n;
    return PPCR_OK;
}

}  // extern "C"

// ---- reporting clouds, voxel filter: entry points ----------------------------------------
extern "C" {

int ppcr_set_companion(ppcr_ctx *c, const float *xyz, int64_t n, int64_t stride_bytes)
{
    CTX_ENTER(c);
    PPCR_TRY(upload_cloud(c, xyz, false, n, stride_bytes, c->companion));
    c->n_companion = n;
    c->have_companion = true;
    c->have_previous = false;
    return PPCR_OK;
}

int ppcr_get_companion(ppcr_ctx *c, float *xyz, int64_t stride_bytes)
{
    CTX_ENTER(c);
    if (!c->have_companion) return fail(c, PPCR_ERR_STATE, "companion cloud not set");
    if (stride_bytes < 12 || stride_bytes % 4) return fail(c, PPCR_ERR_INVALID, "stride_bytes must be a multiple of 4 and >= 12");
    if (c->n_companion == 0) return PPCR_OK;
    if (!xyz) return fail(c, PPCR_ERR_INVALID, "null output");
    std::vector<float4> h((size_t)c->n_companion);
    HIP_TRY(c, hipMemcpyAsync(h.data(), c->companion.p, sizeof(float4) * h.size(), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    unsigned char *out = reinterpret_cast<unsigned char *>(xyz);
    for (size_t r = 0; r < h.size(); r++) {
        float *p = reinterpret_cast<float *>(out + r * (size_t)stride_bytes);
        p[0] = h[r].x;
        p[1] = h[r].y;
        p[2] = h[r].z;
    }
    return PPCR_OK;
}

int ppcr_set_ground_truth(ppcr_ctx *c, const float *xyz, int64_t n, int64_t stride_bytes)
{
    CTX_ENTER(c);
    PPCR_TRY(upload_cloud(c, xyz, false, n, stride_bytes, c->ground_truth));
    c->n_ground_truth = n;
    c->have_ground_truth = true;
    return PPCR_OK;
}

int ppcr_mse_ground_truth(ppcr_ctx *c, double *mse)
{
    CTX_ENTER(c);
    if (!mse) return fail(c, PPCR_ERR_INVALID, "null output");
    if (!c->have_ground_truth) return fail(c, PPCR_ERR_STATE, "ground truth cloud not set");
    Tracked t;
    PPCR_TRY(tracked_cloud(c, t));
    // utilities.hpp:19 asserts equal sizes
    if (t.n != c->n_ground_truth) return fail(c, PPCR_ERR_INVALID, "ground truth and source clouds differ in size");
    return mean_distance(c, t, c->ground_truth.p, mse);
}

int ppcr_mse_previous(ppcr_ctx *c, double *mse)
{
    CTX_ENTER(c);
    Tracked t;
    PPCR_TRY(tracked_cloud(c, t));
    if (mse) {
        if (c->have_previous && c->n_previous == t.n) PPCR_TRY(mean_distance(c, t, c->previous.p, mse));
        else *mse = 0.0;  // first call: nothing to compare with yet
    }
    return snapshot_tracked(c, t);  // *prev_source_cloud_ = *source_cloud_ (cc:122)
}

int ppcr_voxel_filter(int device_id, const float *xyz, int64_t n, int64_t stride_bytes, float leaf, float *out_xyz,
                      int64_t out_stride_bytes, int64_t *n_out)
{
    if (!n_out) return fail(nullptr, PPCR_ERR_INVALID, "null n_out");
    *n_out = 0;
    if (!(leaf > 0) || !std::isfinite(leaf)) return fail(nullptr, PPCR_ERR_INVALID, "leaf size must be positive and finite");
    if (out_stride_bytes < 12 || out_stride_bytes % 4) return fail(nullptr, PPCR_ERR_INVALID, "out_stride_bytes must be a multiple of 4 and >= 12");
    if (n > 0 && !out_xyz) return fail(nullptr, PPCR_ERR_INVALID, "null output");
    ppcr_ctx *c = nullptr;
    PPCR_TRY(ppcr_create(device_id, &c));
    auto body = [&]() -> int {
        DevBuf<float4> pts;
        struct Release {
            DevBuf<float4> &b;
            ~Release() { b.release(); }
        } rel{pts};
        PPCR_TRY(upload_cloud(c, xyz, false, n, stride_bytes, pts));
        if (n == 0) return PPCR_OK;
        float lo[3], hi[3];
        // bounding box of the finite points; an all-non-finite cloud leaves no voxel at all
        {
            const int nbb = std::min(1024, nblocks(n));
            HIP_TRY(c, c->bbox_part.reserve((size_t)nbb * 6));
            bbox_kernel<<<nbb, kBlock, 0, c->stream>>>(pts.p, (int)n, c->bbox_part.p);
            PPCR_TRY(check_launch(c, "bbox_kernel"));
            std::vector<float> part((size_t)nbb * 6);
            HIP_TRY(c, hipMemcpyAsync(part.data(), c->bbox_part.p, part.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            for (int a = 0; a < 3; a++) lo[a] = INFINITY, hi[a] = -INFINITY;
            for (int b = 0; b < nbb; b++)
                for (int a = 0; a < 3; a++) {
                    lo[a] = std::min(lo[a], part[(size_t)b * 6 + a]);
                    hi[a] = std::max(hi[a], part[(size_t)b * 6 + 3 + a]);
                }
            if (!(lo[0] <= hi[0])) return PPCR_OK;  // no finite point
        }
        VoxelDesc v;
        v.inv = 1.0f / leaf;
        int64_t d[3];
        for (int a = 0; a < 3; a++) d[a] = (int64_t)((hi[a] - lo[a]) * v.inv) + 1;
        unsigned char *out = reinterpret_cast<unsigned char *>(out_xyz);
        if (d[0] * d[1] * d[2] > (int64_t)INT32_MAX) {
            // pcl::VoxelGrid: "Leaf size is too small for the input dataset" -> the input is passed through
            const unsigned char *in = reinterpret_cast<const unsigned char *>(xyz);
            for (int64_t i = 0; i < n; i++)
                std::memcpy(out + (size_t)i * (size_t)out_stride_bytes, in + (size_t)i * (size_t)stride_bytes, 12);
            *n_out = n;
            return PPCR_OK;
        }
        int div_b[3];
        for (int a = 0; a < 3; a++) {
            v.min_b[a] = (int)std::floor(lo[a] * v.inv);
            div_b[a] = (int)std::floor(hi[a] * v.inv) - v.min_b[a] + 1;
        }
        v.mul[0] = 1;
        v.mul[1] = div_b[0];
        v.mul[2] = div_b[0] * div_b[1];
        const int ni = (int)n;
        HIP_TRY(c, c->keys_a.reserve((size_t)n + 1));
        HIP_TRY(c, c->keys_b.reserve((size_t)n + 1));
        HIP_TRY(c, c->vals_a.reserve((size_t)n + 1));
        HIP_TRY(c, c->vals_b.reserve((size_t)n + 1));
        voxel_key_kernel<<<nblocks(n), kBlock, 0, c->stream>>>(pts.p, ni, v, c->keys_a.p, c->vals_a.p);
        PPCR_TRY(check_launch(c, "voxel_key_kernel"));
        size_t tmp_bytes = 0;
        HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, c->keys_a.p, c->keys_b.p, c->vals_a.p, c->vals_b.p,
                                                     ni, 0, 32, c->stream));
        HIP_TRY(c, c->cub_tmp.reserve(tmp_bytes + 16));
        HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(c->cub_tmp.p, tmp_bytes, c->keys_a.p, c->keys_b.p, c->vals_a.p,
                                                     c->vals_b.p, ni, 0, 32, c->stream));
        // run heads -> output slots (exclusive scan); keys_a / vals_a are free again
        int *head = reinterpret_cast<int *>(c->keys_a.p), *slot = c->vals_a.p;
        voxel_head_kernel<<<nblocks(n), kBlock, 0, c->stream>>>(c->keys_b.p, ni, head);
        PPCR_TRY(check_launch(c, "voxel_head_kernel"));
        size_t scan_bytes = 0;
        HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, head, slot, ni, c->stream));
        HIP_TRY(c, c->cub_tmp.reserve(scan_bytes + 16));
        HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(c->cub_tmp.p, scan_bytes, head, slot, ni, c->stream));
        int last_head = 0, last_slot = 0;
        HIP_TRY(c, hipMemcpyAsync(&last_head, head + (ni - 1), sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(&last_slot, slot + (ni - 1), sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        const int64_t nv = (int64_t)last_slot + last_head;
        DevBuf<float> cent;
        struct Release2 {
            DevBuf<float> &b;
            ~Release2() { b.release(); }
        } rel2{cent};
        HIP_TRY(c, cent.reserve((size_t)std::max<int64_t>(nv, 1) * 3));
        voxel_centroid_kernel<<<nblocks(n), kBlock, 0, c->stream>>>(pts.p, c->keys_b.p, c->vals_b.p, head, slot, ni, cent.p);
        PPCR_TRY(check_launch(c, "voxel_centroid_kernel"));
        std::vector<float> h((size_t)nv * 3);
        if (nv > 0) HIP_TRY(c, hipMemcpyAsync(h.data(), cent.p, sizeof(float) * h.size(), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (int64_t k = 0; k < nv; k++) std::memcpy(out + (size_t)k * (size_t)out_stride_bytes, &h[(size_t)k * 3], 12);
        *n_out = nv;
        return PPCR_OK;
    };
    const int rc = body();
    if (rc != PPCR_OK) g_create_error = c->err;  // the temporary handle goes away: keep its message
    ppcr_destroy(c);
    return rc;
}

}  // extern "C"

extern "C" {

int ppcr_nearest_sq_distances(int device_id, const float *queries, int64_t nq, int64_t q_stride_bytes, const float *targets,
                              int64_t nt, int64_t t_stride_bytes, float *d2_out)
{
    if (nq < 0 || nt <= 0) return fail(nullptr, PPCR_ERR_INVALID, "ppcr_nearest_sq_distances: needs at least one target point");
    if (nq > 0 && !d2_out) return fail(nullptr, PPCR_ERR_INVALID, "null output");
    ppcr_ctx *c = nullptr;
    PPCR_TRY(ppcr_create(device_id, &c));
    auto body = [&]() -> int {
        c->opt_eager_grid = 0;  // (the cell edge is chosen below, from the box that comes with the upload)
        PPCR_TRY(set_target_common(c, targets, false, nt, t_stride_bytes));
        PPCR_TRY(set_source_common(c, queries, false, nq, q_stride_bytes));
        if (nq == 0) return PPCR_OK;
        // a cubic grid with a few points per cell: the nearest neighbour is then usually in the first shell or two
        const float *lo = c->tgt_lo, *hi = c->tgt_hi;
        double vol = 1, emax = 0;
        for (int a = 0; a < 3; a++) {
            const double e = (double)hi[a] - (double)lo[a];
            emax = std::max(emax, e);
            vol *= std::max(e, 1e-30);
        }
        double h = std::cbrt(vol * 3.0 / (double)nt);
        if (!(h > emax * 1e-4)) h = std::max(emax * 1e-4, 1e-30);  // degenerate (flat) clouds
        if (!std::isfinite(h) || !(h > 0)) h = 1.0;
        c->radius = h;
        c->opt_grid_xf = 1;
        c->opt_two_pass = 0;  // (the cell edge chosen above is what the shell search wants)
        PPCR_TRY(ensure_grid(c));
        DevBuf<float> d2;
        struct Release {
            DevBuf<float> &b;
            ~Release() { b.release(); }
        } rel{d2};
        HIP_TRY(c, d2.reserve((size_t)nq));
        nn1_kernel<<<nblocks(nq), kBlock, 0, c->stream>>>(c->src.p, (int)nq, c->tgt_sorted.p, c->cell_start.p, c->grid, d2.p);
        PPCR_TRY(check_launch(c, "nn1_kernel"));
        HIP_TRY(c, hipMemcpyAsync(d2_out, d2.p, sizeof(float) * (size_t)nq, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return PPCR_OK;
    };
    const int rc = body();
    if (rc != PPCR_OK) g_create_error = c->err;
    ppcr_destroy(c);
    return rc;
}

}  // extern "C"

// ---- batches of independent pairs -------------------------------------------------------------------------------
namespace {

// align() on one handle, keeping only the last cumulative transform
int align_final(ppcr_ctx *c, int n_iter, double thresh, double n_cost_drop_it, const double q0[4], const double t0[3],
                int inner_steps, double f_tol, double *T_final, int32_t *n_done)
{
    int done = 0;
    const int rc = align_impl(c, n_iter, thresh, n_cost_drop_it, q0, t0, inner_steps, f_tol, nullptr, nullptr, nullptr,
                              T_final, &done);
    if (rc != PPCR_OK) return rc;
    if (n_done) *n_done = done;
    return PPCR_OK;
}

struct FirstError {
    std::mutex mu;
    int rc = PPCR_OK;
    std::string text;
    void set(int code, const std::string &msg)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (rc == PPCR_OK) {
            rc = code;
            text = msg;
        }
    }
    bool failed()
    {
        std::lock_guard<std::mutex> lk(mu);
        return rc != PPCR_OK;
    }
};

}  // namespace

// Handles of ppcr_batch_run, kept between calls: creating one (a stream, the pinned mailbox ring, ~30 device buffers
// that grow on first use) costs ~10 ms, and the allocator serialises the lanes — more than the registrations of a small
// batch themselves (16 pairs of 250k with four in flight: 60 ms with fresh handles every call, 12 ms with pooled ones).
namespace {
struct HandlePool {
    std::mutex mu;
    std::vector<std::pair<int, ppcr_ctx *>> idle;  // (device, handle)
    static constexpr size_t kMaxIdle = 64;
    ppcr_ctx *take(int device)
    {
        std::lock_guard<std::mutex> lk(mu);
        for (size_t k = 0; k < idle.size(); k++)
            if (idle[k].first == device) {
                ppcr_ctx *c = idle[k].second;
                idle.erase(idle.begin() + (long)k);
                return c;
            }
        return nullptr;
    }
    bool give(int device, ppcr_ctx *c)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (idle.size() >= kMaxIdle) return false;
        idle.emplace_back(device, c);
        return true;
    }
    std::vector<ppcr_ctx *> drain()
    {
        std::lock_guard<std::mutex> lk(mu);
        std::vector<ppcr_ctx *> all;
        for (auto &e : idle) all.push_back(e.second);
        idle.clear();
        return all;
    }
};
HandlePool &batch_pool()
{
    static HandlePool *pool = new HandlePool;  // never destroyed: the HIP runtime may be gone by the time statics are
    return *pool;
}
}  // namespace

namespace {

// hand-over point between the two threads a device's share of a batch runs on (ppcr_batch_run)
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

struct PreparedPair {
    ppcr_ctx *c;
    int64_t pair;
};

// everything the first association of a registration would wait for, done where waiting costs nothing (the preparing
// thread of ppcr_batch_run): the second half of the grid build, the levels, the source's spatial sort
int prepare_first_association(ppcr_ctx *c)
{
    HIP_TRY(c, hipSetDevice(c->device));
    PPCR_TRY(ensure_grid(c));
    PPCR_TRY(ensure_source_sorted(c));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PPCR_OK;
}

}  // namespace

extern "C" {

int ppcr_align_many(ppcr_ctx *const *ctxs, int n, int lanes, int n_iter, double cost_drop_thresh,
                    double n_cost_drop_it, const double q0[4], const double t0[3], int inner_steps, double f_tol,
                    double *T_final, int32_t *n_done)
{
    if (n < 0 || (n > 0 && (!ctxs || !T_final)) || !q0 || !t0) return PPCR_ERR_INVALID;
    for (int k = 0; k < n; k++) {
        if (!ctxs[k]) return PPCR_ERR_INVALID;
        for (int j = 0; j < k; j++)
            if (ctxs[j] == ctxs[k]) return fail(ctxs[k], PPCR_ERR_INVALID, "ppcr_align_many: the same handle appears twice");
    }
    lanes = std::max(1, std::min(lanes, n));
    // Handles whose loop can run ahead of the host (one inner step per association) are all driven from THIS thread:
    // up to `lanes` of them are in flight at a time, each on its own stream, and the thread only polls their mailboxes
    // — no thread per pair, nothing to oversubscribe when many ranks share few cores.
    {
        std::vector<AlignJob> jobs;
        jobs.reserve((size_t)n);
        bool all_pipelined = true;
        for (int k = 0; k < n && all_pipelined; k++) {
            jobs.push_back(make_job(ctxs[k], n_iter, cost_drop_thresh, n_cost_drop_it, q0, t0, inner_steps, f_tol, nullptr, nullptr, nullptr));
            const int rc = jobs.back().validate();
            if (rc != PPCR_OK) return rc;
            all_pipelined = jobs.back().pipelined;
        }
        if (all_pipelined) {
            int next_job = 0, retired = 0;
            std::vector<int> window;  // indices of the jobs in flight
            while (retired < n) {
                while ((int)window.size() < lanes && next_job < n) window.push_back(next_job++);
                bool any = false;
                for (size_t w = 0; w < window.size();) {
                    AlignJob &j = jobs[(size_t)window[w]];
                    bool progressed = false;
                    const int rc = j.advance(false, &progressed);
                    if (rc != PPCR_OK) {
                        // leave every handle of the batch in a defined state (nothing in flight, no device-resident move)
                        const std::string msg = j.c->err;
                        for (AlignJob &other : jobs) other.abandon();
                        j.c->err = msg;
                        return rc;
                    }
                    any = any || progressed;
                    if (j.finished) {
                        // its last move is queued on its own stream; the synchronising tail runs after the loop
                        window.erase(window.begin() + (long)w);
                        retired++;
                    } else {
                        w++;
                    }
                }
                if (!any) std::this_thread::yield();
            }
            for (int k = 0; k < n; k++) {
                int done = 0;
                const int rc = jobs[(size_t)k].finish(T_final + (size_t)k * 12, &done);
                if (rc != PPCR_OK) return rc;
                if (n_done) n_done[k] = done;
            }
            return PPCR_OK;
        }
    }
    std::atomic<int> next{0};
    FirstError first;
    auto worker = [&]() {
        for (;;) {
            const int k = next.fetch_add(1);
            if (k >= n || first.failed()) return;
            const int rc = align_final(ctxs[k], n_iter, cost_drop_thresh, n_cost_drop_it, q0, t0, inner_steps, f_tol,
                                       T_final + (size_t)k * 12, n_done ? n_done + k : nullptr);
            if (rc != PPCR_OK) first.set(rc, ppcr_last_error(ctxs[k]));
        }
    };
    if (lanes == 1) {
        worker();
    } else {
        std::vector<std::thread> pool;
        for (int l = 0; l < lanes; l++) pool.emplace_back(worker);
        for (auto &th : pool) th.join();
    }
    return first.rc;
}

int ppcr_batch_release(void)
{
    for (ppcr_ctx *c : batch_pool().drain()) ppcr_destroy(c);
    return PPCR_OK;
}

int ppcr_batch_run(const ppcr_pair *pairs, int64_t n_pairs, const ppcr_batch_options *opt, const int *device_ids,
                   int n_devices, int lanes_per_device, double *T_all, int32_t *n_iter_done, char *err,
                   int64_t err_capacity)
{
    auto report = [&](int code, const std::string &msg) {
        if (err && err_capacity > 0) std::snprintf(err, (size_t)err_capacity, "%s", msg.c_str());
        return fail(nullptr, code, msg);
    };
    if (err && err_capacity > 0) err[0] = 0;
    if (n_pairs < 0 || !opt || n_devices <= 0 || !device_ids || lanes_per_device <= 0)
        return report(PPCR_ERR_INVALID, "ppcr_batch_run: bad argument");
    if (n_pairs == 0) return PPCR_OK;
    if (!pairs || !T_all) return report(PPCR_ERR_INVALID, "ppcr_batch_run: null pairs/T_all");
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) {
        (void)hipGetLastError();
        return report(PPCR_ERR_NODEVICE, "no HIP device visible: this library has no CPU fallback");
    }
    for (int d = 0; d < n_devices; d++)
        if (device_ids[d] < 0 || device_ids[d] >= visible) return report(PPCR_ERR_INVALID, "ppcr_batch_run: device id out of range");

    // pair p -> device p % n_devices
    FirstError first;
    // Bounded searches with the device-paced loop (every pair of every batch the command line or batch.py builds): TWO
    // threads per device.  One PREPARES pairs — uploads (synchronous PCIe copies), grid build, levels, source sort: all the
    // waiting of a registration — on handles nobody is iterating on; the other only enqueues and polls, up to
    // `lanes_per_device` registrations in flight on their own streams, as ppcr_align_many does.  A thread per lane
    // (what this was: each doing its own uploads between its iterations) reached 1.07 k pairs/s at two lanes and LESS at
    // four (0.96 k; 64 x 250k on one GPU): every copy held up the other lanes' launches.
    bool two_threads = opt->max_neighbours > 0 && opt->max_neighbours <= kEllMaxWidth;
    for (int64_t p = 0; p < n_pairs && two_threads; p++)
        two_threads = pairs[p].n_source > 0 && (int64_t)opt->max_neighbours < pairs[p].n_target;
    auto device_share = [&](int d) {
        const int dev = device_ids[d];
        const int64_t mine = (n_pairs - d + n_devices - 1) / n_devices;
        const int lanes = (int)std::max<int64_t>(1, std::min<int64_t>(lanes_per_device, mine));
        std::vector<ppcr_ctx *> handles;  // in flight, being prepared, or waiting on either side
        HandOver<ppcr_ctx *> idle;
        HandOver<PreparedPair> ready;
        for (int64_t h = 0; h < std::min<int64_t>(mine, (int64_t)lanes + 2); h++) {
            ppcr_ctx *c = batch_pool().take(dev);
            const int rc = c ? PPCR_OK : ppcr_create(dev, &c);
            if (rc != PPCR_OK) {
                first.set(rc, ppcr_last_error(nullptr));
                break;
            }
            handles.push_back(c);
            idle.push(c);
        }
        if (!first.failed()) {
            std::thread preparing([&]() {
                for (int64_t k = 0; k < mine && !first.failed(); k++) {
                    ppcr_ctx *c = nullptr;
                    if (idle.pop(&c, true) != 1) break;  // (closed: the other thread met an error)
                    const int64_t p = (int64_t)d + k * n_devices;
                    const ppcr_pair &pr = pairs[p];
                    int rc = ppcr_set_params(c, opt->radius, opt->max_neighbours, opt->dof, opt->dim);
                    if (rc == PPCR_OK) rc = ppcr_set_target(c, pr.target, pr.n_target, pr.target_stride_bytes);
                    if (rc == PPCR_OK) rc = ppcr_set_source(c, pr.source, pr.n_source, pr.source_stride_bytes);
                    if (rc == PPCR_OK) rc = prepare_first_association(c);
                    if (rc != PPCR_OK) {
                        first.set(rc, "pair " + std::to_string(p) + ": " + ppcr_last_error(c));
                        break;
                    }
                    ready.push(PreparedPair{c, p});
                }
                ready.close();
            });
            struct Running {
                AlignJob job;
                int64_t pair;
            };
            std::vector<std::unique_ptr<Running>> window;
            bool more = true;
            while ((more || !window.empty()) && !first.failed()) {
                while (more && (int)window.size() < lanes) {
                    PreparedPair pp{nullptr, 0};
                    const int got = ready.pop(&pp, window.empty());
                    if (got < 0) more = false;
                    if (got <= 0) break;
                    std::unique_ptr<Running> r(new Running{make_job(pp.c, opt->n_iter, opt->cost_drop_thresh, opt->n_cost_drop_it, opt->q0, opt->t0,
                                                                    opt->inner_steps, opt->f_tol, nullptr, nullptr, nullptr),
                                                           pp.pair});
                    int rc = r->job.validate();
                    if (rc == PPCR_OK && !r->job.pipelined) rc = r->job.run();  // (a handle with the mailbox switched off: here and now)
                    if (rc != PPCR_OK) {
                        first.set(rc, "pair " + std::to_string(pp.pair) + ": " + ppcr_last_error(pp.c));
                        break;
                    }
                    window.push_back(std::move(r));
                }
                bool any = false;
                for (size_t w = 0; w < window.size() && !first.failed();) {
                    Running &r = *window[w];
                    bool progressed = false;
                    const int rc = r.job.advance(false, &progressed);
                    if (rc != PPCR_OK) {
                        first.set(rc, "pair " + std::to_string(r.pair) + ": " + ppcr_last_error(r.job.c));
                        break;
                    }
                    any = any || progressed;
                    if (r.job.finished) {
                        // the totals are on the host; what is still queued on the handle's stream (its last move is a
                        // host-side pending one) the next upload waits for
                        std::memcpy(T_all + (size_t)r.pair * 12, r.job.Tcum, sizeof(r.job.Tcum));
                        if (n_iter_done) n_iter_done[r.pair] = r.job.rule.iteration;
                        idle.push(r.job.c);
                        window.erase(window.begin() + (long)w);
                    } else {
                        w++;
                    }
                }
                if (!any && !window.empty()) std::this_thread::yield();
            }
            if (first.failed())
                for (auto &r : window) r->job.abandon();
            idle.close();
            preparing.join();
        }
        // handles that worked go back to the pool (their buffers stay allocated for the next batch); after a failure none
        // of them is trusted again
        for (ppcr_ctx *c : handles) {
            (void)hipSetDevice(dev);
            if (first.failed() || hipStreamSynchronize(c->stream) != hipSuccess || !batch_pool().give(dev, c)) ppcr_destroy(c);
        }
    };
    // everything else (unbounded or very wide searches: the host-paced loop): a thread per lane, each with its own handle
    std::vector<std::atomic<int64_t>> next(n_devices);
    for (auto &a : next) a.store(0);
    auto worker = [&](int d) {
        ppcr_ctx *c = batch_pool().take(device_ids[d]);
        int rc = c ? PPCR_OK : ppcr_create(device_ids[d], &c);
        if (rc != PPCR_OK) {
            first.set(rc, ppcr_last_error(nullptr));
            return;
        }
        for (;;) {
            const int64_t k = next[d].fetch_add(1);
            const int64_t p = (int64_t)d + k * n_devices;
            if (p >= n_pairs || first.failed()) break;
            const ppcr_pair &pr = pairs[p];
            rc = ppcr_set_params(c, opt->radius, opt->max_neighbours, opt->dof, opt->dim);
            if (rc == PPCR_OK) rc = ppcr_set_target(c, pr.target, pr.n_target, pr.target_stride_bytes);
            if (rc == PPCR_OK) rc = ppcr_set_source(c, pr.source, pr.n_source, pr.source_stride_bytes);
            if (rc == PPCR_OK)
                rc = align_final(c, opt->n_iter, opt->cost_drop_thresh, opt->n_cost_drop_it, opt->q0, opt->t0,
                                 opt->inner_steps, opt->f_tol, T_all + (size_t)p * 12,
                                 n_iter_done ? n_iter_done + p : nullptr);
            if (rc != PPCR_OK) {
                first.set(rc, "pair " + std::to_string(p) + ": " + ppcr_last_error(c));
                break;
            }
        }
        if (rc != PPCR_OK || hipStreamSynchronize(c->stream) != hipSuccess || !batch_pool().give(device_ids[d], c)) ppcr_destroy(c);
    };
    std::vector<std::thread> pool;
    for (int d = 0; d < n_devices; d++) {
        if (two_threads) {
            pool.emplace_back(device_share, d);
            continue;
        }
        const int64_t mine = (n_pairs - d + n_devices - 1) / n_devices;
        const int lanes = (int)std::max<int64_t>(1, std::min<int64_t>(lanes_per_device, mine));
        for (int l = 0; l < lanes; l++) pool.emplace_back(worker, d);
    }
    for (auto &th : pool) th.join();
    if (first.rc != PPCR_OK) return report(first.rc, first.text);
    return PPCR_OK;
}

}  // extern "C"
