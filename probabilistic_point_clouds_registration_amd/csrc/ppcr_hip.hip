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
#include <cstdlib>
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
#include "ppcr_pool.hpp"
#include "ppcr_batch_sched.hpp"
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

// A device buffer of a handle: a block of the device's pool (ppcr_pool.hpp), grown on demand, never shrunk.
template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    int dev = 0;  // the device whose pool the block came from
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) {
            // growing: launches in flight may still read the old block (what hipFree used to wait for by itself)
            (void)hipDeviceSynchronize();
            device_pool(dev).free(p);
        }
        p = nullptr;
        cap = 0;
        size_t want = n + n / 8 + 64;
        hipError_t e = hipGetDevice(&dev);
        void *block = nullptr;
        if (e == hipSuccess) e = device_pool(dev).alloc(want * sizeof(T), &block);
        if (e == hipSuccess) {
            p = static_cast<T *>(block);
            cap = want;
        }
        return e;
    }
    // the block back to the pool once nothing in flight can touch it
    void release()
    {
        if (p) (void)hipDeviceSynchronize();
        release_idle();
    }
    // (the caller has made sure of that itself: ppcr_destroy, after synchronising the handle's streams)
    void release_idle()
    {
        if (p) device_pool(dev).free(p);
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
    char *h_pinned = nullptr;                   // the handle's one pinned, device-mapped block; the h_* / d_mbox / d_report pointers are cuts of it
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
    DevBuf<unsigned short> level_cap;  // two buffers x two words per 256-query block of the (sorted) source: cap | floor << 4 | split << 8 (UnansweredRows::level_in / level_out)
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
    // Verlet lists (steady state, ppcr_device.hip.h: VerletLists): option "verlet" 1 (default) / 0, "verlet_skin" in 1e-4 of
    // the radius (default 500: lists reach up to 2 x 0.05 radius beyond the cut-off bound — less where more than sixteen targets lie
    // that close; the grid's cells are that much larger)
    int opt_verlet = 1, opt_verlet_skin = -1;  // (-1: by width — verlet_skin_units)
    // option "verlet_levels": lists in MULTI-LEVEL searches too (default 0).  Built, exact (tests/test_gpu_configs.py runs the
    // command line's defaults on both pinned scenes with it on) and measured on them: the rows nn_wide_kernel searches come
    // there once per list lifetime instead of every iteration (slab with blobs: 83 -> 55 us of that kernel, 4.3 k -> 4.7-4.8 k
    // it/s), but a one-round launch lasts as long as its slowest workgroup, and the dense blocks that never keep lists — or
    // build them, twice the plain scan, every few launches — are exactly those (LiDAR-like scene: K1 119 -> 136 us, 4.5 k ->
    // 4.3 k it/s).  Off until the dense blocks' search itself is cheaper.
    int opt_verlet_levels = 0;
    // WHEN lists are built: once they are predicted to outlive the registration's remaining moves (associate_impl:
    // verlet_lists_pay_off).  Known at enqueue time: the largest displacement of the last rigid move this thread has seen
    // (move_estimate, corners of the target's box), how many iterations old it is (move_lag) and the ratio of the last two
    // such moves (move_ratio; NaN: unknown).  Option "verlet_engage" >= 0 replaces the rule by a fixed threshold on
    // move_estimate in 1e-4 of the radius (tests: 100000 = always, 0 = never); -1 (default): the rule.
    int opt_verlet_engage = -1;
    int opt_verlet_room = -1;  // (experiments: the forecast must fit this many per cent of the skin; -1: the rule's own)
    double move_estimate = std::numeric_limits<double>::infinity();
    double move_ratio = std::numeric_limits<double>::quiet_NaN();
    int move_lag = 1;
    bool pending_from_align = false;  // the pending move is an align loop's last transform (its ratio is already known)
    bool move_forecast_pinned = false;  // an align loop set the three values above for the association being enqueued
    // what the last align loop knew about the registration's moves when it ended (AlignJob: estimate_of / ratio_of, the
    // entry's values, iterations so far): a following align call on the untouched handle continues the SAME sequence of
    // forecasts — align(a) + align(b) launches the kernels of align(a + b)
    double al_estimate[4] = {0, 0, 0, 0}, al_ratio[4] = {0, 0, 0, 0}, al_entry_estimate = 0, al_entry_ratio = 0, al_last = 0;
    int al_iterations = 0;
    int opt_verlet_order = 1;    // option "verlet_order": workgroups forecast to search again are dispatched first (default 1)
    double grid_skin2 = 0;       // 2 x skin the grid in use was built for (0: its cells do not cover a list's reach)
    int opt_verlet_dense = 0;      // 1 / 2: keep the lists' cells whatever the halo estimate says, small / large tile (tests)
    unsigned assoc_counter = 0;    // associations that found (nearly) every block handed over (associate_impl: list_all)
    double halo_rho = 0, halo_h = 0;  // what that verdict was reached from (0: this grid was not judged): target density, skinned cell edge
    bool verlet_big_tile = false;  // this grid: lists in the 2240-candidate tile (halo_outgrows_verlet_tile)
    bool verlet_grid_off = false;  // this grid: no skin — the halo of a block would outgrow the Verlet variant's tile (grid_finish)
    bool verlet_ok = false;      // the rows' lists were (re)built or verified by the previous association and nothing moved the source since but K1's own prologue
    DevBuf<int> vl;
    DevBuf<unsigned char> vn;
    DevBuf<unsigned> vmask;
    DevBuf<unsigned> vcount;     // [0]: diagnostic rebuild counter; [16 .. 64): three sets of dispatch-order counters
    DevBuf<int> vorder;          // two dispatch orders (this launch's, the next one's), 2 x 8 x ceil(grid / 8) slots each
    unsigned verlet_launches = 0;  // Verlet launches enqueued on this handle (the rotation of the order buffers and counters)
    bool verlet_order_ok = false;  // the previous association filed a dispatch order for this one
    bool verlet_order_used = false;  // (diagnostic) the last Verlet launch took its workgroups in the filed order
    DevBuf<float> vg2, vacc;
    DevBuf<unsigned char> vstreak;  // VerletLists::streak, one byte per workgroup slot of the largest K1 grid
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
    // PPCR_TRACE=1 in the environment: host wall-clock marks of one ppcr_align (where a call's fixed cost goes), printed
    // to stderr when the call returns
    std::vector<std::pair<const char *, double>> trace;
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
const bool g_trace = [] { const char *e = std::getenv("PPCR_TRACE"); return e && *e && *e != '0'; }();
// (per-iteration marks only while the trace is short — the first dozen iterations; closing marks always)
inline void trace_mark(ppcr_ctx *c, const char *what, bool closing = false)
{
    if (g_trace && (closing || c->trace.size() < 96))
        c->trace.emplace_back(what, std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count());
}
void trace_print(ppcr_ctx *c)
{
    if (!g_trace || c->trace.empty()) return;
    std::string line = "PPCR_TRACE";
    char buf[96];
    for (size_t k = 0; k < c->trace.size(); k++) {
        std::snprintf(buf, sizeof(buf), " %s=%.1f", c->trace[k].first, (c->trace[k].second - c->trace[0].second) * 1e6);
        line += buf;
    }
    std::fprintf(stderr, "%s\n", line.c_str());
    c->trace.clear();
}
}  // namespace

// The rest of this translation unit, in reading order:
#include "ppcr_hip_setup.inc"      // errors, uploads, K0: grid build in two halves, levels, source sort
#include "ppcr_hip_iteration.inc"  // K1 / K23 / fold-and-solve launches of one iteration, association export
#include "ppcr_hip_api.inc"        // C ABI: handles, parameters, options, clouds, associations, single steps
#include "ppcr_hip_align.inc"      // the registration loop (AlignJob) and ppcr_align / ppcr_align_report
#include "ppcr_hip_extras.inc"     // companion cloud, reports, VoxelGrid, 1-NN distances
#include "ppcr_hip_batch.inc"      // ppcr_align_many, ppcr_batch_run
