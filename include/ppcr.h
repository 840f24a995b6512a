/*
 * ppcr.h — C ABI of the MI355X-native probabilistic point-cloud registration hot path.
 *
 * This is the drop-in boundary: everything above it (the C++ classes that mirror
 * prob_point_cloud_registration::*, the CLI, the Python binding) is host code; everything
 * below it is hand-written HIP for gfx950.  Plain pointers and sizes only.
 *
 * The reference (iralabdisco/probabilistic_point_clouds_registration) has no FFI seam of its
 * own — it is one C++ library — so each entry point below names the reference code it
 * replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - every function returns PPCR_OK (0) or a negative ppcr_status; no exception crosses;
 *   - the caller owns host buffers, the handle owns device buffers;
 *   - one handle <-> one device + one HIP stream; a handle is not thread-safe, distinct
 *     handles are independent;
 *   - quaternions are (w,x,y,z) (prob_point_cloud_registration_params.hpp:14), need not be
 *     normalised on input (ceres::QuaternionRotatePoint normalises, error_term.hpp:31);
 *   - T[12] is the top three rows of a 4x4 rigid transform, row-major: [R|t];
 *   - clouds are float32 xyz with a byte stride (12 = packed, 16 = pcl::PointXYZ).
 */
#ifndef PPCR_H
#define PPCR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPCR_ABI_VERSION 1

/* number of doubles in the moment vector produced by ppcr_accumulate():
 *   [0] W=sum w  [1..3] sum w(x-c)  [4..6] sum w(y-c)  [7..15] sum w(x-c)(y-c)^T row-major
 *   [16] sum w*s  [17] sum w|x-c|^2  [18] sum w|y-c|^2      c = ppcr_get_origin() */
#define PPCR_NSUMS 19

typedef enum ppcr_status {
    PPCR_OK = 0,
    PPCR_ERR_INVALID = -1,  /* bad argument */
    PPCR_ERR_STATE = -2,    /* call order: e.g. associate before set_target */
    PPCR_ERR_HIP = -3,      /* a HIP runtime call failed; see ppcr_last_error */
    PPCR_ERR_NOMEM = -4,
    PPCR_ERR_NODEVICE = -5  /* no usable GPU: this library has NO CPU fallback */
} ppcr_status;

typedef struct ppcr_ctx ppcr_ctx;

int ppcr_abi_version(void);
/* number of visible HIP devices (0 without a GPU; never initialises a context) */
int ppcr_device_count(int *count);

int ppcr_create(int device_id, ppcr_ctx **out);
int ppcr_destroy(ppcr_ctx *ctx);
/* message of the last failing call on this handle (ctx may be NULL for create failures) */
const char *ppcr_last_error(const ppcr_ctx *ctx);

/* ProbPointCloudRegistrationParams {radius, max_neighbours, dof} and the DIMENSIONS constant
 * (prob_point_cloud_registration_params.hpp:5-18, ..._iteration.hpp:17,29).
 * dof = +inf selects the Gaussian model (probabilistic_weights.hpp:35, CLI -u ..._ex.cc:93-97).
 * max_neighbours <= 0 or >= N_target means unbounded (PCL radiusSearch max_nn rule). */
int ppcr_set_params(ppcr_ctx *ctx, double radius, int max_neighbours, double dof, int dim);

/* Upload clouds (host pointers).  Replaces the cloud members the reference keeps
 * (prob_point_cloud_registration.h:50-55) and, for the target, the per-iteration kd-tree
 * build (src/prob_point_cloud_registration.cc:66-67): the uniform grid over the target is
 * built once per (target, radius, max_neighbours).  When ppcr_set_params has configured a bounded
 * search, ppcr_set_target starts that build on a second stream before it returns, so that it runs
 * while the caller's ppcr_set_source copies over PCIe (option "eager_grid"); otherwise, and
 * whenever the parameters or a grid option change afterwards, it is built at the first
 * ppcr_associate() / ppcr_align().  Both uploads return once the caller's buffer has been read. */
int ppcr_set_target(ppcr_ctx *ctx, const float *xyz, int64_t n, int64_t stride_bytes);
int ppcr_set_source(ppcr_ctx *ctx, const float *xyz, int64_t n, int64_t stride_bytes);
/* Same, from device memory already resident on the handle's GPU (e.g. a torch tensor). */
int ppcr_set_target_device(ppcr_ctx *ctx, const void *d_xyz, int64_t n, int64_t stride_bytes);
int ppcr_set_source_device(ppcr_ctx *ctx, const void *d_xyz, int64_t n, int64_t stride_bytes);

/* K1 — radius-NN correspondence search for every source point.  Replaces the radiusSearch
 * loop + triplet/CSR assembly of src/prob_point_cloud_registration.cc:66-83. */
int ppcr_associate(ppcr_ctx *ctx);
int ppcr_association_size(ppcr_ctx *ctx, int64_t *n_rows, int64_t *nnz);
/* CSR export in the layout Eigen::SparseMatrix<double,RowMajor> would have after
 * setFromTriplets (cc:82-83): rows = source index, columns ascending = target index.
 * d2 (nullable) = float squared distance source->target for the CURRENT source positions. */
int ppcr_get_association(ppcr_ctx *ctx, int32_t *row_ptr, int32_t *col, float *d2);
/* Install a caller-made association (the data_association argument of
 * ProbPointCloudRegistrationIteration, ..._iteration.hpp:24-27). */
int ppcr_set_association(ppcr_ctx *ctx, const int32_t *row_ptr, const int32_t *col, int64_t n_rows);

/* K2 — WeightUpdaterCallback::operator() (weight_updater_callback.hpp:36-63):
 * s = |y - (R(q)x + t)|^2 per stored pair (ErrorTerm::operator(), error_term.hpp:21-37) and
 * ProbabilisticWeights::updateWeights (probabilistic_weights.hpp:48-105).  Outputs are in the
 * CSR order of ppcr_get_association(); either may be NULL. */
int ppcr_weights(ppcr_ctx *ctx, const double q[4], const double t[3], double *w_out, double *s_out);

/* ProbabilisticWeights(dof, dim, max_nb).updateWeights(data_association, squared_errors)
 * (probabilistic_weights.hpp:30-105) as a stateless call: caller-supplied squared errors in CSR
 * order, weights out in the same order.  (max_neighbours only sizes a reserve() in the reference.) */
int ppcr_update_weights(int device_id, const int32_t *row_ptr, int64_t n_rows, const double *sq_errors,
                        double dof, int dim, double *w_out);

/* K2+K3 fused — weights at (q,t) and the weighted moments the closed-form solve needs
 * (what Ceres would assemble from the residual blocks of ..._iteration.hpp:37-46). */
int ppcr_accumulate(ppcr_ctx *ctx, const double q[4], const double t[3], double sums[PPCR_NSUMS]);
int ppcr_get_origin(ppcr_ctx *ctx, double c[3]);

/* Host-side closed-form minimiser (3x3 SVD) of sum w|y - Rx - t|^2 from the moments: replaces
 * ceres::Solve for fixed weights (..._iteration.hpp:52-57).  Pure host functions. */
int ppcr_solve_moments(const double sums[PPCR_NSUMS], const double origin[3], double R[9], double t[3]);
double ppcr_cost_from_moments(const double sums[PPCR_NSUMS], const double origin[3], const double R[9],
                              const double t[3]);

/* ProbPointCloudRegistrationIteration::solve + transformation() (..._iteration.hpp:52-67) on the
 * current association: IRLS from (q0,t0) — weights, closed-form solve, repeat — until max_steps
 * or relative cost decrease <= f_tol (Ceres function_tolerance, cc:97).
 * cost_out = {initial_cost, final_cost} with Ceres' 1/2 factor; steps_out = solves performed. */
int ppcr_solve(ppcr_ctx *ctx, const double q0[4], const double t0[3], int max_steps, double f_tol,
               double T_out[12], double cost_out[2], int *steps_out);

/* K4 — pcl::transformPointCloud(source, source, T) in place: f64 math, f32 store (cc:110-112). */
int ppcr_apply_transform(ppcr_ctx *ctx, const double T[12]);

/* One outer iteration of ProbPointCloudRegistration::align (cc:65-131), resident on the
 * device: associate -> solve from (q0,t0) -> move the source by the incremental transform. */
int ppcr_iterate(ppcr_ctx *ctx, const double q0[4], const double t0[3], int inner_steps, double f_tol,
                 double T_out[12], double cost_out[2], int *steps_out);

/* ProbPointCloudRegistration::hasConverged (cc:138-158) as a plain state machine, so that ppcr_align, the C++ class
 * and any other host front end share ONE definition of the stopping rule.  Zero-initialise, call ppcr_stop_rule_check
 * before every outer iteration, and after each iteration store its relative cost drop
 * (initial_cost - final_cost) / initial_cost (cc:119) and add one to `iteration` (cc:130).
 * Quirks kept on purpose: cost_drop starts at 0, so the first check already counts as an idle iteration; a NaN drop
 * (empty association, 0/0) compares false and resets the idle count, so such a run lasts n_iter iterations;
 * n_iter < 0 never equals the iteration count: no cap. */
typedef struct ppcr_stop_rule {
    int32_t iteration; /* outer iterations finished so far (current_iteration_) */
    int32_t idle;      /* consecutive checks that saw a cost drop below the threshold */
    double cost_drop;  /* relative cost drop of the last iteration */
} ppcr_stop_rule;
enum { PPCR_CONTINUE = 0, PPCR_STOP_MAX_ITERATIONS = 1, PPCR_STOP_COST_DROP = 2 };
/* returns PPCR_CONTINUE or the reason to stop; pure host arithmetic (no device is touched) */
int ppcr_stop_rule_check(ppcr_stop_rule *rule, int n_iter, double cost_drop_thresh, double n_cost_drop_it);

/* The whole align() loop including hasConverged() (cc:63-158).  history (n_iter*12 doubles,
 * cumulative transforms T_cum <- T_k*T_cum), costs (n_iter*2), steps (n_iter ints) may be NULL.
 * n_done receives the number of outer iterations performed.
 * n_iter < 0 = no iteration cap (the reference's meaning): then history, costs and steps must be NULL and
 * cost_drop_thresh > 0 (PPCR_ERR_INVALID otherwise: the arrays could not be sized / the loop would never end).
 * For every bounded search (0 < max_neighbours <= 32) the whole iteration runs on the device — association, the IRLS
 * inner loop with its function_tolerance test, the solve — and the next association takes its source move from device
 * memory, so the device works one iteration ahead of this thread — but an iteration is only enqueued early when
 * hasConverged() cannot stop before it whatever the pending cost turns out to be, so the iterations performed, and
 * every number returned, are those of the one-at-a-time loop.
 * Like ppcr_iterate, the call leaves its last move pending: it is applied by whatever reads the source next
 * (ppcr_get_source, ppcr_get_association, the reports ...), or carried in the first association of a following
 * ppcr_align / ppcr_iterate, which then continues the loop exactly where this call stopped (same cut-off state). */
int ppcr_align(ppcr_ctx *ctx, int n_iter, double cost_drop_thresh, double n_cost_drop_it,
               const double q0[4], const double t0[3], int inner_steps, double f_tol, double *history,
               double *costs, int32_t *steps, int *n_done);

/* The same loop with what ProbPointCloudRegistration::align() does BETWEEN iterations delivered through a callback
 * (cc:114-129: the verbose log line, the mean distance to the ground-truth cloud, the --dump row with the mean
 * distance every point moved) — so that the C++ class and the CLI run the device-paced loop instead of one blocking
 * call per iteration.  on_iteration (nullable) is called on the calling thread, once per outer iteration and in order,
 * when that iteration's numbers reach the host: the device may already be working on the next one, so the call comes
 * "one iteration late"; what it is given is exactly what the one-at-a-time loop would have computed.
 *   report_flags  PPCR_REPORT_TRUTH: mse_truth = calculateMSE(tracked cloud after the move, ground truth) (cc:115;
 *                 needs ppcr_set_ground_truth), PPCR_REPORT_MOVED: moved = calculateMSE(tracked cloud after the move,
 *                 the same cloud before it) (cc:121); the tracked cloud is the companion when one is set, else the
 *                 source.  Unrequested values are NaN.  Ignored without a callback.
 *   rule_io       (nullable) in: the hasConverged() state to continue from, out: the state the loop stopped in — so a
 *                 front end that owns a ppcr_stop_rule (the C++ class) sees what it would have seen driving
 *                 ppcr_stop_rule_check + ppcr_iterate itself.
 *   T_final       (nullable) the cumulative transform of the iterations of THIS call (identity when none ran).
 *   n_done        (nullable) rule_io->iteration on return: the iterations counted so far, earlier calls included.
 * n_iter < 0 (no cap) is allowed with cost_drop_thresh > 0. */
typedef struct ppcr_iteration_info {
    int32_t iteration;    /* index of the outer iteration (ppcr_stop_rule.iteration before it was counted) */
    int32_t inner_steps;  /* Summary::num_successful_steps analogue: IRLS steps of this iteration */
    double cost[2];       /* initial_cost, final_cost */
    double T_step[12];    /* this iteration's increment */
    double T_cum[12];     /* T_k * ... * T_1 of this call */
    double mse_truth;     /* PPCR_REPORT_TRUTH, else NaN */
    double moved;         /* PPCR_REPORT_MOVED, else NaN */
} ppcr_iteration_info;
typedef void (*ppcr_iteration_fn)(void *user, const ppcr_iteration_info *info);
enum { PPCR_REPORT_TRUTH = 1, PPCR_REPORT_MOVED = 2 };
int ppcr_align_report(ppcr_ctx *ctx, int n_iter, double cost_drop_thresh, double n_cost_drop_it, const double q0[4],
                      const double t0[3], int inner_steps, double f_tol, ppcr_stop_rule *rule_io, int report_flags,
                      ppcr_iteration_fn on_iteration, void *user, double T_final[12], int *n_done);

/* ---- multi-pair batches (BASELINE configs[4]; the reference registers one pair per process run) ----
 * Independent pairs never exchange data, so a batch is a work list: pair p is registered on
 * device_ids[p % n_devices] by one of `lanes_per_device` host worker threads of that device (each with its
 * own handle and HIP stream, so one pair's upload / 3x3 host solve overlaps another pair's kernels).
 * In the one-process-per-GPU deployment (torch.distributed / RCCL) each rank passes its own pairs and
 * n_devices = 1; the final gather of the transforms is the caller's (the only collective of the job). */
typedef struct ppcr_pair {
    const float *source;          /* host, float32 xyz */
    int64_t n_source;
    int64_t source_stride_bytes;
    const float *target;
    int64_t n_target;
    int64_t target_stride_bytes;
} ppcr_pair;

/* the knobs of ProbPointCloudRegistrationParams (..._params.hpp:5-18) that reach the loop */
typedef struct ppcr_batch_options {
    double radius;
    double dof;              /* +inf => Gaussian (-u) */
    double cost_drop_thresh;
    double n_cost_drop_it;
    double f_tol;            /* Ceres function_tolerance, cc:97 */
    double q0[4];            /* initial_rotation (w,x,y,z) */
    double t0[3];            /* initial_translation */
    int32_t max_neighbours;
    int32_t dim;
    int32_t n_iter;
    int32_t inner_steps;
} ppcr_batch_options;

/* T_all: n_pairs*12 doubles (final cumulative [R|t] of each pair, identity when no iteration ran);
 * n_iter_done: n_pairs ints or NULL.  On failure the first error text is copied to err (may be NULL).
 * Pair p runs on device_ids[p % n_devices].  Threads: two or three per device for bounded searches (one — two from three lanes on —
 * prepares pairs on idle handles: uploads, grid build, source sort; one keeps up to lanes_per_device registrations in
 * flight and only enqueues and polls); one per lane for unbounded / very wide searches (host-paced loop). */
int ppcr_batch_run(const ppcr_pair *pairs, int64_t n_pairs, const ppcr_batch_options *opt,
                   const int *device_ids, int n_devices, int lanes_per_device, double *T_all,
                   int32_t *n_iter_done, char *err, int64_t err_capacity);

/* ppcr_batch_run keeps its handles (streams, mailboxes, device buffers) between calls; this frees them.  Call it before
 * unloading the library or to give the memory back; never required for correctness. */
int ppcr_batch_release(void);

/* Device memory of the handles (no counterpart in the reference: its clouds live in host memory owned by PCL).
 * Every handle's buffers are blocks of a per-device pool: the library takes memory from the driver in a few large
 * slabs, sub-allocates them on the host side, and keeps what ppcr_destroy gives back for the next handle (at most
 * PPCR_POOL_KEEP_MB of wholly unused slabs, default 4096; the environment variable is read once) — a fresh handle's
 * first registration then makes no driver call at all.
 * ppcr_memory_stats: bytes held from the driver, bytes in use by live buffers, hipMalloc calls made so far (any pointer
 * may be NULL).  ppcr_memory_trim: hand every unused slab (and the pinned blocks of destroyed handles) back to the
 * driver now — synchronises the device; call it when other users of the GPU need the memory. */
int ppcr_memory_stats(int device_id, uint64_t *reserved_bytes, uint64_t *in_use_bytes, uint64_t *driver_allocs);
int ppcr_memory_trim(int device_id);

/* The same loop over handles whose clouds are already resident (set_source/set_target done by the caller):
 * ppcr_align on each of the n handles, `lanes` of them in flight at a time on their own streams.
 * T_final: n*12 doubles; n_done: n ints or NULL.  Handles may live on different devices.
 * Handles whose loop the device paces (bounded searches: see ppcr_align) are all driven from the CALLING thread (it
 * enqueues and polls their mailboxes), so `lanes` costs no host threads; otherwise `lanes` worker threads are used. */
int ppcr_align_many(ppcr_ctx *const *ctxs, int n, int lanes, int n_iter, double cost_drop_thresh,
                    double n_cost_drop_it, const double q0[4], const double t0[3], int inner_steps,
                    double f_tol, double *T_final, int32_t *n_done);

/* ---- the one collective of the multi-GPU deployment, native (north star: "RCCL over xGMI only for the final gather of
 * transforms") ----
 * One process per GPU: rank r registers the pairs p with p % world == r on its device (ppcr_batch_run with n_devices = 1,
 * or ppcr_align_many), then every rank calls ppcr_gather_transforms and holds all n_pairs transforms.  The communicator
 * is an RCCL communicator (ncclCommInitRank); RCCL is bound at run time (librccl.so.1), so single-GPU users of this
 * library do not need it.  Bootstrap: rank 0 draws an id (ppcr_comm_get_id = ncclGetUniqueId) and hands its 128 bytes to
 * the other ranks by whatever means the launcher has (the command line's --rendezvous file); ppcr_comm_create is
 * collective over the ranks.  The reference has no counterpart (it registers one pair per process run). */
#define PPCR_COMM_ID_BYTES 128
typedef struct ppcr_comm ppcr_comm;
int ppcr_comm_get_id(unsigned char id[PPCR_COMM_ID_BYTES]);
int ppcr_comm_create(int device_id, int rank, int world, const unsigned char id[PPCR_COMM_ID_BYTES], ppcr_comm **out);
int ppcr_comm_destroy(ppcr_comm *comm);
/* T_local: this rank's pairs in ascending pair index, 12 doubles each ([R|t] rows); T_all: n_pairs * 12 doubles, filled
 * on EVERY rank (one ncclAllGather of ceil(n_pairs / world) * 12 doubles per rank: a few hundred bytes, latency-bound). */
int ppcr_gather_transforms(ppcr_comm *comm, const double *T_local, int64_t n_pairs, double *T_all);
/* message of the last failing ppcr_comm_* / ppcr_gather_transforms call on this thread */
const char *ppcr_comm_last_error(void);

/* current (moved) source in the caller's original point order */
int ppcr_get_source(ppcr_ctx *ctx, float *xyz, int64_t stride_bytes);

/* ---- the steps either side of the loop (reporting and down-sampling) ------------------------------------------
 * The reference moves TWO copies of the source every iteration (the full cloud and the voxel-filtered one the
 * association runs on, cc:110-112) and reports mean point distances on the full one (cc:114-122).
 * ppcr_set_companion gives the handle that full-resolution copy: it is moved on the device by every transform
 * applied to the source (same f64 -> f32 arithmetic) and is what the two reports below look at; without a
 * companion they look at the source itself.  All clouds are in the caller's index order. */
int ppcr_set_companion(ppcr_ctx *ctx, const float *xyz, int64_t n, int64_t stride_bytes);
int ppcr_get_companion(ppcr_ctx *ctx, float *xyz, int64_t stride_bytes);
int ppcr_set_ground_truth(ppcr_ctx *ctx, const float *xyz, int64_t n, int64_t stride_bytes);
/* calculateMSE(source, ground_truth) (utilities.hpp:16-26: the mean Euclidean distance of index-paired points);
 * PPCR_ERR_INVALID when the sizes differ (the reference asserts). */
int ppcr_mse_ground_truth(ppcr_ctx *ctx, double *mse);
/* calculateMSE(source, source at the previous call), then remembers the current cloud (cc:121-122).
 * The first call reports 0 and only takes the snapshot; mse may be NULL. */
int ppcr_mse_previous(ppcr_ctx *ctx, double *mse);

/* pcl::VoxelGrid<PointXYZ> centroid down-sampling with leaf (l,l,l) (cc:24-41): one output point per occupied
 * voxel, voxels in ascending index; out_xyz must hold n points.  Stateless (temporary handle on device_id). */
int ppcr_voxel_filter(int device_id, const float *xyz, int64_t n, int64_t stride_bytes, float leaf, float *out_xyz,
                      int64_t out_stride_bytes, int64_t *n_out);

/* Squared distance from every query to its exact nearest target (pcl::KdTreeFLANN::nearestKSearch with k = 1, as the
 * evaluation metrics of utilities.hpp:28-234 call it: averageClosestDistance, sumSquaredError, the robust and median
 * variants).  d2_out: nq floats in query order.  Stateless (temporary handle on device_id); nt must be > 0. */
int ppcr_nearest_sq_distances(int device_id, const float *queries, int64_t nq, int64_t q_stride_bytes,
                              const float *targets, int64_t nt, int64_t t_stride_bytes, float *d2_out);

int ppcr_synchronize(ppcr_ctx *ctx);

/* Per-kernel device timing with HIP events recorded on the handle's own stream. */
typedef struct ppcr_kernel_stat {
    char name[48];
    int64_t launches;
    double total_ms;
} ppcr_kernel_stat;
int ppcr_profile_enable(ppcr_ctx *ctx, int enable); /* also clears accumulated stats */
int ppcr_profile_get(ppcr_ctx *ctx, ppcr_kernel_stat *out, int capacity, int *n_out);

/* Tuning / debugging knobs (never change results): key is one of
 *   "sort_source"  0 keep caller order, 1 brick order, boustrophedon (default; see "brick_x");
 *   "temporal"     1 start each query's cut-off from its previous m-th distance (default), 0 off;
 *   "run_ahead"    1 ppcr_align keeps the device one iteration ahead of the host when the stopping rule allows
 *                  (default), 0 one iteration at a time;
 *   "fuse_k23"     1 ppcr_align's one-step iterations compute the weighted moments inside the association kernel
 *                  (default; a different summation order: results agree to rounding), 0 separate kernel;
 *   "merge_fold"   1 with "fuse_k23": the fold-and-solve step rides in the cleanup launch when the handle has the GPU
 *                  to itself (default), 0 always its own launch;
 *   "defer_moves"  1 ppcr_apply_transform leaves the move to the prologue of the next association (what ppcr_iterate
 *                  and ppcr_align always do; the temporal cut-off then survives the move), 0 moves at once (default);
 *   "inner_dev_steps"  IRLS steps beyond the first that ppcr_align enqueues for the device per outer iteration
 *                  (default 3, 0..8); an inner loop that needs more is finished by the host, one step at a time;
 *   "grid_xf"      x slices per grid cell, 1/2/4/8 (default 4; set before the target): every stencil row is clipped
 *                  to the x window the search sphere needs in that row;
 *   "brick_x"      x extent in cells of the 4x4 (y,z) bricks the source is ordered by: 1 (default), 2 or 4;
 *   "short_lists"  1 once the temporal cut-off is valid K1 runs with 16-slot lists, a 1728-candidate halo and five
 *                  workgroups per CU (default), 0 always 32 slots / three workgroups;
 *   "levels"       -1 / 1 the target is binned at several resolutions when its density varies widely or the radius holds far
 *                  more than max_neighbours points, and every block of queries searches the finest level that covers its
 *                  cut-off radii (default; uniform clouds searched with a radius of a few spacings keep one level),
 *                  0 always one level (set before the first association);
 *   "verlet"       1 the steady-state K1 keeps per-row Verlet lists and answers a workgroup's rows from them for as long as
 *                  every list provably holds every target the exact search could return (default) — one-pass searches of up
 *                  to 20 neighbours (16 slots per row up to 10 neighbours, 32 beyond), two-pass searches of 11 .. 20: the
 *                  command line's own defaults —, 0 always search, 2 build the lists in every launch and never trust them
 *                  (testing);
 *   "verlet_levels"  1 lists in multi-level searches too, 0 not (default: measured on the two pinned non-uniform scenes, a
 *                  gain on one and a loss on the other: csrc/ppcr_hip.hip);
 *   "verlet_skin"  how far a list reaches beyond what the row needs, in 2e-4 of the (first-pass) search radius (1..2000;
 *                  default by width: 500 — 0.1 radius — up to 10 neighbours, 350 beyond; set before the first association:
 *                  the grid's cells grow by twice the skin);
 *   "verlet_engage"  -1 (default) lists are built once the moves the registration still has to make — forecast from the last
 *                  known rigid move and the ratio of the last two — fit the lists' skin, and dropped when the forecast exceeds
 *                  four times that; >= 0: a fixed threshold instead, lists are built once the last known rigid move displaces
 *                  no corner of the target's box by more than this many 1e-4 radii (0 never .. 100000 always: tests);
 *   "verlet_room"  -1 (default) the forecast must fit the whole skin (half of it on grids that are resident all at once, widths
 *                  up to 10), >= 0: this many per cent of it (experiments: tools/exp_verlet_sweep.py);
 *   "verlet_order" 1 workgroups that will probably search are dispatched first (default), 0 launch order;
 *   "verlet_dense" 0 the list variant's LDS tile by an estimate of a 256-query block's halo — the 1920-candidate tile, the
 *                  2240-candidate one for denser clouds (three workgroups per CU), no lists (and radius-sized grid cells)
 *                  where the halo would outgrow that too: a radius that holds ~38 or more target points, a source much
 *                  sparser than the target — (default), 1 lists in the small tile regardless, 2 lists in the large tile
 *                  regardless (tests; set before the first association);
 *   "two_pass"     1 a bounded search whose radius holds far more than max_neighbours target points runs in two passes
 *                  (default): the grid and the tiled kernel work with radius / k, chosen from the target's density, and
 *                  only the rows that find fewer than max_neighbours there are searched again with the full radius;
 *                  0 always one pass with radius-sized cells; 2..8 force that many first-pass cells per radius (set
 *                  before the first association);
 *   "fuse_max_handed_over"  K23 is folded into K1 (and the rows of handed-over workgroups are redone by the cleanup role of
 *                  the second launch) only while the last association heard from handed over at most this many workgroups
 *                  (default 4); beyond, K23 runs as its own kernel and those rows go to nn_wide_kernel, one row per wave;
 *   "first_pass_occupancy"  tenths of a target point per first-pass grid cell the automatic choice allows where the typical
 *                  point lives (default 110); fuller cells: fewer short rows, more workgroups whose halo outgrows the LDS tile;
 *   "first_pass_fill"  tenths: the first-pass sphere of a two-pass search should hold this many times max_neighbours points
 *                  where the density allows (default 22: ~2.2 m candidates answer nearly every row in the first pass);
 *   "eager_grid"   1 ppcr_set_target starts the grid build (default, see there), 0 the first association builds it;
 *   "stamps", "fold_stamps", "level_stats", "debug_mbox_seq"  diagnostics (per-phase cycle counts of K1, wall-clock stamps
 *                  of the fold-and-solve lane, per-level counters of a multi-level search, the sequence counter for tests). */
int ppcr_set_option(ppcr_ctx *ctx, const char *key, int value);

#ifdef __cplusplus
}
#endif
#endif /* PPCR_H */
