// K1, the LDS-tiled radius search with max_neighbours cut-off: nn_fast_kernel (steady state and first association) and
// nn_tile_cleanup_kernel (the general flavour, run on what the fast kernel hands over).  Templates only; instantiated by
// ppcr_nn_tile.hip, one translation unit per compiled-in list width M.
#pragma once
#include "ppcr_device.hip.h"

namespace ppcr {
namespace dev {

// ---------------------------------------------------------------------------------------------
// K1, tiled variant (default).  rocprofv3 on the list variant: the scan is bound by the texture
// address path (TA busy 79 %, 16 cycles per 64-lane dwordx4 load: every distance test pulls 16 B
// per lane through L1) and its selection passes re-gather lines L1 has already evicted.  So the
// candidates are staged in LDS instead:
//   1. the workgroup's 256 (spatially compact) queries -> bounding box in cells, +-1 cell halo;
//   2. every halo row (fixed y,z; contiguous in the cell-sorted target) is copied into LDS with
//      lane-contiguous 16-byte loads — each target point is fetched once per workgroup;
//   3. each lane walks ITS OWN 27-cell stencil (9 runs) out of LDS (ds_read_b128) — the exact
//      candidate set, no extra distance tests;
//   4. in-radius candidates are appended to a lane-private u16 list of LDS indices; the top-m
//      cut-off is applied afterwards with the v_med3 threshold selection (select_top_m below),
//      now reading LDS only.
// A halo that does not fit (sparse or unsorted source) is retried per wave, and as a last resort
// the wave falls back to scanning global memory with the same selection code.
// ---------------------------------------------------------------------------------------------
constexpr int kTileRows = 128;   // halo rows per staging

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL load
// (s_waitcnt vmcnt(0)): in the K1 kernels that serialises the run-bound loads issued in the prologue with the
// row-table and staging loads behind the barrier; with the LDS-only fences they stay in flight across it.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// wave-wide scans and reductions on the DPP row_shr / row_bcast network
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_pull(int identity, int v)
{
    return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xf, false);
}
// inclusive scan over the 64 lanes (ALL lanes must be active); lane 63 ends up with the reduction
template <class Op>
__device__ __forceinline__ int wave_scan(int v, int identity, Op op)
{
    v = op(v, dpp_pull<0x111, 0xf>(identity, v));  // row_shr:1
    v = op(v, dpp_pull<0x112, 0xf>(identity, v));  // row_shr:2
    v = op(v, dpp_pull<0x114, 0xf>(identity, v));  // row_shr:4
    v = op(v, dpp_pull<0x118, 0xf>(identity, v));  // row_shr:8   -> inclusive within each row of 16
    v = op(v, dpp_pull<0x142, 0xa>(identity, v));  // row_bcast:15 into rows 1 and 3
    v = op(v, dpp_pull<0x143, 0xc>(identity, v));  // row_bcast:31 into rows 2 and 3
    return v;
}
struct OpMin { __device__ __forceinline__ int operator()(int a, int b) const { return a < b ? a : b; } };
struct OpMax { __device__ __forceinline__ int operator()(int a, int b) const { return a > b ? a : b; } };
struct OpAdd { __device__ __forceinline__ int operator()(int a, int b) const { return a + b; } };
template <class Op>
__device__ __forceinline__ int wave_reduce(int v, int identity, Op op)  // wave-uniform result
{
    return __builtin_amdgcn_readlane(wave_scan(v, identity, op), 63);
}

template <int BLOCK>
struct LdsCands {  // candidate source = staged halo (SoA in LDS); list entries are LDS indices
    const float *sx, *sy, *sz;
    const unsigned char *srow;  // halo row of every staged candidate: its sorted-target position is
    const int *row_gb, *row_off;  //   row_gb[row] + (LDS index - row_off[row])  (one byte instead of four per candidate)
    unsigned short *list;  // [slot * BLOCK + tid]
    int tid;
    __device__ __forceinline__ float4 get(int e) const { return make_float4(sx[e], sy[e], sz[e], 0.f); }
    __device__ __forceinline__ int load(int t) const { return list[t * BLOCK + tid]; }
    __device__ __forceinline__ void store(int t, int e) const { list[t * BLOCK + tid] = (unsigned short)e; }
    __device__ __forceinline__ int pos_of(int e) const
    {
        const int r = srow[e];
        return row_gb[r] + (e - row_off[r]);
    }
    __device__ __forceinline__ unsigned orig_of(int e, const float4 *__restrict__ tgt) const
    {
        return (unsigned)__float_as_int(tgt[pos_of(e)].w);
    }
};
template <int STRIDE = 64>
struct GlobalCands {  // candidate source = global memory; list entries are sorted-target positions
    const float4 *tgt;
    int *list;  // [slot * STRIDE + lane]
    int lane;
    __device__ __forceinline__ float4 get(int e) const { return tgt[e]; }
    __device__ __forceinline__ int load(int t) const { return list[t * STRIDE + lane]; }
    __device__ __forceinline__ void store(int t, int e) const { list[t * STRIDE + lane] = e; }
    __device__ __forceinline__ int pos_of(int e) const { return e; }
    __device__ __forceinline__ unsigned orig_of(int e, const float4 *__restrict__) const
    {
        return (unsigned)__float_as_int(tgt[e].w);
    }
};

// Visit the lane's list entries [0, n) as f(slot, entry, d2 bits), FOUR entries per trip: their index loads, then
// their coordinate loads, are issued together, so a trip costs two LDS round trips instead of eight (selection
// phase -10 % while the source moves; it is mostly instruction-bound: ~36 instructions per entry over two passes).
// f may store to slots <= the one it is called with (in-place compaction): a trip reads before it writes.
template <class S, class F>
__device__ __forceinline__ void for_each_entry(const S &src, float4 q, int n, F &&f)
{
    for (int t = 0; t < n; t += 4) {
        int e[4];
        float4 p[4];
#pragma unroll
        for (int u = 0; u < 4; u++) e[u] = src.load(min(t + u, n - 1));
#pragma unroll
        for (int u = 0; u < 4; u++) p[u] = src.get(e[u]);
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (t + u < n) f(t + u, e[u], __float_as_uint(dist2_flann(q, p[u])));
    }
}

// reduce a lane's list (n > m entries) to its top-m by (d2, original index); returns the new n and
// the threshold T (bit pattern of the m-th smallest d2)
// (Keeping the first 16 entries' d2 bits in registers between the two passes was measured too: selection
//  -20 %, but 176 VGPRs -> 2 waves/SIMD (190 us), or 168 with spills for a net 1 %: not kept.)
template <int M, class S>
__device__ __forceinline__ int select_top_m(const S &src, const float4 *__restrict__ tgt, float4 q, int n, int m,
                                            unsigned &thr)
{
    unsigned K[M];
#pragma unroll
    for (int j = 0; j < M; j++) K[j] = 0xFFFFFFFFu;
    for_each_entry(src, q, n, [&](int, int, unsigned b) { sorted_insert<M>(K, b); });
    const unsigned T = pick<M>(K, m - 1);
    int w = 0, c_eq = 0;
    for_each_entry(src, q, n, [&](int, int e, unsigned b) {
        if (b <= T) {
            src.store(w, e);
            w++;
            c_eq += (b == T) ? 1 : 0;
        }
    });
    if (w > m) {  // more ties at the cut-off than room: lowest original target indices win
        const int need = m - (w - c_eq);
#pragma unroll
        for (int j = 0; j < M; j++) K[j] = 0xFFFFFFFFu;
        for (int t = 0; t < w; t++) {
            const int e = src.load(t);
            if (__float_as_uint(dist2_flann(q, src.get(e))) == T) sorted_insert<M>(K, src.orig_of(e, tgt));
        }
        const unsigned T2 = pick<M>(K, need - 1);
        int w2 = 0;
        for (int t = 0; t < w; t++) {
            const int e = src.load(t);
            const unsigned b = __float_as_uint(dist2_flann(q, src.get(e)));
            if (b < T || src.orig_of(e, tgt) <= T2) {
                src.store(w2, e);
                w2++;
            }
        }
        w = w2;
    }
    thr = T;
    return w;
}

// Where the n-th accepted candidate of a scan goes: the first C fill the list in order, every further one overwrites a
// pseudo-random slot (a hash of its position).  What is left after an overflowing scan is then a spread-out SAMPLE of the
// accepted candidates, not the first C in scan order (which all come from the first one or two stencil rows: their m-th
// smallest distance hardly bounds anything, and the threshold-and-rescan loop below crawled: 8 rounds, then the compacting
// scan — 490 us for one workgroup).  The m-th smallest of ANY subset is a valid bound; of a random subset of C out of N
// it is about the (m N / C)-th smallest of all: every round shrinks the count by m / C.
template <int C>
__device__ __forceinline__ int overflow_slot(int n, int id)
{
    static_assert((C & (C - 1)) == 0, "the overflow slot is a masked hash");
    return n < C ? n : ((id ^ (id >> 5) ^ (id >> 11)) & (C - 1));
}

// A lane's scan over candidates from global memory into its list, for neighbourhoods that may hold far more than C
// candidates: every accepted candidate (d2 bits <= thr) is counted and kept as overflow_slot() says; on overflow the
// threshold drops to the m-th smallest of the C kept — genuine candidates, so a bound of the final m-th distance — and
// the scan is repeated.  `scan(accept)` must call accept(position, d2) for every candidate.
// No selection inside the scan loop: there, the whole wave pays whenever ANY lane's list fills.  Returns the list length
// (<= C); after eight rounds (exact-tie floods) the compacting scan finishes the job.
template <int M, int C, class Cands, class Scan>
__device__ __forceinline__ int scan_global_with_threshold(const Cands &G, const float4 *__restrict__ tgt, float4 q, int m,
                                                          unsigned &thr, Scan &&scan)
{
    int n = 0;
    for (int attempt = 0;; attempt++) {
        n = 0;
        scan([&](int p, float d2) {
            if (__float_as_uint(d2) <= thr) {
                G.store(overflow_slot<C>(n, p), p);
                n++;
            }
        });
        if (n <= C) return n;
        if (attempt == 7) break;
        (void)select_top_m<M>(G, tgt, q, C, m, thr);
    }
    n = 0;
    scan([&](int p, float d2) {
        if (__float_as_uint(d2) <= thr) {
            G.store(n, p);
            n++;
            if (n == C) n = select_top_m<M>(G, tgt, q, n, m, thr);
        }
    });
    return n;
}

__device__ __forceinline__ int halves_slot(int g);  // (SplitTable::all_halves, below)
// GENERAL flavour of K1, run on the workgroups nn_fast_kernel hands over (ovf_list[0 .. *ovf_count)): halos of any
// shape (up to 128 rows), binary subdivision when a halo does not fit, global-memory scan as the last resort, in-loop
// list compaction for dense neighbourhoods.  The source has already been moved by the fast kernel and the temporal
// cut-off is not used here (the fast kernel may have overwritten some of this block's dm2 entries already).
// Persistent workgroups stride over the list.  The list counters ping-pong: launch k counts in ovf_state[k & 1] and the
// fast kernel of launch k clears ovf_state[(k + 1) & 1] (last used by launch k - 1, whose cleanup has finished by then in
// stream order), so nobody needs an atomic ticket (1024 same-address atomics cost this kernel 20 us when it was tried).
// Entries are (index of the handing-over workgroup in the fast kernel's grid) * 4 + half: half 0 = the whole block,
// 1 / 2 = only the queries of waves 0-1 / 2-3 (split blocks).  FTM >= 0: the fast kernel also folded K23 in, so this one
// finishes the rows it redoes the same way (gathering their neighbours from global memory) and fills the slot of the
// partials the fast workgroup left empty.
// MERGED (with FTM >= 0): the launch also carries the fold-and-solve step as its last kNSums workgroups — they wait until
// the cleanup role has finished every listed entry (nothing to wait for in the common case of an empty list) — which
// saves the ~4 us a dependent launch costs even when it has nothing to do.
template <int M, int C, int BLOCK, int CAP, int FTM = -2, bool MERGED = false>
__global__ __launch_bounds__(BLOCK, (C >= 64 ? 2 : 3)) void nn_tile_cleanup_kernel(const float4 *__restrict__ src, int ns,
                                                         const float4 *__restrict__ tgt,
                                                         const int *__restrict__ cell_start, GridDesc g,
                                                         float r2, int m, int *__restrict__ nbr,
                                                         int *__restrict__ cnt, unsigned *__restrict__ dm2,
                                                         const int *__restrict__ ovf_list,
                                                         const unsigned *__restrict__ ovf_count,
                                                         const int *__restrict__ split_list, int n_extra, FusedMoments fm,
                                                         FoldSolve fs)
{
    static_assert(!MERGED || FTM != -2, "the merged launch folds the partials the fused kernels wrote");
    if (loop_aborted(fs.loop.st)) {  // an earlier launch handed the iteration to the host: step aside (see LoopState)
        if constexpr (MERGED)
            if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) publish_skipped(fs);
        return;
    }
    const unsigned n_listed = *ovf_count;
    const unsigned n_cleanup = MERGED ? gridDim.x - kNSums : gridDim.x;  // workgroups in the cleanup role
    if constexpr (MERGED) {
        if (blockIdx.x >= n_cleanup) {
            // fold role: the partials of the handed-over workgroups must be in place first
            if (n_listed > 0) {
                if (threadIdx.x == 0)
                    while (__hip_atomic_load(fs.ticket + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < n_listed)
                        __builtin_amdgcn_s_sleep(8);
                __syncthreads();
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
            }
            (void)fold_and_solve_block(fs, (int)(blockIdx.x - n_cleanup));
            return;
        }
    }
    for (unsigned listed = blockIdx.x; listed < n_listed; listed += n_cleanup) {
    const int entry = ovf_list[listed];
    const int fast_slot = entry >> 2, half = entry & 3;
    // (n_extra < 0: the fast kernel ran with SplitTable::all_halves)
    const int bid = n_extra < 0 ? xcd_block(halves_slot(fast_slot), (ns + BLOCK - 1) / BLOCK)
                                : (fast_slot < n_extra ? split_list[fast_slot] : xcd_block(fast_slot - n_extra, (ns + BLOCK - 1) / BLOCK));
    static_assert(C > M, "a compaction must leave room in the list");
    static_assert(CAP % 4 == 0 && CAP <= 65536 && C * 64 <= 3 * CAP && kTileRows <= 256, "the global fallback aliases the candidate buffer");
    static_assert((C & (C - 1)) == 0, "overflow_slot() masks");
    static_assert(kTileRows == 128, "row table: two rows per lane of one wave");
    constexpr int kWaves = BLOCK / 64;
    constexpr int kStageUnroll = 8;  // halo rows in flight per wave
    // staged halo, structure-of-arrays: two candidates per ds_read_b64 and per packed-f32 instruction
    __shared__ __attribute__((aligned(16))) float s_halo[3 * CAP + CAP / 4];
    float *const s_x = s_halo, *const s_y = s_halo + CAP, *const s_z = s_halo + 2 * CAP;
    unsigned char *const s_rowid = reinterpret_cast<unsigned char *>(s_halo + 3 * CAP);
    int *const s_glist = reinterpret_cast<int *>(s_halo);  // global-fallback list aliases the halo buffer
    __shared__ unsigned short s_list[C * BLOCK];
    __shared__ int s_row_gb[kTileRows];
    __shared__ int s_row_off[kTileRows + 1];
    __shared__ int s_wlo[kWaves][3], s_whi[kWaves][3];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = bid * BLOCK + tid;
    const bool valid = i < ns && (half == 0 || (wave >> 1) == half - 1);  // lanes whose query the fast workgroup owned
    const float4 q = valid ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned thr0 = 0xFFFFFFFFu;  // no temporal cut-off in this flavour
    const QueryCells qc = query_cells(q, g);

    // this lane's 9 stencil runs [rb, re) in sorted-target positions: issued now, consumed after the
    // halo has been staged, so their latency hides behind the staging phase.
    // Each run is clipped in x: a target of row (dy, dz) is at least (gy, gz) away in y and z (the gap between the
    // query and that row's slab, under-estimated by g.eps), so it can only be within the cut-off radius R if
    // |dx| <= sqrt(R^2 - gy^2 - gz^2); R^2 is the radius or the temporal cut-off, inflated by 4e-6 for the float
    // rounding of d2.  The x slices that window touches are the run; a row with no window is skipped.
    const int x0 = max(qc.cx - g.xr, 0), x1 = min(qc.cx + g.xr, g.n[0] - 1);
    int rb[9], re[9];
    {
        const float R2 = __uint_as_float(min(thr0, __float_as_uint(r2))) * 1.000004f;
        const float fy = q.y - g.org[1], fz = q.z - g.org[2];
        const float gy[3] = {fmaxf(fy - (float)qc.cy * g.h - g.eps, 0.f), 0.f,
                             fmaxf((float)(qc.cy + 1) * g.h - fy - g.eps, 0.f)};
        const float gz[3] = {fmaxf(fz - (float)qc.cz * g.h - g.eps, 0.f), 0.f,
                             fmaxf((float)(qc.cz + 1) * g.h - fz - g.eps, 0.f)};
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int cz = qc.cz + (k / 3 - 1), cy = qc.cy + (k % 3 - 1);
            const float w2 = R2 - (gy[k % 3] * gy[k % 3] + gz[k / 3] * gz[k / 3]);
            const float w = sqrtf(fmaxf(w2, 0.f)) * 1.000001f + g.eps;
            const int fa = max(cell_coord(q.x - w, g.org[0], g.inv_hx, g.n[0]), x0);
            const int fb = min(cell_coord(q.x + w, g.org[0], g.inv_hx, g.n[0]), x1);
            const bool in = valid && w2 >= 0.f && fa <= fb && (unsigned)cz < (unsigned)g.n[2] &&
                            (unsigned)cy < (unsigned)g.n[1];
            const int base = in ? (cz * g.n[1] + cy) * g.n[0] : 0;
            rb[k] = in ? cell_start[base + fa] : 0;
            re[k] = in ? cell_start[base + fb + 1] : 0;
        }
    }

    // per-wave bounding box of the query cells
    {
        int lo[3] = {valid ? qc.cx : INT_MAX, valid ? qc.cy : INT_MAX, valid ? qc.cz : INT_MAX};
        int hi[3] = {valid ? qc.cx : INT_MIN, valid ? qc.cy : INT_MIN, valid ? qc.cz : INT_MIN};
#pragma unroll
        for (int a = 0; a < 3; a++)
            for (int off = 32; off > 0; off >>= 1) {
                lo[a] = min(lo[a], __shfl_xor(lo[a], off));
                hi[a] = max(hi[a], __shfl_xor(hi[a], off));
            }
        if (lane == 0)
            for (int a = 0; a < 3; a++) {
                s_wlo[wave][a] = lo[a];
                s_whi[wave][a] = hi[a];
            }
    }
    lds_barrier();

    int n = 0;
    bool done = !valid;
    // Halo passes, coarse to fine: all waves together; if that halo does not fit, halves, then single
    // waves (binary subdivision of the wave range).  done_mask (uniform over the block) has a bit per
    // finished wave; a pass whose waves are all finished is skipped.
    unsigned done_mask = 0;
    for (int span = kWaves; span >= 1; span >>= 1)
      for (int w0 = 0; w0 < kWaves; w0 += span) {
        const int w1 = w0 + span;
        const unsigned pass_mask = ((1u << span) - 1u) << w0;
        if ((done_mask & pass_mask) == pass_mask) continue;
        const bool last_level = span == 1;
        int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
        for (int w = w0; w < w1; w++)
            for (int a = 0; a < 3; a++) {
                lo[a] = min(lo[a], s_wlo[w][a]);
                hi[a] = max(hi[a], s_whi[w][a]);
            }
        const bool any = lo[0] <= hi[0];  // at least one valid query among these waves
        const int hx0 = max(lo[0] - g.xr, 0), hx1 = min(hi[0] + g.xr, g.n[0] - 1);
        const int hy0 = max(lo[1] - 1, 0), hy1 = min(hi[1] + 1, g.n[1] - 1);
        const int hz0 = max(lo[2] - 1, 0), hz1 = min(hi[2] + 1, g.n[2] - 1);
        const int ny_h = hy1 - hy0 + 1, nz_h = hz1 - hz0 + 1;
        const bool empty = !any || hx0 > hx1 || ny_h <= 0 || nz_h <= 0;
        const long long nrows_ll = empty ? 0 : (long long)ny_h * nz_h;
        const bool rows_ok = nrows_ll <= kTileRows;
        const int nrows = rows_ok ? (int)nrows_ll : 0;

        // Row table, built redundantly by every wave in registers (no barrier before the staging):
        // lane l owns halo rows 2l and 2l+1: global begin, length, exclusive prefix of the lengths.
        int gbA = 0, gbB = 0, lenA = 0, lenB = 0;
        {
            const int rA = 2 * lane, rB = 2 * lane + 1;
            if (rA < nrows) {
                const int base = ((hz0 + rA / ny_h) * g.n[1] + hy0 + rA % ny_h) * g.n[0];
                gbA = cell_start[base + hx0];
                lenA = cell_start[base + hx1 + 1] - gbA;
            }
            if (rB < nrows) {
                const int base = ((hz0 + rB / ny_h) * g.n[1] + hy0 + rB % ny_h) * g.n[0];
                gbB = cell_start[base + hx0];
                lenB = cell_start[base + hx1 + 1] - gbB;
            }
        }
        int incl = lenA + lenB;
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        const int offA = incl - lenA - lenB, offB = offA + lenA;
        const int total = __builtin_amdgcn_readlane(incl, 63);
        const bool ok = rows_ok && total <= CAP;
        if (ok) {
            if (wave == w0) {  // one wave publishes the table for the scan phase
                s_row_gb[2 * lane] = gbA;
                s_row_gb[2 * lane + 1] = gbB;
                s_row_off[2 * lane] = offA;
                s_row_off[2 * lane + 1] = offB;
                if (lane == 63) s_row_off[kTileRows] = incl;
            }
            // stage the halo: one wave per row, lane-contiguous 16-byte loads, kStageUnroll rows in flight
            for (int k0 = 0; wave + kWaves * k0 < nrows; k0 += kStageUnroll) {
                float4 c[kStageUnroll];
                int so[kStageUnroll], sl[kStageUnroll], sg[kStageUnroll];
#pragma unroll
                for (int u = 0; u < kStageUnroll; u++) {
                    const int r = wave + kWaves * (k0 + u);
                    const int rr = min(r, kTileRows - 1);
                    const int gA = __builtin_amdgcn_readlane(gbA, rr >> 1), gB = __builtin_amdgcn_readlane(gbB, rr >> 1);
                    const int oA = __builtin_amdgcn_readlane(offA, rr >> 1), oB = __builtin_amdgcn_readlane(offB, rr >> 1);
                    const int lA = __builtin_amdgcn_readlane(lenA, rr >> 1), lB = __builtin_amdgcn_readlane(lenB, rr >> 1);
                    sg[u] = (rr & 1) ? gB : gA;
                    so[u] = (rr & 1) ? oB : oA;
                    sl[u] = (r < nrows) ? ((rr & 1) ? lB : lA) : 0;
                    c[u] = tgt[(lane < sl[u]) ? sg[u] + lane : 0];  // unconditional load (slot 0 always exists)
                }
#pragma unroll
                for (int u = 0; u < kStageUnroll; u++) {
                    if (lane < sl[u]) {
                        const int d = so[u] + lane;
                        s_x[d] = c[u].x;
                        s_y[d] = c[u].y;
                        s_z[d] = c[u].z;
                        s_rowid[d] = (unsigned char)(wave + kWaves * (k0 + u));
                    }
                    for (int k = lane + 64; k < sl[u]; k += 64) {  // rows longer than a wave (dense data)
                        const float4 t = tgt[sg[u] + k];
                        const int d = so[u] + k;
                        s_x[d] = t.x;
                        s_y[d] = t.y;
                        s_z[d] = t.z;
                        s_rowid[d] = (unsigned char)(wave + kWaves * (k0 + u));
                    }
                }
            }
            lds_barrier();
            if (!done && wave >= w0 && wave < w1) {
                const LdsCands<BLOCK> L{s_x, s_y, s_z, s_rowid, s_row_gb, s_row_off, s_list, tid};
                // d2 >= +0 and r2 > 0, so "d2 < r2" is "bits(d2) <= bits(r2) - 1" (a NaN d2 has larger bits and
                // fails): the radius test and the running cut-off become ONE unsigned compare per candidate
                const unsigned lim0 = min(thr0, __float_as_uint(r2) - 1u);
                unsigned thr = lim0;
                typedef float v2f __attribute__((ext_vector_type(2)));
                const v2f qx2 = {q.x, q.x}, qy2 = {q.y, q.y}, qz2 = {q.z, q.z};
                // The 9-run scan.  Fast flavour (COMPACT = false): an accepted candidate is stored at slot
                // min(n, C-1) and counted, nothing else — the list-full test stays out of the per-candidate path.
                // A lane that ends with n > C overflowed its list (dense neighbourhood and no usable cut-off);
                // only those lanes re-run the scan in the compacting flavour, which reduces a full list to its
                // top-m on the spot and tightens the lane's threshold.
                // The nine runs as (LDS start, length), ordered by DESCENDING length: every lane of the wave then
                // walks its longest run first, its second longest next, ... — a run's trip count is the maximum
                // over the 64 lanes, and the maxima of order statistics add up to far fewer steps than the maxima
                // of arbitrary runs (simulated for this density: 130 steps instead of 161; 121 would be perfect).
                // 25-comparator sorting network (verified with the 0/1 principle).
                // Sorted as ONE 32-bit key per run, (length << 16) | LDS start (both < 65536: list entries are
                // 16-bit LDS indices): a comparator is a v_max_u32 / v_min_u32 pair instead of a compare and four
                // selects on a (length, start) pair (~50 instead of ~200 instructions for the 25 comparators).
                int rf[9], rl[9];
                {
                    unsigned key[9];
#pragma unroll
                    for (int k = 0; k < 9; k++) {
                        const int len = re[k] - rb[k];
                        const int r = (qc.cz + (k / 3 - 1) - hz0) * ny_h + (qc.cy + (k % 3 - 1) - hy0);
                        const int rr = (len > 0) ? r : 0;
                        const int start = s_row_off[rr] + (rb[k] - s_row_gb[rr]);
                        key[k] = len > 0 ? ((unsigned)len << 16) | (unsigned)start : 0u;
                    }
                    constexpr int net[25][2] = {{0, 3}, {1, 7}, {2, 5}, {4, 8}, {0, 7}, {2, 4}, {3, 8}, {5, 6}, {0, 2},
                                                {1, 3}, {4, 5}, {7, 8}, {1, 4}, {3, 6}, {5, 7}, {0, 1}, {2, 4}, {3, 5},
                                                {6, 8}, {2, 3}, {4, 5}, {6, 7}, {1, 2}, {3, 4}, {5, 6}};
#pragma unroll
                    for (int c = 0; c < 25; c++) {
                        const int a = net[c][0], b = net[c][1];
                        const unsigned hi = max(key[a], key[b]), lo = min(key[a], key[b]);  // descending
                        key[a] = hi;
                        key[b] = lo;
                    }
#pragma unroll
                    for (int k = 0; k < 9; k++) {
                        rf[k] = (int)(key[k] & 0xFFFFu);
                        rl[k] = (int)(key[k] >> 16);
                    }
                }
                auto scan_runs = [&](auto compact_tag) {
                    constexpr bool COMPACT = decltype(compact_tag)::value;
                    auto accept = [&](int f, float d2) {
                        if (__float_as_uint(d2) <= thr) {
                            if constexpr (COMPACT) {
                                L.store(n, f);
                                n++;
                                if (n == C) n = select_top_m<M>(L, tgt, q, n, m, thr);
                            } else {
                                L.store(overflow_slot<C>(n, f), f);
                                n++;
                            }
                        }
                    };
                    // One run: ALIGNED pairs from (fb & ~1) while p < fe.  The element below fb (first trip of an
                    // odd start) and the element at fe (last trip of an odd end) belong to other runs: they are
                    // kept out by the two index tests, which replace the odd head / tail singles of the previous
                    // version (two compares per trip instead of ~34 instructions per run, and one code path).
                    // Two candidates per trip: ds_read_b64 x3, packed f32 sub/mul/add (no FMA: the same IEEE
                    // operations per element as dist2_flann, so d2 is bit-identical).
                    auto scan_run = [&](int fb, int len) {
                        if (len <= 0) return;
                        const int fe = fb + len;
                        for (int p = fb & ~1; p < fe; p += 2) {
                            const v2f cx = *reinterpret_cast<const v2f *>(&s_x[p]);
                            const v2f cy = *reinterpret_cast<const v2f *>(&s_y[p]);
                            const v2f cz = *reinterpret_cast<const v2f *>(&s_z[p]);
                            const v2f dx = qx2 - cx, dy = qy2 - cy, dz = qz2 - cz;
                            v2f d = dx * dx;
                            d = d + dy * dy;
                            d = d + dz * dz;
                            if (p >= fb) accept(p, d.x);
                            if (p + 1 < fe) accept(p + 1, d.y);
                        }
                    };
                    if constexpr (COMPACT) {
                        // rare flavour: keep the code small — one loop body, the runs rotated through rf[0] / rl[0]
                        // (after nine rotations they are back in place)
#pragma unroll 1
                        for (int k = 0; k < 9; k++) {
                            const int fb = rf[0], len = rl[0];
#pragma unroll
                            for (int u = 0; u < 8; u++) {
                                rf[u] = rf[u + 1];
                                rl[u] = rl[u + 1];
                            }
                            rf[8] = fb;
                            rl[8] = len;
                            scan_run(fb, len);
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 9; k++) scan_run(rf[k], rl[k]);
                    }
                };
                scan_runs(std::false_type{});
                // List overflow (dense neighbourhood and no usable cut-off): the C entries that were kept — a spread-out
                // sample of the accepted candidates, see overflow_slot() — are genuine in-radius candidates, so the m-th
                // smallest of them bounds the final m-th distance: tighten the threshold to it and scan again.  Each
                // round keeps ~m / C of the candidates.  The compacting scan —
                // whose in-loop selections are paid by the whole wave whenever ANY lane's list fills: ~1 ms per
                // workgroup once every lane overflows — is only the last resort (floods of exact ties).
                for (int attempt = 0; n > C && attempt < 8; attempt++) {
                    (void)select_top_m<M>(L, tgt, q, C, m, thr);
                    n = 0;
                    scan_runs(std::false_type{});
                }
                if (n > C) {
                    n = 0;
                    thr = lim0;
                    scan_runs(std::true_type{});
                }
                unsigned tm = 0xFFFFFFFFu;  // d2 bits of the m-th neighbour (all-ones: fewer than m found)
                if (n > m) {
                    n = select_top_m<M>(L, tgt, q, n, m, thr);
                    tm = thr;
                } else if (n == m) {
                    tm = 0;
                    for_each_entry(L, q, n, [&](int, int, unsigned b) { tm = max(tm, b); });
                }
                for (int j = 0; j < n; j++) nbr[(size_t)j * ns + i] = L.pos_of(L.load(j));
                cnt[i] = n;
                if (dm2) dm2[i] = tm;
                done = true;
            }
            done_mask |= pass_mask;
            if (done_mask == (1u << kWaves) - 1u) break;  // common case: nothing left, no trailing barrier
            lds_barrier();                              // the halo buffer is reused by the next pass
        } else if (last_level) {
            // last resort for this wave: scan global memory (list of positions aliases the halo buffer)
            if (!done && wave == w0) {
                const GlobalCands<64> G{tgt, s_glist, lane};
                unsigned thr = min(thr0, __float_as_uint(r2) - 1u);
                n = scan_global_with_threshold<M, C>(G, tgt, q, m, thr, [&](auto &&accept) {
                    for_each_candidate(q, g, cell_start, tgt, [&](int p, float4 t) { accept(p, dist2_flann(q, t)); });
                });
                unsigned tm = 0xFFFFFFFFu;
                if (n > m) {
                    n = select_top_m<M>(G, tgt, q, n, m, thr);
                    tm = thr;
                } else if (n == m) {
                    tm = 0;
                    for_each_entry(G, q, n, [&](int, int, unsigned b) { tm = max(tm, b); });
                }
                for (int j = 0; j < n; j++) nbr[(size_t)j * ns + i] = G.load(j);
                cnt[i] = n;
                if (dm2) dm2[i] = tm;
                done = true;
            }
            done_mask |= pass_mask;
            lds_barrier();
        }
      }
    __syncthreads();  // the LDS buffers are reused by this workgroup's next listed block (and by the fold below)
    if constexpr (FTM != -2) {
        // K23 for the rows just redone: each lane re-reads its own row (it wrote it itself) and gathers the neighbours
        RowAcc acc;
#pragma unroll
        for (int j = 0; j < kNSums; j++) acc.a[j] = 0.0;
        const int nrow = valid ? cnt[i] : 0;
        if (nrow > 0) {
            double xr[3];
            rotated_point(fm.P, q, xr);
            RowMoments<FTM> row;
            row.begin(fm.md);
            for (int j = 0; j < nrow; j++) {
                const float4 y = tgt[nbr[(size_t)j * ns + i]];
                row.add_pair(fm.md, xr, y.x, y.y, y.z);
            }
            row.finish(acc, fm.P, q, xr);
        }
        double *const scratch = reinterpret_cast<double *>(s_halo);
        block_reduce_scratch(acc, scratch, scratch + 10 * 257, fm.partials + fast_slot, (size_t)fm.nslots, true);
        __syncthreads();
        if constexpr (MERGED) {
            if (threadIdx.x == 0) {  // this entry's partials are written: let the fold role count it
                __atomic_thread_fence(__ATOMIC_RELEASE);
                __hip_atomic_fetch_add(fs.ticket + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    }
}

// ---------------------------------------------------------------------------------------------
// Second pass of a TWO-PASS radius search.  When the radius holds far more than max_neighbours target points (the
// command line's defaults: radius 3, 20 neighbours, ~430 points in radius at the benchmark density), a grid of
// radius-sized cells gives every 256-query block a halo of thousands of candidates — no LDS tile holds it and every block
// used to fall through to the cleanup kernel's global-memory scan (15-50 ms per iteration at 200k points).  The grid is
// then built for a SMALLER search radius r' <= radius (chosen from the target's density so that r' still holds
// ~2.2 max_neighbours points) and the ordinary K1 runs with r': a row that finds max_neighbours points within r' has its
// exact answer (its m nearest overall are among them: d2 < r'^2 <= radius^2, same (d2, index) order).  Only the rows
// that came back SHORT are searched again with the full radius, over a stencil reach cells wide on the same grid (so both
// passes index the same sorted target).  Short rows are the cloud's fringe and its sparse parts: a few per cent of the
// rows, scattered over every wave of the first pass — one lane per short row (the first form of this kernel) left 60 lanes
// of every wave idle behind the slowest one (229 us at 200k points).  So K1 appends the rows it leaves unanswered to a LIST
// (UnansweredRows: short rows, rows of handed-over workgroups) and they are searched ONE ROW PER WAVE: the 64 lanes walk
// the stencil's runs together as one flat candidate sequence, accepted candidates are appended to a wave-shared list in
// LDS (ballot + prefix), and the selection of the m smallest by (d2, original index) is a wave-wide bit-by-bit descent on
// ballot counts.
// ---------------------------------------------------------------------------------------------
// The wave's list [0, n) in LDS (sorted-target position, d2 bits) -> its m smallest by (d2, original index), compacted in
// place; returns the new length (min(n, m)) and the d2 bits of the m-th (thr is left alone when n < m).  PER * 64 >= n.
// The m-th smallest d2 is found bit by bit from the top: "how many entries agree with the prefix so far and have a 0
// here" is a ballot count per 64 entries, nothing else; ties at the m-th distance that do not all fit are settled by the
// same descent on their original indices (the oracle's order).
// (experiment knobs, tools/build_variant.py: loads in flight per lane and list entries per wave of nn_wide_kernel)
#ifndef PPCR_WIDE_U
#define PPCR_WIDE_U 4
#endif
#ifndef PPCR_WIDE_CAPW
#define PPCR_WIDE_CAPW 512
#endif
template <int PER>
__device__ __forceinline__ int wave_select_top_m(int *s_pos, unsigned *s_d2, int n, int m, const float4 *__restrict__ tgt, int lane,
                                                 unsigned &thr)
{
    if (n < m) return n;
    unsigned d[PER];
    int p[PER];
    const int chunks = (n + 63) >> 6;  // uniform
#pragma unroll
    for (int j = 0; j < PER; j++) {
        d[j] = 0xFFFFFFFFu, p[j] = 0;
        if (j < chunks) {
            const int idx = j * 64 + lane;
            d[j] = idx < n ? s_d2[idx] : 0xFFFFFFFFu;  // (a genuine d2 is a finite float: never all-ones)
            p[j] = s_pos[min(idx, n - 1)];
        }
    }
    auto descend = [&](const unsigned(&key)[PER], unsigned long long const(&in)[PER], int need) -> unsigned {
        // the need-th smallest (1-based) key among the entries flagged in `in`
        unsigned prefix = 0;
        for (int bit = 31; bit >= 0; bit--) {
            const unsigned above = bit == 31 ? 0u : ~((2u << bit) - 1u);
            int c = 0;
#pragma unroll
            for (int j = 0; j < PER; j++)
                if (j < chunks) {
                    const bool on = (in[j] >> lane) & 1ull;
                    c += __popcll(__ballot(on && (key[j] & above) == prefix && !((key[j] >> bit) & 1u)));
                }
            if (c < need) {
                need -= c;
                prefix |= 1u << bit;
            }
        }
        return prefix;
    };
    unsigned long long all[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) all[j] = __ballot(j < chunks && j * 64 + lane < n);
    const unsigned T = descend(d, all, m);
    int below = 0, ties = 0;
    unsigned long long tied[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) {
        tied[j] = 0;
        if (j < chunks) {
            below += __popcll(__ballot(d[j] < T));
            tied[j] = __ballot(d[j] == T);
            ties += __popcll(tied[j]);
        }
    }
    const int room = m - below;  // >= 1: T is the m-th smallest
    unsigned U = 0xFFFFFFFFu;    // tied entries with original index <= U stay
    if (ties > room) {
        unsigned oi[PER];
#pragma unroll
        for (int j = 0; j < PER; j++) {
            oi[j] = 0xFFFFFFFFu;
            if (j < chunks && d[j] == T) oi[j] = (unsigned)__float_as_int(tgt[p[j]].w);
        }
        U = descend(oi, tied, room);
#pragma unroll
        for (int j = 0; j < PER; j++) d[j] = (d[j] == T && oi[j] > U) ? 0xFFFFFFFFu : d[j];  // drop the losers of the tie
    }
    int base = 0;
#pragma unroll
    for (int j = 0; j < PER; j++)
        if (j < chunks) {
            const bool keep = d[j] <= T;
            const unsigned long long k = __ballot(keep);
            const int at = base + __builtin_amdgcn_mbcnt_hi((unsigned)(k >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)k, 0u));
            if (keep) s_pos[at] = p[j], s_d2[at] = d[j];
            base += __popcll(k);
        }
    thr = T;
    return base;
}

// vv.vl != nullptr: the rows searched here also get their Verlet lists (VerletLists) — one more sweep over the sphere of
// radius G = the row's m-th distance (or the full radius where it has fewer than m neighbours) + 2 x skin, the nearest
// verlet_slots(M) of what lies inside kept by the wave's selection — so that from the next association on nn_fast_kernel
// answers them from their lists like any other row, for as long as the lists hold: the rows a two-pass search finds short
// every time (a cloud's fringe, its sparse parts: thousands of rows, 13-55 us of this kernel per iteration at the command
// line's defaults) come here once per list lifetime instead of once per iteration.
template <int M>
__global__ __launch_bounds__(256, 6) void nn_wide_kernel(const float4 *__restrict__ src, int ns, const float4 *__restrict__ tgt,
                                                      const int *__restrict__ cell_start, GridDesc g, int reach, float r1_sq, float r2, int m,
                                                      int *__restrict__ nbr, int *__restrict__ cnt, unsigned *__restrict__ dm2,
                                                      const int *__restrict__ short_list, const unsigned *__restrict__ short_count,
                                                      unsigned *__restrict__ short_seen, const LoopState *loop_st, VerletLists vv)
{
    constexpr int U = PPCR_WIDE_U;        // chunks of 64 candidates (loads per lane) in flight
    constexpr int CAPW = PPCR_WIDE_CAPW;  // list entries per wave; compacted whenever a round of U * 64 might not fit
    constexpr int PER = CAPW / 64;
    constexpr int CAPT = 1024;       // candidates of one batch of 64 runs that are walked as ONE flat sequence
    constexpr int CVs = verlet_slots(M);
    static_assert(CAPW >= 2 * U * 64 && CAPW - U * 64 >= M && CAPW - U * 64 >= CVs, "a compaction leaves room for a round");
    if (loop_aborted(loop_st)) return;
    __shared__ int s_pos_all[4][CAPW];
    __shared__ unsigned s_d2_all[4][CAPW];
    __shared__ int s_pre_all[4][64], s_b_all[4][64];
    __shared__ unsigned char s_mark_all[4][CAPT];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int *const s_pos = s_pos_all[wave], *const s_pre = s_pre_all[wave], *const s_b = s_b_all[wave];
    unsigned *const s_d2 = s_d2_all[wave];
    unsigned char *const s_mark = s_mark_all[wave];
    unsigned *const s_mark4 = reinterpret_cast<unsigned *>(s_mark);
    const unsigned n_short = *short_count;
    // A wave's rows one after the other, two deep: the row after next is looked up in the list and the next row's header —
    // count, bound, coordinates: a round trip that depends on the list's — is asked for while this row is searched
    // (a row is ~20 us of dependent round trips; the header's was the first of them).  Nobody but this wave writes its rows.
    const unsigned entry0 = blockIdx.x * 4 + wave, stride = gridDim.x * 4;
    int i_next = entry0 < n_short ? short_list[entry0] : 0;
    int i_after = entry0 + stride < n_short ? short_list[entry0 + stride] : 0;
    int found_next = cnt[i_next];
    unsigned prev_next = dm2[i_next];
    float4 q_next = src[i_next];
    for (unsigned entry = entry0; entry < n_short; entry += stride) {
        const int i = __builtin_amdgcn_readfirstlane(i_next);
        const int found = __builtin_amdgcn_readfirstlane(found_next);  // < 0: its workgroup left the first pass to this one
        const unsigned prev_bits = (unsigned)__builtin_amdgcn_readfirstlane((int)prev_next);
        const float4 q = q_next;
        i_next = i_after;
        i_after = entry + 2 * stride < n_short ? short_list[entry + 2 * stride] : 0;
        if (entry + stride < n_short) found_next = cnt[i_next], prev_next = dm2[i_next], q_next = src[i_next];  // (uniform)
        const QueryCells c = query_cells(q, g);
        const float fy = q.y - g.org[1], fz = q.z - g.org[2];
        // How far this row's m-th neighbour is, nobody knows; the search starts from an ESTIMATE and grows outwards (radius
        // x 1.5 per attempt, up to the full radius), each attempt a scan of just the rows and x windows its sphere touches;
        // the first attempt that holds m candidates ends the search (the m nearest overall are among the candidates within
        // ITS radius: everything beyond is farther than all of them).  The estimate: a row that found `found` < m points
        // within the first pass's radius r1 has m of them within ~r1 cbrt(m / found) if the density holds; an unsearched
        // row (a workgroup whose halo outgrew the LDS tile, a row whose Verlet list ran out) takes its m-th distance of the
        // previous association where it has one — the source has hardly moved since — and the density of its own cell
        // where it has not.
        // (the tiled kernel leaves a BOUND in dm2 for the rows it lists — the previous m-th distance + the row's move: the m
        //  neighbours of the previous association lie inside, one sweep of that sphere is the whole search; the rows of a
        //  workgroup that answered from lists and listed a few of its rows come with the previous m-th distance as it was)
        float R2;
        if (found < 0 && prev_bits != 0xFFFFFFFFu && vv.vl != nullptr) {
            R2 = fminf(__uint_as_float(prev_bits) * 1.05f + 1e-30f, r2);
        } else if (prev_bits != 0xFFFFFFFFu) {
            R2 = fminf(__uint_as_float(prev_bits), r2);
        } else if (found >= 0) {
            const float k = cbrtf((float)m / fmaxf((float)found, 0.5f)) * 1.15f;
            R2 = r1_sq * k * k;
        } else {
            const bool inside = (unsigned)c.cx < (unsigned)g.n[0] && (unsigned)c.cy < (unsigned)g.n[1] && (unsigned)c.cz < (unsigned)g.n[2];
            const int cell = inside ? (c.cz * g.n[1] + c.cy) * g.n[0] + c.cx : 0;
            const float in_cell = inside ? (float)max(cell_start[cell + 1] - cell_start[cell], 1) : 1.f;
            const float cell_vol = g.h * g.h / g.inv_hx;
            const float re = cbrtf(1.5f * (float)m * cell_vol / (4.19f * in_cell));
            R2 = fminf(re * re, r1_sq);
        }
        R2 *= (1.0f / 2.25f);  // (the loop below grows before it scans)
        unsigned thr = 0;
        int n = 0;
        bool cut = false;  // a compaction dropped entries: what is left is complete only below thr
        // one sweep over the sphere of radius^2 R2s around the query: every target whose d2 bits are <= thr is appended to the
        // wave's list, which is compacted to its `keep` smallest (tightening thr) whenever a round might not fit
        auto sweep = [&](const float R2s, const int keep) {
            n = 0;
            const int rows = min(reach, (int)(__builtin_amdgcn_sqrtf(R2s) * g.inv_h) + 1);  // cells the sphere can reach in y / z
            const int side = 2 * rows + 1, nrun = side * side;
            // lane l of batch k0: the (dy, dz) row k0 + l of the stencil, clipped to the x slices the sphere can touch there
            auto run_bounds = [&](int k0, int &rb, int &re) {
                rb = 0, re = 0;
                const int k = k0 + lane;
                const int dz = k / side - rows, dy = k % side - rows;
                const int cz = c.cz + dz, cy = c.cy + dy;
                if (k < nrun && (unsigned)cz < (unsigned)g.n[2] && (unsigned)cy < (unsigned)g.n[1]) {
                    const float gz = dz < 0 ? fmaxf(fz - (float)(cz + 1) * g.h - g.eps, 0.f) : (dz > 0 ? fmaxf((float)cz * g.h - fz - g.eps, 0.f) : 0.f);
                    const float gy = dy < 0 ? fmaxf(fy - (float)(cy + 1) * g.h - g.eps, 0.f) : (dy > 0 ? fmaxf((float)cy * g.h - fy - g.eps, 0.f) : 0.f);
                    const float w2 = R2s - (gy * gy + gz * gz);
                    if (w2 > 0.f) {
                        const float w = __builtin_amdgcn_sqrtf(w2) * 1.000001f + g.eps;
                        const int fa = max(cell_coord(q.x - w, g.org[0], g.inv_hx, g.n[0]), 0);
                        const int fb = min(cell_coord(q.x + w, g.org[0], g.inv_hx, g.n[0]), g.n[0] - 1);
                        if (fa <= fb) {
                            const int base = (cz * g.n[1] + cy) * g.n[0];
                            rb = cell_start[base + fa], re = cell_start[base + fb + 1];
                        }
                    }
                }
            };
            // one candidate per lane: distance, acceptance, append to the wave's list
            auto offer = [&](bool live, int pos, float4 t) {
                const unsigned bits = __float_as_uint(dist2_flann(q, t));
                const bool acc = live && bits <= thr;
                const unsigned long long k = __ballot(acc);
                const int at = n + __builtin_amdgcn_mbcnt_hi((unsigned)(k >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)k, 0u));
                if (acc) s_pos[at] = pos, s_d2[at] = bits;
                n += __popcll(k);
            };
            auto make_room = [&]() {
                if (n > CAPW - U * 64) {
                    cut = cut || n > keep;
                    n = wave_select_top_m<PER>(s_pos, s_d2, n, keep, tgt, lane, thr);
                }
            };
            int nb_, ne_;
            run_bounds(0, nb_, ne_);
            for (int k0 = 0; k0 < nrun; k0 += 64) {
                const int rb = nb_, len = ne_ - nb_;
                if (k0 + 64 < nrun) run_bounds(k0 + 64, nb_, ne_);  // the next batch's bounds travel while this one is walked
                const int incl = wave_scan(len, 0, OpAdd());
                const int total = __builtin_amdgcn_readlane(incl, 63);
                if (total == 0) continue;
                if (total <= CAPT) {
                    // the batch's candidates as ONE flat sequence [0, total): a byte per candidate marks where a run
                    // starts (run + 1), a running maximum spreads the run over its candidates — sparse rows' runs hold
                    // two or three points each, and a round per run is a memory round trip per run
                    for (int w = lane; w * 4 < total; w += 64) s_mark4[w] = 0u;
                    s_pre[lane] = incl - len, s_b[lane] = rb;
                    if (len > 0) s_mark[incl - len] = (unsigned char)(lane + 1);
                    int carry = 0;
                    for (int j0 = 0; j0 < total; j0 += U * 64) {
                        make_room();
                        int pos[U];
                        float4 t[U];
#pragma unroll
                        for (int u = 0; u < U; u++) {
                            const int j = j0 + u * 64 + lane;
                            int run = j < total ? (int)s_mark[min(j, CAPT - 1)] : 0;
                            run = max(wave_scan(run, 0, OpMax()), carry);
                            carry = __builtin_amdgcn_readlane(run, 63);
                            const int r = max(run - 1, 0);
                            pos[u] = j < total ? s_b[r] + (j - s_pre[r]) : 0;
                            t[u] = tgt[pos[u]];
                        }
#pragma unroll
                        for (int u = 0; u < U; u++)
                            if (j0 + u * 64 < total) offer(j0 + u * 64 + lane < total, pos[u], t[u]);
                    }
                } else {
                    // a dense batch: long runs, walked run by run, U at a time
                    unsigned long long live = __ballot(len > 0);
                    while (live) {
                        int b[U], ln[U], longest = 0;
#pragma unroll
                        for (int u = 0; u < U; u++) {
                            b[u] = 0, ln[u] = 0;
                            if (live) {
                                const int r = __builtin_ctzll(live);
                                live &= live - 1;
                                b[u] = __builtin_amdgcn_readlane(rb, r);
                                ln[u] = __builtin_amdgcn_readlane(len, r);
                                longest = max(longest, ln[u]);
                            }
                        }
                        for (int off = 0; off < longest; off += 64) {
                            make_room();
                            float4 t[U];
#pragma unroll
                            for (int u = 0; u < U; u++) t[u] = tgt[b[u] + min(off + lane, max(ln[u] - 1, 0))];
#pragma unroll
                            for (int u = 0; u < U; u++)
                                if (off < ln[u]) offer(off + lane < ln[u], b[u] + off + lane, t[u]);
                        }
                    }
                }
            }
        };
        // A row that comes here because its Verlet list ran out knows where its m-th neighbour was an association ago, and the
        // source has hardly moved since: ONE sweep of the sphere that m-th distance (+ 2 %) + the list's skin reaches gives
        // the new list, and the m nearest of the list inside the radius are the answer — where the sweep holds m of them (it
        // then holds the m nearest of the cloud: everything outside it is farther) or reaches the whole radius.  Half the
        // latency of search-then-list for the handful of rows a launch is left with once the lists answer.
        bool done = false;
        if (vv.vl != nullptr && prev_bits != 0xFFFFFFFFu) {
            const float need0 = __builtin_amdgcn_sqrtf(__uint_as_float(prev_bits)) * 1.02f + 1e-30f;
            const float G1 = fminf(need0 + fmaxf(vv.skin2, vv.skin_rel * need0), (float)reach * g.h * 0.999f);
            const unsigned thr_g = __float_as_uint(G1 * G1);
            thr = thr_g;
            cut = false;
            sweep(G1 * G1 * 1.000004f, CVs);
            if (n > CVs) {
                n = wave_select_top_m<PER>(s_pos, s_d2, n, CVs, tgt, lane, thr);
                cut = true;
            }
            const unsigned in_radius = __float_as_uint(r2) - 1u;
            const unsigned mine = lane < n ? s_d2[lane] : 0xFFFFFFFFu;  // (n <= CVs <= 64: an entry per lane)
            const int n_in = __popcll(__ballot(mine <= in_radius));
            if (n_in >= m || G1 * G1 >= r2) {
                // the list first (every entry the sweep kept), then its m nearest inside the radius
                const unsigned g2_bits = cut ? (thr > 0u ? thr - 1u : 0u) : thr_g;
                if (lane < CVs) vv.vl[(size_t)lane * ns + i] = lane < n ? s_pos[lane] : 0;
                if (lane == 0) {
                    vv.vn[i] = (unsigned char)n;
                    vv.vg2[i] = __uint_as_float(g2_bits);
                    vv.vacc[i] = 0.f;
                    vv.vmask[i] = 0xFFFFFFFFu;
                }
                if (lane < n && mine > in_radius) s_d2[lane] = 0xFFFFFFFFu;  // (outside the radius: they sort last and are not counted)
                unsigned tm1 = 0xFFFFFFFFu;
                int n_ans = n_in;
                if (n_in >= m) {
                    unsigned t1 = 0;
                    n_ans = wave_select_top_m<PER>(s_pos, s_d2, n, m, tgt, lane, t1);
                    tm1 = t1;
                } else if (n_in > 0) {
                    // fewer than m inside the (fully covered) radius: those are the row — compact them to the front
                    const unsigned long long keep = __ballot(lane < n && s_d2[lane] <= in_radius);
                    const int at = __builtin_amdgcn_mbcnt_hi((unsigned)(keep >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)keep, 0u));
                    const int p_l = lane < n ? s_pos[lane] : 0;
                    const bool k_l = lane < n && s_d2[lane] <= in_radius;
                    if (k_l) s_pos[at] = p_l;  // (at <= lane: a lane reads its own entry before anybody writes a later slot... same wave, lockstep)
                }
                if (lane < n_ans) nbr[(size_t)lane * ns + i] = s_pos[lane];
                if (lane == 0) cnt[i] = n_ans, dm2[i] = tm1;
                done = true;
            }
        }
        if (done) continue;
        for (;;) {
            R2 = fminf(R2 * 2.25f, r2);
            // d2 >= +0 and r2 > 0: "d2 < r2" is "bits(d2) <= bits(r2) - 1"; an intermediate radius may include its sphere's surface
            thr = R2 < r2 ? __float_as_uint(R2) : __float_as_uint(r2) - 1u;
            sweep(R2 * 1.000004f, m);
            if (n >= m || !(R2 < r2)) break;
        }
        unsigned tm = 0xFFFFFFFFu;
        if (n >= m) {
            n = wave_select_top_m<PER>(s_pos, s_d2, n, m, tgt, lane, thr);
            tm = thr;
        }
        if (lane < n) nbr[(size_t)lane * ns + i] = s_pos[lane];
        if (lane == 0) cnt[i] = n, dm2[i] = tm;
        if (vv.vl != nullptr) {
            // ---- the row's Verlet list: everything within G of where the row is now -----------------------------------------
            const float need = tm != 0xFFFFFFFFu ? __builtin_amdgcn_sqrtf(__uint_as_float(tm)) : __builtin_amdgcn_sqrtf(r2);
            // (... as far as this grid's stencil reaches: `reach` cells of edge h around the query's own)
            const float G = fminf(need + fmaxf(vv.skin2, vv.skin_rel * need), (float)reach * g.h * 0.999f);
            const unsigned thr_g = __float_as_uint(G * G);
            thr = thr_g;
            cut = false;
            sweep(G * G * 1.000004f, CVs);
            if (n > CVs) {
                n = wave_select_top_m<PER>(s_pos, s_d2, n, CVs, tgt, lane, thr);
                cut = true;
            }
            // (cut: strictly below the farthest kept — equal distances beyond it were dropped; 0: no list)
            const unsigned g2_bits = cut ? (thr > 0u ? thr - 1u : 0u) : thr_g;
            if (lane < CVs) vv.vl[(size_t)lane * ns + i] = lane < n ? s_pos[lane] : 0;  // (every slot a valid position)
            if (lane == 0) {
                vv.vn[i] = (unsigned char)n;
                vv.vg2[i] = __uint_as_float(g2_bits);
                vv.vacc[i] = 0.f;
                vv.vmask[i] = 0xFFFFFFFFu;  // (the association's row was not written from this list)
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *short_seen = n_short;  // (diagnostic: ppcr_debug_get_short_rows)
}

// ---------------------------------------------------------------------------------------------
// K1, FAST flavour (the one every association launches): same algorithm as the general flavour above — spatially
// compact block of 256 queries, target halo staged once into LDS, per-lane scan of the nine clipped stencil runs
// into a lane-private u16 list, v_med3 threshold selection — stripped of everything the common case does not need,
// because the kernel is bound by VALU issue (round-1 counters: 3050 VALU instructions per wave, 45 % of them
// integer bookkeeping):
//   * ONE halo per workgroup of at most 128 (y,z) row slots, slot = rz << ys | ry with ys = 3 (up to 8 x 16 rows) or
//     4 (16 x 8: blocks that straddle two columns of the source order): no division, no subdivision passes, no
//     per-pass state; a block whose halo does not fit that shape or CAP appends itself to ovf_list and
//     nn_tile_cleanup_kernel redoes it;
//   * the nine run windows are computed branch-free in slice units (one v_sqrt_f32 each: a 1-ulp root is inside the
//     slack the window carries anyway) and their 18 cell_start loads are unconditional (index 0 for a dead run);
//   * wave reductions / scans on the DPP row_shr / row_bcast network instead of ds_bpermute trees;
//   * the halo's sorted-target position is ONE table entry per row (gbo[row] = global begin - LDS offset), so a
//     run's LDS start and a winner's position cost one LDS read each;
//   * list entries are BYTE offsets into the halo arrays (the selection passes use them as addresses as they are),
//     pairs are read with 4-byte alignment from the run's true start (no head test, fewer trips), an accepted
//     candidate costs a store and two VALU instructions;
//   * a lane whose list overflows takes the m-th smallest of the C candidates it did store as its new threshold and
//     scans again (no compacting flavour of the scan in the binary); a second overflow (> C exact ties) hands the
//     block to the cleanup kernel.
// Every d2 that is computed is computed with the same IEEE operations as dist2_flann, and the final selection is the
// same code as before, so neighbour sets and cut-off states stay bit-identical to the general flavour and the oracle.
// ---------------------------------------------------------------------------------------------
#ifndef PPCR_LIST_PERM
#define PPCR_LIST_PERM 1
#endif
#ifndef PPCR_VERLET_ROW_ORDER
#define PPCR_VERLET_ROW_ORDER 0
#endif
#ifndef PPCR_VERLET_RETRY
#define PPCR_VERLET_RETRY 1
#endif
#ifndef PPCR_VERLET_MASK
#define PPCR_VERLET_MASK 1
#endif

// (the unclamped list stores lean on gfx950 dropping DS stores beyond the workgroup's allocation — probed by
//  tools/micro/lds_oob.hip and tests/test_gpu_parity.py — so any other device target gets the clamped form; the host
//  pass sees the same value as the gfx950 device pass this library is built for)
#ifndef PPCR_LIST_NOCLAMP
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#define PPCR_LIST_NOCLAMP 0
#else
#define PPCR_LIST_NOCLAMP 1
#endif
#endif
// candidate source of the fast flavour: list entries are byte offsets (4 * LDS index) into the SoA halo
struct HaloList {
    const char *hx;                 // s_x as bytes; y and z follow at fixed strides
    int stride;                     // bytes between the x, y and z arrays
    const unsigned char *rowid;     // per staged point: its row slot
    const char *gbo;                // s_gbo as bytes
    unsigned short *list;           // [slot * 256 + tid]
    __device__ __forceinline__ float4 get(int a) const
    {
        return make_float4(*reinterpret_cast<const float *>(hx + a), *reinterpret_cast<const float *>(hx + stride + a),
                           *reinterpret_cast<const float *>(hx + 2 * stride + a), 0.f);
    }
    __device__ __forceinline__ int load(int t) const { return list[t * 256]; }
    __device__ __forceinline__ void store(int t, int a) const { list[t * 256] = (unsigned short)a; }
    __device__ __forceinline__ int pos_of(int a) const
    {
        const int e = a >> 2;
        return e + *reinterpret_cast<const int *>(gbo + 4 * rowid[e]);
    }
    __device__ __forceinline__ unsigned orig_of(int a, const float4 *__restrict__ tgt) const
    {
        return (unsigned)__float_as_int(tgt[pos_of(a)].w);
    }
};

// Blocks whose halo outgrew CAP once are SPLIT from then on: the block's own workgroup scans waves 0-1's queries, an
// extra workgroup at the front of the grid scans waves 2-3's (both stage with all four waves; a half-block's halo is
// ~60 % of the block's).  The split set lives in device memory, is extended by the workgroup that bails and takes
// effect at the next launch (the bailing block itself goes to the cleanup kernel this once), so a small CAP — five
// workgroups per CU instead of four — costs one cleanup launch per newly outgrown block, not one per iteration.
struct SplitTable {
    unsigned char *flag;      // [nblocks] 0: whole, 1: registered for splitting, 2: split (an extra workgroup scans waves 2-3)
    int *list;                // [kMaxSplit] block ids, in order of registration
    unsigned *total;          // registrations so far (may exceed kMaxSplit: the blocks with the largest ids are not split)
    const unsigned *visible;  // registrations the extra workgroups of THIS launch may act on (set by the cleanup kernel)
    int n_extra;              // extra workgroups at the front of this launch's grid (0: no splitting in this launch)
    int presplit;             // a whole block whose halo exceeds this is registered for splitting BEFORE it overflows
                              // (halos grow a few per cent per iteration as the source drifts: 15/16 of the capacity)
    int all_halves;           // small clouds: EVERY block is scanned as two half-blocks (grid = 2 * roundup8(blocks), no table):
                              // a cloud of 100k points is 391 blocks on 1280 resident slots, and an iteration lasts as long
                              // as one workgroup's dependent chain — two workgroups per block shorten the chain
};
// all_halves: which block and half workgroup g takes.  Workgroups g and g + 8 (same XCD: its L2 holds the halo both
// stage) take the two halves of the block xcd_block() gives slot (g / 16) * 8 + g % 8; slots beyond the last block idle.
__device__ __forceinline__ int halves_slot(int g) { return (g >> 4) * 8 + (g & 7); }
__device__ __forceinline__ int halves_half(int g) { return 1 + ((g >> 3) & 1); }

// the device-paced loop's state as the association kernel sees it (nullable): it only steps aside while abort is up
struct LoopReset {
    const LoopState *st;
};

// The rows this kernel does not answer, for nn_wide_kernel (row-per-wave search): rows of a workgroup that hands itself
// over or of a lane whose list overflowed twice (marked cnt = -1), and rows that found fewer than m_list neighbours (a
// two-pass search's short rows; m_list = 0 in a one-pass search, whose short rows are final).  `list` null: nobody takes
// them (the cleanup kernel redoes handed-over workgroups).  The counters are a ping-pong pair: this launch appends under
// `count` and leaves `next`, the following association's, at zero.
struct UnansweredRows {
    int *list;
    unsigned *count, *next;
    int m_list;
    // multi-level search (nn_fast_kernel<..., MULTI>): the levels in ascending order of r2_cap, which of them is the base
    // (whose positions the association is written in, and on whose grid nn_wide_kernel searches), the full radius^2
    const GridLevel *levels;
    int n_levels, base_level;
    float r2_full;
    float r2_cap[kMaxLevels];  // the levels' r2_cap, by value (the level choice then needs no memory round trip of its own)
    // feedback of halos that did not fit, per 256-query block and HALF of it (two words per block: [2 * block + half]):
    // coarsest level it may pick | finest << 4 | split << 8.  Read from level_in (what the PREVIOUS launch left), written to
    // level_out by every workgroup that runs (a block that is scanned whole writes both words) — two buffers that swap per
    // launch, so that what a workgroup reads never depends on what its sibling or anybody else does in the same launch:
    // which level a block searches, and with it who answers which row, is the same in every run.
    const unsigned short *level_in;
    unsigned short *level_out;
    unsigned *level_dbg;       // diagnostic (nullable): per level {blocks, handed over for the halo's shape, ... for its size,
                               //   short rows listed, staged candidates, rows}, cumulative (ppcr_debug_get_levels)
    int list_all;              // the last association handed (nearly) every block over — a source far sparser than the target:
                               //   256 of its rows span a halo no tile holds —: move the rows, list them all, try no tile
};

// A workgroup's unanswered rows go to the list in one piece (every thread calls; two barriers; `words`: six LDS words nobody
// else touches at that moment).
__device__ __forceinline__ void list_rows_of(const UnansweredRows &un, const bool mine, const int i, int *words)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long b = __ballot(mine);
    if (lane == 0) words[wave] = __popcll(b);
    lds_barrier();
    const int c0 = words[0], c1 = words[1], c2 = words[2], c3 = words[3];
    const int total_listed = c0 + c1 + c2 + c3;
    if (total_listed == 0) return;  // (uniform)
    if (threadIdx.x == 0) words[4] = (int)atomicAdd(un.count, (unsigned)total_listed);
    lds_barrier();
    const unsigned base = (unsigned)words[4] + (unsigned)(wave > 0 ? c0 : 0) + (unsigned)(wave > 1 ? c1 : 0) + (unsigned)(wave > 2 ? c2 : 0);
    if (mine) un.list[base + __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u))] = i;
}

// Dispatch order of the Verlet variant's workgroups (VerletLists::order_*).  A workgroup that has to search again runs
// ~3 times as long as one that answers from its lists; dispatched late, it IS the launch's tail (493 of 4035 workgroups
// searching: 100 us where their share of the work is 69).  So every workgroup, when it is done, files its slot for the
// NEXT launch under "will probably search" (front) or "will probably not" (back) — a forecast from the rows' remaining
// room and their last move; being wrong costs time, never correctness: what a workgroup does is decided by the test on
// the spot — and the next launch's g-th workgroup takes the g-th slot of front + back.  Per XCD class (slot % 8 ==
// workgroup % 8): a slot's block keeps the XCD whose L2 its neighbours' halos are in.
__device__ __forceinline__ unsigned verlet_slot(const VerletLists &vv, bool *filed_front)
{
    *filed_front = false;
    if (vv.order_now == nullptr) return blockIdx.x;
    const unsigned c = blockIdx.x & 7u, j = blockIdx.x >> 3, per = (gridDim.x + 7u) >> 3;
    const unsigned nf = vv.count_now[2 * c];
    *filed_front = j < nf;
    return j < nf ? (unsigned)vv.order_now[c * per + j] : (unsigned)vv.order_now[8 * per + c * per + (j - nf)];
}
__device__ __forceinline__ void verlet_file_slot(const VerletLists &vv, unsigned wg, bool front)
{
    if (vv.order_next == nullptr) return;
    const unsigned c = wg & 7u, per = (gridDim.x + 7u) >> 3;
    const unsigned pos = atomicAdd(vv.count_next + 2 * c + (front ? 0u : 1u), 1u);
    vv.order_next[(front ? 0u : 8 * per) + c * per + pos] = (int)wg;
}
// the workgroup's forecast: any lane's `mine` -> one verdict, through the LDS word s_word (two barriers; every thread calls)
__device__ __forceinline__ void verlet_forecast(const VerletLists &vv, unsigned wg, bool mine, int *s_word)
{
    if (vv.order_next == nullptr) return;  // (uniform)
    if (threadIdx.x == 0) *s_word = 0;
    lds_barrier();
    const bool any = __ballot(mine) != 0ull;
    if ((threadIdx.x & 63) == 0 && any) *s_word = 1;
    lds_barrier();
    if (threadIdx.x == 0) verlet_file_slot(vv, wg, *s_word != 0);
}

// Answer a workgroup's rows from their Verlet lists (see VerletLists; called by nn_fast_kernel<..., VERLET> once every row
// of the workgroup has passed the completeness test): verlet_slots(M) gathers of 16 bytes per row, the same float d2 as
// everywhere, the m smallest by (d2, original index), the association's row written in list order, K23 folded in from the
// winners (compacted through LDS so that the f64 phase runs without the 48 coordinate registers).
// thr: the association's threshold (radius and temporal cut-off in one, as the scan uses it); s_mem: at least
// VerletLds<M>::kBytes of LDS nobody else is using.
template <int M>
struct VerletLds {
    // per lane: the coordinates of the list entries within the association's threshold, stored as they are measured — up
    // to M + 2 of them (the m winners and two more that the selection then drops) and one spare slot that everything
    // beyond lands in (a row with more: its winners are gathered once more, see below)
    static constexpr int kSlots = M + 3;
    static constexpr int kWinBytes = 3 * kSlots * 256 * 4;
    static constexpr int kBytes = kWinBytes > kFoldScratchBytes ? kWinBytes : kFoldScratchBytes;
};
template <int M, int FTM>
__device__ __forceinline__ void verlet_answer_rows(const int tid, const int i, const bool valid, const float4 q, const int ns,
                                                   const float4 *__restrict__ tgt, const unsigned thr, const int m,
                                                   int *__restrict__ nbr, int *__restrict__ cnt, unsigned *__restrict__ dm2,
                                                   const FusedMoments &fm, const VerletLists &vv, unsigned char *s_mem,
                                                   const unsigned wg, const float g2, const float acc, const float moved, const float radius)
{
    constexpr int BLOCK = 256, CV = verlet_slots(M), S = VerletLds<M>::kSlots;
    static_assert(M <= CV && CV <= 32, "a list holds at least the m winners; the winners' slots fit a 32-bit mask");
    const int nl = valid ? (int)vv.vn[i] : 0;
    // which list slots the row's association (nbr / cnt) was written from last time (all-ones: somebody else wrote it)
    const unsigned was = valid ? vv.vmask[i] : 0xFFFFFFFFu;
    // ---- re-measure the list: all index loads, then all gathers, in flight together --------------------------------
    // Buffer loads (uniform descriptor + ONE 32-bit offset register per address): the list slots of a row share the lane's
    // row offset (the slot is the scalar offset), a gather's address is the position * 16 — no 64-bit address pairs, so
    // that all the gathers are in flight together inside the register budget of four workgroups per CU.
    // (host: ns < 2^26 and nt < 2^28 in this mode, so both byte ranges fit 32 bits)
    // (Issuing these loads earlier — next to the query load, ahead of the completeness test, for the workgroups the
    //  previous launch did not expect to search — was measured: 1M windows 10.55 k -> 9.71 k it/s, converged 13.35 k ->
    //  12.74 k: 32 more loads in flight per lane crowd out the ones the test waits for.  Not kept.)
    unsigned d2b[CV];
    const unsigned row4 = (unsigned)(valid ? i : 0) * 4u;
    const auto rs_vl = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(vv.vl), 0, (int)((unsigned)ns * (unsigned)(CV * 4)), 0x00020000);
    const auto rs_tgt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(tgt), 0, (int)vv.tgt_bytes, 0x00020000);
    // list entry k of this row (a coalesced, cached load)
    auto list_pos = [&](int k) { return (int)__builtin_amdgcn_raw_buffer_load_b32(rs_vl, (int)row4, (int)((unsigned)k * (unsigned)ns * 4u), 0); };
    typedef unsigned v3u __attribute__((ext_vector_type(3)));
    float *const s_wx = reinterpret_cast<float *>(s_mem), *const s_wy = s_wx + S * BLOCK, *const s_wz = s_wy + S * BLOCK;
    unsigned in = 0;  // bit k: list entry k is (still) among the answer
    // Lists of up to 16 slots keep their entries' positions in registers until the association's row is written (re-loading
    // them there one by one — predicated load-then-store pairs, each a round trip of its own — cost every wave with ONE row
    // whose members had changed 15 us: a third of the launch while the source still moves); wider lists are measured in
    // helpings of 16 entries and load the positions once more, all in flight together, where a row is written.
    constexpr int H = CV <= 16 ? CV : 16;
    constexpr bool kKeepPos = CV <= 16;
    static_assert(CV % H == 0, "helpings of equal size");
    int pos_kept[kKeepPos ? CV : 1];
    {
        int w = tid;
        const int w_spare = tid + (S - 1) * BLOCK;
#pragma unroll
        for (int h0 = 0; h0 < CV; h0 += H) {
            // every slot of a list is a valid position (the builders pad with 0), so nothing here is predicated
            int pos[H];
#pragma unroll
            for (int u = 0; u < H; u++) {
                const int p = list_pos(h0 + u);
                pos[u] = valid ? p : 0;  // (a lane without a query read row 0's slots: whatever they hold is not a position to gather from)
                if constexpr (kKeepPos) pos_kept[h0 + u] = pos[u];
            }
            float cx[H], cy[H], cz[H];
#pragma unroll
            for (int u = 0; u < H; u++) {
                const v3u g = __builtin_amdgcn_raw_buffer_load_b96(rs_tgt, (int)((unsigned)pos[u] << 4), 0, 0);
                cx[u] = __uint_as_float(g.x), cy[u] = __uint_as_float(g.y), cz[u] = __uint_as_float(g.z);
            }
            // Each entry is measured and — K23 folded in — its coordinates go to the lane's LDS cursor at once; only an entry
            // within the threshold moves the cursor (as in nn_fast_kernel's scan), so the coordinate registers are free again
            // entry by entry.  The cursor stops at the spare slot.
            // (two entries per packed-f32 instruction — the same IEEE subtract / multiply / add per element as dist2_flann, no
            //  FMA —, and every slot measured: a slot beyond the list's length holds a valid position and is masked out
            //  afterwards, which is cheaper than a branch per entry)
            typedef float v2f __attribute__((ext_vector_type(2)));
            unsigned d2_h[H];
#pragma unroll
            for (int u = 0; u + 1 < H; u += 2) {
                const v2f ex = v2f{q.x, q.x} - v2f{cx[u], cx[u + 1]}, ey = v2f{q.y, q.y} - v2f{cy[u], cy[u + 1]}, ez = v2f{q.z, q.z} - v2f{cz[u], cz[u + 1]};
                v2f d = ex * ex;
                d = d + ey * ey;
                d = d + ez * ez;
                d2_h[u] = __float_as_uint(d.x), d2_h[u + 1] = __float_as_uint(d.y);
            }
            if constexpr (H % 2) d2_h[H - 1] = __float_as_uint(dist2_flann(q, make_float4(cx[H - 1], cy[H - 1], cz[H - 1], 0.f)));
#pragma unroll
            for (int u = 0; u < H; u++) {
                const int k = h0 + u;
                d2b[k] = (k < nl) ? d2_h[u] : 0xFFFFFFFFu;
                const bool hit = d2b[k] <= thr;
                in |= hit ? (1u << k) : 0u;
                if constexpr (FTM != -2) {
                    s_wx[w] = cx[u];
                    s_wy[w] = cy[u];
                    s_wz[w] = cz[u];
                    w = hit ? min(w + BLOCK, w_spare) : w;
                }
            }
        }
    }
    const unsigned in_first = in;           // the entries within the threshold, in list order: LDS slot j holds the j-th of them
    const int n_first = __popc(in_first);
    unsigned slot_alive = n_first >= 32 ? 0xFFFFFFFFu : ((1u << n_first) - 1u);  // bit j: LDS slot j is (still) among the answer
    int n = n_first;
    // ---- more than m within the threshold: the largest by (d2, original index) leave, one per round ------------------
    int surplus = n - m;
    while (__ballot(surplus > 0) != 0ull) {
        if (surplus > 0) {
            unsigned best = 0;
#pragma unroll
            for (int k = 0; k < CV; k++) best = max(best, ((in >> k) & 1u) ? d2b[k] : 0u);
            int bk = 0, ties = 0;
#pragma unroll
            for (int k = 0; k < CV; k++) {
                const bool hit = ((in >> k) & 1u) && d2b[k] == best;
                bk = hit ? k : bk;
                ties += hit ? 1 : 0;
            }
            if (ties > 1) {  // equal distances at the boundary: the larger original index leaves (the oracle's order)
                unsigned worst = 0;
#pragma unroll
                for (int k = 0; k < CV; k++)
                    if (((in >> k) & 1u) && d2b[k] == best) {
                        const unsigned o = __builtin_amdgcn_raw_buffer_load_b32(rs_tgt, (int)(((unsigned)list_pos(k) << 4) + 12u), 0, 0);
                        if (o >= worst) worst = o, bk = k;
                    }
            }
            in &= ~(1u << bk);
            slot_alive &= ~(1u << __popc(in_first & ((1u << bk) - 1u)));  // (its LDS slot: how many accepted entries precede it)
            surplus--;
        }
    }
    n = min(n, m);
    unsigned tm = 0xFFFFFFFFu;  // d2 bits of the m-th neighbour (all-ones: fewer than m)
    if (n == m) {
        tm = 0;
#pragma unroll
        for (int k = 0; k < CV; k++) tm = max(tm, ((in >> k) & 1u) ? d2b[k] : 0u);
    }
    if (valid) {
        // the association's row only when its members changed: in a registration that has all but converged they do not,
        // and 44 bytes per row of writes stay away (the m-th distance, which does change, always goes out)
        if (!PPCR_VERLET_MASK || in != was) {
            int *out = nbr + i;
            if constexpr (kKeepPos) {
#pragma unroll
                for (int k = 0; k < CV; k++)
                    if ((in >> k) & 1u) {
                        *out = pos_kept[k];
                        out += ns;
                    }
            } else {
#pragma unroll
                for (int h0 = 0; h0 < CV; h0 += H) {
                    int pos[H];
#pragma unroll
                    for (int u = 0; u < H; u++) pos[u] = list_pos(h0 + u);  // (all in flight, then the stores)
#pragma unroll
                    for (int u = 0; u < H; u++)
                        if ((in >> (h0 + u)) & 1u) {
                            *out = pos[u];
                            out += ns;
                        }
                }
            }
            cnt[i] = n;
            vv.vmask[i] = in;
        }
        dm2[i] = tm;
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FTM != -2) {
        // ---- K23 for this row at fm.P, from the coordinates in LDS ----------------------------------------------------
        // a row with more entries within the threshold than the lane's slots hold (three or more beyond the m winners: a few
        // rows in a thousand while the source still moves, none once it has stopped): its winners' coordinates are
        // gathered once more, into slots 0 .. n - 1
        const bool lost = n_first > S - 1;
        if (__ballot(lost) != 0ull) {
            // (in two helpings of CV / 2 entries, each with its loads in flight together: one entry at a time was two dependent
            //  round trips per winner, 10 us for every wave that holds such a row — four in ten while the source moves 0.005 radii)
            int w = tid;
#pragma unroll
            for (int h0 = 0; h0 < CV; h0 += CV / 2) {
                int pos[CV / 2];
#pragma unroll
                for (int u = 0; u < CV / 2; u++) pos[u] = lost ? list_pos(h0 + u) : 0;
                v3u g[CV / 2];
#pragma unroll
                for (int u = 0; u < CV / 2; u++) g[u] = __builtin_amdgcn_raw_buffer_load_b96(rs_tgt, (int)((unsigned)pos[u] << 4), 0, 0);
#pragma unroll
                for (int u = 0; u < CV / 2; u++)
                    if (lost) {
                        s_wx[w] = __uint_as_float(g[u].x), s_wy[w] = __uint_as_float(g[u].y), s_wz[w] = __uint_as_float(g[u].z);
                        w += ((in >> (h0 + u)) & 1u) ? BLOCK : 0;  // (at most m winners: the cursor stays inside the lane's slots)
                    }
            }
            if (lost) slot_alive = (1u << n) - 1u;
        }
        RowAcc sums;
#pragma unroll
        for (int j = 0; j < kNSums; j++) sums.a[j] = 0.0;
        if (valid && n > 0) {
            double xr[3];
            rotated_point(fm.P, q, xr);
            RowMoments<FTM> row;
            row.begin(fm.md);
            // the j-th winner sits in the slot of the j-th set bit (a per-lane LDS address): m trips, whatever was dropped
            unsigned rem = slot_alive;
#pragma unroll
            for (int j = 0; j < M; j++)
                if (j < n) {  // nearly every row is full: the branch is uniform for most waves
                    const int sl = __builtin_ctz(rem) * BLOCK + tid;
                    rem &= rem - 1u;
                    row.add_pair(fm.md, xr, s_wx[sl], s_wy[sl], s_wz[sl]);  // (a lane reads what it wrote)
                }
            row.finish(sums, fm.P, q, xr);
        }
        __syncthreads();  // every lane is through with its winners: the fold borrows the memory
        double *const scratch = reinterpret_cast<double *>(s_mem);
        block_reduce_scratch(sums, scratch, scratch + 10 * 257, fm.partials + wg, (size_t)fm.nslots, true);
        __syncthreads();  // (the forecast below borrows a word of the same memory)
    }
    // will this row's list still do after one more move like the last one?  (VerletLists' test with the new m-th distance)
    const float need_next = (tm != 0xFFFFFFFFu ? __builtin_amdgcn_sqrtf(__uint_as_float(tm)) : radius) + moved;
    const float reach_next = (need_next + acc + moved) * 1.0001f;
    verlet_forecast(vv, wg, valid && !(reach_next * reach_next < g2), reinterpret_cast<int *>(s_mem));
}

// PER-ROW REBUILD.  The completeness test is per row; sending a workgroup's 256 rows through the search because ONE of them
// failed (the ~30 us chain of the tiled search, against ~10 us for answering) was the steady state's largest cost: a row
// whose list has little room — sixteen targets hardly farther than its tenth — fails every few iterations, and one
// workgroup in eleven held such a row in the benchmark's timed windows.  A workgroup with at most PPCR_VERLET_ROWS failing
// rows now rebuilds just THOSE rows' lists, one row per wave (verlet_rebuild_row: the nine clipped stencil runs walked by
// the 64 lanes as one flat sequence straight from global memory — no halo staging —, accepted candidates appended to a
// wave-shared LDS list by ballot, the sixteen nearest by (d2, original index) kept by wave_select_top_m), and then answers
// ALL its rows from their lists as any other workgroup does: the association, the order K23 adds in, the dispatch
// forecast are the list path's own, whatever the failing rows were.  A fresh list answers its own build position exactly
// whatever its room (the m <= 16 nearest are among the sixteen nearest).  More failing rows than that: the tiled search,
// as before (it rebuilds all 256 lists in one go).
#ifndef PPCR_VERLET_ROWS
#define PPCR_VERLET_ROWS 0
#endif
// ... where the kernel's unanswered rows are searched one row per wave anyway (nn_wide_kernel: two-pass and multi-level
// searches), a workgroup with at most this many failing rows lists them there and answers the others
#ifndef PPCR_VERLET_LIST_ROWS
#define PPCR_VERLET_LIST_ROWS 4
#endif
constexpr int kVerletRowCap = 128;  // entries of a wave's candidate list (compacted to the list's slots when a round might not fit)
template <int ROWS>
struct VerletRowLds {
    int pos[4][kVerletRowCap];
    unsigned d2[4][kVerletRowCap];
    float4 q[ROWS > 0 ? ROWS : 1];     // the failing rows: query (w: the bound `need` on its m-th distance) ...
    int row[ROWS > 0 ? ROWS : 1];      // ... row index ...
    float g2[ROWS > 0 ? ROWS : 1];     // ... and, coming back, the new list's reach
};
// rows a workgroup rebuilds one by one before it gives in and searches.  Where launches last many residency rounds (1M rows,
// K23 folded in) none: the search renews all 256 lists of a workgroup whose rows age together, and hides behind the other
// workgroups' work (measured: PPCR_VERLET_ROWS above).  Where the whole grid is resident at once — the mid widths' clouds of a
// few hundred blocks, any cloud of up to ~230k rows — a searching workgroup IS the launch's length (and a launch of
// nn_wide_kernel for a handful of rows costs 20 us of latency): eight (kernel variant 2).
constexpr int verlet_rows_in_kernel(int M, int variant) { return (M > 12 || variant == 2) ? 8 : PPCR_VERLET_ROWS; }
// one wave, one row: i / q / need are wave-uniform (read from LDS)
template <int CVS>
__device__ __forceinline__ float verlet_rebuild_row(const int i, const float4 q, const float need, const int ns, const float4 *__restrict__ tgt,
                                                    const int *__restrict__ cell_start, const GridDesc &g, const VerletLists &vv,
                                                    const int lane, int *s_pos, unsigned *s_d2)
{
    const float bnd = need + fmaxf(vv.skin2, vv.skin_rel * need);
    const unsigned thr_v = __float_as_uint(bnd * bnd);  // the scan's acceptance threshold (nn_fast_kernel: thr_v)
    unsigned thr = thr_v;
    const float R2s = bnd * bnd * 1.000004f;
    const QueryCells c = query_cells(q, g);
    // lane l < 9: the (dy, dz) row l of the stencil, clipped to the x slices the sphere of radius G can touch there
    // (the arithmetic of nn_wide_kernel's run_bounds with a reach of one cell)
    int rb = 0, re = 0;
    if (lane < 9) {
        const int dz = lane / 3 - 1, dy = lane % 3 - 1;
        const int cz = c.cz + dz, cy = c.cy + dy;
        const float fy = q.y - g.org[1], fz = q.z - g.org[2];
        if ((unsigned)cz < (unsigned)g.n[2] && (unsigned)cy < (unsigned)g.n[1]) {
            const float gz = dz < 0 ? fmaxf(fz - (float)(cz + 1) * g.h - g.eps, 0.f) : (dz > 0 ? fmaxf((float)cz * g.h - fz - g.eps, 0.f) : 0.f);
            const float gy = dy < 0 ? fmaxf(fy - (float)(cy + 1) * g.h - g.eps, 0.f) : (dy > 0 ? fmaxf((float)cy * g.h - fy - g.eps, 0.f) : 0.f);
            const float w2 = R2s - (gy * gy + gz * gz);
            if (w2 > 0.f) {
                const float w = __builtin_amdgcn_sqrtf(w2) * 1.000001f + g.eps;
                const int fa = max(cell_coord(q.x - w, g.org[0], g.inv_hx, g.n[0]), 0);
                const int fb = min(cell_coord(q.x + w, g.org[0], g.inv_hx, g.n[0]), g.n[0] - 1);
                if (fa <= fb) {
                    const int base = (cz * g.n[1] + cy) * g.n[0];
                    rb = cell_start[base + fa], re = cell_start[base + fb + 1];
                }
            }
        }
    }
    const int len = re - rb;
    const int incl = wave_scan(len, 0, OpAdd());
    const int total = __builtin_amdgcn_readlane(incl, 63);
    int sb[9], se[9];  // the nine runs' first positions and their offsets in the flat sequence (scalars)
#pragma unroll
    for (int r = 0; r < 9; r++) {
        sb[r] = __builtin_amdgcn_readlane(rb, r);
        se[r] = __builtin_amdgcn_readlane(incl - len, r);
    }
    int n = 0;
    bool cut = false;  // entries were dropped: the list is complete only strictly below the farthest kept
    for (int j0 = 0; j0 < total; j0 += 64) {
        if (n > kVerletRowCap - 64) {
            n = wave_select_top_m<kVerletRowCap / 64>(s_pos, s_d2, n, CVS, tgt, lane, thr);
            cut = true;
        }
        const int j = j0 + lane;
        int pos = 0;
#pragma unroll
        for (int r = 0; r < 9; r++) pos = (j >= se[r]) ? sb[r] + (j - se[r]) : pos;  // (the last run that starts at or before j: never an empty one while j < total)
        const bool live = j < total;
        const float4 t = tgt[live ? pos : 0];
        const unsigned bits = __float_as_uint(dist2_flann(q, t));
        const bool acc = live && bits <= thr;
        const unsigned long long k = __ballot(acc);
        const int at = n + __builtin_amdgcn_mbcnt_hi((unsigned)(k >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)k, 0u));
        if (acc) s_pos[at] = pos, s_d2[at] = bits;
        n += __popcll(k);
    }
    if (n > CVS) {
        n = wave_select_top_m<kVerletRowCap / 64>(s_pos, s_d2, n, CVS, tgt, lane, thr);
        cut = true;
    }
    // (strictly below the farthest kept: equal distances beyond it were dropped; 0: no list — sixteen targets at distance 0)
    const unsigned g2_bits = cut ? (thr > 0u ? thr - 1u : 0u) : thr_v;
    if (lane < CVS) vv.vl[(size_t)lane * ns + i] = lane < n ? s_pos[lane] : 0;  // (every slot a valid position)
    if (lane == 0) {
        vv.vn[i] = (unsigned char)n;
        vv.vg2[i] = __uint_as_float(g2_bits);
        vv.vacc[i] = 0.f;
        vv.vmask[i] = 0xFFFFFFFFu;  // (the association's row was not written from this list yet)
    }
    return __uint_as_float(g2_bits);
}

// FTM != -2 (0: Gaussian, k > 0: t model with v + dim = k, -3: t model with an integer v + dim read at run time) folds
// K23 into this kernel: each lane finishes its row's
// contribution to the 19 moments from the winners' coordinates while they are still in LDS (no neighbour gathers, no
// second pass over the source, no K23 launch) and the workgroup folds them into fm.partials.  FTM = -2: plain K1.
// The ONE LDS allocation of nn_fast_kernel (PPCR_LIST_NOCLAMP): halo (x[], y[], z[], row bytes), row table, boxes, flags, and
// LAST the lists — the unclamped stores of the scan rely on nothing of the workgroup's lying behind them.  The host checks
// the compiled kernels' static LDS size against kBytes before the first launch (check_fast_kernel_lds): a __shared__
// variable that slipped into the kernel some other way would sit behind the list and be overwritten silently.
// (kAllocBytes: what the kernel allocates — the Verlet variant answers rows from their lists in the same memory and needs a
//  little more than the search does; the padding behind the list is nobody's)
template <int C, int CAP, bool MULTI, int VERLET_M = 0>
struct FastLds {
    static constexpr int kHaloBytes = (3 * CAP + CAP / 4) * 4, kListBytes = (C + 1) * 256 * 2;
    static constexpr int kOffGbo = kHaloBytes, kOffBox = kOffGbo + 128 * 4, kOffBail = kOffBox + 4 * 6 * 4, kOffNeed = kOffBail + 4,
                         kOffList = (kOffNeed + (MULTI ? 4 * 16 * 4 : 4) + 15) & ~15;  // (one word at kOffNeed also without MULTI: the Verlet forecast's)
    static constexpr int kBytes = kOffList + kListBytes;
    static constexpr int kVerletBytes = VERLET_M > 0 ? VerletLds<(VERLET_M > 0 ? VERLET_M : 1)>::kBytes : 0;
    static constexpr int kAllocBytes = kBytes > kVerletBytes ? kBytes : kVerletBytes;
};

// MULTI: the grid is chosen per workgroup from un.levels (GridLevel): tgt / cell_start / g / r2 of the arguments are the
// base level's and only used by the first association (no cut-offs yet: every block searches the base level).
// VERLET (steady-state variant only): the rows keep Verlet lists (see VerletLists).  A workgroup whose rows all pass the
// completeness test answers them from their lists (verlet_answer_rows) and is done; one that holds a failing row — or any
// workgroup when vv.build_all says that no lists exist yet — searches as usual and leaves fresh lists behind.  One kernel
// for both: a rebuild is one workgroup's chain of dependent round trips (~25 us), which a launch of its own would add to
// every iteration in which a single row of the cloud fails — inside the launch it hides behind the other workgroups' work.
// Four workgroups per CU (the list path holds 48 coordinates in registers).
#ifndef PPCR_CUT_TIES_BY_INDEX
#define PPCR_CUT_TIES_BY_INDEX 1  // (0: the defect tests/test_gpu_parity.py::test_wide_lists_keep_the_tie_rule_on_a_lattice was written against)
#endif
#ifndef PPCR_LEVEL_Q16
#define PPCR_LEVEL_Q16 15  // sixteenths of a block's rows whose cut-offs the block's level must cover (multi-level search)
#endif
template <int M, int C, int CAP, bool STAMPS, int FTM = -2, bool MULTI = false, int VERLET_K = 0>
__global__ __launch_bounds__(256, (VERLET_K != 0 ? (CAP * 13 + C * 512 > 40960 ? 3 : 4) : C <= 16 ? (CAP * 13 + C * 512 <= 30900 ? 5 : 4) : (CAP * 13 + C * 512 <= 39400 ? 4 : 3))) void nn_fast_kernel(float4 *__restrict__ src, int ns,
                                                         const float4 *__restrict__ tgt0,
                                                         const int *__restrict__ cell_start0, GridDesc g0,
                                                         float r2_0, int m, int *__restrict__ nbr,
                                                         int *__restrict__ cnt, PendingMove pm,
                                                         unsigned *__restrict__ dm2, int dm2_valid,
                                                         int *__restrict__ ovf_list, unsigned *__restrict__ ovf_count,
                                                         unsigned *__restrict__ ovf_count_next, SplitTable split,
                                                         unsigned long long *__restrict__ stamps, FusedMoments fm,
                                                         LoopReset lr, UnansweredRows un, VerletLists vv)
{
    // VERLET_K: 0 no lists, 1 Verlet lists, 2 Verlet lists and a few failing rows rebuilt inside the workgroup (grids that are
    // resident all at once: see verlet_rows_in_kernel)
    constexpr bool VERLET = VERLET_K != 0;
    // an earlier launch may have handed the iteration to the host (LoopState::abort, set before this kernel started):
    // then this one must touch nothing.  A uniform scalar load, tested below once the query load is in flight.
    const unsigned aborted = lr.st ? lr.st->abort : 0u;
    static_assert(C > M, "a re-scan must leave room in the list");
    static_assert(!VERLET || (C >= verlet_slots(M) && !STAMPS), "Verlet lists are built by the steady-state variant: a list is the nearest part of the scan's LDS list");
    static_assert(!MULTI || FTM == -2, "a multi-level search leaves rows to nn_wide_kernel: K23 is its own kernel");
    static_assert(FTM == -2 || (3 * CAP + CAP / 4) * 4 >= kFoldScratchBytes, "the final fold borrows the halo buffer");
    static_assert(CAP % 4 == 0 && CAP * 4 < 65536, "list entries are 16-bit byte offsets into the halo arrays");
    // (MULTI, three workgroups per CU: the curve-ordered blocks of a multi-level search have up to 128 short rows, and
    //  registers to keep twelve of them in flight)
    constexpr int BLOCK = 256, kWaves = 4, kRows = 128, kStageUnroll = (MULTI && C > 16) ? 12 : 6;
#if PPCR_LIST_NOCLAMP
    // ONE allocation with the list LAST: a store beyond the list's spare slot then leaves the workgroup's LDS allocation,
    // where the hardware drops it (gfx950, as the ISA documents; probed by tools/micro/lds_oob.hip, which
    // tests/test_gpu_parity.py builds and runs) — the scan stores at its cursor without clamping it to the list's end:
    // two v_min_u32 less on a 22-VALU trip (2508 -> 2427 VALU per wave, +1.6 % iterations/s; -DPPCR_LIST_NOCLAMP=0 is
    // the clamped form with separate arrays).
    using Lds = FastLds<C, CAP, MULTI, ((VERLET && FTM != -2) ? M : 0)>;  // (the list path keeps coordinates in LDS only where K23 is folded in)
    constexpr int kListBytes = Lds::kListBytes, kOffGbo = Lds::kOffGbo, kOffBox = Lds::kOffBox, kOffBail = Lds::kOffBail,
                  kOffNeed = Lds::kOffNeed, kOffList = Lds::kOffList;
    static_assert(BLOCK == 256 && kWaves == 4 && kRows == 128, "FastLds mirrors these");
    __shared__ __attribute__((aligned(16))) unsigned char s_all[Lds::kAllocBytes];
    float *const s_halo = reinterpret_cast<float *>(s_all);
    int *const s_gbo = reinterpret_cast<int *>(s_all + kOffGbo);
    int(*const s_box)[6] = reinterpret_cast<int(*)[6]>(s_all + kOffBox);
    int &s_bail = *reinterpret_cast<int *>(s_all + kOffBail);
    int *const s_need = reinterpret_cast<int *>(s_all + kOffNeed);  // (MULTI: the waves' largest cut-offs)
    unsigned short *const s_list = reinterpret_cast<unsigned short *>(s_all + kOffList);
    static_assert(sizeof(int2) * kRows <= kListBytes, "row table aliases the list area");
    int2 *const s_rowtab = reinterpret_cast<int2 *>(s_list);
#else
    __shared__ __attribute__((aligned(16))) float s_halo[3 * CAP + CAP / 4];
    __shared__ unsigned short s_list[(C + 1) * BLOCK];  // C slots per lane + one that rejected candidates land in
    __shared__ int s_gbo[kRows];
    // non-empty rows, compacted: {global begin, LDS offset << 19 | length << 7 | slot}; only alive between the row
    // table and the staging barrier, so it borrows the (not yet written) list area
    static_assert(sizeof(int2) * kRows <= sizeof(s_list), "row table aliases the list area");
    int2 *const s_rowtab = reinterpret_cast<int2 *>(s_list);
    __shared__ int s_box[kWaves][6];
    __shared__ int s_bail;
    __shared__ int s_need[kWaves * 16];
#endif
    float *const s_x = s_halo, *const s_y = s_halo + CAP, *const s_z = s_halo + 2 * CAP;
    unsigned char *const s_rowid = reinterpret_cast<unsigned char *>(s_halo + 3 * CAP);

    // diagnostic only (STAMPS instantiation): per-wave, per-phase cycle counts
    unsigned long long t_prev = 0, t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (STAMPS) t_prev = clock64();
    auto stamp = [&](int phase) {
        if constexpr (STAMPS) {
            const unsigned long long now = clock64();
#pragma unroll
            for (int k = 0; k < 8; k++) t_acc[k] += (k == phase) ? now - t_prev : 0ull;
            t_prev = now;
        }
    };
    auto flush_stamps = [&]() {
        if constexpr (STAMPS)
            if (stamps && (threadIdx.x & 63) == 0)
                for (int k = 0; k < 8; k++) stamps[((size_t)blockIdx.x * kWaves + (threadIdx.x >> 6)) * 8 + k] = t_acc[k];
    };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the workgroup's SLOT: its index in the launch, or (Verlet variant) the slot the previous launch filed for this place
    // in the dispatch order (verlet_slot).  The slot names the block, the partial sums' column, the hand-over entry.
    unsigned wg = blockIdx.x;
    // (a launch that steps aside does not look at the order: the launch that should have filed it may have stepped aside too)
    bool filed_front = false;  // (Verlet dispatch order: the previous launch expected this workgroup to search again)
    if constexpr (VERLET)
        if (!aborted) wg = (unsigned)__builtin_amdgcn_readfirstlane((int)verlet_slot(vv, &filed_front));

    // which block, and which of its waves' queries, this workgroup scans (uniform)
    int bid, half = 0;  // half: 0 whole block, 1 waves 0-1, 2 waves 2-3
    unsigned level_fb = 0x0Fu;  // MULTI: the block's feedback word (UnansweredRows::level_in)
    float4 q_early = make_float4(0.f, 0.f, 0.f, 0.f);
    if (split.all_halves) {
        const int slot = halves_slot((int)wg), nb = (ns + BLOCK - 1) / BLOCK;
        if (slot >= nb) {  // (padding of the grid to a multiple of sixteen workgroups)
            if constexpr (FTM != -2)
                if (tid < kNSums) fm.partials[(size_t)tid * fm.nslots + wg] = 0.0;
            return;
        }
        bid = xcd_block(slot, nb);
        half = halves_half((int)wg);
        if constexpr (MULTI) {
            // multi-level searches launch the two-workgroups-per-block grid and use the second workgroup only for the
            // blocks marked split (bit 8 of the feedback word: their halo outgrew the tile at the level their cut-offs ask for;
            // half a block's halo is ~60 % of the block's): everybody else's second workgroup leaves at once
            if (un.level_in != nullptr) {
                // the two halves' words as one: the finer cap, the coarser floor, split if either asked for it
                const unsigned fa = un.level_in[2 * bid], fb = un.level_in[2 * bid + 1];
                level_fb = min(fa & 15u, fb & 15u) | (max((fa >> 4) & 15u, (fb >> 4) & 15u) << 4) | ((fa | fb) & 0x100u);
            }
            // (the query is asked for in the same breath: which lanes are valid depends on the byte, the load need not wait for it)
            q_early = src[min(bid * BLOCK + tid, ns - 1)];
            if (!(level_fb & 0x100u)) {
                if (half == 2) return;
                half = 0;
            }
        }
    } else if ((int)wg < split.n_extra) {
        if (wg >= min(*split.visible, (unsigned)kMaxSplit)) {
            if constexpr (FTM != -2)  // an idle slot of the partials still has to read as zero
                if (tid < kNSums) fm.partials[(size_t)tid * fm.nslots + wg] = 0.0;
            if constexpr (VERLET)
                if (tid == 0 && !aborted) verlet_file_slot(vv, wg, false);
            return;
        }
        bid = split.list[wg];
        half = 2;
    } else {
        bid = xcd_block((int)wg - split.n_extra, (ns + BLOCK - 1) / BLOCK);
        if (split.n_extra > 0) {
            // (the block's flag byte through a SCALAR load of the word that holds it — a uniform address; pool blocks are
            //  multiples of 512 bytes, so the word exists —: as a vector load it was a round trip of its own ahead of the query's)
            typedef const __attribute__((address_space(4))) unsigned *const_u32p;
            const unsigned word = ((const_u32p)(__UINTPTR_TYPE__)split.flag)[bid >> 2];
            if (((word >> (8 * (bid & 3))) & 0xFFu) == 2u) half = 1;
        }
    }
    const int i = bid * BLOCK + tid;
    const bool valid = i < ns && (half == 0 || (wave >> 1) == half - 1);  // lanes whose query this workgroup owns
    if (tid == 0) s_bail = 0;

    // ---- prologue: query, pending move, temporal cut-off ---------------------------------------------------------
    // The row's state — cut-off of the previous association, the list's reach and path — and the pending move are asked for
    // in the same breath as the query: none of them depends on it, and taken where they are used each was a memory round
    // trip of its own on every workgroup's critical path (query -> move -> cut-off -> list state: four in a row).
    // Unconditional loads (row 0 for a lane without a query: every buffer has it), so that they are ONE block of loads.
    float4 q;
    unsigned prev_dm2 = 0xFFFFFFFFu;
    float g2_row = 0.f, acc_row = 0.f;
    if constexpr (VERLET) {
        const int ic = valid ? i : 0;
        const float4 q_ld = src[ic];
        const unsigned prev_ld = dm2[ic];
        const float g2_ld = vv.vg2[ic], acc_ld = vv.vacc[ic];
        q = valid ? q_ld : make_float4(0.f, 0.f, 0.f, 0.f);
        prev_dm2 = (valid && dm2_valid) ? prev_ld : 0xFFFFFFFFu;
        g2_row = (valid && !vv.build_all) ? g2_ld : 0.f;
        acc_row = (valid && !vv.build_all) ? acc_ld : 0.f;
    } else if constexpr (!MULTI) {
        const int ic = valid ? i : 0;
        const float4 q_ld = src[ic];
        const unsigned prev_ld = dm2[ic];
        q = valid ? q_ld : make_float4(0.f, 0.f, 0.f, 0.f);
        prev_dm2 = (valid && dm2_valid) ? prev_ld : 0xFFFFFFFFu;
    } else {
        q = valid ? (split.all_halves ? q_early : src[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid && dm2_valid) prev_dm2 = dm2[i];
    }
    Pose P_move = pm.P;
    if (pm.enabled == 2) {
        // (the previous iteration's transform, left in device memory by the lane that solved it: a uniform address, read with
        //  SCALAR loads through the constant address space — the launch boundary orders the write before them)
        typedef const __attribute__((address_space(4))) double *const_f64p;
        const const_f64p pd = (const_f64p)(__UINTPTR_TYPE__)pm.dev;
#pragma unroll
        for (int k = 0; k < 9; k++) P_move.R[k] = pd[k];
#pragma unroll
        for (int k = 0; k < 3; k++) P_move.t[k] = pd[9 + k];
    }
    if (aborted) return;
    // the other counter of the ping-pong pair: idle during this launch  (the last workgroup is never an idle split slot;
    // with all_halves the first one never idles)
    if (tid == 0 && wg == (split.all_halves ? 0u : gridDim.x - 1)) {
        *ovf_count_next = 0;
        if (un.next != nullptr) *un.next = 0;
        if constexpr (VERLET)
            if (vv.count_clear != nullptr)
                for (int k = 0; k < 16; k++) vv.count_clear[k] = 0;  // (the dispatch-order counters of the launch after next)
        if constexpr (VERLET)
            if (vv.searched_clear != nullptr) *vv.searched_clear = 0;
    }
    float moved = 0.f;  // how far this query travelled since the association that produced dm2
    if (pm.enabled && valid) {
        const float4 q0 = q;
        q = move_point(q, P_move);
        src[i] = q;
        const float ex = q.x - q0.x, ey = q.y - q0.y, ez = q.z - q0.z;
        moved = __builtin_amdgcn_sqrtf(ex * ex + ey * ey + ez * ez);  // 1 ulp: far inside the 1e-5 inflation below
    }
    // Temporal cut-off.  dm2[i] holds the float d2 of this query's m-th neighbour in the previous association
    // (all-ones when it had fewer than m).  Those m targets are now at most dm + |move| away, so the new m-th distance
    // is <= dm + |move|: a candidate farther than that cannot be among the m closest and is never appended.  The
    // bound is inflated by 1e-5 (float rounding of d2 and of the two roots is ~1e-6 relative); the final selection is
    // exact — only the amount of list traffic changes.
    unsigned thr0 = 0xFFFFFFFFu;
    // VERLET: the same bound against the FULL radius (a two-pass search's first-pass radius r2_0 is smaller: a row whose m-th
    // neighbour lies beyond it has no cut-off for the tiled scan, but nn_wide_kernel gave it one — and a list — all the same)
    // (K23 folded in: a one-pass search — its radius IS the full radius, and the argument is at hand in a register)
    const float r2_far = (FTM != -2) ? r2_0 : un.r2_full;
    float need2_full = r2_far;
    if (dm2_valid && valid) {
        const unsigned prev = prev_dm2;
        if (prev != 0xFFFFFFFFu) {
            const float bound = __builtin_amdgcn_sqrtf(__uint_as_float(prev)) + moved;
            const float t2 = bound * bound * 1.00001f + 1e-30f;
            thr0 = (t2 < (MULTI ? un.r2_full : r2_0)) ? __float_as_uint(t2) : 0xFFFFFFFFu;
            need2_full = fminf(t2, r2_far);
        }
    }
    // What a row this kernel leaves to nn_wide_kernel finds in dm2: the BOUND just formed — the m neighbours of the previous
    // association lie within it where the row is now, so one sweep of that sphere holds the row's answer (the kernel had
    // the row's density to go by: an estimate, a sweep too wide or one too many) —, all-ones where there is none.
    // (K23 folded in: nothing goes to that kernel)
    const unsigned bound_bits = (FTM == -2 && need2_full < r2_far) ? __float_as_uint(need2_full) : 0xFFFFFFFFu;
    if constexpr (FTM == -2 && !VERLET) {
        if (un.list_all && un.list != nullptr) {
            // (uniform) every row to nn_wide_kernel, marked unsearched, every block counted as handed over.  The list is the
            // identity and both counters are known: no atomics (4 000 waves drawing list positions from one counter at the
            // same moment cost this path 20 us)
            if (valid) cnt[i] = -1, un.list[i] = i, dm2[i] = bound_bits;
            // (written by a workgroup that always gets here: slot 0 of the steady-state grids is a split slot and leaves
            //  above when no block is registered — every row stayed marked unsearched and nn_wide_kernel saw an empty list:
            //  found by the association soak under another seed)
            if (tid == 0 && wg == (split.all_halves ? 0u : gridDim.x - 1)) {
                *un.count = (unsigned)ns;
                *ovf_count = (unsigned)((ns + BLOCK - 1) / BLOCK);
            }
            flush_stamps();
            return;
        }
    }
    // VERLET: the scan collects every target within G = bound + 2 skin (the bound on the m-th distance, or the radius for a
    // row that has none) — where its stencil covers G: the grid's cells are at least (first-pass) radius + 2 skin wide, so
    // in a one-pass search always, in a two-pass search for the rows whose bound lies inside the first-pass radius (the
    // others' lists are built by nn_wide_kernel, which searches them anyway)
    unsigned thr_v = 0;
    float need = 0.f, bnd_v = 0.f;
    bool scan_lists = false;  // this launch's scan (if it comes to that) builds this row's list (decided once the grid is known)
    bool plain_search = false;  // (VerletLists::streak: this block keeps searching — no lists from its scan)
    if constexpr (VERLET) {
        // how far the m-th neighbour (or, without a bound, the radius) can be from the query where it is now
        need = __builtin_amdgcn_sqrtf(need2_full);
        bnd_v = need + (MULTI ? fmaxf(vv.skin2, vv.skin_rel * need) : vv.skin2);
        // `near`: the 27-cell stencil of the BASE grid covers the list's reach — the row's list can be rebuilt in this
        // kernel, one row per wave; the others' bounds lie beyond (a two-pass search's short rows, a multi-level search's
        // rows of coarser levels): nn_wide_kernel's
        // (K23 folded in: a one-pass, single-level search — every row is near, and nothing below is compiled for the others)
        const bool near = (FTM != -2) ? true : (bnd_v <= g0.h);
        if constexpr (FTM == -2)
            if (vv.build_all && tid == 0 && vv.streak != nullptr) vv.streak[wg] = 0;  // (a launch that builds every list starts the count over)
        if (!vv.build_all) {
            // ---- are the lists still complete?  (VerletLists: need + path travelled < the list's reach, with float slack) ----
            float g2 = g2_row, acc = valid ? acc_row + moved : 0.f;
            const float reach = (need + acc) * 1.0001f;
            // (no list: g2 = 0; a NaN anywhere fails — except in the query itself: a row that is not a point has no neighbours
            //  whatever its list says, and must not send its workgroup through the search in every iteration)
            const bool is_point = (q.x - q.x) == 0.f && (q.y - q.y) == 0.f && (q.z - q.z) == 0.f;
            const bool ok = !valid || reach * reach < g2 || !is_point;
            // the workgroup's verdict through four words of s_gbo (idle until the row table is built, two barriers from here)
            // failing rows: `near` — the list can be rebuilt from the 27-cell stencil, here or by the tiled scan — and `far` —
            // the row's bound lies beyond the first pass's radius: nn_wide_kernel's
            const unsigned long long failing = __ballot(!ok && near), failing_far = __ballot(!ok && !near);  // (every lane votes)
            if (lane == 0) s_gbo[wave] = __popcll(failing) | (__popcll(failing_far) << 16);
            lds_barrier();
            const int w0 = s_gbo[0], w1 = s_gbo[1], w2 = s_gbo[2], w3 = s_gbo[3];
            const int f0 = w0 & 0xFFFF, f1 = w1 & 0xFFFF, f2 = w2 & 0xFFFF, f3 = w3 & 0xFFFF;
            const int n_near = f0 + f1 + f2 + f3, n_far = (w0 >> 16) + (w1 >> 16) + (w2 >> 16) + (w3 >> 16);
            const int n_fail = n_near + n_far;
            lds_barrier();  // (everybody has read the verdict: the list path writes its winners over these words)
            bool all_ok = n_fail == 0;
            // (diagnostic: how many rows fail where any does — 1, 2-4, 5-16, more)
            if (tid == 0 && n_fail > 0 && vv.rebuilt != nullptr) atomicAdd(vv.rebuilt + 12 + (n_fail > 16 ? 3 : n_fail > 4 ? 2 : n_fail > 1 ? 1 : 0), 1u);
            bool answers = true;  // this lane's row is answered from its list
            constexpr int kRows = verlet_rows_in_kernel(M, VERLET_K);
            // A FEW failing rows: the near ones are rebuilt here, one row per wave (verlet_rebuild_row), and answered with
            // the others; the far ones go to nn_wide_kernel's list (where there is one: two-pass and multi-level searches,
            // every launch that folds nothing in), which answers them and leaves them fresh lists.  More than a few: the
            // tiled search, which renews all 256 lists at once (a workgroup's rows age together; tools/exp_verlet_rows.py).
            // (K23 folded in and no rows rebuilt in the kernel: none of this is compiled — the verdict is all or nothing)
            bool few = false;
            if constexpr (FTM == -2 || kRows > 0) {
                const bool far_ok = n_far == 0 || (FTM == -2 && un.list != nullptr && n_far <= PPCR_VERLET_LIST_ROWS);
                const bool near_ok = n_near == 0 || n_near <= kRows || (FTM == -2 && un.list != nullptr && n_fail <= PPCR_VERLET_LIST_ROWS);
                few = !all_ok && far_ok && near_ok;
            }
            if (few) {
                bool to_wide = !ok && !near;  // this lane's row goes to nn_wide_kernel
                if constexpr (kRows > 0) {
                    if (n_near > 0 && n_near <= kRows) {
                        // the failing rows, in row order, into LDS; each wave rebuilds every fourth of them
                        VerletRowLds<kRows> &rl = *reinterpret_cast<VerletRowLds<kRows> *>(s_all);
                        static_assert(sizeof(VerletRowLds<kRows>) <= (size_t)Lds::kOffGbo, "the row-rebuild scratch sits below the verdict words");
                        const int slot = (wave > 0 ? f0 : 0) + (wave > 1 ? f1 : 0) + (wave > 2 ? f2 : 0) +
                                         __builtin_amdgcn_mbcnt_hi((unsigned)(failing >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)failing, 0u));
                        const bool mine = !ok && near;
                        if (mine) {
                            rl.q[slot] = make_float4(q.x, q.y, q.z, need);
                            rl.row[slot] = i;
                        }
                        lds_barrier();
                        for (int f = wave; f < n_near; f += kWaves) {
                            const float4 fq = rl.q[f];
                            const int fi = __builtin_amdgcn_readfirstlane(rl.row[f]);
                            const float fg = verlet_rebuild_row<verlet_slots(M)>(fi, make_float4(fq.x, fq.y, fq.z, 0.f), fq.w, ns, tgt0, cell_start0, g0, vv, lane,
                                                                                 rl.pos[wave], rl.d2[wave]);
                            if (lane == 0) rl.g2[f] = fg;
                        }
                        __syncthreads();  // (the lists the other waves wrote: global memory, read back by the list path below)
                        if (mine) {
                            g2 = rl.g2[slot];
                            acc = 0.f;
                        }
                        __syncthreads();  // (everybody has its g2: the list path may write over this memory)
                    } else {
                        to_wide = !ok;  // (more near rows than are rebuilt here, few enough for nn_wide_kernel)
                    }
                } else {
                    to_wide = !ok;
                }
                if constexpr (FTM == -2) {
                    if (un.list != nullptr && (n_far > 0 || (n_near > 0 && !(kRows > 0 && n_near <= kRows)))) {
                        answers = !to_wide;
                        if (valid && to_wide) cnt[i] = -1;
                        list_rows_of(un, valid && to_wide, i, s_gbo);
                    }
                }
                if (tid == 0 && vv.rebuilt != nullptr) {
                    atomicAdd(vv.rebuilt + 1, (unsigned)n_fail);
                    atomicAdd(vv.rebuilt + 2, 1u);
                }
                all_ok = true;
            }
            if (all_ok) {
                if (valid) vv.vacc[i] = acc;
                if constexpr (FTM == -2)
                    if (tid == 0 && vv.streak != nullptr) {
                        const unsigned st = vv.streak[wg];
                        if (st != 0) vv.streak[wg] = (unsigned char)(st > 2u ? st - 2u : 0u);
                    }
                // (the association's own threshold: the bound where there is one, strictly inside the full radius)
                const unsigned thr_row = need2_full < r2_far ? __float_as_uint(need2_full) : __float_as_uint(r2_far) - 1u;
                verlet_answer_rows<M, FTM>(tid, i, valid && answers, q, ns, tgt0, thr_row, m, nbr, cnt, dm2, fm, vv, s_all, wg, g2, acc, moved,
                                           __builtin_amdgcn_sqrtf(r2_far));
                return;
            }
            if (tid == 0 && vv.rebuilt != nullptr) {  // (diagnostic: workgroups that searched again)
                atomicAdd(vv.rebuilt, 1u);
                atomicAdd(vv.searched_now, 1u);
            }
            if (FTM == -2 && vv.streak != nullptr) {  // (uniform: one byte per workgroup, read by everybody, written by one)
                const unsigned st = vv.streak[wg];
                plain_search = st >= 8u && (wg & 15u) != vv.launch_tag;
                __builtin_amdgcn_s_barrier();  // (everybody has read it)
                if (tid == 0) vv.streak[wg] = (unsigned char)min(st + 4u, 16u);
            }
        }
    }
    // ---- MULTI: which level of the grid this block searches ------------------------------------------------------
    // The finest level whose stencil covers the block's largest cut-off radius (a row without a cut-off — it found fewer
    // than m within the full radius last time, or there is no last time — needs the full radius).  The first association
    // of a registration has no cut-offs at all: every block searches the base level with its first-pass radius, and the
    // rows that come back short go to nn_wide_kernel, as in the single-level two-pass search.
    GridDesc g_l;
    const float4 *tgt_l = nullptr;
    const int *cell_start_l = nullptr, *to_base = nullptr;
    float r2_l = 0.f;
    int level = 0;
    if constexpr (MULTI) {
        level = un.base_level;
        if (dm2_valid) {
            // The finest level that covers the cut-offs of at least 15/16 of the block's rows (7/8 until the rows left to
            // nn_wide_kernel came with their bound: 12/16, 13/16, 14/16, 15/16 — LiDAR-like scene 4.85 / 4.94 / 5.08 / 5.14 k
            // it/s, slab 5.09 / 5.17 / 5.29 / 5.34 k) (a row whose cut-off reaches
            // beyond the level searches the level's radius all the same: it is exact when it finds m there, and goes to
            // nn_wide_kernel when it does not).  Going by the block's LARGEST cut-off was measured first: one row at the
            // cloud's edge then drags its whole block to a level whose halo no tile holds (uniform 200k cloud at radius 3:
            // 22 % of the rows sit in such blocks).  Each lane finds the finest level that covers its own cut-off, the
            // waves count their lanes per level, the block adds the counts up.
            // (VERLET: by the reach of the row's list — the level that covers it builds it)
            const float need_f = (VERLET && !plain_search) ? bnd_v * bnd_v : __uint_as_float(min(thr0, __float_as_uint(un.r2_full)));  // no cut-off: the full radius
            int mine = un.n_levels - 1;
#pragma unroll
            for (int l = kMaxLevels - 2; l >= 0; l--)
                if (l < un.n_levels - 1 && un.r2_cap[l] >= need_f) mine = l;
            int below[kMaxLevels];  // lanes of this wave whose cut-off level l covers
#pragma unroll
            for (int l = 0; l < kMaxLevels; l++) below[l] = __popcll(__ballot(valid && mine <= l));
            if (lane == 0) {
#pragma unroll
                for (int l = 0; l < kMaxLevels; l++) s_need[wave * 16 + l] = below[l];
            }
            lds_barrier();
            const int n_rows = s_need[un.n_levels - 1] + s_need[16 + un.n_levels - 1] + s_need[32 + un.n_levels - 1] + s_need[48 + un.n_levels - 1];
            level = un.n_levels - 1;
            for (int l = un.n_levels - 2; l >= 0; l--)
                if (16 * (s_need[l] + s_need[16 + l] + s_need[32 + l] + s_need[48 + l]) >= PPCR_LEVEL_Q16 * n_rows) level = l;
            // ... but never a level at which this block's halo has outgrown the tile before (the feedback word, see below)
            // cap | floor << 4 | split << 8 (a block that met both keeps the floor)
            level = max(min(level, (int)(level_fb & 15u)), (int)((level_fb >> 4) & 15u));
        }
        level = __builtin_amdgcn_readfirstlane(level);
        const GridLevel *lv = un.levels + level;
        g_l = lv->g;
        tgt_l = lv->tgt, cell_start_l = lv->cell_start, to_base = lv->to_base;
        // the first association searches the base level with the first-pass radius (r2_0 <= the level's cap)
        r2_l = dm2_valid ? fminf(lv->r2_cap, un.r2_full) : r2_0;
    }
    const GridDesc &g = MULTI ? g_l : g0;
    const float4 *__restrict__ const tgt = MULTI ? tgt_l : tgt0;
    const int *__restrict__ const cell_start = MULTI ? cell_start_l : cell_start0;
    const float r2 = MULTI ? r2_l : r2_0;
    if constexpr (VERLET) {
        // the scan builds the lists of the rows whose reach its stencil covers (cells of at least the (first-pass / level's)
        // radius + 2 skin: in a one-pass search every row); the others are scanned as the plain search would — their own
        // cut-off, no margin — and get their lists from nn_wide_kernel when it searches them
        scan_lists = (FTM != -2) ? true : (bnd_v <= g.h && !plain_search);
        thr_v = scan_lists ? __float_as_uint(bnd_v * bnd_v) : min(thr0, __float_as_uint(r2));
    }
    const QueryCells qc = query_cells(q, g);

    // ---- the nine stencil runs [rb, re) in sorted-target positions, clipped in x --------------------------------
    // A target of row (dy, dz) is at least (gy, gz) away in y and z (gap between the query and that row's slab,
    // under-estimated by g.eps), so it can only be within the cut-off radius R if |dx| <= sqrt(R^2 - gy^2 - gz^2);
    // R^2 is the radius or the temporal cut-off, inflated by 4e-6 for the float rounding of d2.  The window is taken
    // in SLICE units: targets were binned by floor((x - org) * inv_hx), a monotone map, so every in-window target has
    // its slice in [floor(ux - ws), floor(ux + ws)] up to the rounding of ux, ws and the root (a few ulp of the
    // largest slice coordinate), which 2 * g.eps (64 ulp of the cloud's extent) covers several times over.
    // Loads are unconditional: a dead run reads cell_start[0] twice (= 0, 0: empty).
    const int x0 = max(qc.cx - g.xr, 0), x1 = min(qc.cx + g.xr, g.n[0] - 1);
    int rb[9], re[9];
    // the part of the grid this query's LIVE runs touch: rows [ylo, yhi] x [zlo, zhi], slices [xlo, xhi].  The
    // workgroup's halo is the union of these boxes — tighter than "cell bounding box +- 1": a query that has drifted a
    // little way into a cell does not need the row beyond it.
    int xlo = INT_MAX, xhi = INT_MIN, ylo = INT_MAX, yhi = INT_MIN, zlo = INT_MAX, zhi = INT_MIN;
    {
        const float R2 = __uint_as_float(VERLET ? thr_v : min(thr0, __float_as_uint(r2))) * 1.000004f;
        const float ux = (q.x - g.org[0]) * g.inv_hx;
        const float fy = q.y - g.org[1], fz = q.z - g.org[2];
        const float gy0 = fmaxf(fy - (float)qc.cy * g.h - g.eps, 0.f), gy2 = fmaxf((float)(qc.cy + 1) * g.h - fy - g.eps, 0.f);
        const float gz0 = fmaxf(fz - (float)qc.cz * g.h - g.eps, 0.f), gz2 = fmaxf((float)(qc.cz + 1) * g.h - fz - g.eps, 0.f);
        const float gy_sq[3] = {gy0 * gy0, 0.f, gy2 * gy2}, gz_sq[3] = {gz0 * gz0, 0.f, gz2 * gz2};
        const float k_s = g.inv_hx * 1.000001f, eps_s = 2.0f * g.eps * g.inv_hx;
        const bool x_ok = valid & (x0 <= x1);
        const bool oky[3] = {bool(x_ok & ((unsigned)(qc.cy - 1) < (unsigned)g.n[1])), bool(x_ok & ((unsigned)qc.cy < (unsigned)g.n[1])),
                             bool(x_ok & ((unsigned)(qc.cy + 1) < (unsigned)g.n[1]))};
        const bool okz[3] = {(unsigned)(qc.cz - 1) < (unsigned)g.n[2], (unsigned)qc.cz < (unsigned)g.n[2],
                             (unsigned)(qc.cz + 1) < (unsigned)g.n[2]};
        bool live[9];
        int cfa = 0, cfb = -1;
        int base_c = (qc.cz * g.n[1] + qc.cy) * g.n[0];
        asm volatile("" : "+v"(base_c));  // keep the nine row bases as base_c + uniform offset (not nine multiplies)
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int dz = k / 3 - 1, dy = k % 3 - 1;
            const float w2 = R2 - (gy_sq[k % 3] + gz_sq[k / 3]);
            const float ws = __builtin_fmaf(__builtin_amdgcn_sqrtf(fmaxf(w2, 0.f)), k_s, eps_s);
            const int fa = max((int)floorf(ux - ws), x0), fb = min((int)floorf(ux + ws), x1);
            const bool in = bool(oky[k % 3] & okz[k / 3]) & bool((w2 >= 0.f) & (fa <= fb));  // no short circuit: no branches
            const int row_base = base_c + (dz * g.n[1] + dy) * g.n[0];  // uniform offset from the centre row
            rb[k] = cell_start[(unsigned)(in ? row_base + fa : 0)];
            re[k] = cell_start[(unsigned)(in ? row_base + fb + 1 : 0)];
            live[k] = in;
            if (k == 4) cfa = fa, cfb = fb;
        }
        // box of the live runs.  x: the centre run has the widest window (its w2 is the largest), so [cfa, cfb] covers
        // every live run's slices (it is computed whether or not the centre row itself is inside the grid).
        const bool ym = live[0] | live[3] | live[6], y0 = live[1] | live[4] | live[7], yp = live[2] | live[5] | live[8];
        const bool zm = live[0] | live[1] | live[2], z0 = live[3] | live[4] | live[5], zp = live[6] | live[7] | live[8];
        if (ym | y0 | yp) {
            xlo = cfa, xhi = cfb;
            ylo = qc.cy + (ym ? -1 : (y0 ? 0 : 1));
            yhi = qc.cy + (yp ? 1 : (y0 ? 0 : -1));
            zlo = qc.cz + (zm ? -1 : (z0 ? 0 : 1));
            zhi = qc.cz + (zp ? 1 : (z0 ? 0 : -1));
        }
    }

    // ---- per-wave union of the queries' boxes -> LDS ------------------------------------------------------------
    {
        const int lx = wave_reduce(xlo, INT_MAX, OpMin()), hx = wave_reduce(xhi, INT_MIN, OpMax());
        const int ly = wave_reduce(ylo, INT_MAX, OpMin()), hy = wave_reduce(yhi, INT_MIN, OpMax());
        const int lz = wave_reduce(zlo, INT_MAX, OpMin()), hz = wave_reduce(zhi, INT_MIN, OpMax());
        if (lane == 0) {
            s_box[wave][0] = lx, s_box[wave][1] = ly, s_box[wave][2] = lz;
            s_box[wave][3] = hx, s_box[wave][4] = hy, s_box[wave][5] = hz;
        }
    }
    lds_barrier();
    stamp(0);

    // ---- halo box and row table (every wave builds it for itself: no barrier before the staging) -----------------
    int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
#pragma unroll
    for (int w = 0; w < kWaves; w++)
#pragma unroll
        for (int a = 0; a < 3; a++) {
            lo[a] = min(lo[a], s_box[w][a]);
            hi[a] = max(hi[a], s_box[w][3 + a]);
        }
    // live runs only name rows and slices inside the grid; a workgroup without any live run has an empty box
    const bool any_live = lo[0] <= hi[0];
    const int hx0 = any_live ? lo[0] : 0, hx1 = any_live ? hi[0] : 0;
    const int hy0 = any_live ? lo[1] : 0, hy1 = any_live ? hi[1] : 0;
    const int hz0 = any_live ? lo[2] : 0, hz1 = any_live ? hi[2] : 0;
    const int ny_h = hy1 - hy0 + 1, nz_h = hz1 - hz0 + 1;
    // row slot = rz << ys | ry: 8 (y) x 16 (z) slots, or 16 x 8 for the blocks that straddle two columns of the source
    // order in y (a block that straddles in z as well, one in a few hundred, goes to the cleanup kernel)
    // MULTI: any box of at most 128 rows, slot = rz * ny_h + ry (the blocks of surfaces — a wall's 9 x 9 rows, a ground
    // plane's 18 x 3 — fit neither power-of-two shape): the lane -> (ry, rz) split is a multiplication by a reciprocal
    // that is exact for slots below 128 (slot * ny_h < 2^16), the run bases a multiplication by the uniform ny_h.
    const int ys = (ny_h <= 8) ? 3 : 4;
    const int yw = MULTI ? max(ny_h, 1) : (1 << ys);  // slots per z step
    const bool shape_ok = MULTI ? (hx0 <= hx1 && ny_h >= 1 && nz_h >= 1 && ny_h <= kRows && ny_h * nz_h <= kRows)
                                : (hx0 <= hx1 && ny_h >= 1 && nz_h >= 1 && ny_h <= 16 && nz_h <= (kRows >> ys));
    // lane l owns halo row slots l (A) and l + 64 (B): global begin, length; LDS offsets from a scan of A + B
    int gbA, lenA, gbB, lenB;
    {
        const int ymask = (1 << ys) - 1;
        int ryA = lane & ymask, rzA = lane >> ys, ryB = (lane + 64) & ymask, rzB = (lane + 64) >> ys;
        if constexpr (MULTI) {
            const unsigned magic = 65536u / (unsigned)yw + 1u;  // floor(s / yw) = (s * magic) >> 16 for s < 128, yw <= 128
            rzA = (int)(((unsigned)lane * magic) >> 16), rzB = (int)(((unsigned)(lane + 64) * magic) >> 16);
            ryA = lane - rzA * yw, ryB = lane + 64 - rzB * yw;
        }
        const bool hasA = shape_ok && ryA < ny_h && rzA < nz_h, hasB = shape_ok && ryB < ny_h && rzB < nz_h;
        const int baseA = ((hz0 + rzA) * g.n[1] + hy0 + ryA) * g.n[0], baseB = ((hz0 + rzB) * g.n[1] + hy0 + ryB) * g.n[0];
        gbA = cell_start[(unsigned)(hasA ? baseA + hx0 : 0)];
        lenA = cell_start[(unsigned)(hasA ? baseA + hx1 + 1 : 0)] - gbA;
        gbB = cell_start[(unsigned)(hasB ? baseB + hx0 : 0)];
        lenB = cell_start[(unsigned)(hasB ? baseB + hx1 + 1 : 0)] - gbB;
    }
    const int incl = wave_scan(lenA + lenB, 0, OpAdd());
    const int exclA = incl - lenA - lenB, exclB = exclA + lenA;
    const int total = __builtin_amdgcn_readlane(incl, 63);
    stamp(1);
    if constexpr (STAMPS) {  // diagnostic: slot 6 = staged candidates, slot 7 = (ny_h << 8) | nz_h of this block's halo
        t_acc[6] = (unsigned long long)total;
        t_acc[7] = (unsigned long long)((ny_h << 8) | nz_h);
    }
    const bool handed_over = !shape_ok || total > CAP;  // uniform: derived from the shared boxes and cell_start only
    if constexpr (MULTI) {
        // the block's feedback for the NEXT launch (UnansweredRows::level_out): what it was, or — after a hand-over —
        // a halo too LARGE for the tile: a finer level next time (more short rows, but a halo that fits), first the same
        // level on two workgroups; a halo of too many ROWS (a surface seen at too fine a level): a coarser one
        if (tid == 0 && un.level_out != nullptr) {
            unsigned cap = level_fb & 15u, floor_ = (level_fb >> 4) & 15u, split_ = level_fb & 0x100u;
            if (handed_over) {
                if (!shape_ok) floor_ = (unsigned)min(level + 1, un.n_levels - 1);
                else if (!split_ && split.all_halves) split_ = 0x100u;
                else cap = (unsigned)max(level - 1, 0);
            }
            const unsigned short word = (unsigned short)(cap | (floor_ << 4) | split_);
            if (half == 0) un.level_out[2 * bid] = word, un.level_out[2 * bid + 1] = word;
            else un.level_out[2 * bid + (half == 2 ? 1 : 0)] = word;
        }
    }
    if (tid == 0 && half == 0 && split.flag != nullptr && (handed_over || total > split.presplit) && !split.flag[bid]) {
        // once the fold-and-solve step has rebuilt the list from the flags (the kMaxSplit registered blocks with the
        // smallest ids are split, flag 2: which ones does not depend on the order the registrations arrived in) this
        // block is scanned in two halves
        split.flag[bid] = 1;
        atomicAdd(split.total, 1u);  // (only says that there is something new)
    }
    // the workgroup's unanswered rows go to the list in one piece: ONE atomic per workgroup that has any (every thread
    // calls; two barriers; the counts travel through six LDS words nobody else touches at that moment — `words`: the box
    // words at the kernel's end, idle since the halo was staged; the row-offset table for a workgroup that hands its block
    // over before that table is written, while slower waves may still be reading the boxes).  One atomic per WAVE
    // was 3 000 of them on one counter within a few microseconds at the end of a one-round launch (a 200k-point cloud: every
    // workgroup reaches its end at about the same time), and same-address atomics are served one by one.
    auto list_rows = [&](bool mine, int *words) { list_rows_of(un, mine, i, words); };
    if (handed_over) {
        // the cleanup kernel redoes this workgroup's queries (this launch): entry = grid index * 4 + half — or, when the
        // unanswered rows are listed, nn_wide_kernel searches them (marked unsearched)
        if (tid == 0) ovf_list[atomicAdd(ovf_count, 1u)] = (int)wg * 4 + half;
        if constexpr (VERLET) {
            if (valid) vv.vg2[i] = 0.f;  // (whoever redoes these rows builds no lists: they come back here next time)
            if (tid == 0) verlet_file_slot(vv, wg, true);
        }
        if constexpr (MULTI) {
            if (tid == 0 && un.level_dbg != nullptr) {
                atomicAdd(un.level_dbg + level * kLevelDbgWords + 0, 1u);
                atomicAdd(un.level_dbg + level * kLevelDbgWords + (shape_ok ? 2 : 1), 1u);
            }
        }
        if (valid) cnt[i] = -1;
        if constexpr (FTM == -2)
            if (valid) dm2[i] = bound_bits;  // (see bound_bits: what nn_wide_kernel starts from)
        if constexpr (FTM == -2)  // (a launch that folds K23 in never lists: the cleanup role redoes its hand-overs)
            if (un.list != nullptr) list_rows(valid, s_gbo);
        flush_stamps();
        return;
    }
    // sorted-target position of a staged point = its LDS index + gbo[row slot]
    if (wave == 0) {
        s_gbo[lane] = gbA - exclA;
        s_gbo[lane + 64] = gbB - exclB;
    }

    // ---- stage the halo: the non-empty rows are dealt round-robin to the four waves, kStageUnroll rows in flight --
    {
        // compact the non-empty rows into s_rowtab (every wave writes the same values; each reads back its own writes)
        const unsigned long long neA = __ballot(lenA > 0), neB = __ballot(lenB > 0);
        const int nA = __popcll(neA), nrows = nA + __popcll(neB);
        const int rankA = __builtin_amdgcn_mbcnt_hi((unsigned)(neA >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)neA, 0u));
        const int rankB = nA + __builtin_amdgcn_mbcnt_hi((unsigned)(neB >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)neB, 0u));
        if (lenA > 0) s_rowtab[rankA] = make_int2(gbA, (exclA << 19) | (lenA << 7) | lane);          // 12 + 12 + 7 bits
        if (lenB > 0) s_rowtab[rankB] = make_int2(gbB, (exclB << 19) | (lenB << 7) | (lane + 64));
        for (int j0 = 0; wave + kWaves * j0 < nrows; j0 += kStageUnroll) {
            float cx[kStageUnroll], cy[kStageUnroll], cz[kStageUnroll];  // 12 of a record's 16 bytes: six rows in flight (four: one more round trip; eight: no better)
            int so[kStageUnroll], sl[kStageUnroll], sg[kStageUnroll], sr[kStageUnroll];
#pragma unroll
            for (int u = 0; u < kStageUnroll; u++) {
                const int t = wave + kWaves * (j0 + u);
                const int2 row = s_rowtab[min(t, kRows - 1)];  // uniform address: one broadcast read
                const int pk = __builtin_amdgcn_readfirstlane(row.y);
                sg[u] = __builtin_amdgcn_readfirstlane(row.x);
                so[u] = (int)((unsigned)pk >> 19);
                sl[u] = (t < nrows) ? ((pk >> 7) & 0xFFF) : 0;
                sr[u] = pk & 127;
                const float *g = reinterpret_cast<const float *>(tgt + ((lane < sl[u]) ? sg[u] + lane : 0));  // slot 0 always exists
                cx[u] = g[0], cy[u] = g[1], cz[u] = g[2];
            }
#pragma unroll
            for (int u = 0; u < kStageUnroll; u++) {
                if (lane < sl[u]) {
                    const int d = so[u] + lane;
                    s_x[d] = cx[u];
                    s_y[d] = cy[u];
                    s_z[d] = cz[u];
                    s_rowid[d] = (unsigned char)sr[u];
                }
                // rows longer than a wave (dense data): four more loads in flight per trip (one at a time each 64 points
                // cost a memory round trip of their own)
                for (int k0 = 64; k0 < sl[u]; k0 += 256) {
                    float tx[4], ty[4], tz[4];
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        const float *gp = reinterpret_cast<const float *>(tgt + sg[u] + min(k0 + 64 * v + lane, sl[u] - 1));
                        tx[v] = gp[0], ty[v] = gp[1], tz[v] = gp[2];
                    }
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        const int k = k0 + 64 * v + lane;
                        if (k < sl[u]) {
                            const int d = so[u] + k;
                            s_x[d] = tx[v];
                            s_y[d] = ty[v];
                            s_z[d] = tz[v];
                            s_rowid[d] = (unsigned char)sr[u];
                        }
                    }
                }
            }
        }
    }
    lds_barrier();
    stamp(2);

    int n = 0;
    unsigned tm = 0xFFFFFFFFu;  // d2 bits of the m-th neighbour (all-ones: fewer than m found)
    float vg_new = 0.f;         // VERLET: the reach (as d2) of the list this launch built for the row
    // A lane's column of the list.  With column = tid the lanes 2k and 2k + 1 of a wave share a 32-bit LDS word, and in
    // the scan every lane writes at its own cursor (another slot = another address in the same bank): the two u16 stores
    // of the pair collide whenever their cursors differ.  The columns of a wave are therefore dealt so that a word is
    // shared by lanes l and l + 32 — the two halves of the wave the LDS serves in different cycles.
#if PPCR_LIST_PERM
    const int lcol = (tid & ~63) | ((tid & 31) << 1) | ((tid >> 5) & 1);
#else
    const int lcol = tid;
#endif
    const HaloList L{reinterpret_cast<const char *>(s_x), CAP * 4, s_rowid, reinterpret_cast<const char *>(s_gbo), s_list + lcol};
    if (valid) {
        // the nine runs as ONE 32-bit key each, (length << 16) | LDS byte offset of the run's first candidate, sorted
        // by DESCENDING length: every lane of the wave walks its longest run first, ... — a run's trip count is the
        // maximum over the 64 lanes, and the maxima of order statistics add up to far fewer steps than the maxima of
        // arbitrary runs.  25-comparator network (0/1 principle), a comparator is a v_max_u32 / v_min_u32 pair.
        unsigned key[9];
        {
            const char *gbo_c = reinterpret_cast<const char *>(s_gbo) + 4 * ((qc.cz - hz0) * yw + (qc.cy - hy0));
#pragma unroll
            for (int k = 0; k < 9; k++) {
                const int rl = re[k] - rb[k];
                const int start = rb[k] - *reinterpret_cast<const int *>(gbo_c + 4 * ((k / 3 - 1) * yw + (k % 3 - 1)));
                key[k] = rl > 0 ? ((unsigned)rl << 16) | (unsigned)(start << 2) : 0u;
            }
            constexpr int net[25][2] = {{0, 3}, {1, 7}, {2, 5}, {4, 8}, {0, 7}, {2, 4}, {3, 8}, {5, 6}, {0, 2},
                                        {1, 3}, {4, 5}, {7, 8}, {1, 4}, {3, 6}, {5, 7}, {0, 1}, {2, 4}, {3, 5},
                                        {6, 8}, {2, 3}, {4, 5}, {6, 7}, {1, 2}, {3, 4}, {5, 6}};
            // (VERLET: the runs stay in row order — ascending positions in the sorted target — so that every row's list, and
            //  with it the association, is sorted by position: lanes next to each other then gather their k-th entries from
            //  the same few cache lines in nn_verify_kernel and K23.  Only (re)building launches pay the less even walk.)
            if constexpr (!PPCR_VERLET_ROW_ORDER || !VERLET) {
#pragma unroll
                for (int c = 0; c < 25; c++) {
                    const int a = net[c][0], b = net[c][1];
                    const unsigned kh = max(key[a], key[b]), kl = min(key[a], key[b]);  // descending
                    key[a] = kh;
                    key[b] = kl;
                }
            }
        }
        if constexpr (STAMPS) {  // diagnostic: this lane's nine sorted run lengths, 7 bits each, behind the wave records
            if (stamps) {
                unsigned long long pk = 0;
#pragma unroll
                for (int k = 0; k < 9; k++) pk |= (unsigned long long)min(key[k] >> 16, 127u) << (7 * k);
                stamps[((size_t)gridDim.x * kWaves + 64) * 8 + (size_t)wg * BLOCK + tid] = pk;
            }
        }
        // d2 >= +0 and r2 > 0, so "d2 < r2" is "bits(d2) <= bits(r2) - 1" (a NaN d2 has larger bits and fails): the
        // radius test and the running cut-off are ONE unsigned compare per candidate
        const unsigned thr_a = min(thr0, __float_as_uint(r2) - 1u);  // the association's own threshold
        unsigned thr = (VERLET && scan_lists) ? thr_v : thr_a;       // what the scan accepts
        typedef float v2f __attribute__((ext_vector_type(2)));
        const v2f qx2 = {q.x, q.x}, qy2 = {q.y, q.y}, qz2 = {q.z, q.z};
        // LDS addresses as plain 32-bit integers (address space 3): the write cursor and the candidate cursor are
        // one VGPR each and a candidate costs v_min + ds_write + v_cndmask + v_add, accepted or not
        typedef __attribute__((address_space(3))) unsigned short *lds_u16p;
        typedef __attribute__((address_space(3))) const float *lds_f32p;
        // (the casts go through uintptr_t so that the host pass, where every pointer is 64-bit, parses them too)
        const unsigned list0 = (unsigned)(__UINTPTR_TYPE__)(lds_u16p)(s_list + lcol), list_last = list0 + C * 512;
        const unsigned halo0 = (unsigned)(__UINTPTR_TYPE__)(lds_f32p)s_x;
        bool no_room = false;
        for (int attempt = 0;; attempt++) {
            // the list's write cursor counts every accepted candidate (so n is exact), the store slot is clamped to
            // the spare slot C: an overflowing lane keeps its first C entries
            unsigned wp = list0;
#pragma unroll
            for (int k = 0; k < 9; k++) {
                if (key[k] > 0xFFFFu) {
                    const unsigned a0 = halo0 + (key[k] & 0xFFFFu), a_end = a0 + 4 * (key[k] >> 16), a_pair = a_end - 4;
                    // two candidates per trip from the run's true start (4-byte aligned reads), packed f32
                    // sub / mul / add (no FMA: the same IEEE operations per element as dist2_flann)
                    for (unsigned a = a0; a < a_end; a += 8) {
                        const lds_f32p px = (lds_f32p)(__UINTPTR_TYPE__)a, py = (lds_f32p)(__UINTPTR_TYPE__)(a + CAP * 4),
                                       pz = (lds_f32p)(__UINTPTR_TYPE__)(a + CAP * 8);
                        const v2f cx = {px[0], px[1]}, cy = {py[0], py[1]}, cz = {pz[0], pz[1]};
                        const v2f dx = qx2 - cx, dy = qy2 - cy, dz = qz2 - cz;
                        v2f d = dx * dx;
                        d = d + dy * dy;
                        d = d + dz * dz;
                        // every candidate is written to the list's next free slot; only an accepted one moves the cursor
                        // (a rejected one is overwritten by whatever comes next): no branch, no exec juggling
#if PPCR_LIST_NOCLAMP
                        (void)list_last;
                        const unsigned slot_x = wp;
#else
                        const unsigned slot_x = min(wp, list_last);
#endif
                        *(lds_u16p)(__UINTPTR_TYPE__)slot_x = (unsigned short)(a - halo0);
                        wp += (__float_as_uint(d.x) <= thr) ? 512u : 0u;
#if PPCR_LIST_NOCLAMP
                        const unsigned slot_y = wp;
#else
                        const unsigned slot_y = min(wp, list_last);
#endif
                        *(lds_u16p)(__UINTPTR_TYPE__)slot_y = (unsigned short)(a - halo0 + 4);
                        wp += (bool(a < a_pair) & bool(__float_as_uint(d.y) <= thr)) ? 512u : 0u;
                    }
                }
            }
            n = (int)((wp - list0) >> 9);
            if (n <= C) break;
            if constexpr (VERLET) {
                // more than C targets within the list's reach (one row in 10^5 at the benchmark's density): a quarter of the
                // skin first — a list with less room, rebuilt sooner, but not one without any room, which would send its
                // workgroup through the search in every iteration
                if (PPCR_VERLET_RETRY && attempt == 0 && scan_lists) {
                    const float b = need + 0.25f * (bnd_v - need);
                    thr = __float_as_uint(b * b);
                    continue;
                }
            }
            if (attempt == ((VERLET && PPCR_VERLET_RETRY) ? 2 : 1)) {  // more than C candidates tie at the threshold: leave the block to the general flavour
                n = -1;
                break;
            }
            // list overflow (dense neighbourhood, or no usable cut-off yet): the C entries that were kept are genuine
            // in-radius candidates, so the m-th smallest of them bounds the final m-th distance: scan again
            (void)select_top_m<M>(L, tgt, q, C, m, thr);
            if constexpr (VERLET) {
                thr = min(thr, thr_a);  // (what is left of such a row's scan ends at its cut-off: nothing a list could live on)
                no_room = true;
            }
        }
        if constexpr (VERLET) {
            // The row's Verlet list: what the scan accepted (every target whose d2 bits are <= thr; the scan's lists are half
            // as long again as a Verlet list), cut back to the list's slots where there are more.
            constexpr int CVs = verlet_slots(M);
            bool listed = scan_lists && n >= 0 && !no_room;  // (no_room: more targets in reach than the scan's list holds, twice: no list)
            if (listed && n > CVs) {
                if constexpr (M <= 12) {
                    // the sixteen NEAREST stay; the list is complete below the farthest of them (for the typical row the 16th
                    // neighbour's distance against the 10th's: 0.14 radii of room at the benchmark's density).  Cutting by reach
                    // instead — the largest of need + skin / 2, / 4, / 8 that holds at most sixteen: a quarter of the selection's
                    // instructions — was measured: the lists' room shrinks, a fifth more workgroups search per iteration and the
                    // timed windows lose 10 % (11.47 k -> 10.32 k it/s at 1M).
                    unsigned t_far = 0;
                    n = select_top_m<CVs>(L, tgt, q, n, CVs, t_far);
                    thr = t_far > 0u ? t_far - 1u : 0u;  // (strictly below the farthest kept: equal distances beyond it were dropped)
                }
            }
            if constexpr (M > 12) {
                // wide lists (32 slots of a 36-slot scan list): the few FARTHEST entries leave, one per pass — no 32-register
                // selection network in this kernel — and the list is complete strictly below the nearest of those that left
                int drop = (listed && n > CVs) ? n - CVs : 0;
                unsigned gone = 0xFFFFFFFFu;  // smallest d2 among the entries dropped
                while (__ballot(drop > 0) != 0ull) {
                    if (drop > 0) {
                        unsigned far = 0;
                        int at = 0, ties = 0;
                        for_each_entry(L, q, n, [&](int t, int, unsigned b) {
                            ties = b == far ? ties + 1 : (b > far ? 1 : ties);
                            at = b >= far ? t : at;
                            far = max(far, b);
                        });
                        // (THE list is also what the association's row is selected from, by (d2, original index): of several
                        //  entries at the farthest distance the one with the LARGEST original index leaves — whichever went,
                        //  the Verlet list would be as good; the row would not: a lattice cloud's m-th neighbour ties with a
                        //  dozen others, and the soak under another seed found rows that kept the wrong one)
                        if (PPCR_CUT_TIES_BY_INDEX && ties > 1) {
                            unsigned worst = 0;
                            for_each_entry(L, q, n, [&](int t, int e, unsigned b) {
                                if (b == far) {
                                    const unsigned o = L.orig_of(e, tgt);
                                    at = o >= worst ? t : at;
                                    worst = max(worst, o);
                                }
                            });
                        }
                        L.store(at, L.load(n - 1));
                        n -= 1;
                        gone = min(gone, far);
                        drop -= 1;
                    }
                }
                if (gone != 0xFFFFFFFFu) thr = gone > 0u ? gone - 1u : 0u;
            }
            if (listed) {
                int *out = vv.vl + i;
                for (int j = 0; j < n; j++) {
                    int pos = L.pos_of(L.load(j));
                    if constexpr (MULTI)
                        if (to_base != nullptr) pos = to_base[pos];  // (lists hold base positions whatever level built them)
                    *out = pos;
                    out += ns;
                }
                for (int j = n; j < CVs; j++) {  // (every slot a valid position: the verification loads all of them)
                    *out = 0;
                    out += ns;
                }
                vv.vn[i] = (unsigned char)n;
                vv.vacc[i] = 0.f;
                vv.vmask[i] = 0xFFFFFFFFu;  // (the association's row is written from the scan's list below, not from list slots)
            }
            vg_new = listed ? __uint_as_float(thr) : 0.f;  // (0: no list — also the degenerate row whose nearest are all at distance 0)
            vv.vg2[i] = vg_new;
            // ... and what lies beyond the association's own threshold (radius, temporal cut-off) leaves the list now
            if (n > 0 && thr > thr_a) {
                int w = 0;
                for_each_entry(L, q, n, [&](int, int e, unsigned b) {
                    if (b <= thr_a) {
                        L.store(w, e);
                        w++;
                    }
                });
                n = w;
            }
        }
        stamp(3);
        // ---- selection: the list -> its m smallest by (d2, original index); tm = d2 bits of the m-th -----------------
        // With a valid cut-off nearly every list holds m or a few more candidates (the cut-off is the previous m-th
        // distance plus the query's own displacement).  Such a list does not need the two-pass threshold selection: ONE
        // pass finds its three largest d2 and the slots of the first two; a list of m takes the largest as its new
        // cut-off state, a list of m + 1 drops the slot of the largest (the last entry moves into it) and takes the
        // second, a list of m + 2 drops two and takes the third, a list of m + 3 goes round once more.  Candidates tying
        // at a dropped distance (the rule then asks for their original indices), or any lane of the wave with m + 4 or
        // more, take the general selection.
        const int surplus = n - m;  // < 0: the list is the answer, no cut-off state
        bool general = surplus >= 4;
        if (__ballot(general) == 0ull) {
            int left = surplus;
            for (bool first_pass = true;; first_pass = false) {
                const bool active = !general && (first_pass ? left >= 0 : left > 0);
                if (__ballot(active) == 0ull) break;
                if (active) {
                    unsigned f1 = 0, f2 = 0, f3 = 0;  // the three largest, descending
                    int s1 = 0, s2 = 0;               // slots of the first two
                    for_each_entry(L, q, n, [&](int t, int, unsigned b) {
                        const bool gt1 = b > f1, gt2 = b > f2;
                        s2 = gt1 ? s1 : (gt2 ? t : s2);
                        s1 = gt1 ? t : s1;
                        f3 = umed3(f2, f3, b);  // f2 >= f3: max(f3, min(f2, b))
                        f2 = umed3(f1, f2, b);  // f1 >= f2: max(f2, min(f1, b))
                        f1 = max(f1, b);
                    });
                    if (left == 0) {
                        tm = f1;
                    } else if (f1 == f2 || (left >= 2 && f2 == f3)) {
                        general = true;
                    } else {
                        L.store(s1, L.load(n - 1));
                        n -= 1;
                        tm = f2;
                        if (left >= 2) {
                            s2 = (s2 == n) ? s1 : s2;  // the second largest was the last entry: it has just moved
                            L.store(s2, L.load(n - 1));
                            n -= 1;
                            tm = f3;
                        }
                    }
                }
                left -= 2;
            }
        } else {
            general = surplus > 0;
            if (surplus == 0) {
                tm = 0;
                for_each_entry(L, q, n, [&](int, int, unsigned b) { tm = max(tm, b); });
            }
        }
        if (general) {
            n = select_top_m<M>(L, tgt, q, n, m, thr);
            tm = thr;
        }
        stamp(4);
    }
    // a wave with a twice-overflowed lane registers the block (once) for the cleanup kernel; its other results are
    // simply overwritten there with identical values
    const bool unanswered = n < 0;  // (marked unsearched, like the rows of a workgroup that bailed: see nn_wide_kernel)
    if (__ballot(unanswered) != 0ull) {
        if (lane == 0 && atomicExch(&s_bail, 1) == 0) ovf_list[atomicAdd(ovf_count, 1u)] = (int)wg * 4 + half;
        n = max(n, 0);
    }
    if (valid) {
        int *out = nbr + i;
        for (int j = 0; j < n; j++) {
            int pos = L.pos_of(L.load(j));
            if constexpr (MULTI)
                if (to_base != nullptr) pos = to_base[pos];  // the association is kept in the base level's positions
            *out = pos;
            out += ns;
        }
        // MULTI: a row that found fewer than m inside a level's radius is final only when that radius is the full one;
        // elsewhere it goes to nn_wide_kernel, marked unsearched (its count says nothing about the base level's radius)
        const bool short_here = MULTI && n < un.m_list && r2 < un.r2_full && to_base != nullptr;
        cnt[i] = (unanswered || short_here) ? -1 : n;
        // (a row that goes on to nn_wide_kernel takes the bound along, see bound_bits; that kernel writes the row's own)
        bool listed = false;
        if constexpr (FTM == -2) listed = un.list != nullptr && (unanswered || (n < un.m_list && (!MULTI || r2 < un.r2_full)));
        dm2[i] = listed ? bound_bits : tm;
    }
    if constexpr (FTM == -2)
        if (un.list != nullptr) list_rows(valid && (unanswered || (n < un.m_list && (!MULTI || r2 < un.r2_full))), &s_box[0][0]);
    if constexpr (MULTI) {
        if (un.level_dbg != nullptr) {
            const int n_short = __popcll(__ballot(valid && (unanswered || (n < un.m_list && r2 < un.r2_full))));
            const int n_valid = __popcll(__ballot(valid));
            if (lane == 0) {
                atomicAdd(un.level_dbg + level * kLevelDbgWords + 3, (unsigned)n_short);
                atomicAdd(un.level_dbg + level * kLevelDbgWords + 5, (unsigned)n_valid);
                if (wave == 0) {
                    atomicAdd(un.level_dbg + level * kLevelDbgWords + 0, 1u);
                    atomicAdd(un.level_dbg + level * kLevelDbgWords + 4, (unsigned)total);
                }
            }
        }
    }
    stamp(5);
    if constexpr (FTM != -2) {
        // ---- K23 for this row, from LDS: weights at fm.P, the row's share of the 19 moments ----------------------
        RowAcc acc;
#pragma unroll
        for (int j = 0; j < kNSums; j++) acc.a[j] = 0.0;
        if (valid && n > 0) {
            double xr[3];
            rotated_point(fm.P, q, xr);
            RowMoments<FTM> row;
            row.begin(fm.md);
#pragma unroll
            for (int j = 0; j < M; j++) {
                if (j < n) {  // nearly every row is full: the branch is uniform for most waves
                    const float4 y = L.get(L.load(j));
                    row.add_pair(fm.md, xr, y.x, y.y, y.z);
                }
            }
            row.finish(acc, fm.P, q, xr);
        }
        __syncthreads();  // every wave is through with the halo: the fold borrows its memory
        double *const scratch = reinterpret_cast<double *>(s_halo);
        // a block that was handed to the cleanup kernel (s_bail) leaves its slot to that kernel
        block_reduce_scratch(acc, scratch, scratch + 10 * 257, fm.partials + wg, (size_t)fm.nslots, s_bail == 0);
        stamp(6);
    }
    if constexpr (VERLET) {
        // the forecast for the next launch's dispatch order (verlet_forecast): will the fresh list still do after one more
        // move like this one?  (a row without a list, or a block the cleanup kernel redoes: no)
        const float need_next = (tm != 0xFFFFFFFFu ? __builtin_amdgcn_sqrtf(__uint_as_float(tm)) : __builtin_amdgcn_sqrtf(r2_far)) + moved;
        const float reach_next = (need_next + moved) * 1.0001f;
        verlet_forecast(vv, wg, valid && !(reach_next * reach_next < vg_new), s_need);
    }
    flush_stamps();
}


}  // namespace dev
}  // namespace ppcr
