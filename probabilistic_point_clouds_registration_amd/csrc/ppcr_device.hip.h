// Device-side building blocks shared by every translation unit of the library (structs the kernels take by value,
// distance / selection / weight / moment helpers, the block folds, the one-lane rigid solve and the fold-and-solve
// step).  Everything here is a template or __device__ __forceinline__: no kernel is defined in this header, so it can
// be included from several translation units.
//
// Float contraction is OFF for every translation unit (-ffp-contract=off): neighbour membership is decided by a float
// d^2 accumulated x->y->z (FLANN L2_Simple<float>); f64 code asks for fma() explicitly where wanted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include <type_traits>

#include "ppcr_host_math.hpp"

namespace ppcr {
namespace dev {

constexpr int kNSums = 19;
constexpr int kBlock = 256;
constexpr int kMaxSplit = 128;  // K1: blocks that can be scanned as two half-blocks (SplitTable in ppcr_nn_tile.hip.h)

// Uniform grid over the target's bounding box.  Cells are cubes of edge h >= radius in y and z; in x every cell is
// split into xr slices (edge h / xr, inv_hx = xr * inv_h): a (dy, dz) row of the stencil is one contiguous run
// of the cell-sorted target whatever xr is, and a finer x lets every query clip each of its nine runs to the
// x window the sphere really needs in that row (nn_fast_kernel) instead of three full cells.
// n[0] counts x SLICES; the stencil reaches xr slices either side of the query's slice.
struct GridDesc {
    float org[3];
    float inv_h;
    int n[3];
    int ncells;
    float inv_hx;  // xr * inv_h
    float h;       // cell edge in y and z
    float eps;     // absolute slack that covers the float rounding of cell coordinates and gaps
    int xr;        // x slices per cell edge (1, 2, 4, 8) = stencil reach in x slices
    int xr_shift;  // log2(xr)
};

// One resolution of the target's grid.  A cloud whose density varies a hundredfold has no single good cell size: where
// it is dense a radius-sized cell holds hundreds of points (no LDS tile holds a block's halo), where it is sparse the m-th
// neighbour lies several cells away.  The target is therefore binned at several resolutions (cell edges a factor sqrt(2)
// apart, each with its own sorted copy) and every 256-query block of K1 picks, per launch, the FINEST level whose 27-cell
// stencil still covers the largest cut-off radius among its rows (nn_fast_kernel<..., MULTI>).
struct GridLevel {
    GridDesc g;
    const float4 *tgt;      // the target sorted by this level's cells
    const int *cell_start;
    const int *to_base;     // position in this level's order -> position in the base level's order (nullptr: this IS the base)
    float r2_cap;           // largest search radius^2 this level's stencil covers, capped at the full radius^2
    int pad;
};
constexpr int kMaxLevels = 12;  // cell edges a factor sqrt(2) apart: a 45-fold range of cut-off radii
constexpr int kLevelDbgWords = 6;  // diagnostic counters per level (UnansweredRows::level_dbg)

struct Pose {  // y ~ R x + t ; c = fixed origin of the moments
    double R[9];
    double t[3];
    double c[3];
};

struct Model {  // ProbabilisticWeights constants (probabilistic_weights.hpp:30-46)
    int is_normal;
    int vpd_int;   // v + dim when that is an integer in [1,64], else 0 (hot-path fast power)
    double v;      // dof
    double texp;   // -(v + dim)/2
    double vpd;    // v + dim
};

// integer cell coordinate clamped to [-1, n]; NaN -> -1.  (v-org)*inv_h is a float sub then a
// float mul in every kernel that bins points, so targets and queries bin consistently.
__device__ __forceinline__ int cell_coord(float v, float org, float inv_h, int n)
{
    float f = floorf((v - org) * inv_h);
    f = fminf(fmaxf(f, -1.0f), (float)n);
    return (int)f;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---------------------------------------------------------------------------------------------
// K1: radius-NN with top-m cut-off.  One lane per query; the queries were spatially sorted once
// (x-fastest cell order of the target grid) so the 64 lanes of a wave walk the same few cell
// rows and their candidate loads hit the same cache lines.  Per (dy,dz) the three x-adjacent
// cells form ONE contiguous run of the cell-sorted target, so a query scans 9 runs.
// Candidates are ranked by the packed key (float_bits(d2) << 32 | target_index): d2 >= +0 so
// float bits order like unsigned ints, and ties fall to the lower target index — the order the
// oracle defines (FLANN's own tie order is traversal dependent).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float dist2_flann(float4 q, float4 t)
{
    // L2_Simple<float>: result += diff*diff for x, y, z in turn; no fused multiply-add
    float dx = __fsub_rn(q.x, t.x), dy = __fsub_rn(q.y, t.y), dz = __fsub_rn(q.z, t.z);
    float r = __fmul_rn(dx, dx);
    r = __fadd_rn(r, __fmul_rn(dy, dy));
    r = __fadd_rn(r, __fmul_rn(dz, dz));
    return r;
}

struct QueryCells {
    int cx, cy, cz;
};

__device__ __forceinline__ QueryCells query_cells(float4 q, const GridDesc &g)
{
    QueryCells c;
    c.cx = cell_coord(q.x, g.org[0], g.inv_hx, g.n[0]);
    c.cy = cell_coord(q.y, g.org[1], g.inv_h, g.n[1]);
    c.cz = cell_coord(q.z, g.org[2], g.inv_h, g.n[2]);
    return c;
}

// One contiguous run [b, e) of the sorted target, eight loads in flight per lane (a lane that walks its runs one point at a
// time is bound by memory latency: one round trip per candidate)
template <int BATCH = 8, class F>
__device__ __forceinline__ void for_each_in_run(int b, int e, const float4 *__restrict__ tgt, F &&f)
{
    for (int p = b; p < e; p += BATCH) {
        float4 t[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; u++) t[u] = tgt[min(p + u, e - 1)];
#pragma unroll
        for (int u = 0; u < BATCH; u++)
            if (p + u < e) f(p + u, t[u]);
    }
}

// Visits every candidate of the 27-cell stencil: f(position_in_sorted_target, float4 point)
template <class F>
__device__ __forceinline__ void for_each_candidate(float4 q, const GridDesc &g,
                                                   const int *__restrict__ cell_start,
                                                   const float4 *__restrict__ tgt, F &&f)
{
    const QueryCells c = query_cells(q, g);
    const int x0 = max(c.cx - g.xr, 0), x1 = min(c.cx + g.xr, g.n[0] - 1);
    if (x0 > x1) return;
#pragma unroll 1
    for (int dz = -1; dz <= 1; dz++) {
        const int cz = c.cz + dz;
        if ((unsigned)cz >= (unsigned)g.n[2]) continue;
#pragma unroll 1
        for (int dy = -1; dy <= 1; dy++) {
            const int cy = c.cy + dy;
            if ((unsigned)cy >= (unsigned)g.n[1]) continue;
            const int base = (cz * g.n[1] + cy) * g.n[0];
            const int b = cell_start[base + x0], e = cell_start[base + x1 + 1];
            for_each_in_run(b, e, tgt, f);
        }
    }
}

// The same over a stencil that reaches up to `reach` cells either way (a search radius sqrt(R2) <= reach * h on a grid
// whose cells were sized for a SMALLER radius: nn_wide_kernel).  Only the rows and, in every row, the x slices that the
// sphere of squared radius R2 can touch are visited: a (dy, dz) row is at least (gy, gz) away in y and z (the gap between
// the query and that row's slab, under-estimated by g.eps), so it matters only if gy^2 + gz^2 < R2, and then only for
// |dx| <= sqrt(R2 - gy^2 - gz^2) (inflated for the float rounding of d2, as in nn_fast_kernel's windows).
template <class F>
__device__ __forceinline__ void for_each_candidate_wide(float4 q, const GridDesc &g, int reach, float R2,
                                                        const int *__restrict__ cell_start,
                                                        const float4 *__restrict__ tgt, F &&f)
{
    const QueryCells c = query_cells(q, g);
    const float R2s = R2 * 1.000004f;
    const int rows = min(reach, (int)(__builtin_amdgcn_sqrtf(R2s) * g.inv_h) + 1);  // cells the sphere can reach in y / z
    const float fy = q.y - g.org[1], fz = q.z - g.org[2];
    for (int dz = -rows; dz <= rows; dz++) {
        const int cz = c.cz + dz;
        if ((unsigned)cz >= (unsigned)g.n[2]) continue;
        const float gz = dz < 0 ? fmaxf(fz - (float)(cz + 1) * g.h - g.eps, 0.f) : (dz > 0 ? fmaxf((float)cz * g.h - fz - g.eps, 0.f) : 0.f);
        for (int dy = -rows; dy <= rows; dy++) {
            const int cy = c.cy + dy;
            if ((unsigned)cy >= (unsigned)g.n[1]) continue;
            const float gy = dy < 0 ? fmaxf(fy - (float)(cy + 1) * g.h - g.eps, 0.f) : (dy > 0 ? fmaxf((float)cy * g.h - fy - g.eps, 0.f) : 0.f);
            const float w2 = R2s - (gy * gy + gz * gz);
            if (!(w2 > 0.f)) continue;
            const float w = __builtin_amdgcn_sqrtf(w2) * 1.000001f + g.eps;
            const int fa = max(cell_coord(q.x - w, g.org[0], g.inv_hx, g.n[0]), 0);
            const int fb = min(cell_coord(q.x + w, g.org[0], g.inv_hx, g.n[0]), g.n[0] - 1);
            if (fa > fb) continue;
            const int base = (cz * g.n[1] + cy) * g.n[0];
            const int b = cell_start[base + fa], e = cell_start[base + fb + 1];
            for_each_in_run<16>(b, e, tgt, f);  // (few lanes of a wave are short rows: latency, not issue, is what counts)
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K1, list variant (default).  Measured on MI355X: the candidate scan alone costs ~90 us at
// 1M<->1M while keeping a sorted top-m list inside the scan loop costs another ~320 us (every
// step some lane of the wave inserts, so the whole wave pays the insertion).  So the scan only
// APPENDS in-radius candidates to a lane-private list in LDS ([slot][lane]: conflict-free) and
// the cut-off is applied afterwards:
//   pass A  threshold T = m-th smallest d2 of the list, by inserting the d2 bit patterns into a
//           sorted register list with v_med3_u32:  L'_j = med3(L_{j-1}, k, L_j)  — one
//           instruction per slot and no carry chain (the list stays sorted, duplicates allowed);
//   pass B  keep the entries with d2 <= T (in place);
//   ties    only if more entries tie at T than there is room for: keep the tied entries with the
//           smallest original target index (same med3 trick on the indices) — the oracle's
//           (d2, index) order, exactly.
// A list that fills up (C entries) is compacted on the spot and the lane's acceptance
// threshold drops to T, so dense neighbourhoods cost a few compactions instead of overflowing.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int M>
__device__ __forceinline__ void sorted_insert(unsigned (&K)[M], unsigned k)
{
#pragma unroll
    for (int j = M - 1; j >= 1; --j) K[j] = umed3(K[j - 1], k, K[j]);
    K[0] = min(K[0], k);
}

template <int M>
__device__ __forceinline__ unsigned pick(const unsigned (&K)[M], int j)
{
    unsigned r = 0;
#pragma unroll
    for (int a = 0; a < M; a++) r = (a == j) ? K[a] : r;
    return r;
}

__device__ __forceinline__ double log_prob(const Model &md, double s)
{
    // additive constants cancel in the row softmax (probabilistic_weights.hpp:39-41,44,69,71-72)
    return md.is_normal ? -0.5 * s : md.texp * log1p(s / md.v);
}

__device__ __forceinline__ double sq_residual(const float4 y, const double xr[3])
{
    const double r0 = (double)y.x - xr[0], r1 = (double)y.y - xr[1], r2 = (double)y.z - xr[2];
    return r0 * r0 + r1 * r1 + r2 * r2;
}

__device__ __forceinline__ void rotate_point(const Pose &P, float4 xf, double xr[3])
{
    const double px = xf.x, py = xf.y, pz = xf.z;
    xr[0] = (P.R[0] * px + P.R[1] * py + P.R[2] * pz) + P.t[0];
    xr[1] = (P.R[3] * px + P.R[4] * py + P.R[5] * pz) + P.t[1];
    xr[2] = (P.R[6] * px + P.R[7] * py + P.R[8] * pz) + P.t[2];
}

__device__ __forceinline__ double fast_rsqrt(double x)  // x > 0, finite
{
    double r = __builtin_amdgcn_rsq(x);
    r = r * fma(-0.5 * x * r, r, 1.5);
    r = r * fma(-0.5 * x * r, r, 1.5);
    return r;
}

// 1/x to ~1 ulp without the IEEE division sequence: v_rcp_f64 seed + two Newton steps (x finite, > 0)
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// exp(lp(s) - lp(smin)) for the hot path.  t model: (u_min/u)^((v+d)/2) with u = 1 + s/v; when
// v + d is an integer (every practical dof) this is an integer power times at most one sqrt — no
// log1p/exp at all; otherwise the reference's exp(texp * log1p(s/v)) form.  Gaussian: exp(-(s-smin)/2).
// inv_vs = 1/(v + s) (shared with the expected-weight factor; unused by the Gaussian model)
// TM (compile-time model): -1 = read md at run time; 0 = Gaussian; k > 0 = t model with v + dim == k;
// -3 = t model with an integer v + dim read at run time (md.vpd_int).
// (The run-time form keeps the odd-power sqrt behind an opaque branch: as a plain ?: the compiler if-converts
//  it and every pair pays the 20-instruction f64 sqrt expansion — measured: 220 of 970 VALU instructions per row.)
// exp(x) for x <= 0 (the Gaussian model's likelihood ratios), ~1 ulp: x = k ln2 + r with |r| <= ln2 / 2, a degree-13
// Taylor polynomial in r (remainder < 4e-18) and v_ldexp_f64; arguments below -745 give 0 like exp().  About 20
// instructions against ~45 for the library routine, which has to serve the whole real line.
__device__ __forceinline__ double exp_nonpositive(double x)
{
    x = fmax(x, -800.0);
    const double kf = rint(x * 1.4426950408889634);            // log2(e)
    double r = fma(kf, -6.93147180369123816490e-01, x);        // ln2 high part (exact product for |k| < 2^11)
    r = fma(kf, -1.90821492927058770002e-10, r);               // ln2 low part
    double p = 1.0 / 6227020800.0;                              // 1/13!
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)kf);
}

template <int TM = -1>
__device__ __forceinline__ double rel_likelihood(const Model &md, double s, double smin, double lp_max, double inv_vs)
{
    if constexpr (TM == 0) return exp_nonpositive(-0.5 * (s - smin));
    if constexpr (TM == -3) {
        // t model whose v + dim is an integer known only at run time (md.vpd_int: -d 3 -> 6, -d 10 -> 13 ...):
        // rho^((v + dim) / 2) by squaring (wave-uniform trip count), the half power of an odd v + dim as
        // rho * rsqrt(rho) from the v_rsq_f64 seed + two Newton steps — no sqrt / exp / log1p expansion in the kernel
        // this is folded into
        const double rho = (md.v + smin) * inv_vs;  // in (0, 1]
        double r = 1.0;
        if (md.vpd_int & 1) r = rho * fast_rsqrt(rho);  // (uniform branch)
        double base = rho;
        for (int k = md.vpd_int >> 1; k; k >>= 1) {
            if (k & 1) r *= base;
            base *= base;
        }
        return r;
    }
    if constexpr (TM > 0) {
        const double rho = (md.v + smin) * inv_vs;
        double r = 1.0, base = rho;  // same multiplication sequence as the run-time loop below (1.0 * x is exact)
        if constexpr (TM & 1) r = sqrt(rho);
#pragma unroll
        for (int k = TM >> 1; k; k >>= 1) {
            if (k & 1) r *= base;
            base *= base;
        }
        return r;
    }
    if (md.is_normal) return exp(-0.5 * (s - smin));
    if (md.vpd_int) {
        const double rho = (md.v + smin) * inv_vs;  // = u_min / u  in (0, 1]
        double r = 1.0;
        if (md.vpd_int & 1) {
            r = sqrt(rho);
            asm volatile("" : "+v"(r));  // not speculatable: keeps the sqrt out of the even-power path
        }
        double base = rho;
        for (int k = md.vpd_int >> 1; k; k >>= 1) {     // wave-uniform trip count
            if (k & 1) r *= base;
            base *= base;
        }
        return r;
    }
    return exp(md.texp * log1p(s / md.v) - lp_max);
}

struct RowAcc {  // per-lane running moments
    double a[kNSums];
};

__device__ __forceinline__ void row_finish(RowAcc &acc, const Pose &P, float4 xf, double Z, double G, double Gs,
                                           double Gyy, const double Gy[3])
{
    const double iz = fast_rcp(Z);  // w_k = g_k / Z
    const double Wi = G * iz;
    const double xc[3] = {(double)xf.x - P.c[0], (double)xf.y - P.c[1], (double)xf.z - P.c[2]};
    const double wy[3] = {Gy[0] * iz, Gy[1] * iz, Gy[2] * iz};
    acc.a[0] += Wi;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        acc.a[1 + d] = fma(Wi, xc[d], acc.a[1 + d]);
        acc.a[4 + d] += wy[d];
#pragma unroll
        for (int b = 0; b < 3; b++) acc.a[7 + 3 * d + b] = fma(xc[d], wy[b], acc.a[7 + 3 * d + b]);
    }
    acc.a[16] += Gs * iz;
    acc.a[17] = fma(Wi, xc[0] * xc[0] + xc[1] * xc[1] + xc[2] * xc[2], acc.a[17]);
    acc.a[18] += Gyy * iz;
}

// Block fold of the 19 per-lane accumulators -> partials[j * nblocks + block], through an LDS transpose.
// (A shuffle tree costs 19 sums x 6 steps x 2 ds_bpermute + add per WAVE — a third of all instructions of the
// one-row-per-lane K23 kernel.)  Every lane parks its 19 doubles in sh[j][tid]; thread (j = t % 32 < 19,
// part = t / 32) then adds 32 consecutive entries of row j, and 19 threads add the 8 parts: ~90 instructions per
// wave, fixed summation order, no atomics.  Row stride 257 doubles: lanes j = 0..18 of a half-wave hit
// consecutive 8-byte bank pairs.
// The same two-round fold on scratch memory the caller provides (kernels that fold at their very end lend the buffers
// they no longer need): sh holds 10 * 257 doubles, part 19 * 8.  All 256 threads must call it.
__device__ __forceinline__ void block_reduce_scratch(const RowAcc &acc, double *sh, double *part, double *__restrict__ out,
                                                     size_t out_stride, bool write)
{
    constexpr int STRIDE = 257, ROUND = (kNSums + 1) / 2;
    const int tid = threadIdx.x;
    const int j = tid & 31, p = tid >> 5;
#pragma unroll
    for (int j0 = 0; j0 < kNSums; j0 += ROUND) {
        if (j0 > 0) __syncthreads();
#pragma unroll
        for (int q = 0; q < ROUND; q++)
            if (j0 + q < kNSums) sh[q * STRIDE + tid] = acc.a[j0 + q];
        __syncthreads();
        if (j < ROUND && j0 + j < kNSums) {
            const double *row = sh + j * STRIDE + p * 32;
            double v0 = row[0], v1 = row[1], v2 = row[2], v3 = row[3];
#pragma unroll
            for (int k = 4; k < 32; k += 4) {
                v0 += row[k];
                v1 += row[k + 1];
                v2 += row[k + 2];
                v3 += row[k + 3];
            }
            part[(j0 + j) * 8 + p] = (v0 + v1) + (v2 + v3);
        }
    }
    __syncthreads();
    if (write && tid < kNSums) {
        double v = part[tid * 8];
#pragma unroll
        for (int q = 1; q < 8; q++) v += part[tid * 8 + q];
        out[(size_t)tid * out_stride] = v;
    }
}
constexpr int kFoldScratchBytes = (10 * 257 + kNSums * 8) * 8;  // 21 776

// 1/x to ~2e-15 (v_rcp_f64 seed, 4.5e-8 measured on gfx950, + ONE Newton step): the per-pair reciprocal of the folded
// K23, whose results are held to 1e-10
__device__ __forceinline__ double rcp_1step(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}

// One row's contribution to the moments for a compiled-in model, pairs handed over one at a time (the one-pass form of
// accumulate_ell_kernel: likelihoods relative to s = 0).  Used by the kernels that fold K23 into the association.
// t model with v + dim = 8 (the reference's defaults): e = (v / (v + s))^4 and g = e (v + dim) / (v + s); the constant
// factors v^4 and (v + dim) v^4 are left out of the per-pair sums and applied once per row in finish().
template <int TM>
struct RowMoments {
    double Z = 0, G = 0, Gs = 0, Gr[3] = {0, 0, 0};  // sum e, sum g, sum g s, sum g r   (r = y - (R x + t))
    // xr = R x + t.  The centred target never appears: sum g (y - c) = sum g r + (xr - c) sum g.
    // One stored pair (callers skip the unused slots of a row by control flow: no selects in here).
    __device__ __forceinline__ void add_pair(const Model &md, const double (&xr)[3], float yx, float yy, float yz)
    {
        const double r0 = (double)yx - xr[0], r1 = (double)yy - xr[1], r2 = (double)yz - xr[2];
        const double sk = fma(r2, r2, fma(r1, r1, r0 * r0));
        double gk;
        if constexpr (TM == 8) {
            const double inv = rcp_1step(md.v + sk);
            const double c = inv * inv, p = c * c;
            Z += p;
            gk = p * inv;
        } else {
            const double inv_vs = (TM == 0) ? 0.0 : rcp_1step(md.v + sk);
            const double e = rel_likelihood<TM>(md, sk, 0.0, 0.0, inv_vs);
            Z += e;
            gk = (TM == 0 || (TM == -1 && md.is_normal)) ? e : e * (md.vpd * inv_vs);
        }
        G += gk;
        Gs = fma(gk, sk, Gs);
        Gr[0] = fma(gk, r0, Gr[0]);
        Gr[1] = fma(gk, r1, Gr[1]);
        Gr[2] = fma(gk, r2, Gr[2]);
    }
    __device__ __forceinline__ void add(const Model &md, const double (&xr)[3], float yx, float yy, float yz, bool live)
    {
        if (live) add_pair(md, xr, yx, yy, yz);
    }
    __device__ __forceinline__ void finish(RowAcc &acc, const Pose &P, float4 xf, const double (&xr)[3]) const
    {
        double Zs = Z, Gk = G, Gsk = Gs, Grk[3] = {Gr[0], Gr[1], Gr[2]};
        if constexpr (TM == 8) {
            const double v2 = md_v2, v4 = v2 * v2, k = md_vpd * v4;
            Zs *= v4, Gk *= k, Gsk *= k, Grk[0] *= k, Grk[1] *= k, Grk[2] *= k;
        }
        const double xrc[3] = {xr[0] - P.c[0], xr[1] - P.c[1], xr[2] - P.c[2]};
        const double Gy[3] = {fma(xrc[0], Gk, Grk[0]), fma(xrc[1], Gk, Grk[1]), fma(xrc[2], Gk, Grk[2])};
        // sum g |y - c|^2 with y - c = r + xrc:  Gs + 2 xrc . Gr + |xrc|^2 G
        const double x2 = fma(xrc[2], xrc[2], fma(xrc[1], xrc[1], xrc[0] * xrc[0]));
        const double Gyy = fma(x2, Gk, fma(2.0, fma(xrc[2], Grk[2], fma(xrc[1], Grk[1], xrc[0] * Grk[0])), Gsk));
        row_finish(acc, P, xf, Zs, Gk, Gsk, Gyy, Gy);
    }
    double md_v2 = 0, md_vpd = 0;  // set by begin() for TM == 8
    __device__ __forceinline__ void begin(const Model &md)
    {
        md_v2 = md.v * md.v;
        md_vpd = md.vpd;
    }
};
__device__ __forceinline__ void rotated_point(const Pose &P, float4 xf, double (&xr)[3])
{
    const double px = xf.x, py = xf.y, pz = xf.z;
    xr[0] = fma(P.R[2], pz, fma(P.R[1], py, fma(P.R[0], px, P.t[0])));
    xr[1] = fma(P.R[5], pz, fma(P.R[4], py, fma(P.R[3], px, P.t[1])));
    xr[2] = fma(P.R[8], pz, fma(P.R[7], py, fma(P.R[6], px, P.t[2])));
}
// K23 folded into the association: the pose and model the first IRLS half-step is evaluated at, and where this
// workgroup's 19 partial sums go (slot = its index in the FAST kernel's grid; stride = number of slots)
struct FusedMoments {
    Pose P;
    Model md;
    double *partials;
    int nslots;
};

// HALVES = true folds ten sums, then nine, through a buffer half the size (20.6 KB instead of 39 KB: six instead of
// four workgroups per CU for a kernel that is otherwise lean in registers) at the price of two more barriers.
// The block's sums go to partials[j * stride + slot]; COHERENT: with agent-scope atomic stores (readers in the same
// launch, on another XCD: see inner_steps_kernel).  lds_acc (nullable): the sums are ADDED to lds_acc[j] in LDS instead
// (a workgroup that walks several tiles keeps its running sums there, not in 38 VGPRs across the loop).
template <int BLOCK = kBlock, bool HALVES = false, bool COHERENT = false>
__device__ __forceinline__ void block_reduce_store(const RowAcc &acc, double *__restrict__ partials, int stride, int slot,
                                                   double *lds_acc = nullptr)
{
    static_assert(BLOCK == 256, "fold layout assumes 256 lanes (8 parts of 32)");
    constexpr int STRIDE = BLOCK + 1;
    constexpr int ROUND = HALVES ? (kNSums + 1) / 2 : kNSums;  // sums per round
    __shared__ double sh[ROUND * STRIDE];
    __shared__ double part[kNSums][8];
    const int tid = threadIdx.x;
    const int j = tid & 31, p = tid >> 5;
#pragma unroll
    for (int j0 = 0; j0 < kNSums; j0 += ROUND) {
        if (j0 > 0) __syncthreads();  // the buffer is reused
#pragma unroll
        for (int q = 0; q < ROUND; q++)
            if (j0 + q < kNSums) sh[q * STRIDE + tid] = acc.a[j0 + q];
        __syncthreads();
        if (j < ROUND && j0 + j < kNSums) {
            const double *row = sh + j * STRIDE + p * 32;
            double v0 = row[0], v1 = row[1], v2 = row[2], v3 = row[3];
#pragma unroll
            for (int k = 4; k < 32; k += 4) {
                v0 += row[k];
                v1 += row[k + 1];
                v2 += row[k + 2];
                v3 += row[k + 3];
            }
            part[j0 + j][p] = (v0 + v1) + (v2 + v3);
        }
    }
    __syncthreads();
    if (tid < kNSums) {
        double v = part[tid][0];
#pragma unroll
        for (int q = 1; q < 8; q++) v += part[tid][q];
        if (lds_acc) lds_acc[tid] += v;
        else if constexpr (COHERENT) __hip_atomic_store(&partials[(size_t)tid * stride + slot], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else partials[(size_t)tid * stride + slot] = v;
    }
}


// Which block (K1) or tile (K23) of 256 source rows the g-th workgroup of a launch takes.  The dispatcher deals consecutive workgroups to the
// eight XCDs in turn; blocks are spatially coherent in their index, and neighbouring blocks stage largely the same
// target rows.  Giving XCD x the x-th eighth of the blocks (instead of every eighth block) lets those rows hit in that
// XCD's L2 instead of being fetched once per XCD.  A bijection on [0, nb): the first 8 * (nb / 8) ids are permuted,
// the remainder keeps its place.  K23 uses the same map for its tiles of rows: a tile's gathers touch the target points
// around its rows, and with every eighth tile per XCD each XCD's 4 MB L2 saw the whole 16 MB target.
// (Chunks of kXcdChunk blocks, dealt to the XCDs in turn, rather than one contiguous eighth of the blocks per XCD: work
//  that depends on WHERE a block lies — Verlet lists expire first where the cloud moves most, dense regions hand blocks
//  over — then spreads over all eight XCDs instead of queueing on one, while a chunk, two columns of the source order,
//  still shares its halo in one L2.)
constexpr int kXcdChunk = 32;
__device__ __forceinline__ int xcd_block(int g, int nb)
{
    const int full = nb / (8 * kXcdChunk) * (8 * kXcdChunk);  // the part of the range that is whole super-chunks
    if (g >= full) {
        // the rest: one contiguous eighth per XCD, as far as it divides
        const int r = g - full, per = (nb - full) >> 3;
        return full + (r < 8 * per ? (r & 7) * per + (r >> 3) : r);
    }
    const int xcd = g & 7, idx = g >> 3;  // the idx-th workgroup this XCD receives
    return ((idx / kXcdChunk) * 8 + xcd) * kXcdChunk + idx % kXcdChunk;
}

// rigid move of one point: f64 arithmetic summed left to right, f32 store (pcl::transformPointCloud
// semantics, src/prob_point_cloud_registration.cc:110-112); the w lane (original index) is preserved
__device__ __forceinline__ float4 move_point(float4 p, const Pose &P)
{
    const double x = p.x, y = p.y, z = p.z;
    p.x = (float)(((P.R[0] * x + P.R[1] * y) + P.R[2] * z) + P.t[0]);
    p.y = (float)(((P.R[3] * x + P.R[4] * y) + P.R[5] * z) + P.t[1]);
    p.z = (float)(((P.R[6] * x + P.R[7] * y) + P.R[8] * z) + P.t[2]);
    return p;
}

// Pending in-place move of the source (K4) folded into K1's prologue: the previous iteration's rigid
// transform is applied while the query is loaded and the moved point is written back, which saves one
// kernel launch and one 32 MB read+write pass per iteration.
struct PendingMove {
    int enabled;      // 0 none, 1 P below, 2 *dev (written by reduce_solve_kernel of the previous iteration)
    Pose P;
    const Pose *dev;
};

// ---------------------------------------------------------------------------------------------
// Verlet lists: the steady state of a registration.  Once the source barely moves between two associations, searching the
// grid again finds the same neighbours again.  Instead, an association made by nn_fast_kernel also leaves, per query, the
// list of EVERY target within G of the query's position — G = (bound on its m-th distance, or the radius) + 2 * skin, or the
// distance of the 16th nearest target when more than sixteen lie that close — and the
// next associations only re-measure that list (verlet_answer_rows: ~12 gathers per query instead of a 43-candidate scan)
// for as long as it provably still holds every target the exact search could return:
//   at the build position p0 the list holds all y with |y - p0| <= G;  later, at p, with a = path length travelled since
//   (>= |p - p0|), the query needs all y within n = min(r_m(p), radius) of p, and r_m(p) <= r_m(previous) + |last move|;
//   such a y has |y - p0| <= n + a, so the list is complete while  n + a < G  (float slack: the test inflates by 1e-4).
// The m nearest of the list by (d2, original index) are then the m nearest of the cloud, ties included (a target tying at
// the m-th distance is within n as well).  The test is per row, the decision per workgroup: one row that fails makes its
// workgroup search the grid as before, which rebuilds its rows' lists around their new positions.  Rigid moves shrink as a registration converges, so lists built with skin ~ the last move's length outlive
// the rest of it; the first iterations of a registration rebuild every time and cost what they cost before plus the
// list's 64 bytes per query.
struct VerletLists {
    int *vl;              // [verlet_slots(M)][ns], k-major: base positions (sorted target) of the listed targets
    unsigned char *vn;    // [ns] how many
    unsigned *vmask;      // [ns] the list slots the row's association was last written from (bit k: slot k; all-ones: unknown)
    float *vg2;           // [ns] the list holds every target whose float d2 at the BUILD position is <= this; 0: no list
    float *vacc;          // [ns] path length the query has travelled since the build (bounds its displacement)
    unsigned char *streak; // [workgroup slots] (nullable) how HOT the slot's block is: + 4 (up to 16) whenever it searches, - 2 whenever it
                          //     answers.  A block that keeps searching — rows that can keep no list (a dense blob's, a handed-over
                          //     block's), lists that last one launch — stops paying for lists it does not get to use (the
                          //     list-building scan costs twice the plain one, and in a one-round launch the slowest workgroup IS the
                          //     launch): at 8 and above it searches as the plain kernel does; one launch in sixteen (by slot:
                          //     launch_tag) it builds again, in case things have calmed down
    unsigned launch_tag;  // (the launch's index mod 16)
    unsigned *rebuilt;    // diagnostic (nullable): [0] workgroups that failed the test and searched again, cumulative
    unsigned *searched_now;   // diagnostic (nullable): the same count for THIS launch alone (a ring of eight: ppcr_debug_get_verlet) ...
    unsigned *searched_clear; // ... and the next launch's entry of that ring, zeroed by this one
    // dispatch order (ppcr_nn_tile.hip.h: verlet_slot): per XCD class c the slots filed "front" then those filed "back",
    // [2][8][ceil(grid / 8)] ints, with their counts [8][2]; order_now == nullptr: slot = workgroup index
    const int *order_now;
    const unsigned *count_now;
    int *order_next;       // where this launch's workgroups file their slots for the next one (nullable)
    unsigned *count_next;
    unsigned *count_clear; // the counters of the launch after next: zeroed by this launch (nullable)
    unsigned tgt_bytes;   // size of the sorted target in bytes (the gathers go through a buffer descriptor)
    float skin2;          // 2 * skin: how far beyond the cut-off bound a list is built ...
    float skin_rel;       // ... or this fraction of the bound, whichever is more (multi-level searches: the rows of a cloud whose
                          //     density varies a hundredfold have bounds from a twentieth of the radius to all of it, and a skin
                          //     sized for the finest level's rows is nothing to the coarsest's; 0 elsewhere)
    int build_all;        // no lists exist yet (or they are not trusted): every workgroup searches and builds
};
#ifndef PPCR_VERLET_SLOTS
#define PPCR_VERLET_SLOTS 16
#endif
// list slots per row by the association's compiled-in width M (10 neighbours: 16 slots hold the ~13.5 targets a list's reach
// holds at the benchmark's density; the command line's 20: 32), and the LDS list slots of the search that builds the lists
constexpr int verlet_slots(int M) { return M <= 12 ? PPCR_VERLET_SLOTS : 32; }
constexpr int verlet_scan_slots(int M) { return M <= 12 ? (PPCR_VERLET_SLOTS <= 16 ? 24 : 28) : 36; }
constexpr int kVerletSlotsMax = 32;

// ---------------------------------------------------------------------------------------------
// The closed-form weighted rigid solve for ONE lane (it sits on the iteration's critical path right behind the moment
// fold, with the whole chip waiting): the algorithm of solve_rigid_from_moments / svd3 / cost_from_moments in
// ppcr_host_math.hpp — one-sided Jacobi SVD of the 3x3 cross-covariance, rank handling, R = V diag(1,1,d) U^T — written
// for latency: every index is static (the shared source indexes small arrays dynamically, which lands in scratch
// memory: ~9 us measured), reciprocals and roots are v_rcp_f64 / v_rsq_f64 seeds with two Newton steps instead of the
// IEEE sequences (a Jacobi rotation only has to be orthogonal to rounding, and it is: c^2 (1 + t^2) = 1 to ~1 ulp).
// Agrees with the host solve to a few ulp of the moments; the oracle tolerance on transforms is 1e-5.
// ---------------------------------------------------------------------------------------------

struct DeviceSolve {
    double R[9], t[3], cost;
    bool degenerate;
};

__device__ __forceinline__ void jacobi_pair(double (&w)[3][3], double (&v)[3][3], const int p, const int q, bool &rotated)
{
    const double alpha = w[0][p] * w[0][p] + w[1][p] * w[1][p] + w[2][p] * w[2][p];
    const double beta = w[0][q] * w[0][q] + w[1][q] * w[1][q] + w[2][q] * w[2][q];
    const double gamma = w[0][p] * w[0][q] + w[1][p] * w[1][q] + w[2][p] * w[2][q];
    if (gamma * gamma <= 1e-32 * (alpha * beta)) return;  // columns orthogonal to rounding (also gamma == 0)
    rotated = true;
    const double zeta = (beta - alpha) * fast_rcp(2.0 * fabs(gamma)) * (gamma < 0 ? -1.0 : 1.0);
    const double az = fabs(zeta), h2 = fma(zeta, zeta, 1.0);
    const double tn = (zeta < 0 ? -1.0 : 1.0) * fast_rcp(az + h2 * fast_rsqrt(h2));
    const double c = fast_rsqrt(fma(tn, tn, 1.0)), sn = c * tn;
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const double wp = w[r][p], wq = w[r][q];
        w[r][p] = c * wp - sn * wq;
        w[r][q] = sn * wp + c * wq;
        const double vp = v[r][p], vq = v[r][q];
        v[r][p] = c * vp - sn * vq;
        v[r][q] = sn * vp + c * vq;
    }
}

__device__ __forceinline__ void swap_cols(double (&w)[3][3], double (&v)[3][3], double (&len)[3], const int a, const int b)
{
    if (len[b] > len[a]) {
        double tmp = len[a];
        len[a] = len[b];
        len[b] = tmp;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            tmp = w[r][a], w[r][a] = w[r][b], w[r][b] = tmp;
            tmp = v[r][a], v[r][a] = v[r][b], v[r][b] = tmp;
        }
    }
}

// Rotation of the weighted Kabsch problem (R maximising trace(R H), H = sum w (x - mx)(y - my)^T) on ONE lane with the
// whole chip waiting, so written for the length of its dependent chain:
//   1. Horn's quaternion form: the optimal unit quaternion is the eigenvector of the symmetric 4x4 matrix N(H) for its
//      largest eigenvalue; the eigenvalue by Newton's iteration on the characteristic quartic l^4 + c2 l^2 + c1 l + c0
//      (trace N = 0) from the upper bound (Gx + Gy) / 2 — the start is within the residual of the root, two or three
//      iterations of eight operations (Theobald's QCP); the eigenvector as the largest column of adj(N - l I), ten 3x3
//      cofactors that are all independent of each other;
//   2. ONE Newton step on SO(3) for the same objective (R H must be symmetric at the optimum): it squares the error of
//      step 1, which is eps / (eigenvalue gap): 1e-12 for thin clouds, 1e-16 typically (checked against an SVD on
//      4000 random weighted problems: worst 2e-13 rad, the conditioning of those problems).
// About 150 dependent operations where the scaled Newton iteration for the polar factor (round 2) needed ~90 per
// iteration and five to seven iterations.  Handles reflections (det H < 0) by itself.  Returns false — and the caller
// takes the Jacobi SVD route with its rank handling — when the largest eigenvalue is (numerically) not simple: planar,
// collinear or empty clouds.
__device__ __forceinline__ double det3_rows(double a0, double a1, double a2, double b0, double b1, double b2, double c0, double c1, double c2)
{
    return a0 * (b1 * c2 - b2 * c1) - a1 * (b0 * c2 - b2 * c0) + a2 * (b0 * c1 - b1 * c0);
}
__device__ __forceinline__ void quat_to_R(double w, double x, double y, double z, double (&R)[9])
{
    const double ww = w * w, xx = x * x, yy = y * y, zz = z * z;
    const double xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
    R[0] = ww + xx - yy - zz, R[1] = 2.0 * (xy - wz), R[2] = 2.0 * (xz + wy);
    R[3] = 2.0 * (xy + wz), R[4] = ww - xx + yy - zz, R[5] = 2.0 * (yz - wx);
    R[6] = 2.0 * (xz - wy), R[7] = 2.0 * (yz + wx), R[8] = ww - xx - yy + zz;
}
__device__ __forceinline__ bool quaternion_rotation(const double (&h)[3][3], double G, double (&R)[9])
{
    if (!(G > 0) || !isfinite(G)) return false;
    const double iG = fast_rcp(G);
    const double Sxx = h[0][0] * iG, Sxy = h[0][1] * iG, Sxz = h[0][2] * iG;
    const double Syx = h[1][0] * iG, Syy = h[1][1] * iG, Syz = h[1][2] * iG;
    const double Szx = h[2][0] * iG, Szy = h[2][1] * iG, Szz = h[2][2] * iG;
    // N (symmetric): n00 n01 n02 n03 / n11 n12 n13 / n22 n23 / n33
    const double n00 = Sxx + Syy + Szz, n01 = Syz - Szy, n02 = Szx - Sxz, n03 = Sxy - Syx;
    const double n11 = Sxx - Syy - Szz, n12 = Sxy + Syx, n13 = Szx + Sxz;
    const double n22 = -Sxx + Syy - Szz, n23 = Syz + Szy;
    const double n33 = -Sxx - Syy + Szz;
    const double c2 = -2.0 * (((Sxx * Sxx + Sxy * Sxy) + (Sxz * Sxz + Syx * Syx)) + ((Syy * Syy + Syz * Syz) + (Szx * Szx + Szy * Szy)) + Szz * Szz);
    const double c1 = -8.0 * det3_rows(Sxx, Sxy, Sxz, Syx, Syy, Syz, Szx, Szy, Szz);
    // det N by cofactors of its first row
    const double c0 = n00 * det3_rows(n11, n12, n13, n12, n22, n23, n13, n23, n33) - n01 * det3_rows(n01, n12, n13, n02, n22, n23, n03, n23, n33) +
                      n02 * det3_rows(n01, n11, n13, n02, n12, n23, n03, n13, n33) - n03 * det3_rows(n01, n11, n12, n02, n12, n22, n03, n13, n23);
    // lam = 1 is (Gx + Gy) / 2 in these units: an upper bound of the largest root, and within the fit's residual of it;
    // Newton descends onto the root monotonically (two or three steps for a registration that fits, more for a poor
    // fit).  Steps with the raw v_rcp_f64 seed (1e-8: it only scales the step), stopped at a relative 1e-9: whatever
    // error is left in lam reaches the eigenvector divided by the eigenvalue gap, and the Newton step on SO(3) below
    // squares it.
    double lam = 1.0;
    bool settled = false;
    for (int it = 0; it < 48; ++it) {
        const double l2 = lam * lam;
        const double P = (l2 + c2) * l2 + (c1 * lam + c0);
        const double dP = (4.0 * l2 + 2.0 * c2) * lam + c1;
        const double d = P * __builtin_amdgcn_rcp(dP);
        lam -= d;
        if (fabs(d) <= 1e-9 * fabs(lam)) {
            settled = true;
            break;
        }
    }
    if (!settled || !isfinite(lam)) return false;
    // A = N - lam I; its adjugate is (up to scale) q q^T: take the column with the largest diagonal cofactor
    const double a00 = n00 - lam, a11 = n11 - lam, a22 = n22 - lam, a33 = n33 - lam;
    const double d0 = det3_rows(a11, n12, n13, n12, a22, n23, n13, n23, a33);
    const double d1 = det3_rows(a00, n02, n03, n02, a22, n23, n03, n23, a33);
    const double d2 = det3_rows(a00, n01, n03, n01, a11, n13, n03, n13, a33);
    const double d3 = det3_rows(a00, n01, n02, n01, a11, n12, n02, n12, a22);
    // off-diagonal cofactors C_rc = (-1)^(r+c) det(A without row r and column c)
    const double k01 = -det3_rows(n01, n12, n13, n02, a22, n23, n03, n23, a33);
    const double k02 = det3_rows(n01, a11, n13, n02, n12, n23, n03, n13, a33);
    const double k03 = -det3_rows(n01, a11, n12, n02, n12, a22, n03, n13, n23);
    const double k12 = -det3_rows(a00, n01, n03, n02, n12, n23, n03, n13, a33);
    const double k13 = det3_rows(a00, n01, n02, n02, n12, a22, n03, n13, n23);
    const double k23 = -det3_rows(a00, n01, n02, n01, a11, n12, n03, n13, n23);
    const double m0 = fabs(d0), m1 = fabs(d1), m2 = fabs(d2), m3 = fabs(d3);
    double qw, qx, qy, qz, dm;
    if (m0 >= m1 && m0 >= m2 && m0 >= m3) qw = d0, qx = k01, qy = k02, qz = k03, dm = m0;
    else if (m1 >= m2 && m1 >= m3) qw = k01, qx = d1, qy = k12, qz = k13, dm = m1;
    else if (m2 >= m3) qw = k02, qx = k12, qy = d2, qz = k23, dm = m2;
    else qw = k03, qx = k13, qy = k23, qz = d3, dm = m3;
    if (!(dm > 1e-10)) return false;  // adj ~ 0: the largest eigenvalue is (nearly) double — rank-deficient H
    const double qn = fast_rsqrt((qw * qw + qx * qx) + (qy * qy + qz * qz));
    double R0[9];
    quat_to_R(qw * qn, qx * qn, qy * qn, qz * qn, R0);
    // Newton step on SO(3): M = R0 S must become symmetric; omega = (tr(M) I - sym(M))^-1 axial(M^T - M)
    double M[3][3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        M[r][0] = R0[3 * r] * Sxx + R0[3 * r + 1] * Syx + R0[3 * r + 2] * Szx;
        M[r][1] = R0[3 * r] * Sxy + R0[3 * r + 1] * Syy + R0[3 * r + 2] * Szy;
        M[r][2] = R0[3 * r] * Sxz + R0[3 * r + 1] * Syz + R0[3 * r + 2] * Szz;
    }
    const double tr = M[0][0] + M[1][1] + M[2][2];
    const double b00 = tr - M[0][0], b11 = tr - M[1][1], b22 = tr - M[2][2];
    const double b01 = -0.5 * (M[0][1] + M[1][0]), b02 = -0.5 * (M[0][2] + M[2][0]), b12 = -0.5 * (M[1][2] + M[2][1]);
    const double ax = M[1][2] - M[2][1], ay = M[2][0] - M[0][2], az = M[0][1] - M[1][0];
    // symmetric 3x3 solve by cofactors
    const double e00 = b11 * b22 - b12 * b12, e01 = b02 * b12 - b01 * b22, e02 = b01 * b12 - b02 * b11;
    const double e11 = b00 * b22 - b02 * b02, e12 = b01 * b02 - b00 * b12, e22 = b00 * b11 - b01 * b01;
    const double detB = b00 * e00 + b01 * e01 + b02 * e02;
    if (!(fabs(detB) > 1e-300)) return false;
    const double iB = __builtin_amdgcn_rcp(detB);  // (omega is a correction of ~1e-6 .. 1e-12: a 1e-8 relative error in it is nothing)
    const double ox = (e00 * ax + e01 * ay + e02 * az) * iB, oy = (e01 * ax + e11 * ay + e12 * az) * iB,
                 oz = (e02 * ax + e12 * ay + e22 * az) * iB;
    // retraction exp([omega]x) ~ the rotation of the unit quaternion (1, omega / 2) / |.|  (omega is ~1e-12 .. 1e-6)
    const double hx = 0.5 * ox, hy = 0.5 * oy, hz = 0.5 * oz;
    const double hn = fast_rsqrt(1.0 + (hx * hx + hy * hy + hz * hz));
    double dR[9];
    quat_to_R(hn, hx * hn, hy * hn, hz * hn, dR);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) R[3 * r + c] = dR[3 * r] * R0[c] + dR[3 * r + 1] * R0[3 + c] + dR[3 * r + 2] * R0[6 + c];
    bool finite = true;
#pragma unroll
    for (int k = 0; k < 9; k++) finite = finite && isfinite(R[k]);
    return finite;
}

// S points to the 19 moments.  Two uses: a plain pointer to a register array (the fold kernels, where the solve sits on
// the outer loop's critical path: 120 VGPRs, everything at hand), or a VOLATILE pointer into LDS, read where a value is
// needed and not hoisted, so that nothing of S occupies registers while the rotation is solved: 74 VGPRs — the form for
// inner_steps_kernel, whose K23 role must not pay for the solve's registers (the LDS round trips cost the lane ~4 us).
template <class SumsPtr>
__device__ inline DeviceSolve solve_rigid_device(SumsPtr S, const double (&c)[3])
{
    DeviceSolve out;
#pragma unroll
    for (int k = 0; k < 9; k++) out.R[k] = (k % 4 == 0) ? 1.0 : 0.0;
    out.t[0] = out.t[1] = out.t[2] = 0.0;
    out.cost = 0.5 * S[16];
    out.degenerate = true;
    const double W = S[0];
    if (!(W > 0) || !isfinite(W)) return out;
    out.degenerate = false;
    const double iW = fast_rcp(W);
    // H = sum w (x - mx)(y - my)^T
    auto load_H = [&](double (&h)[3][3]) {
        const double my0 = S[4] * iW, my1 = S[5] * iW, my2 = S[6] * iW;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const double sx = S[1 + a];
            h[a][0] = S[7 + 3 * a] - sx * my0;
            h[a][1] = S[8 + 3 * a] - sx * my1;
            h[a][2] = S[9 + 3 * a] - sx * my2;
        }
    };
    bool polar_ok;
    {
        double h[3][3];
        load_H(h);
        // (Gx + Gy) / 2 with Gx = sum w |x - mx|^2 = S17 - |S1..3|^2 / W, Gy likewise: bounds the largest eigenvalue
        const double s1 = S[1], s2 = S[2], s3 = S[3], s4 = S[4], s5 = S[5], s6 = S[6];
        const double G = 0.5 * ((S[17] + S[18]) - ((s1 * s1 + s2 * s2 + s3 * s3) + (s4 * s4 + s5 * s5 + s6 * s6)) * iW);
        polar_ok = quaternion_rotation(h, G, out.R);
    }
    if (!polar_ok) {
    double w[3][3], v[3][3];  // w = H, columns rotated in place; v accumulates V
    load_H(w);
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) v[a][b] = (a == b) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < 9; k++) out.R[k] = (k % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; ++sweep) {
        bool rotated = false;
        jacobi_pair(w, v, 0, 1, rotated);
        jacobi_pair(w, v, 0, 2, rotated);
        jacobi_pair(w, v, 1, 2, rotated);
        if (!rotated) break;
    }
    double len[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const double n2 = w[0][j] * w[0][j] + w[1][j] * w[1][j] + w[2][j] * w[2][j];
        len[j] = n2 > 0 ? n2 * fast_rsqrt(n2) : 0.0;
    }
    swap_cols(w, v, len, 0, 1);  // singular values descending
    swap_cols(w, v, len, 0, 2);
    swap_cols(w, v, len, 1, 2);
    if (len[0] > 0) {
        double u[3][3];  // columns of U
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const double il = len[j] > 0 ? fast_rcp(len[j]) : 0.0;
#pragma unroll
            for (int r = 0; r < 3; r++) u[r][j] = w[r][j] * il;
        }
        const double tiny = len[0] * 1e-14;
        if (len[1] <= tiny) {  // rank 1: any unit vector orthogonal to u0 (cross with the axis u0 is least aligned with)
            const double a0 = fabs(u[0][0]), a1 = fabs(u[1][0]), a2 = fabs(u[2][0]);
            const bool pick1 = a1 < a0, pick2 = a2 < (pick1 ? a1 : a0);
            const double e0 = (!pick1 && !pick2) ? 1.0 : 0.0, e1 = (pick1 && !pick2) ? 1.0 : 0.0, e2 = pick2 ? 1.0 : 0.0;
            double x0 = u[1][0] * e2 - u[2][0] * e1, x1 = u[2][0] * e0 - u[0][0] * e2, x2 = u[0][0] * e1 - u[1][0] * e0;
            const double in = fast_rsqrt(x0 * x0 + x1 * x1 + x2 * x2);
            u[0][1] = x0 * in, u[1][1] = x1 * in, u[2][1] = x2 * in;
        }
        if (len[2] <= tiny || len[1] <= tiny) {
            u[0][2] = u[1][0] * u[2][1] - u[2][0] * u[1][1];
            u[1][2] = u[2][0] * u[0][1] - u[0][0] * u[2][1];
            u[2][2] = u[0][0] * u[1][1] - u[1][0] * u[0][1];
        }
        auto det3 = [](const double (&m)[3][3]) {
            return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                   m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
        };
        // H = U S V^T with H = sum x y^T  =>  R = V diag(1,1,d) U^T maps x onto y
        const double d = (det3(u) * det3(v) < 0) ? -1.0 : 1.0;
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) out.R[3 * a + b] = v[a][0] * u[b][0] + v[a][1] * u[b][1] + d * v[a][2] * u[b][2];
    }
    }
    // translation and cost: the moments come back from LDS now
    const double mx[3] = {S[1] * iW, S[2] * iW, S[3] * iW}, my[3] = {S[4] * iW, S[5] * iW, S[6] * iW};
    double Rmx[3], Rc[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        Rmx[a] = out.R[3 * a] * mx[0] + out.R[3 * a + 1] * mx[1] + out.R[3 * a + 2] * mx[2];
        Rc[a] = out.R[3 * a] * c[0] + out.R[3 * a + 1] * c[1] + out.R[3 * a + 2] * c[2];
        out.t[a] = (my[a] - Rmx[a]) + c[a] - Rc[a];
    }
    // 0.5 * sum w |y - R x - t|^2 from the moments (cost_from_moments)
    double tp[3], RSx[3], yRx = 0, tpRSx = 0, tptp = 0, tpSy = 0;
    const double S1 = S[1], S2 = S[2], S3 = S[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        tp[a] = out.t[a] + Rc[a] - c[a];
        RSx[a] = out.R[3 * a] * S1 + out.R[3 * a + 1] * S2 + out.R[3 * a + 2] * S3;
#pragma unroll
        for (int b = 0; b < 3; b++) yRx += out.R[3 * a + b] * S[7 + 3 * b + a];
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        tpRSx += tp[a] * RSx[a];
        tptp += tp[a] * tp[a];
        tpSy += tp[a] * S[4 + a];
    }
    out.cost = 0.5 * (S[18] + S[17] + 2 * tpRSx + W * tptp - 2 * yRx - 2 * tpSy);
    return out;
}

// Host mailbox in pinned, device-mapped memory: the fold-and-solve kernel writes the moments, the rigid transform it
// solved from them and its cost there, then the sequence number (system-scope release); the host spins on `seq` — no
// copy kernel and no stream synchronisation on the iteration's critical path.
enum MailboxStatus : unsigned {
    kStepResult = 0,   // one IRLS half-step, the host decides what comes next (host-paced paths)
    kIterationDone = 1,  // the device finished the outer iteration's inner loop: T, cost_init, cost, steps are final
    kIterationPending = 2,  // the inner loop ran out of device steps before it converged: the host carries on from T
    kLaunchSkipped = 3,  // this launch stepped aside because an earlier one raised LoopState::abort
};
struct HostMailbox {
    double sums[kNSums];
    double T[12];        // [R|t] minimising sum w |y - R x - t|^2 for these moments (identity when degenerate)
    double cost;         // 0.5 * sum w |y - R x - t|^2 at that transform
    double cost_init;    // device-paced inner loop: 0.5 * sum w s at the pose the outer iteration started from
    int steps;           // device-paced inner loop: IRLS steps done in this outer iteration
    unsigned status;     // MailboxStatus
    unsigned degenerate; // no weight mass
    unsigned handed_over; // blocks the association's fast kernel left to the cleanup kernel (sizes the next cleanup grid)
    unsigned seq;
};

// Inner loop of one outer iteration run by the DEVICE (ProbPointCloudRegistrationIteration::solve iterated to Ceres'
// function_tolerance, ..._iteration.hpp:52-57 with cc:96-100): the lane that solves a step also decides whether the
// loop is over — the same test the host loop makes (solve_impl) on the same numbers — and says so in device memory,
// where the launches already enqueued behind it look before they do anything:
//   finished  (= the iteration's sequence number) the inner loop is over: the remaining step workgroups of this
//             iteration return at once;
//   abort     the device cannot finish this iteration on its own (kIterationPending): EVERY later launch of the
//             stream steps aside untouched (K1, cleanup, fold, inner steps, companion move) until the host, which has
//             taken the iteration over, clears the flag in stream order.
struct LoopState {
    unsigned abort;
    unsigned finished;  // sequence number (FoldSolve::seq) of the last outer iteration whose inner loop ended on the device
    int steps;
    int pad;
    double cost_init;
};
struct LoopCtl {  // by value with every fold-and-solve launch
    LoopState *st;    // nullptr: host-paced (every step is published as kStepResult, nothing is decided here)
    double f_tol;
    int max_steps;
    int first;        // this step is the first of its outer iteration
    int last_dev;     // no further device step is enqueued behind this one: unfinished here means kIterationPending
};
__device__ __forceinline__ bool loop_aborted(const LoopState *st)
{
    return st != nullptr && __hip_atomic_load(&st->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}

// Fold of partials[19][nblocks] (one block per sum, fixed order: deterministic, no float atomics) FOLLOWED BY THE SOLVE:
// the block that draws the last ticket reads the 19 moments back and one lane runs the closed-form weighted rigid
// solve (solve_rigid_device above) and the cost at the solution.  The
// transform goes to *pose_out in device memory, where the next association's prologue picks it up as its pending
// source move (PendingMove::dev): the outer loop no longer waits for the host between iterations.  The host gets
// everything through the mailbox and only trails behind for hasConverged() and the history.
struct FoldSolve {  // everything the fold-and-solve step needs
    const double *partials;
    int nslots;
    double *sums;
    double3 origin;
    Pose *pose_out;
    HostMailbox *mbox;
    unsigned *ticket;   // [0] ticket of the fold blocks, [1] list entries the cleanup role has finished (merged kernel)
    unsigned seq;
    const unsigned *handed_over;
    // the split table of the fast K1 (nullable): registrations made by the association that just ran become visible
    // to the next launch here, after the list has been put in ascending order of block id — the order in which blocks
    // register within one launch depends on atomics, the order of the partial slots (and with it every sum) must not
    int *split_list;
    unsigned char *split_flag;
    const unsigned *split_total;
    unsigned *split_visible;
    unsigned *split_rebuilt;  // the value of *split_total the list was last rebuilt for
    int split_nblocks;        // 256-query blocks of the source (length of split_flag)
    LoopCtl loop;
    unsigned long long *dbg;  // diagnostic (nullable): wall-clock stamps of the solve lane
};

// a launch that steps aside still owes the host its mailbox slot (the host counts sequence numbers)
__device__ __forceinline__ void publish_skipped(const FoldSolve &fs)
{
    fs.mbox->status = kLaunchSkipped;
    __hip_atomic_store(&fs.mbox->seq, fs.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <class SumsPtr>
__device__ __forceinline__ void solve_and_publish(const FoldSolve &fs, SumsPtr S);

// The split table after an association that registered new blocks: ALL threads of one workgroup rebuild the list from
// the flags — the registered blocks in ascending id, the first kMaxSplit of them split (flag 2), the rest registered
// but whole (flag 1) — so that which blocks are split, the order of their partial-sum slots and with it every sum do not
// depend on the order in which the registrations arrived (an atomic counter used to hand out the slots: with more than
// kMaxSplit candidates two runs of one registration differed in their last bits).
__device__ __forceinline__ void rebuild_split_list(const FoldSolve &fs, unsigned total)
{
    __shared__ int s_cnt[kBlock];
    const int nb = fs.split_nblocks, per = (nb + kBlock - 1) / kBlock;
    const int b0 = min((int)threadIdx.x * per, nb), b1 = min(b0 + per, nb);
    int mine = 0;
    for (int b = b0; b < b1; b++) mine += fs.split_flag[b] != 0 ? 1 : 0;
    s_cnt[threadIdx.x] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {  // exclusive prefix over 256 counts (once per change of the table: not worth a tree)
        int run = 0;
        for (int t = 0; t < kBlock; t++) {
            const int v = s_cnt[t];
            s_cnt[t] = run;
            run += v;
        }
        *fs.split_visible = (unsigned)min(run, kMaxSplit);
        *fs.split_rebuilt = total;
    }
    __syncthreads();
    int at = s_cnt[threadIdx.x];
    for (int b = b0; b < b1; b++) {
        if (fs.split_flag[b] != 0) {
            if (at < kMaxSplit) {
                fs.split_list[at] = b;
                fs.split_flag[b] = 2;
            } else {
                fs.split_flag[b] = 1;
            }
            at++;
        }
    }
    __syncthreads();
}

// one of the kNSums fold blocks (256 threads): fold row `sum_index` of the partials; the last block to finish solves.
// Returns true on the ONE lane that solved (after everything it had to write is written).  LEAN: see solve_rigid_device.
template <bool LEAN = false, int IN_FLIGHT = 0>
__device__ __forceinline__ bool fold_and_solve_block(const FoldSolve &fs, int sum_index)
{
    __shared__ double sh[kBlock / 64];
    const unsigned long long t_entry = fs.dbg ? wall_clock64() : 0ull;
    // (the first fold block also keeps the split table: the two words that say whether it has to are asked for here and
    //  looked at behind the row's loads)
    const bool keeps_table = sum_index == 0 && fs.split_visible != nullptr;
    const unsigned split_total = keeps_table ? *fs.split_total : 0u, split_rebuilt = keeps_table ? *fs.split_rebuilt : 0u;
    const double *row = fs.partials + (size_t)sum_index * fs.nslots;
    double v = 0.0;
    // sixteen loads per lane in flight (4096 slots: ONE memory round trip at 1M points), added in slot order
    // (LEAN: eight — inner_steps_kernel has at most 2048 slots, and registers to save)
    constexpr int kInFlight = IN_FLIGHT > 0 ? IN_FLIGHT : (LEAN ? 8 : 16);
    for (int b0 = 0; b0 < fs.nslots; b0 += kInFlight * kBlock) {
        double t[kInFlight];
#pragma unroll
        for (int u = 0; u < kInFlight; u++) {
            const int b = b0 + u * kBlock + threadIdx.x;
            t[u] = (b < fs.nslots) ? row[b] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kInFlight; u++) v += t[u];
    }
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (split_total != split_rebuilt) rebuild_split_list(fs, split_total);  // (uniform; a handful of launches per registration)
    if (threadIdx.x != 0) return false;
    const unsigned long long t_folded = fs.dbg ? wall_clock64() : 0ull;
    double x = sh[0];
    for (int w = 1; w < kBlock / 64; w++) x += sh[w];
    // The sum travels as an agent-scope atomic store (written through to the memory side) and is waited for before the
    // ticket is drawn; the ticket itself is relaxed.  (A release fence + acq_rel ticket, as in round 2, write back and
    // invalidate the whole L2 for the sake of one double: ~2 us on the outer loop's critical path.)
    __hip_atomic_store(&fs.sums[sum_index], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned tk = __hip_atomic_fetch_add(fs.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tk != kNSums - 1) return false;
    const unsigned long long t_ticket = fs.dbg ? wall_clock64() : 0ull;
    __hip_atomic_store(fs.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(fs.ticket + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // every fold block is past its wait
    if constexpr (LEAN) {
        __shared__ double s_S[kNSums];  // the moments for the one lane that solves (see solve_rigid_device)
        {   // (all nineteen loads in flight, then the LDS stores: load-store pairs were nineteen round trips in a row, 2.1 us)
            double t[kNSums];
#pragma unroll
            for (int j = 0; j < kNSums; j++) t[j] = __hip_atomic_load(&fs.sums[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int j = 0; j < kNSums; j++) s_S[j] = t[j];
        }
        if (fs.dbg) {
            __builtin_amdgcn_s_waitcnt(0);
            fs.dbg[0] = t_entry, fs.dbg[1] = t_folded, fs.dbg[2] = t_ticket, fs.dbg[3] = wall_clock64();
        }
        solve_and_publish(fs, static_cast<const volatile double *>(s_S));
        if (fs.dbg) {
            __builtin_amdgcn_s_waitcnt(0);
            fs.dbg[5] = wall_clock64();
        }
    } else {
        double S[kNSums];
#pragma unroll
        for (int j = 0; j < kNSums; j++) S[j] = __hip_atomic_load(&fs.sums[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (fs.dbg) {
            __builtin_amdgcn_s_waitcnt(0);
            fs.dbg[0] = t_entry, fs.dbg[1] = t_folded, fs.dbg[2] = t_ticket, fs.dbg[3] = wall_clock64();
        }
        solve_and_publish(fs, static_cast<const double *>(S));
        if (fs.dbg) {
            __builtin_amdgcn_s_waitcnt(0);
            fs.dbg[5] = wall_clock64();
        }
    }
    return true;
}

// The solve lane's work once the 19 moments are in LDS (S): closed-form solve, pose for the next association, the
// device-paced loop's decision, the mailbox.  One lane; everything it must write is written when it returns.
template <class SumsPtr>
__device__ __forceinline__ void solve_and_publish(const FoldSolve &fs, SumsPtr S)
{
    // what the publication needs from device memory is asked for BEFORE the solve, so that these round trips (~0.5 us
    // each, and dependent on nothing) run under it instead of behind it
    const unsigned handed = fs.handed_over ? *fs.handed_over : 0u;
    int prev_steps = 0;
    double prev_cost_init = 0.0;
    if (fs.loop.st != nullptr && !fs.loop.first) {
        prev_steps = fs.loop.st->steps;
        prev_cost_init = fs.loop.st->cost_init;
    }
    const double c[3] = {fs.origin.x, fs.origin.y, fs.origin.z};
    const DeviceSolve rs = solve_rigid_device(S, c);
    if (fs.dbg) fs.dbg[4] = wall_clock64();
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int b = 0; b < 3; b++) fs.pose_out->R[3 * a + b] = rs.R[3 * a + b];
        fs.pose_out->t[a] = rs.t[a];
        fs.pose_out->c[a] = 0.0;
    }
    // device-paced inner loop: the test of solve_impl / the oracle's po_solve, on the same numbers
    unsigned status = kStepResult;
    bool publish = true;
    int steps = 0;
    double c0 = 0.0;
    if (fs.loop.st != nullptr) {
        LoopState *st = fs.loop.st;
        const double cost_old = 0.5 * S[16];
        const double fc = rs.degenerate ? cost_old : rs.cost;
        steps = fs.loop.first ? 1 : prev_steps + 1;
        c0 = fs.loop.first ? cost_old : prev_cost_init;
        const bool fin = rs.degenerate || steps >= fs.loop.max_steps ||
                         (cost_old - fc) <= fmax(fs.loop.f_tol * cost_old, 1e-14 * 0.5 * (S[17] + S[18]));
        st->steps = steps;
        st->cost_init = c0;
        if (fin) {
            status = kIterationDone;
            __hip_atomic_store(&st->finished, fs.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (fs.loop.last_dev) {
            status = kIterationPending;
            __hip_atomic_store(&st->abort, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            publish = false;  // an intermediate step: the next device step takes the pose from *pose_out
        }
    }
    // (an intermediate step sends nothing to the host: its 33 words over PCIe used to sit between two device steps)
    if (!publish) return;
    {   // (S may be a volatile LDS pointer: read in one go, not one read per store)
        double Sl[kNSums];
#pragma unroll
        for (int j = 0; j < kNSums; j++) Sl[j] = S[j];
#pragma unroll
        for (int j = 0; j < kNSums; j++) fs.mbox->sums[j] = Sl[j];
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int b = 0; b < 3; b++) fs.mbox->T[4 * a + b] = rs.R[3 * a + b];
        fs.mbox->T[4 * a + 3] = rs.t[a];
    }
    if (fs.loop.st != nullptr) {
        fs.mbox->cost_init = c0;
        fs.mbox->steps = steps;
    }
    fs.mbox->cost = rs.cost;
    fs.mbox->status = status;
    fs.mbox->degenerate = rs.degenerate ? 1u : 0u;
    fs.mbox->handed_over = handed;
    __hip_atomic_store(&fs.mbox->seq, fs.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace dev
}  // namespace ppcr
