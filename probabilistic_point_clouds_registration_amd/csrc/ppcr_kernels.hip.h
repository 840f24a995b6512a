// Device kernels of the registration hot path for gfx950 (MI355X, wave64).
//
//   K0  uniform-grid build over the static target (once per pair and radius)
//   K1  radius-NN correspondence search with max_neighbours cut-off
//   K2  squared residuals + t/Gaussian soft-assignment weights (materialised, API path)
//   K23 fused weights + weighted-moment accumulation (hot path; no MFMA: it is a reduction)
//   K4  in-place rigid move of the source (f64 math, f32 store)
//
// Reference loops replaced: see include/ppcr.h and DESIGN.md.  Float contraction is OFF for the
// whole translation unit (-ffp-contract=off): neighbour membership is decided by a float d^2
// accumulated x->y->z (FLANN L2_Simple<float>); f64 code asks for fma() explicitly where wanted.
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

// Uniform grid over the target's bounding box.  Cells are cubes of edge h >= radius in y and z; in x every cell is
// split into xr slices (edge h / xr, inv_hx = xr * inv_h): a (dy, dz) row of the stencil is one contiguous run
// of the cell-sorted target whatever xr is, and a finer x lets every query clip each of its nine runs to the
// x window the sphere really needs in that row (nn_tile_kernel) instead of three full cells.
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

// ---------------------------------------------------------------------------------------------
// upload helpers
// ---------------------------------------------------------------------------------------------
__global__ void repack_kernel(const unsigned char *__restrict__ raw, int64_t n, int64_t stride,
                              float4 *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = reinterpret_cast<const float *>(raw + i * stride);
    out[i] = make_float4(p[0], p[1], p[2], __int_as_float((int)i));
}

// per-block bounding box of the finite points: out[block][6] = {minx,miny,minz,maxx,maxy,maxz}
__global__ void bbox_kernel(const float4 *__restrict__ pts, int n, float *__restrict__ out)
{
    __shared__ float sh[kBlock / 64][6];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 p = pts[i];
        float v[3] = {p.x, p.y, p.z};
#pragma unroll
        for (int a = 0; a < 3; a++)
            if (isfinite(v[a])) {
                lo[a] = fminf(lo[a], v[a]);
                hi[a] = fmaxf(hi[a], v[a]);
            }
    }
#pragma unroll
    for (int a = 0; a < 3; a++)
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_down(lo[a], off));
            hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off));
        }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
        for (int a = 0; a < 3; a++) {
            sh[wave][a] = lo[a];
            sh[wave][3 + a] = hi[a];
        }
    __syncthreads();
    if (threadIdx.x < 6) {
        float r = sh[0][threadIdx.x];
        for (int w = 1; w < kBlock / 64; w++)
            r = (threadIdx.x < 3) ? fminf(r, sh[w][threadIdx.x]) : fmaxf(r, sh[w][threadIdx.x]);
        out[blockIdx.x * 6 + threadIdx.x] = r;
    }
}

// integer cell coordinate clamped to [-1, n]; NaN -> -1.  (v-org)*inv_h is a float sub then a
// float mul in every kernel that bins points, so targets and queries bin consistently.
__device__ __forceinline__ int cell_coord(float v, float org, float inv_h, int n)
{
    float f = floorf((v - org) * inv_h);
    f = fminf(fmaxf(f, -1.0f), (float)n);
    return (int)f;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// K0a: linear cell id (x fastest) of every point, clamped into the grid
__global__ void cell_key_kernel(const float4 *__restrict__ pts, int n, GridDesc g,
                                unsigned *__restrict__ keys, int *__restrict__ vals)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = pts[i];
    int cx = clampi(cell_coord(p.x, g.org[0], g.inv_hx, g.n[0]), 0, g.n[0] - 1);
    int cy = clampi(cell_coord(p.y, g.org[1], g.inv_h, g.n[1]), 0, g.n[1] - 1);
    int cz = clampi(cell_coord(p.z, g.org[2], g.inv_h, g.n[2]), 0, g.n[2] - 1);
    keys[i] = (unsigned)((cz * g.n[1] + cy) * g.n[0] + cx);
    vals[i] = i;
}

// Source ordering: (2^bxs)x4x4-cell bricks visited boustrophedon (x snakes per brick row, y snakes per
// brick plane), cells x-fastest inside a brick.  Default bxs = 0: 4x4 columns of cells in (y,z) walked along x.  Any 256 consecutive queries then sit in one or two
// ADJACENT bricks, so the cell bounding box of a workgroup — and with it the target halo it stages
// into LDS (nn_tile_kernel) — stays small.  Only the order of the source changes, never a result.
__global__ void brick_key_kernel(const float4 *__restrict__ pts, int n, GridDesc g,
                                 unsigned *__restrict__ keys, int *__restrict__ vals, int bxs)
{
    // bxs = log2 of the brick's x extent in cells (2: 4x4x4 bricks; 0: 1x4x4 "bricks", i.e. 4x4 yz columns walked
    // along x: 256 consecutive queries then span ~4 cells in x instead of up to 8)
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = pts[i];
    // bricks are measured in whole cells: x slices are folded back to cells first
    const int cx = clampi(cell_coord(p.x, g.org[0], g.inv_hx, g.n[0]), 0, g.n[0] - 1) >> g.xr_shift;
    const int cy = clampi(cell_coord(p.y, g.org[1], g.inv_h, g.n[1]), 0, g.n[1] - 1);
    const int cz = clampi(cell_coord(p.z, g.org[2], g.inv_h, g.n[2]), 0, g.n[2] - 1);
    const int nbx = ((g.n[0] >> g.xr_shift) + (1 << bxs) - 1) >> bxs, nby = (g.n[1] + 3) >> 2;
    const int bx = cx >> bxs, by = cy >> 2, bz = cz >> 2;
    const int byy = (bz & 1) ? nby - 1 - by : by;
    const int row = bz * nby + byy;
    const int bxx = (row & 1) ? nbx - 1 - bx : bx;
    const unsigned brick = (unsigned)(row * nbx + bxx);
    keys[i] = (brick << (4 + bxs)) | (unsigned)(((cz & 3) << (2 + bxs)) | ((cy & 3) << bxs) | (cx & ((1 << bxs) - 1)));
    vals[i] = i;
}

// K0c: permute points into sorted order (the w lane keeps the caller's original index)
__global__ void gather_points_kernel(const float4 *__restrict__ in, const int *__restrict__ order, int n,
                                     float4 *__restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = in[order[i]];
}

// K0d: cell_start[c] = first sorted position with key >= c  (cell_start has ncells+1 entries)
__global__ void cell_start_kernel(const unsigned *__restrict__ keys_sorted, int n, int ncells,
                                  int *__restrict__ cell_start)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    int prev = (i == 0) ? -1 : (int)keys_sorted[i - 1];
    int cur = (i == n) ? ncells : (int)keys_sorted[i];
    for (int c = prev + 1; c <= cur; c++) cell_start[c] = i;
}

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
            for (int p = b; p < e; p++) f(p, tgt[p]);
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
// TM (compile-time model): -1 = read md at run time; 0 = Gaussian; k > 0 = t model with v + dim == k.
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

// One row's contribution to the moments for a compiled-in model, pairs handed over one at a time (the one-pass form of
// accumulate_ell_kernel: likelihoods relative to s = 0).  Used by the kernels that fold K23 into the association.
template <int TM>
struct RowMoments {
    double Z = 0, G = 0, Gs = 0, Gr[3] = {0, 0, 0};  // sum e, sum g, sum g s, sum g r   (r = y - (R x + t))
    // xr = R x + t.  The centred target never appears: sum g (y - c) = sum g r + (xr - c) sum g.
    __device__ __forceinline__ void add(const Model &md, const double (&xr)[3], float yx, float yy, float yz, bool live)
    {
        const double r0 = (double)yx - xr[0], r1 = (double)yy - xr[1], r2 = (double)yz - xr[2];
        const double sk = fma(r2, r2, fma(r1, r1, r0 * r0));
        const double sv = live ? sk : 1e300;
        const double inv_vs = (TM == 0) ? 0.0 : fast_rcp(md.v + sv);
        const double e = rel_likelihood<TM>(md, sv, 0.0, 0.0, inv_vs);
        Z += e;
        const double gk = (TM == 0) ? e : e * (md.vpd * inv_vs);
        G += gk;
        Gs = fma(gk, live ? sk : 0.0, Gs);
        Gr[0] = fma(gk, r0, Gr[0]);
        Gr[1] = fma(gk, r1, Gr[1]);
        Gr[2] = fma(gk, r2, Gr[2]);
    }
    __device__ __forceinline__ void finish(RowAcc &acc, const Pose &P, float4 xf, const double (&xr)[3]) const
    {
        const double xrc[3] = {xr[0] - P.c[0], xr[1] - P.c[1], xr[2] - P.c[2]};
        const double Gy[3] = {fma(xrc[0], G, Gr[0]), fma(xrc[1], G, Gr[1]), fma(xrc[2], G, Gr[2])};
        // sum g |y - c|^2 with y - c = r + xrc:  Gs + 2 xrc . Gr + |xrc|^2 G
        const double x2 = fma(xrc[2], xrc[2], fma(xrc[1], xrc[1], xrc[0] * xrc[0]));
        const double Gyy = fma(x2, G, fma(2.0, fma(xrc[2], Gr[2], fma(xrc[1], Gr[1], xrc[0] * Gr[0])), Gs));
        row_finish(acc, P, xf, Z, G, Gs, Gyy, Gy);
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
template <int BLOCK = kBlock, bool HALVES = false>
__device__ __forceinline__ void block_reduce_store(const RowAcc &acc, double *__restrict__ partials)
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
        partials[(size_t)tid * gridDim.x + blockIdx.x] = v;
    }
}


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
//      cut-off is applied afterwards with the v_med3 threshold selection (see nn_list_kernel),
//      now reading LDS only.
// A halo that does not fit (sparse or unsorted source) is retried per wave, and as a last resort
// the wave falls back to scanning global memory with the same selection code.
// ---------------------------------------------------------------------------------------------
constexpr int kTileRows = 128;   // halo rows per staging

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL load
// (s_waitcnt vmcnt(0)): in nn_tile_kernel that serialises the run-bound loads issued in the prologue with the
// row-table and staging loads behind the barrier; with the LDS-only fences they stay in flight across it.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
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
// The closed-form weighted rigid solve for ONE lane (it sits on the iteration's critical path right behind the moment
// fold, with the whole chip waiting): the algorithm of solve_rigid_from_moments / svd3 / cost_from_moments in
// ppcr_host_math.hpp — one-sided Jacobi SVD of the 3x3 cross-covariance, rank handling, R = V diag(1,1,d) U^T — written
// for latency: every index is static (the shared source indexes small arrays dynamically, which lands in scratch
// memory: ~9 us measured), reciprocals and roots are v_rcp_f64 / v_rsq_f64 seeds with two Newton steps instead of the
// IEEE sequences (a Jacobi rotation only has to be orthogonal to rounding, and it is: c^2 (1 + t^2) = 1 to ~1 ulp).
// Agrees with the host solve to a few ulp of the moments; the oracle tolerance on transforms is 1e-5.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double fast_rsqrt(double x)  // x > 0, finite
{
    double r = __builtin_amdgcn_rsq(x);
    r = r * fma(-0.5 * x * r, r, 1.5);
    r = r * fma(-0.5 * x * r, r, 1.5);
    return r;
}

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

// Rotation of the weighted Kabsch problem by Newton's iteration for the polar decomposition,
//     X <- (z X + X^-T / z) / 2,   z = sqrt(|X^-1|_F / |X|_F),   X_0 = H^T,
// which converges quadratically to the orthogonal factor V U^T of H^T = V S U^T: the same R as the SVD route whenever
// det H > 0 (no reflection to repair) — i.e. for every well-posed registration.  An iteration is a 3x3 adjugate with
// all nine cofactors independent, so the dependent chain is ~a dozen operations (a Jacobi sweep is three rotations of
// ~70 dependent operations each).  Returns false (and the caller takes the Jacobi SVD route with its rank handling)
// when H is singular to working precision, contains a reflection, or the iteration has not settled.
__device__ __forceinline__ bool polar_rotation(const double (&h)[3][3], double (&R)[9])
{
    double x[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) x[a][b] = h[b][a];
    bool settled = false;
    for (int it = 0; it < 24; ++it) {
        double cf[3][3];  // cofactors: X^-T = cf / det
        cf[0][0] = x[1][1] * x[2][2] - x[1][2] * x[2][1];
        cf[0][1] = x[1][2] * x[2][0] - x[1][0] * x[2][2];
        cf[0][2] = x[1][0] * x[2][1] - x[1][1] * x[2][0];
        cf[1][0] = x[0][2] * x[2][1] - x[0][1] * x[2][2];
        cf[1][1] = x[0][0] * x[2][2] - x[0][2] * x[2][0];
        cf[1][2] = x[0][1] * x[2][0] - x[0][0] * x[2][1];
        cf[2][0] = x[0][1] * x[1][2] - x[0][2] * x[1][1];
        cf[2][1] = x[0][2] * x[1][0] - x[0][0] * x[1][2];
        cf[2][2] = x[0][0] * x[1][1] - x[0][1] * x[1][0];
        const double det = x[0][0] * cf[0][0] + x[0][1] * cf[0][1] + x[0][2] * cf[0][2];
        double nx = 0, nc = 0;
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) {
                nx = fma(x[a][b], x[a][b], nx);
                nc = fma(cf[a][b], cf[a][b], nc);
            }
        // well conditioned and orientation preserving?  (|X|_F^3 bounds |det|; 1e-9 leaves cond(H) up to ~1e4-1e9 here)
        if (!(det > 1e-9 * nx * sqrt(nx))) return false;
        const double idet = fast_rcp(det);
        // z^2 = |X^-T|_F / |X|_F = sqrt(nc) / (det sqrt(nx))
        const double z2 = sqrt(nc) * idet * fast_rsqrt(nx);
        const double z = sqrt(z2), a_x = 0.5 * z, a_c = 0.5 * idet * fast_rcp(z);
        double diff = 0, nn = 0;
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) {
                const double nv = a_x * x[a][b] + a_c * cf[a][b];
                const double d = nv - x[a][b];
                diff = fma(d, d, diff);
                nn = fma(nv, nv, nn);
                x[a][b] = nv;
            }
        if (diff <= 1e-30 * nn) {  // |X_{k+1} - X_k| <= 1e-15 |X|: converged to rounding
            settled = true;
            break;
        }
    }
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) R[3 * a + b] = x[a][b];
    return settled;
}

__device__ inline DeviceSolve solve_rigid_device(const double (&S)[kNSums], const double (&c)[3])
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
    const double iW = 1.0 / W;
    const double mx[3] = {S[1] * iW, S[2] * iW, S[3] * iW}, my[3] = {S[4] * iW, S[5] * iW, S[6] * iW};
    double w[3][3], v[3][3];  // w = H = sum w (x - mx)(y - my)^T, columns rotated in place; v accumulates V
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
            w[a][b] = S[7 + 3 * a + b] - S[1 + a] * my[b];
            v[a][b] = (a == b) ? 1.0 : 0.0;
        }
    const bool polar_ok = polar_rotation(w, out.R);
    if (!polar_ok) {
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
    double Rmx[3], Rc[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        Rmx[a] = out.R[3 * a] * mx[0] + out.R[3 * a + 1] * mx[1] + out.R[3 * a + 2] * mx[2];
        Rc[a] = out.R[3 * a] * c[0] + out.R[3 * a + 1] * c[1] + out.R[3 * a + 2] * c[2];
        out.t[a] = (my[a] - Rmx[a]) + c[a] - Rc[a];
    }
    // 0.5 * sum w |y - R x - t|^2 from the moments (cost_from_moments)
    double tp[3], RSx[3], yRx = 0, tpRSx = 0, tptp = 0, tpSy = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        tp[a] = out.t[a] + Rc[a] - c[a];
        RSx[a] = out.R[3 * a] * S[1] + out.R[3 * a + 1] * S[2] + out.R[3 * a + 2] * S[3];
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
struct HostMailbox {
    double sums[kNSums];
    double T[12];        // [R|t] minimising sum w |y - R x - t|^2 for these moments (identity when degenerate)
    double cost;         // 0.5 * sum w |y - R x - t|^2 at that transform
    unsigned degenerate; // no weight mass
    unsigned handed_over; // blocks the association's fast kernel left to the cleanup kernel (sizes the next cleanup grid)
    unsigned seq;
};

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
};

// one of the kNSums fold blocks (256 threads): fold row `sum_index` of the partials; the last block to finish solves
__device__ __forceinline__ void fold_and_solve_block(const FoldSolve &fs, int sum_index)
{
    __shared__ double sh[kBlock / 64];
    const double *row = fs.partials + (size_t)sum_index * fs.nslots;
    double v = 0.0;
    for (int b0 = 0; b0 < fs.nslots; b0 += 8 * kBlock) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int b = b0 + u * kBlock + threadIdx.x;
            t[u] = (b < fs.nslots) ? row[b] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) v += t[u];
    }
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x != 0) return;
    double x = sh[0];
    for (int w = 1; w < kBlock / 64; w++) x += sh[w];
    __hip_atomic_store(&fs.sums[sum_index], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __atomic_thread_fence(__ATOMIC_RELEASE);
    const unsigned tk = __hip_atomic_fetch_add(fs.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (tk != kNSums - 1) return;
    __hip_atomic_store(fs.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(fs.ticket + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // every fold block is past its wait
    double S[kNSums];
#pragma unroll
    for (int j = 0; j < kNSums; j++) S[j] = __hip_atomic_load(&fs.sums[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double c[3] = {fs.origin.x, fs.origin.y, fs.origin.z};
    const DeviceSolve rs = solve_rigid_device(S, c);
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int b = 0; b < 3; b++) fs.pose_out->R[3 * a + b] = rs.R[3 * a + b];
        fs.pose_out->t[a] = rs.t[a];
        fs.pose_out->c[a] = 0.0;
    }
#pragma unroll
    for (int j = 0; j < kNSums; j++) fs.mbox->sums[j] = S[j];
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int b = 0; b < 3; b++) fs.mbox->T[4 * a + b] = rs.R[3 * a + b];
        fs.mbox->T[4 * a + 3] = rs.t[a];
    }
    if (fs.split_visible) {
        const int n_split = (int)min(*fs.split_total, 64u);
        for (int a = 1; a < n_split; a++) {  // insertion sort: the list is nearly sorted, at most 64 long
            const int key = fs.split_list[a];
            int b = a - 1;
            for (; b >= 0 && fs.split_list[b] > key; b--) fs.split_list[b + 1] = fs.split_list[b];
            fs.split_list[b + 1] = key;
        }
        for (int a = 0; a < n_split; a++) fs.split_flag[fs.split_list[a]] = 2;  // ... and from now on they ARE split
        *fs.split_visible = (unsigned)n_split;
    }
    fs.mbox->cost = rs.cost;
    fs.mbox->degenerate = rs.degenerate ? 1u : 0u;
    fs.mbox->handed_over = fs.handed_over ? *fs.handed_over : 0u;
    __hip_atomic_store(&fs.mbox->seq, fs.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(kBlock) void reduce_solve_kernel(FoldSolve fs) { fold_and_solve_block(fs, (int)blockIdx.x); }

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
__global__ __launch_bounds__(BLOCK, 3) void nn_tile_cleanup_kernel(const float4 *__restrict__ src, int ns,
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
            fold_and_solve_block(fs, (int)(blockIdx.x - n_cleanup));
            return;
        }
    }
    for (unsigned listed = blockIdx.x; listed < n_listed; listed += n_cleanup) {
    const int entry = ovf_list[listed];
    const int fast_slot = entry >> 2, half = entry & 3;
    const int bid = fast_slot < n_extra ? split_list[fast_slot] : fast_slot - n_extra;
    static_assert(C > M, "a compaction must leave room in the list");
    static_assert(CAP % 4 == 0 && CAP <= 65536 && C * 64 <= 3 * CAP && kTileRows <= 256, "the global fallback aliases the candidate buffer");
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
                                L.store(min(n, C - 1), f);
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
                if (n > C) {  // list overflow (slot C-1 was overwritten): redo this lane with in-loop compaction
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
                unsigned thr = thr0;
                for_each_candidate(q, g, cell_start, tgt, [&](int p, float4 t) {
                    const float d2 = dist2_flann(q, t);
                    if (d2 < r2 && __float_as_uint(d2) <= thr) {
                        G.store(n, p);
                        n++;
                        if (n == C) n = select_top_m<M>(G, tgt, q, n, m, thr);
                    }
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
            for (int j = 0; j < nrow; j++) {
                const float4 y = tgt[nbr[(size_t)j * ns + i]];
                row.add(fm.md, xr, y.x, y.y, y.z, true);
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
    unsigned *total;          // registrations so far (may exceed kMaxSplit: the surplus is not split)
    const unsigned *visible;  // registrations the extra workgroups of THIS launch may act on (set by the cleanup kernel)
    int n_extra;              // extra workgroups at the front of this launch's grid (0: no splitting in this launch)
    int presplit;             // a whole block whose halo exceeds this is registered for splitting BEFORE it overflows
                              // (halos grow a few per cent per iteration as the source drifts: 15/16 of the capacity)
};
constexpr int kMaxSplit = 64;

// FTM >= 0 (0: Gaussian, k: t model with v + dim = k) folds K23 into this kernel: each lane finishes its row's
// contribution to the 19 moments from the winners' coordinates while they are still in LDS (no neighbour gathers, no
// second pass over the source, no K23 launch) and the workgroup folds them into fm.partials.  FTM = -2: plain K1.
template <int M, int C, int CAP, bool STAMPS, int FTM = -2>
__global__ __launch_bounds__(256, (C <= 16 ? (CAP * 13 + C * 512 <= 30900 ? 5 : 4) : 3)) void nn_fast_kernel(float4 *__restrict__ src, int ns,
                                                         const float4 *__restrict__ tgt,
                                                         const int *__restrict__ cell_start, GridDesc g,
                                                         float r2, int m, int *__restrict__ nbr,
                                                         int *__restrict__ cnt, PendingMove pm,
                                                         unsigned *__restrict__ dm2, int dm2_valid,
                                                         int *__restrict__ ovf_list, unsigned *__restrict__ ovf_count,
                                                         unsigned *__restrict__ ovf_count_next, SplitTable split,
                                                         unsigned long long *__restrict__ stamps, FusedMoments fm)
{
    static_assert(C > M, "a re-scan must leave room in the list");
    static_assert(FTM == -2 || (3 * CAP + CAP / 4) * 4 >= kFoldScratchBytes, "the final fold borrows the halo buffer");
    static_assert(CAP % 4 == 0 && CAP * 4 < 65536, "list entries are 16-bit byte offsets into the halo arrays");
    constexpr int BLOCK = 256, kWaves = 4, kRows = 128, kStageUnroll = 4;
    __shared__ __attribute__((aligned(16))) float s_halo[3 * CAP + CAP / 4];
    __shared__ unsigned short s_list[C * BLOCK];
    __shared__ int s_gbo[kRows];
    // non-empty rows, compacted: {global begin, LDS offset << 19 | length << 7 | slot}; only alive between the row
    // table and the staging barrier, so it borrows the (not yet written) list area
    static_assert(sizeof(int2) * kRows <= sizeof(s_list), "row table aliases the list area");
    int2 *const s_rowtab = reinterpret_cast<int2 *>(s_list);
    __shared__ int s_box[kWaves][6];
    __shared__ int s_bail;
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
    if (tid == 0 && blockIdx.x == 0) *ovf_count_next = 0;  // the other counter of the ping-pong pair: idle during this launch
    // which block, and which of its waves' queries, this workgroup scans (uniform)
    int bid, half = 0;  // half: 0 whole block, 1 waves 0-1, 2 waves 2-3
    if ((int)blockIdx.x < split.n_extra) {
        if (blockIdx.x >= min(*split.visible, (unsigned)kMaxSplit)) {
            if constexpr (FTM != -2)  // an idle slot of the partials still has to read as zero
                if (tid < kNSums) fm.partials[(size_t)tid * fm.nslots + blockIdx.x] = 0.0;
            return;
        }
        bid = split.list[blockIdx.x];
        half = 2;
    } else {
        bid = (int)blockIdx.x - split.n_extra;
        if (split.n_extra > 0 && split.flag[bid] == 2) half = 1;
    }
    const int i = bid * BLOCK + tid;
    const bool valid = i < ns && (half == 0 || (wave >> 1) == half - 1);  // lanes whose query this workgroup owns
    if (tid == 0) s_bail = 0;

    // ---- prologue: query, pending move, temporal cut-off ---------------------------------------------------------
    float4 q = valid ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float moved = 0.f;  // how far this query travelled since the association that produced dm2
    if (pm.enabled && valid) {
        const float4 q0 = q;
        q = move_point(q, pm.enabled == 2 ? *pm.dev : pm.P);  // uniform choice, scalar loads
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
    if (dm2_valid && valid) {
        const unsigned prev = dm2[i];
        if (prev != 0xFFFFFFFFu) {
            const float bound = __builtin_amdgcn_sqrtf(__uint_as_float(prev)) + moved;
            const float t2 = bound * bound * 1.00001f + 1e-30f;
            thr0 = (t2 < r2) ? __float_as_uint(t2) : 0xFFFFFFFFu;
        }
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
        const float R2 = __uint_as_float(min(thr0, __float_as_uint(r2))) * 1.000004f;
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
    const int ys = (ny_h <= 8) ? 3 : 4;
    const bool shape_ok = hx0 <= hx1 && ny_h >= 1 && nz_h >= 1 && ny_h <= 16 && nz_h <= (kRows >> ys);
    // lane l owns halo row slots l (A) and l + 64 (B): global begin, length; LDS offsets from a scan of A + B
    int gbA, lenA, gbB, lenB;
    {
        const int ymask = (1 << ys) - 1;
        const int ryA = lane & ymask, rzA = lane >> ys, ryB = (lane + 64) & ymask, rzB = (lane + 64) >> ys;
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
    if (tid == 0 && half == 0 && split.flag != nullptr && (handed_over || total > split.presplit) && !split.flag[bid]) {
        // once the fold-and-solve step has published the registration (flag 2, sorted list) this block is scanned in
        // two halves
        const unsigned slot = atomicAdd(split.total, 1u);
        if (slot < (unsigned)kMaxSplit) {
            split.list[slot] = bid;
            split.flag[bid] = 1;
        }
    }
    if (handed_over) {
        // the cleanup kernel redoes this workgroup's queries (this launch): entry = grid index * 4 + half
        if (tid == 0) ovf_list[atomicAdd(ovf_count, 1u)] = (int)blockIdx.x * 4 + half;
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
            float4 c[kStageUnroll];
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
                c[u] = tgt[(lane < sl[u]) ? sg[u] + lane : 0];  // unconditional load (slot 0 always exists)
            }
#pragma unroll
            for (int u = 0; u < kStageUnroll; u++) {
                if (lane < sl[u]) {
                    const int d = so[u] + lane;
                    s_x[d] = c[u].x;
                    s_y[d] = c[u].y;
                    s_z[d] = c[u].z;
                    s_rowid[d] = (unsigned char)sr[u];
                }
                for (int k = lane + 64; k < sl[u]; k += 64) {  // rows longer than a wave (dense data)
                    const float4 t = tgt[sg[u] + k];
                    const int d = so[u] + k;
                    s_x[d] = t.x;
                    s_y[d] = t.y;
                    s_z[d] = t.z;
                    s_rowid[d] = (unsigned char)sr[u];
                }
            }
        }
    }
    lds_barrier();
    stamp(2);

    int n = 0;
    unsigned tm = 0xFFFFFFFFu;  // d2 bits of the m-th neighbour (all-ones: fewer than m found)
    const HaloList L{reinterpret_cast<const char *>(s_x), CAP * 4, s_rowid, reinterpret_cast<const char *>(s_gbo), s_list + tid};
    if (valid) {
        // the nine runs as ONE 32-bit key each, (length << 16) | LDS byte offset of the run's first candidate, sorted
        // by DESCENDING length: every lane of the wave walks its longest run first, ... — a run's trip count is the
        // maximum over the 64 lanes, and the maxima of order statistics add up to far fewer steps than the maxima of
        // arbitrary runs.  25-comparator network (0/1 principle), a comparator is a v_max_u32 / v_min_u32 pair.
        unsigned key[9];
        {
            const char *gbo_c = reinterpret_cast<const char *>(s_gbo) + 4 * ((qc.cz - hz0) * (1 << ys) + (qc.cy - hy0));
#pragma unroll
            for (int k = 0; k < 9; k++) {
                const int rl = re[k] - rb[k];
                const int start = rb[k] - *reinterpret_cast<const int *>(gbo_c + 4 * ((k / 3 - 1) * (1 << ys) + (k % 3 - 1)));
                key[k] = rl > 0 ? ((unsigned)rl << 16) | (unsigned)(start << 2) : 0u;
            }
            constexpr int net[25][2] = {{0, 3}, {1, 7}, {2, 5}, {4, 8}, {0, 7}, {2, 4}, {3, 8}, {5, 6}, {0, 2},
                                        {1, 3}, {4, 5}, {7, 8}, {1, 4}, {3, 6}, {5, 7}, {0, 1}, {2, 4}, {3, 5},
                                        {6, 8}, {2, 3}, {4, 5}, {6, 7}, {1, 2}, {3, 4}, {5, 6}};
#pragma unroll
            for (int c = 0; c < 25; c++) {
                const int a = net[c][0], b = net[c][1];
                const unsigned kh = max(key[a], key[b]), kl = min(key[a], key[b]);  // descending
                key[a] = kh;
                key[b] = kl;
            }
        }
        if constexpr (STAMPS) {  // diagnostic: this lane's nine sorted run lengths, 7 bits each, behind the wave records
            if (stamps) {
                unsigned long long pk = 0;
#pragma unroll
                for (int k = 0; k < 9; k++) pk |= (unsigned long long)min(key[k] >> 16, 127u) << (7 * k);
                stamps[((size_t)gridDim.x * kWaves + 64) * 8 + (size_t)blockIdx.x * BLOCK + tid] = pk;
            }
        }
        // d2 >= +0 and r2 > 0, so "d2 < r2" is "bits(d2) <= bits(r2) - 1" (a NaN d2 has larger bits and fails): the
        // radius test and the running cut-off are ONE unsigned compare per candidate
        unsigned thr = min(thr0, __float_as_uint(r2) - 1u);
        typedef float v2f __attribute__((ext_vector_type(2)));
        const v2f qx2 = {q.x, q.x}, qy2 = {q.y, q.y}, qz2 = {q.z, q.z};
        // LDS addresses as plain 32-bit integers (address space 3): the write cursor and the candidate cursor are
        // one VGPR each and an accepted candidate costs v_min + ds_write + v_add
        typedef __attribute__((address_space(3))) unsigned short *lds_u16p;
        typedef __attribute__((address_space(3))) const float *lds_f32p;
        // (the casts go through uintptr_t so that the host pass, where every pointer is 64-bit, parses them too)
        const unsigned list0 = (unsigned)(__UINTPTR_TYPE__)(lds_u16p)(s_list + tid), list_last = list0 + (C - 1) * 512;
        const unsigned halo0 = (unsigned)(__UINTPTR_TYPE__)(lds_f32p)s_x;
        for (int attempt = 0;; attempt++) {
            // the list's write cursor counts every accepted candidate (so n is exact), the store slot is clamped: an
            // overflowing lane keeps its first C - 1 entries and scribbles over the last slot
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
                        const unsigned slot_x = min(wp, list_last);  // outside the branch: the wave pays it either way
                        if (__float_as_uint(d.x) <= thr) {
                            *(lds_u16p)(__UINTPTR_TYPE__)slot_x = (unsigned short)(a - halo0);
                            wp += 512;
                        }
                        const unsigned slot_y = min(wp, list_last);
                        if (a < a_pair && __float_as_uint(d.y) <= thr) {
                            *(lds_u16p)(__UINTPTR_TYPE__)slot_y = (unsigned short)(a - halo0 + 4);
                            wp += 512;
                        }
                    }
                }
            }
            n = (int)((wp - list0) >> 9);
            if (n <= C) break;
            if (attempt == 1) {  // more than C candidates tie at the threshold: leave the block to the general flavour
                n = -1;
                break;
            }
            // list overflow (dense neighbourhood, or no usable cut-off yet): the C - 1 entries that were kept are
            // genuine in-radius candidates, so the m-th smallest of them bounds the final m-th distance: scan again
            (void)select_top_m<M>(L, tgt, q, C - 1, m, thr);
        }
        stamp(3);
        if (n > m) {
            n = select_top_m<M>(L, tgt, q, n, m, thr);
            tm = thr;
        } else if (n == m) {
            tm = 0;
            for_each_entry(L, q, n, [&](int, int, unsigned b) { tm = max(tm, b); });
        }
        stamp(4);
    }
    // a wave with a twice-overflowed lane registers the block (once) for the cleanup kernel; its other results are
    // simply overwritten there with identical values
    if (__ballot(n < 0) != 0ull) {
        if (lane == 0 && atomicExch(&s_bail, 1) == 0) ovf_list[atomicAdd(ovf_count, 1u)] = (int)blockIdx.x * 4 + half;
        n = max(n, 0);
    }
    if (valid) {
        int *out = nbr + i;
        for (int j = 0; j < n; j++) {
            *out = L.pos_of(L.load(j));
            out += ns;
        }
        cnt[i] = n;
        dm2[i] = tm;
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
#pragma unroll
            for (int j = 0; j < M; j++) {
                const bool live = j < n;
                const float4 y = L.get(live ? L.load(j) : 0);  // slot 0 of the halo for the unused pairs: finite, weight 0
                row.add(fm.md, xr, y.x, y.y, y.z, live);
            }
            row.finish(acc, fm.P, q, xr);
        }
        __syncthreads();  // every wave is through with the halo: the fold borrows its memory
        double *const scratch = reinterpret_cast<double *>(s_halo);
        // a block that was handed to the cleanup kernel (s_bail) leaves its slot to that kernel
        block_reduce_scratch(acc, scratch, scratch + 10 * 257, fm.partials + blockIdx.x, (size_t)fm.nslots, s_bail == 0);
        stamp(6);
    }
    flush_stamps();
}

// Generic path (unbounded, or max_neighbours above the register-list variants):
//   count -> exclusive scan -> fill (keys + positions) [-> per-row select of the m smallest]
__global__ void nn_count_kernel(const float4 *__restrict__ src, int ns, const float4 *__restrict__ tgt,
                                const int *__restrict__ cell_start, GridDesc g, float r2,
                                int *__restrict__ counts)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const float4 q = src[i];
    int c = 0;
    for_each_candidate(q, g, cell_start, tgt, [&](int, float4 t) { c += (dist2_flann(q, t) < r2) ? 1 : 0; });
    counts[i] = c;
}

__global__ void nn_fill_kernel(const float4 *__restrict__ src, int ns, const float4 *__restrict__ tgt,
                               const int *__restrict__ cell_start, GridDesc g, float r2,
                               const int *__restrict__ row_ptr, unsigned long long *__restrict__ keys,
                               int *__restrict__ pos)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const float4 q = src[i];
    int w = row_ptr[i];
    for_each_candidate(q, g, cell_start, tgt, [&](int p, float4 t) {
        const float d2 = dist2_flann(q, t);
        if (d2 < r2) {
            keys[w] = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)__float_as_int(t.w);
            pos[w] = p;
            w++;
        }
    });
}

// per row: move the min(m, n) smallest keys to the front (selection sort in global memory);
// counts_out[i] = min(m, n).  Only used for max_neighbours > 32 — a rare, slow-path setting.
__global__ void nn_select_kernel(int ns, const int *__restrict__ row_ptr, unsigned long long *__restrict__ keys,
                                 int *__restrict__ pos, int m, int *__restrict__ counts_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const int b = row_ptr[i], e = row_ptr[i + 1];
    const int n = e - b, keep = n < m ? n : m;
    if (n > m)
        for (int a = 0; a < keep; a++) {
            int best = b + a;
            unsigned long long kb = keys[best];
            for (int j = b + a + 1; j < e; j++)
                if (keys[j] < kb) {
                    kb = keys[j];
                    best = j;
                }
            if (best != b + a) {
                unsigned long long tk = keys[b + a];
                keys[b + a] = keys[best];
                keys[best] = tk;
                int tp = pos[b + a];
                pos[b + a] = pos[best];
                pos[best] = tp;
            }
        }
    counts_out[i] = keep;
}

__global__ void csr_compact_kernel(int ns, const int *__restrict__ row_ptr_in, const int *__restrict__ pos_in,
                                   const int *__restrict__ row_ptr_out, int *__restrict__ pos_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const int bi = row_ptr_in[i], bo = row_ptr_out[i], n = row_ptr_out[i + 1] - bo;
    for (int k = 0; k < n; k++) pos_out[bo + k] = pos_in[bi + k];
}

__global__ void ell_count_sum_kernel(const int *__restrict__ cnt, int ns, unsigned long long *__restrict__ total)
{
    __shared__ unsigned long long sh[kBlock / 64];
    unsigned long long s = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) s += (unsigned)cnt[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = 0;
        for (int w = 0; w < kBlock / 64; w++) b += sh[w];
        if (b) atomicAdd(total, b);
    }
}

// ---------------------------------------------------------------------------------------------
// association accessors: ELL (k-major: slot k of row i at nbr[k*ns + i] -> coalesced for one
// lane per row) and CSR (rows of arbitrary length)
// ---------------------------------------------------------------------------------------------
struct EllAssoc {
    const int *nbr;
    const int *cnt;
    int ns;
    __device__ __forceinline__ int count(int i) const { return cnt[i]; }
    __device__ __forceinline__ size_t slot(int i, int k) const { return (size_t)k * ns + i; }
};
struct CsrAssoc {
    const int *nbr;
    const int *row_ptr;
    __device__ __forceinline__ int count(int i) const { return row_ptr[i + 1] - row_ptr[i]; }
    __device__ __forceinline__ size_t slot(int i, int k) const { return (size_t)row_ptr[i] + k; }
};

// K2 (API path): materialise s and w per stored pair with the reference's exact formula:
//   lp, row max, mll = log(sum exp(lp - max)) + max, w = exp(lp - mll) [* (v+d)/(v+s)]
template <class A>
__global__ void weights_kernel(A a, const float4 *__restrict__ src, const float4 *__restrict__ tgt, int ns,
                               Pose P, Model md, double *__restrict__ w_out, double *__restrict__ s_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const int n = a.count(i);
    if (n == 0) return;
    double xr[3];
    rotate_point(P, src[i], xr);
    double max_lp = -INFINITY;
    for (int k = 0; k < n; k++) {
        const double s = sq_residual(tgt[a.nbr[a.slot(i, k)]], xr);
        const double lp = log_prob(md, s);
        max_lp = lp > max_lp ? lp : max_lp;
        if (s_out) s_out[a.slot(i, k)] = s;
    }
    double z = 0;
    for (int k = 0; k < n; k++) z += exp(log_prob(md, sq_residual(tgt[a.nbr[a.slot(i, k)]], xr)) - max_lp);
    const double mll = log(z) + max_lp;
    if (w_out)
        for (int k = 0; k < n; k++) {
            const double s = sq_residual(tgt[a.nbr[a.slot(i, k)]], xr);
            double w = exp(log_prob(md, s) - mll);
            if (!md.is_normal) w *= md.vpd / (md.v + s);
            w_out[a.slot(i, k)] = w;
        }
}

// K23 (hot path), generic rows (CSR or wide ELL): one lane per source row, grid-stride; two sweeps
// over the row (min s, then the softmax sums).
template <class A>
__global__ __launch_bounds__(kBlock) void accumulate_kernel(A a, const float4 *__restrict__ src,
                                                            const float4 *__restrict__ tgt, int ns, Pose P,
                                                            Model md, double *__restrict__ partials)
{
    RowAcc acc;
#pragma unroll
    for (int j = 0; j < kNSums; j++) acc.a[j] = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < ns; i += gridDim.x * kBlock) {
        const int n = a.count(i);
        if (n == 0) continue;
        const float4 xf = src[i];
        double xr[3];
        rotate_point(P, xf, xr);
        // max lp <=> min s: both models are monotone decreasing in s
        double smin = INFINITY;
        for (int k = 0; k < n; k++) {
            const double s = sq_residual(tgt[a.nbr[a.slot(i, k)]], xr);
            smin = s < smin ? s : smin;
        }
        const double lp_max = log_prob(md, smin);
        double Z = 0, G = 0, Gs = 0, Gyy = 0, Gy[3] = {0, 0, 0};
        for (int k = 0; k < n; k++) {
            const float4 y = tgt[a.nbr[a.slot(i, k)]];
            const double s = sq_residual(y, xr);
            const double inv_vs = md.is_normal ? 0.0 : fast_rcp(md.v + s);
            const double e = rel_likelihood(md, s, smin, lp_max, inv_vs);
            Z += e;
            const double gk = md.is_normal ? e : e * (md.vpd * inv_vs);
            const double yc0 = (double)y.x - P.c[0], yc1 = (double)y.y - P.c[1], yc2 = (double)y.z - P.c[2];
            G += gk;
            Gs = fma(gk, s, Gs);
            Gy[0] = fma(gk, yc0, Gy[0]);
            Gy[1] = fma(gk, yc1, Gy[1]);
            Gy[2] = fma(gk, yc2, Gy[2]);
            Gyy = fma(gk, yc0 * yc0 + yc1 * yc1 + yc2 * yc2, Gyy);
        }
        row_finish(acc, P, xf, Z, G, Gs, Gyy, Gy);
    }
    block_reduce_store(acc, partials);
}

// K23 (hot path), ELL rows of width <= W.  Latency, not arithmetic, bounds this kernel (three dependent
// memory round trips per row: row header -> neighbour indices -> target points), so each lane owns ROWS
// rows and issues ALL their loads before any arithmetic: 2 + ROWS*W index loads in flight, then ROWS*W
// gathers in flight; the rows are then finished from registers in a single sweep (never re-read).  The
// grid covers every row exactly once (no grid-stride loop).  Measured at 1M rows, W = 10: ROWS = 1
// (124 VGPRs, 4 waves/SIMD) 58.6 us; ROWS = 2 (168 VGPRs, 3 waves/SIMD) 67 us; forcing 96 VGPRs spills: 74 us.
// ONEPASS (compile-time models only): likelihoods are taken relative to s = 0 instead of the row's smallest s —
// (v / (v + s))^((v+d)/2) or exp(-s / 2) — so a pair is finished the moment its point arrives and nothing per pair
// stays in registers (the two-pass form keeps s[W] and the centred points: 130 VGPRs, three waves per SIMD).  The host
// picks it only when that ratio cannot underflow for any s below radius^2; the weights w = g / Z are the same numbers
// up to rounding.
template <int W, int ROWS, int BLOCK, int TM = -1, bool ONEPASS = false>
__global__ __launch_bounds__(BLOCK) void accumulate_ell_kernel(const int *__restrict__ nbr,
                                                                const int *__restrict__ cnt,
                                                                const float4 *__restrict__ src,
                                                                const float4 *__restrict__ tgt, int ns, Pose P,
                                                                Model md, double *__restrict__ partials, int width)
{
    // width = slots the association really has per row (<= W): slots beyond it do not exist in nbr
    RowAcc acc;
#pragma unroll
    for (int j = 0; j < kNSums; j++) acc.a[j] = 0.0;
    const int base = blockIdx.x * (BLOCK * ROWS) + threadIdx.x;
    int n[ROWS];
    float4 xf[ROWS];
    float yx[ROWS][W], yy[ROWS][W], yz[ROWS][W];
    {
        int idx[ROWS][W];
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const int i = base + r * BLOCK;
            const bool ok = i < ns;
            n[r] = ok ? cnt[i] : 0;
            xf[r] = src[ok ? i : 0];
#pragma unroll
            for (int k = 0; k < W; k++) idx[r][k] = (ok && k < width) ? nbr[(size_t)k * ns + i] : 0;  // slots >= cnt: stale
        }
#pragma unroll
        for (int r = 0; r < ROWS; r++)
#pragma unroll
            for (int k = 0; k < W; k++) {
                const float4 y = tgt[(k < n[r]) ? idx[r][k] : 0];  // slot 0 of the target for unused slots: masked below
                yx[r][k] = y.x;
                yy[r][k] = y.y;
                yz[r][k] = y.z;
            }
    }
    // Row arithmetic, written for instruction count (the kernel is bound by VALU issue: 599 instructions per wave for
    // one row per lane at W = 10 before this form, ~57 % VALU busy):
    //   * explicit FMAs (the translation unit is compiled with contraction off for K1's sake);
    //   * residual and centred target share their work: yc = y - c, r = yc - (R x + t - c);
    //   * unused slots are given s = 1e300, for which every model's likelihood ratio underflows to exactly 0: no
    //     per-pair masks (their y is slot 0 of the target: finite, so 0 * y stays 0);
    //   * sum g |y - c|^2 is not accumulated: with yc = r + xrc it equals Gs + 2 xrc . Gy - |xrc|^2 G.
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        if (n[r] == 0) continue;
        const double px = xf[r].x, py = xf[r].y, pz = xf[r].z;
        // xrc = R x + t - c
        const double xrc[3] = {fma(P.R[2], pz, fma(P.R[1], py, fma(P.R[0], px, P.t[0] - P.c[0]))),
                               fma(P.R[5], pz, fma(P.R[4], py, fma(P.R[3], px, P.t[1] - P.c[1]))),
                               fma(P.R[8], pz, fma(P.R[7], py, fma(P.R[6], px, P.t[2] - P.c[2])))};
        if constexpr (ONEPASS && TM >= 0) {
            double xr[3];
            rotated_point(P, xf[r], xr);
            RowMoments<TM> row;
#pragma unroll
            for (int k = 0; k < W; k++) row.add(md, xr, yx[r][k], yy[r][k], yz[r][k], k < n[r]);
            row.finish(acc, P, xf[r], xr);
            continue;
        }
        double s[W], yc[W][3];
        double smin = 1e300;
#pragma unroll
        for (int k = 0; k < W; k++) {
            yc[k][0] = (double)yx[r][k] - P.c[0];
            yc[k][1] = (double)yy[r][k] - P.c[1];
            yc[k][2] = (double)yz[r][k] - P.c[2];
            const double r0 = yc[k][0] - xrc[0], r1 = yc[k][1] - xrc[1], r2 = yc[k][2] - xrc[2];
            const double sk = fma(r2, r2, fma(r1, r1, r0 * r0));
            s[k] = (k < n[r]) ? sk : 1e300;
            smin = fmin(smin, s[k]);
        }
        // model known at compile time (TM >= 0): no run-time model tests inside the pair loop
        const bool normal = (TM >= 0) ? (TM == 0) : (md.is_normal != 0);
        const double lp_max = (TM >= 0 || md.is_normal || md.vpd_int) ? 0.0 : log_prob(md, smin);
        double Z = 0, G = 0, Gs = 0, Gy[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < W; k++) {
            const double inv_vs = normal ? 0.0 : fast_rcp(md.v + s[k]);
            const double e = rel_likelihood<TM>(md, s[k], smin, lp_max, inv_vs);
            Z += e;
            const double gk = normal ? e : e * (md.vpd * inv_vs);
            G += gk;
            Gs = fma(gk, (k < n[r]) ? s[k] : 0.0, Gs);  // (0 * 1e300 is 0 already; the select keeps inf/NaN models out)
            Gy[0] = fma(gk, yc[k][0], Gy[0]);
            Gy[1] = fma(gk, yc[k][1], Gy[1]);
            Gy[2] = fma(gk, yc[k][2], Gy[2]);
        }
        const double x2 = fma(xrc[2], xrc[2], fma(xrc[1], xrc[1], xrc[0] * xrc[0]));
        const double Gyy = fma(-x2, G, fma(2.0, fma(xrc[2], Gy[2], fma(xrc[1], Gy[1], xrc[0] * Gy[0])), Gs));
        row_finish(acc, P, xf[r], Z, G, Gs, Gyy, Gy);
    }
    // (an in-kernel last-block fold was measured and removed: its register footprint cost this kernel more than
    //  the separate fold kernel does: 101.6 us vs 67.6 + 17.2 us at the time; the fold kernel is 4.4 us now)
    block_reduce_store<BLOCK, (ONEPASS && TM >= 0)>(acc, partials);  // the lean form is worth six workgroups per CU
}

// fold partials[19][nblocks] -> sums[19].  One 256-thread block PER SUM (grid = 19): every thread issues its
// (up to 8 per trip) loads of a contiguous row back to back, so the fold costs about one memory round trip;
// then a wave shuffle tree and a fixed-order LDS combine: deterministic summation order, no atomics.
__global__ __launch_bounds__(kBlock) void reduce_partials_kernel(const double *__restrict__ partials, int nblocks,
                                                                 double *__restrict__ sums)
{
    __shared__ double sh[kBlock / 64];
    const double *row = partials + (size_t)blockIdx.x * nblocks;
    double v = 0.0;
    for (int b0 = 0; b0 < nblocks; b0 += 8 * kBlock) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int b = b0 + u * kBlock + threadIdx.x;
            t[u] = (b < nblocks) ? row[b] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) v += t[u];
    }
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double x = sh[0];
        for (int w = 1; w < kBlock / 64; w++) x += sh[w];
        sums[blockIdx.x] = x;
    }
}

// ProbabilisticWeights::updateWeights on caller-supplied squared errors (probabilistic_weights.hpp:48-105):
// one lane per CSR row, the reference's exact formula (lp, row max, mll, exp(lp - mll) [* (v+d)/(v+s)]).
__global__ void weights_from_errors_kernel(const int *__restrict__ row_ptr, int64_t n_rows,
                                           const double *__restrict__ s, Model md, double *__restrict__ w)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const int b = row_ptr[i], e = row_ptr[i + 1];
    if (b >= e) return;
    double max_lp = -INFINITY;
    for (int k = b; k < e; k++) {
        const double lp = log_prob(md, s[k]);
        max_lp = lp > max_lp ? lp : max_lp;
    }
    double z = 0;
    for (int k = b; k < e; k++) z += exp(log_prob(md, s[k]) - max_lp);
    const double mll = log(z) + max_lp;
    for (int k = b; k < e; k++) {
        double wk = exp(log_prob(md, s[k]) - mll);
        if (!md.is_normal) wk *= md.vpd / (md.v + s[k]);
        w[k] = wk;
    }
}

// K4: x <- float(R x + t), f64 arithmetic summed left to right, f32 store, in place
// (pcl::transformPointCloud semantics, src/prob_point_cloud_registration.cc:110-112).
// The w lane (original index) is preserved.
__global__ void transform_kernel(float4 *__restrict__ pts, int n, Pose P)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = pts[i];
    const double x = p.x, y = p.y, z = p.z;
    p.x = (float)(((P.R[0] * x + P.R[1] * y) + P.R[2] * z) + P.t[0]);
    p.y = (float)(((P.R[3] * x + P.R[4] * y) + P.R[5] * z) + P.t[1]);
    p.z = (float)(((P.R[6] * x + P.R[7] * y) + P.R[8] * z) + P.t[2]);
    pts[i] = p;
}

// ---------------------------------------------------------------------------------------------
// pcl::VoxelGrid centroid down-sampling (the step before the path: src/prob_point_cloud_registration.cc:24-41).
// One fixed definition (the CPU checker in the test tree restates it bit for bit): voxel index from float floor(p * inv),
// points of one voxel added in ascending original index (the radix sort is stable), float sums, float division.
// ---------------------------------------------------------------------------------------------
struct VoxelDesc {
    float inv;
    int min_b[3];
    int mul[3];
};
constexpr unsigned kVoxelInvalid = 0xFFFFFFFFu;  // non-finite points: sorted to the end and dropped

__global__ void voxel_key_kernel(const float4 *__restrict__ pts, int n, VoxelDesc v, unsigned *__restrict__ keys,
                                 int *__restrict__ vals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    unsigned key = kVoxelInvalid;
    if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
        const int ix = (int)(floorf(__fmul_rn(p.x, v.inv)) - (float)v.min_b[0]);
        const int iy = (int)(floorf(__fmul_rn(p.y, v.inv)) - (float)v.min_b[1]);
        const int iz = (int)(floorf(__fmul_rn(p.z, v.inv)) - (float)v.min_b[2]);
        key = (unsigned)(ix * v.mul[0] + iy * v.mul[1] + iz * v.mul[2]);
    }
    keys[i] = key;
    vals[i] = i;
}

// head[i] = 1 where a voxel's run starts in the sorted keys
__global__ void voxel_head_kernel(const unsigned *__restrict__ keys, int n, int *__restrict__ head)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned k = keys[i];
    head[i] = (k != kVoxelInvalid && (i == 0 || keys[i - 1] != k)) ? 1 : 0;
}

// one lane per voxel run: sequential float sum in sorted (= ascending original index) order
__global__ void voxel_centroid_kernel(const float4 *__restrict__ pts, const unsigned *__restrict__ keys,
                                      const int *__restrict__ order, const int *__restrict__ head,
                                      const int *__restrict__ slot, int n, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !head[i]) return;
    const unsigned k = keys[i];
    float cx = 0.f, cy = 0.f, cz = 0.f;
    int e = i;
    while (e < n && keys[e] == k) {
        const float4 p = pts[order[e]];
        cx = __fadd_rn(cx, p.x);
        cy = __fadd_rn(cy, p.y);
        cz = __fadd_rn(cz, p.z);
        e++;
    }
    const float cnt = (float)(e - i);
    float *o = out + (size_t)slot[i] * 3;
    o[0] = __fdiv_rn(cx, cnt);
    o[1] = __fdiv_rn(cy, cnt);
    o[2] = __fdiv_rn(cz, cnt);
}

// ---------------------------------------------------------------------------------------------
// calculateMSE (utilities.hpp:16-26; despite the name: the MEAN EUCLIDEAN DISTANCE of index-paired points, float
// distance as pcl::euclideanDistance, double sum).  a is either a plain cloud (pair i <-> b[i]) or the handle's
// sorted source, whose w lane holds the caller's index (pair r <-> b[w(r)]).  partials[block] = block sum.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void mean_distance_kernel(const float4 *__restrict__ a, int n,
                                                               const float4 *__restrict__ b, int a_is_sorted_source,
                                                               double *__restrict__ partials)
{
    __shared__ double sh[kBlock / 64];
    double acc = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float4 p = a[i];
        const float4 q = b[a_is_sorted_source ? __float_as_int(p.w) : i];
        const float dx = __fsub_rn(p.x, q.x), dy = __fsub_rn(p.y, q.y), dz = __fsub_rn(p.z, q.z);
        // sqrtf, not __fsqrt_rn: the intrinsic maps to the 1-ulp hardware sqrt, sqrtf is correctly rounded
        acc += (double)sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double v = sh[0];
        for (int w = 1; w < kBlock / 64; w++) v += sh[w];
        partials[blockIdx.x] = v;
    }
}

// snapshot of the tracked cloud in the caller's index order: dst[w(r)] = a[r] (sorted source) or dst[i] = a[i]
__global__ void snapshot_kernel(const float4 *__restrict__ a, int n, int a_is_sorted_source, float4 *__restrict__ dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = a[i];
    dst[a_is_sorted_source ? __float_as_int(p.w) : i] = p;
}

// ---------------------------------------------------------------------------------------------
// Exact nearest neighbour (k = 1, no radius) for the evaluation metrics of utilities.hpp:28-234
// (averageClosestDistance, sumSquaredError, the robust / median variants: all built on nearestKSearch(…, 1, …)).
// One lane per query: shells of cells of growing Chebyshev radius around the query's (clamped) cell are scanned
// until the best d2 found is no larger than the distance to everything not yet scanned — the gap between the query
// and the faces of the scanned block that are not grid faces (under-estimated by g.eps).  Same float d2 as the
// association (dist2_flann).  Queries far outside the cloud degrade to a full scan, which is still exact.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void nn1_kernel(const float4 *__restrict__ queries, int nq,
                                                     const float4 *__restrict__ tgt,
                                                     const int *__restrict__ cell_start, GridDesc g,
                                                     float *__restrict__ d2_out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nq) return;
    const float4 q = queries[i];
    const float qp[3] = {q.x, q.y, q.z};
    const float inv[3] = {g.inv_hx, g.inv_h, g.inv_h};
    int c0[3];
    for (int a = 0; a < 3; a++) c0[a] = clampi(cell_coord(qp[a], g.org[a], inv[a], g.n[a]), 0, g.n[a] - 1);
    const float hx = g.h / (float)g.xr;
    const float edge[3] = {hx, g.h, g.h};
    float best = INFINITY;
    auto scan_cells = [&](int base, int xa, int xb) {
        const int b = cell_start[base + xa], e = cell_start[base + xb + 1];
        for (int p = b; p < e; p++) best = fminf(best, dist2_flann(q, tgt[p]));
    };
    for (int rho = 0;; rho++) {
        int lo[3], hi[3];
        bool whole = true;
        for (int a = 0; a < 3; a++) {
            lo[a] = max(c0[a] - rho, 0);
            hi[a] = min(c0[a] + rho, g.n[a] - 1);
            whole = whole && lo[a] == 0 && hi[a] == g.n[a] - 1;
        }
        for (int z = lo[2]; z <= hi[2]; z++)
            for (int y = lo[1]; y <= hi[1]; y++) {
                const int base = (z * g.n[1] + y) * g.n[0];
                const bool shell_row = (z == c0[2] - rho) || (z == c0[2] + rho) || (y == c0[1] - rho) || (y == c0[1] + rho);
                if (shell_row || rho == 0) {
                    scan_cells(base, lo[0], hi[0]);
                } else {
                    if (c0[0] - rho >= 0) scan_cells(base, c0[0] - rho, c0[0] - rho);
                    if (c0[0] + rho <= g.n[0] - 1) scan_cells(base, c0[0] + rho, c0[0] + rho);
                }
            }
        if (whole) break;
        // everything not scanned yet lies beyond a face of the block that is not a face of the grid
        float gap = INFINITY;
        for (int a = 0; a < 3; a++) {
            const float f = qp[a] - g.org[a];
            if (lo[a] > 0) gap = fminf(gap, f - (float)lo[a] * edge[a]);
            if (hi[a] < g.n[a] - 1) gap = fminf(gap, (float)(hi[a] + 1) * edge[a] - f);
        }
        gap -= g.eps;
        if (gap > 0.f && best <= gap * gap * 0.999999f) break;
    }
    d2_out[i] = best;
}

}  // namespace dev
}  // namespace ppcr
