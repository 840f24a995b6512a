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

namespace ppcr {
namespace dev {

constexpr int kNSums = 19;
constexpr int kBlock = 256;

struct GridDesc {
    float org[3];
    float inv_h;
    int n[3];
    int ncells;
};

struct Pose {  // y ~ R x + t ; c = fixed origin of the moments
    double R[9];
    double t[3];
    double c[3];
};

struct Model {  // ProbabilisticWeights constants (probabilistic_weights.hpp:30-46)
    int is_normal;
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
    int cx = clampi(cell_coord(p.x, g.org[0], g.inv_h, g.n[0]), 0, g.n[0] - 1);
    int cy = clampi(cell_coord(p.y, g.org[1], g.inv_h, g.n[1]), 0, g.n[1] - 1);
    int cz = clampi(cell_coord(p.z, g.org[2], g.inv_h, g.n[2]), 0, g.n[2] - 1);
    keys[i] = (unsigned)((cz * g.n[1] + cy) * g.n[0] + cx);
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
    c.cx = cell_coord(q.x, g.org[0], g.inv_h, g.n[0]);
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
    const int x0 = max(c.cx - 1, 0), x1 = min(c.cx + 1, g.n[0] - 1);
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

template <int M>
__global__ __launch_bounds__(kBlock) void nn_topm_kernel(const float4 *__restrict__ src, int ns,
                                                         const float4 *__restrict__ tgt,
                                                         const int *__restrict__ cell_start, GridDesc g,
                                                         float r2, int m, int *__restrict__ nbr,
                                                         int *__restrict__ cnt)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= ns) return;
    const float4 q = src[i];
    unsigned long long key[M];
    int pos[M];
#pragma unroll
    for (int j = 0; j < M; j++) {
        key[j] = ~0ull;
        pos[j] = -1;
    }
    for_each_candidate(q, g, cell_start, tgt, [&](int p, float4 t) {
        const float d2 = dist2_flann(q, t);
        if (d2 < r2) {
            unsigned long long k = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)__float_as_int(t.w);
            if (k < key[M - 1]) {
                int pp = p;
                // bubble the new key through the ascending list; the largest falls off the end
#pragma unroll
                for (int j = 0; j < M; j++) {
                    const bool lt = k < key[j];
                    const unsigned long long kk = lt ? key[j] : k;
                    const int pk = lt ? pos[j] : pp;
                    key[j] = lt ? k : key[j];
                    pos[j] = lt ? pp : pos[j];
                    k = kk;
                    pp = pk;
                }
            }
        }
    });
    int c = 0;
#pragma unroll
    for (int j = 0; j < M; j++) {
        if (j < m) {
            nbr[(size_t)j * ns + i] = pos[j];
            c += (pos[j] >= 0) ? 1 : 0;
        }
    }
    cnt[i] = c;
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
    unsigned long long s = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) s += (unsigned)cnt[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(total, s);
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

// K23 (hot path): one lane per source row, grid-stride; per row two sweeps over its <= m
// neighbours (min s, then the softmax sums), per lane 19 f64 accumulators, then wave shuffle
// reduction -> LDS across the block's waves -> one partial vector per block.  A second tiny
// kernel folds the per-block partials in a fixed order (deterministic, no float atomics).
template <class A>
__global__ __launch_bounds__(kBlock) void accumulate_kernel(A a, const float4 *__restrict__ src,
                                                            const float4 *__restrict__ tgt, int ns, Pose P,
                                                            Model md, double *__restrict__ partials)
{
    double acc[kNSums];
#pragma unroll
    for (int j = 0; j < kNSums; j++) acc[j] = 0.0;
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
            const double e = exp(log_prob(md, s) - lp_max);
            Z += e;
            const double gk = md.is_normal ? e : e * (md.vpd / (md.v + s));
            const double yc0 = (double)y.x - P.c[0], yc1 = (double)y.y - P.c[1], yc2 = (double)y.z - P.c[2];
            G += gk;
            Gs = fma(gk, s, Gs);
            Gy[0] = fma(gk, yc0, Gy[0]);
            Gy[1] = fma(gk, yc1, Gy[1]);
            Gy[2] = fma(gk, yc2, Gy[2]);
            Gyy = fma(gk, yc0 * yc0 + yc1 * yc1 + yc2 * yc2, Gyy);
        }
        const double iz = 1.0 / Z;  // w_k = g_k / Z
        const double Wi = G * iz;
        const double xc[3] = {(double)xf.x - P.c[0], (double)xf.y - P.c[1], (double)xf.z - P.c[2]};
        const double wy[3] = {Gy[0] * iz, Gy[1] * iz, Gy[2] * iz};
        acc[0] += Wi;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            acc[1 + d] = fma(Wi, xc[d], acc[1 + d]);
            acc[4 + d] += wy[d];
#pragma unroll
            for (int b = 0; b < 3; b++) acc[7 + 3 * d + b] = fma(xc[d], wy[b], acc[7 + 3 * d + b]);
        }
        acc[16] += Gs * iz;
        acc[17] = fma(Wi, xc[0] * xc[0] + xc[1] * xc[1] + xc[2] * xc[2], acc[17]);
        acc[18] += Gyy * iz;
    }
    __shared__ double sh[kBlock / 64][kNSums];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < kNSums; j++) {
        double v = acc[j];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) sh[wave][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < kNSums) {
        double v = sh[0][threadIdx.x];
        for (int w = 1; w < kBlock / 64; w++) v += sh[w][threadIdx.x];
        partials[(size_t)blockIdx.x * kNSums + threadIdx.x] = v;
    }
}

// fold [nblocks][19] partials -> sums[19]; one block, fixed summation tree
__global__ __launch_bounds__(kBlock) void reduce_partials_kernel(const double *__restrict__ partials,
                                                                 int nblocks, double *__restrict__ sums)
{
    __shared__ double sh[kBlock];
    for (int j = 0; j < kNSums; j++) {
        double v = 0;
        for (int b = threadIdx.x; b < nblocks; b += kBlock) v += partials[(size_t)b * kNSums + j];
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int off = kBlock / 2; off > 0; off >>= 1) {
            if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) sums[j] = sh[0];
        __syncthreads();
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

}  // namespace dev
}  // namespace ppcr
