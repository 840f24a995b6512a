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
#include "ppcr_device.hip.h"

namespace ppcr {
namespace dev {

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

// Source ordering: bricks of (2^bxs) x ~4 x ~4 cells visited boustrophedon (x snakes per brick row, y snakes per brick
// plane), cells x-fastest inside a brick.  Default bxs = 0: ~4x4 columns of cells in (y,z) walked along x.  Any 256
// consecutive queries then sit in one or two ADJACENT bricks, so the cell bounding box of a workgroup — and with it the
// target halo it stages into LDS (nn_fast_kernel) — stays small.  Only the order of the source changes, never a result.
// The n cells of an axis are dealt EVENLY to round(n / 4) bricks (widths 3, 4 or 5: b = c * nb / n), not cut into fours
// with a remainder: a grid of 41 x 41 cells (250k points at the benchmark density) used to end in a column ONE cell
// wide, whose 256-query blocks stretch over ~17 cells in x, never fit a halo, and sent ~5 % of the workgroups to the
// cleanup kernel in EVERY iteration (56 us of a 92 us iteration; 64-cell grids such as the 1M benchmark never showed it).
__host__ __device__ inline int brick_count(int ncells) { return ncells + 2 >= 4 ? (ncells + 2) / 4 : 1; }
__global__ void brick_key_kernel(const float4 *__restrict__ pts, int n, GridDesc g,
                                 unsigned *__restrict__ keys, int *__restrict__ vals, int bxs)
{
    // bxs = log2 of the brick's x extent in cells (2: 4 cells; 0: 1 cell, i.e. (y,z) columns walked along x: 256
    // consecutive queries then span ~4 cells in x instead of up to 8)
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = pts[i];
    // bricks are measured in whole cells: x slices are folded back to cells first
    const int cx = clampi(cell_coord(p.x, g.org[0], g.inv_hx, g.n[0]), 0, g.n[0] - 1) >> g.xr_shift;
    const int cy = clampi(cell_coord(p.y, g.org[1], g.inv_h, g.n[1]), 0, g.n[1] - 1);
    const int cz = clampi(cell_coord(p.z, g.org[2], g.inv_h, g.n[2]), 0, g.n[2] - 1);
    const int nbx = ((g.n[0] >> g.xr_shift) + (1 << bxs) - 1) >> bxs, nby = brick_count(g.n[1]), nbz = brick_count(g.n[2]);
    const int bx = cx >> bxs, by = (cy * nby) / g.n[1], bz = (cz * nbz) / g.n[2];
    const int ly = cy - (by * g.n[1] + nby - 1) / nby, lz = cz - (bz * g.n[2] + nbz - 1) / nbz;  // 0 .. 4 inside the brick
    const int byy = (bz & 1) ? nby - 1 - by : by;
    const int row = bz * nby + byy;
    const int bxx = (row & 1) ? nbx - 1 - bx : bx;
    const unsigned brick = (unsigned)(row * nbx + bxx);
    keys[i] = (brick << (6 + bxs)) | (unsigned)((lz << (3 + bxs)) | (ly << bxs) | (cx & ((1 << bxs) - 1)));
    vals[i] = i;
}

// Source ordering for MULTI-LEVEL searches: the Hilbert curve over the whole cells of the finest level.  Clouds whose
// density varies a hundredfold have no brick size that suits them all; along a Hilbert curve ANY 256 consecutive points
// form one connected, compact region at whatever the local density is — a few fine cells inside a dense blob, many in a
// sparse stretch — so every block's halo is small at the level its cut-offs select.  (Skilling's transpose form of the
// 3-D Hilbert index: `bits` bits per axis, 3 * bits <= 30.)
__global__ void hilbert_key_kernel(const float4 *__restrict__ pts, int n, GridDesc g, unsigned *__restrict__ keys,
                                   int *__restrict__ vals, int bits, int shift)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    unsigned X[3];
    X[0] = (unsigned)(clampi(cell_coord(p.x, g.org[0], g.inv_hx, g.n[0]), 0, g.n[0] - 1) >> g.xr_shift) >> shift;
    X[1] = (unsigned)clampi(cell_coord(p.y, g.org[1], g.inv_h, g.n[1]), 0, g.n[1] - 1) >> shift;
    X[2] = (unsigned)clampi(cell_coord(p.z, g.org[2], g.inv_h, g.n[2]), 0, g.n[2] - 1) >> shift;
    const unsigned M = 1u << (bits - 1);
    for (unsigned Q = M; Q > 1; Q >>= 1) {  // inverse undo
        const unsigned P = Q - 1;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            if (X[a] & Q) {
                X[0] ^= P;
            } else {
                const unsigned t = (X[0] ^ X[a]) & P;
                X[0] ^= t;
                X[a] ^= t;
            }
        }
    }
    X[1] ^= X[0];  // Gray encode
    X[2] ^= X[1];
    unsigned t = 0;
    for (unsigned Q = M; Q > 1; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1;
    X[0] ^= t, X[1] ^= t, X[2] ^= t;
    unsigned key = 0;  // interleave: bit b of X[0] is the most significant of its triple
    for (int b = bits - 1; b >= 0; b--) key = (key << 3) | (((X[0] >> b) & 1u) << 2) | (((X[1] >> b) & 1u) << 1) | ((X[2] >> b) & 1u);
    keys[i] = key;
    vals[i] = i;
}

// multi-level search: inv[original index of the base level's p-th point] = p, then to_base[q] = inv[original index of
// another level's q-th point] (the association is kept in base-level positions whichever level found the neighbour)
__global__ void level_inverse_kernel(const float4 *__restrict__ base_sorted, int n, int *__restrict__ inv)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) inv[__float_as_int(base_sorted[p].w)] = p;
}
__global__ void level_to_base_kernel(const float4 *__restrict__ level_sorted, int n, const int *__restrict__ inv, int *__restrict__ to_base)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) to_base[q] = inv[__float_as_int(level_sorted[q].w)];
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
// Lane i owns the cells between two consecutive sorted keys.  A few empty cells it fills itself; a long run of them — the
// empty space between the parts of a scan, 10^5 cells and more on the finer levels of a multi-level search — is filled by
// the whole workgroup, 256 cells per step (one lane walking it alone made this kernel 0.43 ms per level on a 200k-point
// scan, nine tenths of the target's set-up).
constexpr int kCellRunLong = 32;
__global__ __launch_bounds__(kBlock) void cell_start_kernel(const unsigned *__restrict__ keys_sorted, int n, int ncells,
                                                            int *__restrict__ cell_start)
{
    __shared__ int s_runs, s_first[kBlock], s_last[kBlock], s_value[kBlock];
    if (threadIdx.x == 0) s_runs = 0;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n) {
        const int prev = (i == 0) ? -1 : (int)keys_sorted[i - 1];
        const int cur = (i == n) ? ncells : (int)keys_sorted[i];
        if (cur - prev <= kCellRunLong) {
            for (int c = prev + 1; c <= cur; c++) cell_start[c] = i;
        } else {
            const int k = atomicAdd(&s_runs, 1);
            s_first[k] = prev + 1, s_last[k] = cur, s_value[k] = i;
        }
    }
    __syncthreads();
    const int runs = s_runs;
    for (int k = 0; k < runs; k++) {
        const int last = s_last[k], v = s_value[k];
        for (int c = s_first[k] + (int)threadIdx.x; c <= last; c += kBlock) cell_start[c] = v;
    }
}

// Occupancy of the grid's WHOLE cells (x slices folded back) as the POINTS see it: hist[occ_bin(count)] += count
// for every cell, i.e. how many points live in a cell of that many.  The host takes the median over the points — the
// cell the typical row finds itself in (q + 1 for a uniform cloud of q points per cell; unmoved by a few blobs a hundred
// times denser than the rest, which weighted means follow) — and sizes the first-pass search radius for THAT row: rows of
// much sparser and much denser neighbourhoods are what the second pass is for.
constexpr int kOccBins = 512;  // counts below 256 one per bin, above in steps of 16 (up to 4336 points per cell)
__host__ __device__ inline int occ_bin(int cnt) { return cnt < 256 ? cnt : min(256 + (cnt - 256) / 16, kOccBins - 1); }
__host__ inline double occ_bin_value(int b) { return b < 256 ? (double)b : 256.0 + 16.0 * (b - 256) + 8.0; }
__global__ __launch_bounds__(kBlock) void cell_occupancy_kernel(const int *__restrict__ cell_start, GridDesc g, unsigned long long *__restrict__ hist)
{
    __shared__ unsigned s_hist[kOccBins];
    for (int b = threadIdx.x; b < kOccBins; b += kBlock) s_hist[b] = 0;
    __syncthreads();
    const int ncx = g.n[0] >> g.xr_shift;
    const long long nwhole = (long long)ncx * g.n[1] * g.n[2];
    for (long long w = (long long)blockIdx.x * kBlock + threadIdx.x; w < nwhole; w += (long long)gridDim.x * kBlock) {
        const long long row = w / ncx;
        const int cx = (int)(w - row * ncx);
        const long long base = row * g.n[0] + ((long long)cx << g.xr_shift);
        const int cnt = cell_start[base + g.xr] - cell_start[base];
        if (cnt > 0) atomicAdd(&s_hist[occ_bin(cnt)], (unsigned)cnt);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < kOccBins; b += kBlock)
        if (s_hist[b]) atomicAdd(hist + b, (unsigned long long)s_hist[b]);
}

__global__ __launch_bounds__(kBlock) void reduce_solve_kernel(FoldSolve fs)
{
    if (loop_aborted(fs.loop.st)) {  // an earlier launch handed the iteration to the host: step aside
        if (blockIdx.x == 0 && threadIdx.x == 0) publish_skipped(fs);
        return;
    }
    (void)fold_and_solve_block(fs, (int)blockIdx.x);
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
    block_reduce_store(acc, partials, gridDim.x, blockIdx.x);
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
// what a lane holds of its rows [base + r * BLOCK | r < ROWS] after the first memory round trip: header and neighbour slots
template <int W, int ROWS>
struct EllRowsHead {
    int n[ROWS];
    float4 xf[ROWS];
    int idx[ROWS][W];
};
template <int W, int ROWS, int BLOCK>
__device__ __forceinline__ void load_ell_rows_head(EllRowsHead<W, ROWS> &h, int base, const int *__restrict__ nbr,
                                                   const int *__restrict__ cnt, const float4 *__restrict__ src, int ns, int width)
{
    // width = slots the association really has per row (<= W): slots beyond it do not exist in nbr
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        const int i = base + r * BLOCK;
        const bool ok = i < ns;
        h.n[r] = ok ? cnt[i] : 0;
        h.xf[r] = src[ok ? i : 0];
#pragma unroll
        for (int k = 0; k < W; k++) h.idx[r][k] = (ok && k < width) ? nbr[(size_t)k * ns + i] : 0;  // slots >= cnt: stale
    }
}
// the neighbours' coordinates of a lane's rows (second round trip: the gathers)
template <int W, int ROWS>
struct EllRowsPoints {
    float yx[ROWS][W], yy[ROWS][W], yz[ROWS][W];
};
template <int W, int ROWS>
__device__ __forceinline__ void gather_ell_rows(EllRowsPoints<W, ROWS> &p, const EllRowsHead<W, ROWS> &h, const float4 *__restrict__ tgt)
{
#pragma unroll
    for (int r = 0; r < ROWS; r++)
#pragma unroll
        for (int k = 0; k < W; k++) {
            const float4 y = tgt[(k < h.n[r]) ? h.idx[r][k] : 0];  // slot 0 of the target for unused slots: masked below
            p.yx[r][k] = y.x;
            p.yy[r][k] = y.y;
            p.yz[r][k] = y.z;
        }
}
// the rows of one lane added to acc: the row arithmetic, from registers
template <int W, int ROWS, int BLOCK, int TM, bool ONEPASS>
__device__ __forceinline__ void finish_ell_rows(RowAcc &acc, const EllRowsHead<W, ROWS> &h, const EllRowsPoints<W, ROWS> &pts,
                                                const Pose &P, const Model &md)
{
    const int(&n)[ROWS] = h.n;
    const float4(&xf)[ROWS] = h.xf;
    const float(&yx)[ROWS][W] = pts.yx;
    const float(&yy)[ROWS][W] = pts.yy;
    const float(&yz)[ROWS][W] = pts.yz;
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
        if constexpr (ONEPASS) {
            double xr[3];
            rotated_point(P, xf[r], xr);
            RowMoments<TM> row;
            row.begin(md);
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
}

// ONE-PASS rows wider than 10 slots, in two helpings: the first half of the row's neighbours is gathered and added, then
// the second — pair by pair in slot order, as finish_ell_rows adds them: the same sums.  All 20 slots of the command
// line's default width at once are 60 coordinate registers on top of the 20 slots: 110 VGPRs for accumulate_ell_kernel
// (four workgroups per CU), 92 SPILLED registers in inner_steps_kernel's 96-register budget.
template <int W>
constexpr bool kEllHelpings = W > 10;
template <int W, int TM>
__device__ __forceinline__ void finish_ell_row_helpings(RowAcc &acc, const EllRowsHead<W, 1> &h, const float4 *__restrict__ tgt,
                                                        const Pose &P, const Model &md)
{
    constexpr int CH = (W + 1) / 2;
    const int n = h.n[0];
    double xr[3];
    rotated_point(P, h.xf[0], xr);
    RowMoments<TM> row;
    row.begin(md);
#pragma unroll
    for (int c0 = 0; c0 < W; c0 += CH) {
        float yx[CH], yy[CH], yz[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const float4 y = tgt[(c0 + j < W && c0 + j < n) ? h.idx[0][(c0 + j < W) ? c0 + j : 0] : 0];  // slot 0 of the target for unused slots: masked below
            yx[j] = y.x, yy[j] = y.y, yz[j] = y.z;
        }
#pragma unroll
        for (int j = 0; j < CH; j++)
            if (c0 + j < W) row.add(md, xr, yx[j], yy[j], yz[j], c0 + j < n);
        if (c0 + CH < W) __builtin_amdgcn_sched_barrier(0);  // (the second helping's gathers stay behind the first's arithmetic)
    }
    if (n != 0) row.finish(acc, P, h.xf[0], xr);
}

// the rows [base + r * BLOCK | r < ROWS] of one lane, added to acc
template <int W, int ROWS, int BLOCK, int TM, bool ONEPASS>
__device__ __forceinline__ void accumulate_ell_rows(RowAcc &acc, int base, const int *__restrict__ nbr,
                                                    const int *__restrict__ cnt, const float4 *__restrict__ src,
                                                    const float4 *__restrict__ tgt, int ns, const Pose &P,
                                                    const Model &md, int width)
{
    EllRowsHead<W, ROWS> h;
    load_ell_rows_head<W, ROWS, BLOCK>(h, base, nbr, cnt, src, ns, width);
    if constexpr (ONEPASS && ROWS == 1 && kEllHelpings<W>) {
        finish_ell_row_helpings<W, TM>(acc, h, tgt, P, md);
    } else {
        EllRowsPoints<W, ROWS> pts;
        gather_ell_rows<W, ROWS>(pts, h, tgt);
        finish_ell_rows<W, ROWS, BLOCK, TM, ONEPASS>(acc, h, pts, P, md);
    }
}

template <int W, int ROWS, int BLOCK, int TM = -1, bool ONEPASS = false>
__global__ __launch_bounds__(BLOCK) void accumulate_ell_kernel(const int *__restrict__ nbr,
                                                                const int *__restrict__ cnt,
                                                                const float4 *__restrict__ src,
                                                                const float4 *__restrict__ tgt, int ns, Pose P,
                                                                Model md, double *__restrict__ partials, int width,
                                                                const LoopState *loop_st)
{
    if (loop_aborted(loop_st)) return;
    RowAcc acc;
#pragma unroll
    for (int j = 0; j < kNSums; j++) acc.a[j] = 0.0;
    // (XCD-aware tile map; the partial sums stay in launch order: slot = blockIdx)
    accumulate_ell_rows<W, ROWS, BLOCK, TM, ONEPASS>(acc, xcd_block((int)blockIdx.x, (int)gridDim.x) * (BLOCK * ROWS) + threadIdx.x, nbr, cnt,
                                                     src, tgt, ns, P, md, width);
    // (an in-kernel last-block fold was measured and removed: its register footprint cost this kernel more than
    //  the separate fold kernel does: 101.6 us vs 67.6 + 17.2 us at the time; the fold kernel is 4.4 us now)
    block_reduce_store<BLOCK, ONEPASS>(acc, partials, gridDim.x, blockIdx.x);  // the lean form is worth six workgroups per CU
}

// ---------------------------------------------------------------------------------------------
// The inner loop's later steps, paced by the device (ProbPointCloudRegistrationIteration::solve iterated to
// function_tolerance, cc:96-100).  Step 1 of an outer iteration rides in the association (K23 folded into K1, or
// accumulate_ell_kernel) and its fold-and-solve lane decides whether the loop is over (LoopCtl).  This ONE launch,
// enqueued behind it, holds the next `n_steps` IRLS steps: per step G workgroups redo K23 at the pose the previous step
// solved (rows dealt G-strided in tiles of 256) and kNSums workgroups fold and solve.  When the loop is over (the common case at
// its very first look) every remaining workgroup returns at once.
// ---------------------------------------------------------------------------------------------
#ifndef PPCR_INNER_LEAN
#define PPCR_INNER_LEAN 1
#endif
#ifndef PPCR_INNER_PER_CU
#define PPCR_INNER_PER_CU 5
#endif
// K23 workgroups of one device step that are resident together (with the step's fold slots): five per CU, or four
constexpr int kInnerStepWgs = PPCR_INNER_PER_CU >= 5 ? 1200 : 1000;
constexpr int kMaxDevSteps = 8;
constexpr int kInnerMaxG = 2048;  // upper bound of the K23 workgroups per step (rows are dealt G-strided in tiles of 256)
// workgroups a step carries behind its G K23 workgroups: kNSums fold workgroups + idle ones, so that a step is a multiple
// of eight workgroups and workgroup r of EVERY step runs on XCD r % 8 (the tile map below relies on it)
constexpr int kInnerFoldSlots = 24;
static_assert(kInnerFoldSlots >= kNSums && kInnerFoldSlots % 8 == 0, "a step's workgroups: a multiple of the XCD count");
// what the launch reads from DEVICE memory instead of taking it as kernel arguments (as by-value arguments FoldSolve's
// ~40 words sat in SGPRs through the K23 role and pushed its uniforms into VGPRs: 138 VGPRs, three waves per SIMD)
struct InnerConst {
    FoldSolve fs;        // partials = [kNSums][G], nslots = G, loop.first = 0; seq, mbox, handed_over, last_dev: per launch / step
    unsigned *flags;     // [n_steps][G]: == seq once that workgroup's partial sums of this launch are in place
    unsigned long long *step_done;  // seq * kMaxDevSteps + (device steps of this launch that have been solved), as a 64-bit
                                    // word: the product outgrows 32 bits long before seq does (pooled handles live for days)
    HostMailbox *mbox_ring;
    int mbox_slots;
    unsigned *ovf_state;
    int G;
};
struct InnerArgs {
    const int *nbr, *cnt;
    const float4 *src, *tgt;
    int ns, width;
    Model md;
    const InnerConst *ic;
    unsigned seq;
    int ovf_index, n_steps;
};
// the fold role (inlined: a called function is compiled without the kernel's register budget and took 210 VGPRs)
__device__ __forceinline__ void inner_fold_role(const InnerConst *ic, unsigned seq, int ovf_index, int u, int row, int n_steps)
{
    const int G = ic->G;
    // every partial of this step must be in place
    for (int k = threadIdx.x; k < G; k += kBlock)
        while (__hip_atomic_load(&ic->flags[(size_t)u * G + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq)
            __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // kNSums of these per step: the fold reads the partials plainly
    FoldSolve fs = ic->fs;
    fs.seq = seq;
    fs.mbox = ic->mbox_ring + (seq % (unsigned)ic->mbox_slots);
    fs.handed_over = ic->ovf_state + ovf_index;
    fs.loop.first = 0;
    fs.loop.last_dev = (u == n_steps - 1) ? 1 : 0;
    if (fold_and_solve_block<PPCR_INNER_LEAN != 0, 8>(fs, row)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // one per step: pose, loop state
        __hip_atomic_store(ic->step_done, (unsigned long long)seq * kMaxDevSteps + (unsigned)(u + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// Workgroups take their role from blockIdx: [step][G K23 workgroups, kNSums fold workgroups].  A workgroup only ever
// waits for workgroups with SMALLER indices, which the dispatcher has started before it (workgroups of a launch are
// dispatched in index order on every XCD) — running or done: no co-residency is assumed and several such kernels (other
// handles) can share the chip.  No read-modify-write atomics on shared counters: 3000 workgroups drawing tickets from
// one counter cost ~150 us (same-address atomics serialise at the memory side); completion is one flag word per
// workgroup, stamped with the launch's sequence number so that nothing has to be cleared between launches.
// Cross-workgroup traffic inside the launch goes through agent-scope ATOMIC loads and stores (the chip's eight L2s are
// not coherent with each other for ordinary accesses; an acquire / release FENCE invalidates / writes back a whole L2,
// which K23's target gathers live in): waiting workgroups poll with relaxed loads — polling with acquire loads, tried
// first, took a step from ~35 us to ~370 us.
template <int W, int TM, bool ONEPASS>
__global__ __launch_bounds__(kBlock, ONEPASS ? PPCR_INNER_PER_CU : 3) void inner_steps_kernel(InnerArgs a)
{
    const InnerConst *const ic = a.ic;
    LoopState *const st = ic->fs.loop.st;
    const int G = ic->G;
    const int per_step = G + kInnerFoldSlots;
    const int u = (int)blockIdx.x / per_step, r = (int)blockIdx.x % per_step;
    if (r >= G + kNSums) return;  // (padding: see kInnerFoldSlots)
    const unsigned seq = a.seq;
    // The common case: step 1 (an earlier launch) ended the loop, all these workgroups have nothing to do and their
    // number times their lifetime is what the launch costs — an ordinary cached load is enough for a value written
    // before the launch (the agent-scope loads below go to the memory side).
    if (*reinterpret_cast<const volatile unsigned *>(&st->finished) == seq) return;
    auto over = [&]() {  // the inner loop of THIS outer iteration has ended, or the device gave the iteration up
        return __hip_atomic_load(&st->finished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == seq || loop_aborted(st);
    };
    if (u > 0) {  // the previous device step must have been solved (step 1 was: it ran in an earlier launch)
        // (or the loop ended at an earlier step: then step u - 1 never runs and `finished` / `abort` is the news)
        if (threadIdx.x == 0)
            while (__hip_atomic_load(ic->step_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)seq * kMaxDevSteps + (unsigned)u && !over())
                __builtin_amdgcn_s_sleep(16);
        __syncthreads();
    }
    if (over()) return;
    if (ic->fs.dbg && u == 0 && threadIdx.x == 0) {  // diagnostic stamps of the launch's first step
        if (r == 0) ic->fs.dbg[7] = wall_clock64();   // its first K23 workgroup starts
        if (r == G) ic->fs.dbg[6] = wall_clock64();   // its first fold workgroup starts to wait
    }
    if (r >= G) {
        inner_fold_role(ic, seq, a.ovf_index, u, r - G, a.n_steps);
        return;
    }
    // K23 at the pose the previous step solved (written by another workgroup, possibly of this launch).  Uniform: moved
    // to scalar registers, where accumulate_ell_kernel has it as a kernel argument.
    auto uniform = [](double v) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    Pose P;
    {
        const Pose *pose = ic->fs.pose_out;
#pragma unroll
        for (int k = 0; k < 9; k++) P.R[k] = uniform(__hip_atomic_load(&pose->R[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
        for (int k = 0; k < 3; k++) P.t[k] = uniform(__hip_atomic_load(&pose->t[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        P.c[0] = ic->fs.origin.x, P.c[1] = ic->fs.origin.y, P.c[2] = ic->fs.origin.z;
    }
    // tile by tile, exactly as accumulate_ell_kernel does one tile; the running sums of the workgroup live in LDS
    __shared__ double s_acc[kNSums];
    if (threadIdx.x < kNSums) s_acc[threadIdx.x] = 0.0;
    const int ntiles = (a.ns + kBlock - 1) / kBlock;
    // (the next tile's header and neighbour slots are asked for before this tile's gathers and arithmetic: one of a
    //  tile's three dependent round trips runs under the previous tile.  Keeping the next tile's GATHERS in flight as
    //  well was measured: 72 spilled VGPRs even at a 128-register budget, the K23 phase 36 -> 40 us; so was the register
    //  form of the solve at that budget: 70 spills.)
    // XCD-aware tile map (xcd_block): workgroup r runs on XCD r % 8 (a step is a multiple of eight workgroups, G is one
    // too whenever a workgroup walks several tiles), and tile g = r + j G goes to the (g % 8)-th eighth of the rows
    // (wide one-pass rows — the command line's 20 neighbours — go in two helpings and without the next tile's header in
    //  flight: 92 spilled registers otherwise; a 200k-point cloud is one tile per workgroup anyway)
    constexpr bool kHelpings = ONEPASS && kEllHelpings<W>;
    // (narrow one-pass rows in two helpings as well, the next tile's header still in flight: 8 spilled registers -> 2 at
    //  this kernel's 96-register budget, the device-paced inner loop at 1M 7.72 k -> 7.89 k it/s)
    constexpr bool kNarrowHelpings = ONEPASS && !kEllHelpings<W>;
    EllRowsHead<W, 1> head;
    if constexpr (!kHelpings)
        load_ell_rows_head<W, 1, kBlock>(head, xcd_block(r, ntiles) * kBlock + (int)threadIdx.x, a.nbr, a.cnt, a.src, a.ns, a.width);
    for (int tile = r; tile < ntiles; tile += G) {
        RowAcc acc;
#pragma unroll
        for (int j = 0; j < kNSums; j++) acc.a[j] = 0.0;
        if constexpr (kHelpings) {
            load_ell_rows_head<W, 1, kBlock>(head, xcd_block(tile, ntiles) * kBlock + (int)threadIdx.x, a.nbr, a.cnt, a.src, a.ns, a.width);
            finish_ell_row_helpings<W, TM>(acc, head, a.tgt, P, a.md);
        } else {
            const EllRowsHead<W, 1> cur = head;
            if (tile + G < ntiles) load_ell_rows_head<W, 1, kBlock>(head, xcd_block(tile + G, ntiles) * kBlock + (int)threadIdx.x, a.nbr, a.cnt, a.src, a.ns, a.width);
            if constexpr (kNarrowHelpings) {
                finish_ell_row_helpings<W, TM>(acc, cur, a.tgt, P, a.md);
            } else {
                EllRowsPoints<W, 1> pts;
                gather_ell_rows<W, 1>(pts, cur, a.tgt);
                finish_ell_rows<W, 1, kBlock, TM, ONEPASS>(acc, cur, pts, P, a.md);
            }
        }
        block_reduce_store<kBlock, true>(acc, nullptr, 0, 0, s_acc);  // (its own thread adds to s_acc[tid]: no barrier needed)
    }
    if (threadIdx.x < kNSums)
        __hip_atomic_store(const_cast<double *>(&ic->fs.partials[(size_t)threadIdx.x * G + r]), s_acc[threadIdx.x], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();  // (waits for the 19 stores as well: vmcnt(0))
    if (threadIdx.x == 0) __hip_atomic_store(&ic->flags[(size_t)u * G + r], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// ---------------------------------------------------------------------------------------------
// The step AFTER each outer iteration inside the device-paced loop (cc:110-122): the cloud the reference reports on —
// the full-resolution companion (moved here, in place, by the transform the iteration just solved: the second
// pcl::transformPointCloud of cc:110-112) or the source itself (its move rides in the next association's prologue, so
// here the moved point is only formed, not stored) — and the two mean distances the reference prints per iteration:
// calculateMSE(cloud, ground truth) (cc:114-117) and calculateMSE(cloud, cloud before the move) (cc:121).
// Same per-point float arithmetic, block sums and summation order as mean_distance_kernel + the host sum of
// ppcr_mse_ground_truth / ppcr_mse_previous, so the numbers are those of the one-call-at-a-time path.  The last
// workgroup (ticket) adds the block sums in ascending order and hands the means to the host in pinned memory.
// ---------------------------------------------------------------------------------------------
struct HostReport {
    double mse_truth, moved;
    unsigned seq, pad;
};
struct TrackArgs {
    float4 *cloud;
    int n;
    int sorted_source;   // 1: the handle's sorted source (w lane = caller's index; never written here)
    int write_back;      // 1: store the moved points (companion)
    const Pose *pose;    // device memory: written by the iteration's last solve
    const float4 *truth; // caller's index order; nullptr: no ground-truth distance
    int want_moved;
    double *part;        // [2][gridDim.x]
    unsigned *ticket;
    HostReport *out;     // pinned, device-mapped; nullptr: nothing to report (the companion is only moved)
    unsigned seq;
    const LoopState *st;
};
__global__ __launch_bounds__(kBlock) void track_kernel(TrackArgs a)
{
    if (loop_aborted(a.st)) return;
    __shared__ double sh[2][kBlock / 64];
    const Pose P = *a.pose;
    double acc_t = 0.0, acc_m = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < a.n; i += gridDim.x * kBlock) {
        const float4 p = a.cloud[i];
        const float4 p2 = move_point(p, P);
        if (a.write_back) a.cloud[i] = p2;
        if (a.truth) {
            const float4 q = a.truth[a.sorted_source ? __float_as_int(p.w) : i];
            const float dx = __fsub_rn(p2.x, q.x), dy = __fsub_rn(p2.y, q.y), dz = __fsub_rn(p2.z, q.z);
            acc_t += (double)sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
        }
        if (a.want_moved) {
            const float dx = __fsub_rn(p2.x, p.x), dy = __fsub_rn(p2.y, p.y), dz = __fsub_rn(p2.z, p.z);
            acc_m += (double)sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
        }
    }
    if (!a.out) return;
    for (int off = 32; off > 0; off >>= 1) {
        acc_t += __shfl_down(acc_t, off);
        acc_m += __shfl_down(acc_m, off);
    }
    if ((threadIdx.x & 63) == 0) {
        sh[0][threadIdx.x >> 6] = acc_t;
        sh[1][threadIdx.x >> 6] = acc_m;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double vt = sh[0][0], vm = sh[1][0];
    for (int w = 1; w < kBlock / 64; w++) {
        vt += sh[0][w];
        vm += sh[1][w];
    }
    __hip_atomic_store(&a.part[blockIdx.x], vt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&a.part[gridDim.x + blockIdx.x], vm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned tk = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (tk != gridDim.x - 1) return;
    __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    double st = 0.0, sm = 0.0;
    for (unsigned b = 0; b < gridDim.x; b++) {
        st += __hip_atomic_load(&a.part[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sm += __hip_atomic_load(&a.part[gridDim.x + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    a.out->mse_truth = a.truth ? st / (double)a.n : nan;
    a.out->moved = a.want_moved ? sm / (double)a.n : nan;
    __hip_atomic_store(&a.out->seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
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
