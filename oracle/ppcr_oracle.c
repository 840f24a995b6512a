/*
 * ppcr_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the hot path of
 * iralabdisco/probabilistic_point_clouds_registration (reference mounted at
 * /root/reference while developing; all file:line citations below are relative
 * to that tree).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the shipped path
 * (probabilistic_point_clouds_registration_amd/csrc) never links or calls it.
 *
 * PARITY PINNING STATUS
 *   pinned   : ProbabilisticWeights::updateWeights — both golden-value tests of
 *              test/ProbabilisticWeightsTest.cc:35-66 (tests/test_oracle_golden.py).
 *   pinned   : fixed-association solve — the two exact-association recoveries of
 *              test/PointCloudRegistrationTest.cc:30-116 (minimiser only; the
 *              Ceres LM trajectory is not restatable without Ceres).
 *   UNPINNED : kd-tree radiusSearch, pcl::transformPointCloud, the outer loop and
 *              hasConverged.  The reference has no live test for them
 *              (test/PointCloudRegistrationTest.cc:118-193 is commented out) and
 *              the reference cannot be built here (PCL/FLANN/Ceres/Eigen absent,
 *              no network).  They are restated from the published PCL/FLANN
 *              semantics and cross-checked against scipy.spatial.cKDTree
 *              (tests/golden/make_golden.py).  => "parity unpinned" for those rows.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 * Float contraction MUST stay off: neighbour membership is decided by a float
 * d^2 accumulated x->y->z exactly as FLANN's L2_Simple<float> does.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PO_NSUMS 19
/* layout of the moment vector (all double):
 *  [0]      W    = sum w
 *  [1..3]   Sx   = sum w * (x - c)
 *  [4..6]   Sy   = sum w * (y - c)
 *  [7..15]  Sxy  = sum w * (x - c)_a (y - c)_b   (row-major a,b)
 *  [16]     Sws  = sum w * s          (s at the theta the weights were made at)
 *  [17]     Sxx  = sum w * |x - c|^2
 *  [18]     Syy  = sum w * |y - c|^2
 * c = caller-chosen fixed origin (kills cancellation for clouds far from 0). */

/* ------------------------------------------------------------------------- */
/* small fixed-size math                                                      */
/* ------------------------------------------------------------------------- */

/* Rotation matrix of q/|q|, q = (w,x,y,z).  Order of q follows
 * prob_point_cloud_registration_params.hpp:14 ({1,0,0,0} = identity) and
 * prob_point_cloud_registration_iteration.hpp:62 (Quaternion(w,x,y,z)),
 * normalisation follows :63 and ceres::QuaternionRotatePoint used at
 * error_term.hpp:31 (it scales q to unit length before rotating). */
void po_quat_to_R(const double q[4], double R[9])
{
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    double w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}

/* Shepperd's method, returns q with w >= 0 */
void po_R_to_quat(const double R[9], double q[4])
{
    double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        double s = sqrt(tr + 1.0) * 2;
        q[0] = 0.25 * s; q[1] = (R[7] - R[5]) / s; q[2] = (R[2] - R[6]) / s; q[3] = (R[3] - R[1]) / s;
    } else if (R[0] > R[4] && R[0] > R[8]) {
        double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2;
        q[0] = (R[7] - R[5]) / s; q[1] = 0.25 * s; q[2] = (R[1] + R[3]) / s; q[3] = (R[2] + R[6]) / s;
    } else if (R[4] > R[8]) {
        double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2;
        q[0] = (R[2] - R[6]) / s; q[1] = (R[1] + R[3]) / s; q[2] = 0.25 * s; q[3] = (R[5] + R[7]) / s;
    } else {
        double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2;
        q[0] = (R[3] - R[1]) / s; q[1] = (R[2] + R[6]) / s; q[2] = (R[5] + R[7]) / s; q[3] = 0.25 * s;
    }
    if (q[0] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
}

/* ErrorTerm::operator()<double>  (error_term.hpp:21-37):
 * r = y - (R(q/|q|) x + t); points are widened float->double (error_term.hpp:15-16). */
void po_error_term(const float x[3], const float y[3], const double q[4], const double t[3],
                   double r[3])
{
    double R[9];
    po_quat_to_R(q, R);
    double px = x[0], py = x[1], pz = x[2];
    r[0] = (double)y[0] - ((R[0] * px + R[1] * py + R[2] * pz) + t[0]);
    r[1] = (double)y[1] - ((R[3] * px + R[4] * py + R[5] * pz) + t[1]);
    r[2] = (double)y[2] - ((R[6] * px + R[7] * py + R[8] * pz) + t[2]);
}

/* One-sided (Hestenes) Jacobi SVD of a 3x3: A = U diag(s) V^T, s >= 0 unsorted. */
static void svd3(const double A[9], double U[9], double s[3], double V[9])
{
    double B[9];
    memcpy(B, A, sizeof(B));
    for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int p = 0; p < 2; p++)
            for (int qq = p + 1; qq < 3; qq++) {
                double a = 0, b = 0, c = 0;
                for (int k = 0; k < 3; k++) {
                    a += B[3 * k + p] * B[3 * k + p];
                    b += B[3 * k + qq] * B[3 * k + qq];
                    c += B[3 * k + p] * B[3 * k + qq];
                }
                if (c == 0.0) continue;
                if (fabs(c) <= 1e-300 || fabs(c) <= 4e-32 * sqrt(a) * sqrt(b)) continue;
                off += fabs(c) / sqrt(a * b);
                double zeta = (b - a) / (2.0 * c);
                double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
                for (int k = 0; k < 3; k++) {
                    double bp = B[3 * k + p], bq = B[3 * k + qq];
                    B[3 * k + p] = cs * bp - sn * bq;
                    B[3 * k + qq] = sn * bp + cs * bq;
                    double vp = V[3 * k + p], vq = V[3 * k + qq];
                    V[3 * k + p] = cs * vp - sn * vq;
                    V[3 * k + qq] = sn * vp + cs * vq;
                }
            }
        if (off < 1e-15) break;
    }
    for (int j = 0; j < 3; j++) {
        s[j] = sqrt(B[j] * B[j] + B[3 + j] * B[3 + j] + B[6 + j] * B[6 + j]);
    }
    /* columns of U; rank-deficient columns are completed below by the caller */
    for (int j = 0; j < 3; j++)
        for (int k = 0; k < 3; k++) U[3 * k + j] = (s[j] > 0) ? B[3 * k + j] / s[j] : 0.0;
}

static void cross3(const double a[3], const double b[3], double c[3])
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

static double det3(const double M[9])
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
           M[2] * (M[3] * M[7] - M[4] * M[6]);
}

/* Closed-form minimiser of  sum w |y - R x - t|^2  for fixed w (weighted
 * Kabsch/Horn).  This replaces the Ceres solve of
 * prob_point_cloud_registration_iteration.hpp:52-57 for fixed weights: the
 * objective built at :37-46 (+ ScaledLoss weights, error_term.hpp:17-19,39-43)
 * is exactly this weighted point-to-point problem.
 * sums: PO_NSUMS moments about origin c.  Outputs R (row-major), t (in the
 * un-shifted frame).  Returns 0, or 1 when W<=0 / non-finite (R=I, t=0). */
int po_kabsch(const double sums[PO_NSUMS], const double c[3], double R[9], double t[3])
{
    double W = sums[0];
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    t[0] = t[1] = t[2] = 0;
    if (!(W > 0) || !isfinite(W)) return 1;
    double mx[3], my[3];
    for (int a = 0; a < 3; a++) { mx[a] = sums[1 + a] / W; my[a] = sums[4 + a] / W; }
    /* H[a][b] = sum w (x-mx)_a (y-my)_b */
    double H[9];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) H[3 * a + b] = sums[7 + 3 * a + b] - sums[1 + a] * my[b];
    double U[9], s[3], V[9];
    svd3(H, U, s, V);
    /* order singular values descending so the rank-deficient directions are last */
    int idx[3] = {0, 1, 2};
    for (int i = 0; i < 2; i++)
        for (int j = i + 1; j < 3; j++)
            if (s[idx[j]] > s[idx[i]]) { int tmp = idx[i]; idx[i] = idx[j]; idx[j] = tmp; }
    double Us[9], Vs[9], ss[3];
    for (int j = 0; j < 3; j++) {
        ss[j] = s[idx[j]];
        for (int k = 0; k < 3; k++) { Us[3 * k + j] = U[3 * k + idx[j]]; Vs[3 * k + j] = V[3 * k + idx[j]]; }
    }
    double smax = ss[0];
    if (!(smax > 0)) {
        /* no rotational information at all: keep R = I */
    } else {
        double tiny = smax * 1e-14;
        /* complete U for vanishing singular values (V is always orthogonal) */
        if (ss[1] <= tiny) {
            /* rank 1: pick any unit vector orthogonal to u0 */
            double u0[3] = {Us[0], Us[3], Us[6]};
            double e[3] = {0, 0, 0};
            int m = 0;
            if (fabs(u0[1]) < fabs(u0[m])) m = 1;
            if (fabs(u0[2]) < fabs(u0[m])) m = 2;
            e[m] = 1;
            double u1[3];
            cross3(u0, e, u1);
            double n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
            for (int k = 0; k < 3; k++) Us[3 * k + 1] = u1[k] / n1;
        }
        if (ss[2] <= tiny || ss[1] <= tiny) {
            double u0[3] = {Us[0], Us[3], Us[6]}, u1[3] = {Us[1], Us[4], Us[7]}, u2[3];
            cross3(u0, u1, u2);
            for (int k = 0; k < 3; k++) Us[3 * k + 2] = u2[k];
        }
        /* H = U S V^T with H = sum x y^T  =>  R = V D U^T maps x to y */
        double dU = det3(Us), dV = det3(Vs);
        double dsign = (dU * dV < 0) ? -1.0 : 1.0;
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++)
                R[3 * a + b] = Vs[3 * a + 0] * Us[3 * b + 0] + Vs[3 * a + 1] * Us[3 * b + 1] +
                               dsign * Vs[3 * a + 2] * Us[3 * b + 2];
    }
    /* shifted frame: y' = R x' + t'  with x' = x - c  =>  t = t' + c - R c */
    double tp[3];
    for (int a = 0; a < 3; a++)
        tp[a] = my[a] - (R[3 * a] * mx[0] + R[3 * a + 1] * mx[1] + R[3 * a + 2] * mx[2]);
    for (int a = 0; a < 3; a++)
        t[a] = tp[a] + c[a] - (R[3 * a] * c[0] + R[3 * a + 1] * c[1] + R[3 * a + 2] * c[2]);
    return 0;
}

/* 0.5 * sum w |y - R x - t|^2 evaluated from the moments (weights fixed).
 * The 0.5 is the Ceres cost convention that Summary::initial_cost/final_cost
 * carry into cost_drop_ (src/prob_point_cloud_registration.cc:119). */
double po_cost_from_sums(const double sums[PO_NSUMS], const double c[3], const double R[9],
                         const double t[3])
{
    double W = sums[0];
    double tp[3]; /* translation in the shifted frame: t' = t + R c - c */
    for (int a = 0; a < 3; a++)
        tp[a] = t[a] + (R[3 * a] * c[0] + R[3 * a + 1] * c[1] + R[3 * a + 2] * c[2]) - c[a];
    const double *Sx = sums + 1, *Sy = sums + 4, *Sxy = sums + 7;
    double RSx[3];
    for (int a = 0; a < 3; a++) RSx[a] = R[3 * a] * Sx[0] + R[3 * a + 1] * Sx[1] + R[3 * a + 2] * Sx[2];
    double yRx = 0; /* sum w y^T R x = sum_ab R_ab Sxy[b][a] */
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) yRx += R[3 * a + b] * Sxy[3 * b + a];
    double tt = tp[0] * tp[0] + tp[1] * tp[1] + tp[2] * tp[2];
    double tRSx = tp[0] * RSx[0] + tp[1] * RSx[1] + tp[2] * RSx[2];
    double tSy = tp[0] * Sy[0] + tp[1] * Sy[1] + tp[2] * Sy[2];
    double ssum = sums[18] + sums[17] + 2 * tRSx + W * tt - 2 * yRx - 2 * tSy;
    return 0.5 * ssum;
}

/* ------------------------------------------------------------------------- */
/* a-2  radius search (PCL KdTreeFLANN::radiusSearch semantics)               */
/* ------------------------------------------------------------------------- */
/* src/prob_point_cloud_registration.cc:66-81.  PCL -> FLANN (third-party, not
 * under /root/reference; PCL >= 1.7 per CMakeLists.txt:5, FLANN unpinned).
 * Published semantics restated:
 *   - distance functor L2_Simple<float>: d2 = 0; for k in x,y,z: diff = a-b; d2 += diff*diff
 *     (float, sequential, no fused multiply-add);
 *   - PCL passes static_cast<float>(radius*radius) and FLANN keeps a point iff d2 < that
 *     (strict);
 *   - max_nn == 0 or max_nn >= N_target  => unbounded, else the max_nn closest are kept;
 *   - tie order at the cut-off is traversal dependent in FLANN; this restatement DEFINES
 *     it as lexicographic (d2, target index);
 *   - Eigen setFromTriplets + makeCompressed (:82-83) stores each row by ascending column,
 *     so rows are returned sorted by target index. */

typedef struct { float d2; int idx; } po_cand;

static inline int cand_less(po_cand a, po_cand b)
{
    return (a.d2 < b.d2) || (a.d2 == b.d2 && a.idx < b.idx);
}

static inline float dist2f(const float *a, const float *b)
{
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    float r = dx * dx;
    r = r + dy * dy;
    r = r + dz * dz;
    return r;
}

/* bounded insertion into a list sorted ascending by (d2, idx); returns new count */
static inline int topm_insert(po_cand *list, int cnt, int m, po_cand c)
{
    if (cnt == m) {
        if (!cand_less(c, list[m - 1])) return cnt;
        cnt = m - 1;
    }
    int i = cnt;
    while (i > 0 && cand_less(c, list[i - 1])) { list[i] = list[i - 1]; i--; }
    list[i] = c;
    return cnt + 1;
}

static int cmp_cand_idx(const void *a, const void *b)
{
    int ia = ((const po_cand *)a)->idx, ib = ((const po_cand *)b)->idx;
    return (ia > ib) - (ia < ib);
}

typedef struct {
    float org[3];
    float inv_h;
    int n[3];
    int *cell_start; /* n0*n1*n2 + 1 */
    int *order;      /* target indices sorted by cell */
} po_grid;

static inline int cell_of(const po_grid *g, float v, int ax)
{
    float f = floorf((v - g->org[ax]) * g->inv_h);
    if (!(f >= 0)) return (f < 0) ? -1 : -2; /* NaN -> -2 */
    if (f >= (float)g->n[ax]) return g->n[ax];
    return (int)f;
}

static int grid_build(po_grid *g, const float *tgt, int64_t nt, int ts, float radius)
{
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int64_t j = 0; j < nt; j++)
        for (int a = 0; a < 3; a++) {
            float v = tgt[j * ts + a];
            if (v < lo[a]) lo[a] = v;
            if (v > hi[a]) hi[a] = v;
        }
    if (nt == 0) { for (int a = 0; a < 3; a++) { lo[a] = 0; hi[a] = 0; } }
    float amax = 0;
    for (int a = 0; a < 3; a++) { amax = fmaxf(amax, fabsf(lo[a])); amax = fmaxf(amax, fabsf(hi[a])); }
    /* cell edge a little larger than r so float rounding in the cell index can
     * never hide an in-radius point outside the 27-cell stencil */
    float h = radius * 1.001f + 16.0f * FLT_EPSILON * amax;
    if (!(h > 0)) h = 1.0f;
    double ext[3];
    for (int a = 0; a < 3; a++) ext[a] = (double)hi[a] - lo[a];
    /* bound the table: at most ~4 cells per target point (+ slack) */
    double max_cells = 4.0 * (double)nt + 4096.0;
    for (;;) {
        double nc = 1;
        for (int a = 0; a < 3; a++) nc *= floor(ext[a] / h) + 1;
        if (nc <= max_cells) break;
        h *= 1.26f;
    }
    for (int a = 0; a < 3; a++) { g->org[a] = lo[a]; g->n[a] = (int)floor(ext[a] / h) + 1; }
    g->inv_h = 1.0f / h;
    int64_t nc = (int64_t)g->n[0] * g->n[1] * g->n[2];
    g->cell_start = (int *)calloc((size_t)nc + 1, sizeof(int));
    g->order = (int *)malloc(sizeof(int) * (size_t)(nt > 0 ? nt : 1));
    int *cid = (int *)malloc(sizeof(int) * (size_t)(nt > 0 ? nt : 1));
    if (!g->cell_start || !g->order || !cid) return -1;
    for (int64_t j = 0; j < nt; j++) {
        int c[3];
        for (int a = 0; a < 3; a++) {
            c[a] = cell_of(g, tgt[j * ts + a], a);
            if (c[a] < 0) c[a] = 0; /* NaN / below: park in cell 0; d2 test rejects NaN anyway */
            if (c[a] >= g->n[a]) c[a] = g->n[a] - 1;
        }
        cid[j] = (c[2] * g->n[1] + c[1]) * g->n[0] + c[0];
        g->cell_start[cid[j] + 1]++;
    }
    for (int64_t c = 0; c < nc; c++) g->cell_start[c + 1] += g->cell_start[c];
    int *fill = (int *)malloc(sizeof(int) * (size_t)(nc > 0 ? nc : 1));
    memcpy(fill, g->cell_start, sizeof(int) * (size_t)nc);
    for (int64_t j = 0; j < nt; j++) g->order[fill[cid[j]]++] = (int)j;
    free(fill);
    free(cid);
    return 0;
}

static void grid_free(po_grid *g) { free(g->cell_start); free(g->order); }

/* Returns nnz (>= 0), or -(needed capacity) - 1 when `cap` is too small, or
 * INT64_MIN on allocation failure.  method: 0 = brute force O(Ns*Nt),
 * 1 = uniform grid.  row_ptr has ns+1 entries; col/d2 have `cap` entries
 * (d2 may be NULL).  Strides are in floats (3 = packed xyz, 4 = pcl::PointXYZ). */
int64_t po_radius_search(const float *src, int64_t ns, int ss, const float *tgt, int64_t nt,
                         int ts, double radius, int max_nn, int method, int threads,
                         int *row_ptr, int *col, float *d2, int64_t cap)
{
    const float r2 = (float)(radius * radius);
    int unbounded = (max_nn <= 0 || (int64_t)max_nn >= nt);
    int m = unbounded ? 0 : max_nn;
    po_grid g;
    memset(&g, 0, sizeof(g));
    if (method == 1) {
        if (grid_build(&g, tgt, nt, ts, (float)radius) != 0) return INT64_MIN;
    }
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
    int *cnt = (int *)calloc((size_t)ns + 1, sizeof(int));
    /* bounded: fixed ns*m scratch; unbounded: two passes (count, then fill) */
    po_cand *scratch = NULL;
    if (!unbounded) {
        scratch = (po_cand *)malloc(sizeof(po_cand) * (size_t)(ns > 0 ? ns : 1) * (size_t)m);
        if (!scratch) { free(cnt); return INT64_MIN; }
    }
    for (int pass = 0; pass < 2; pass++) {
        if (!unbounded && pass == 1) break;
        if (unbounded && pass == 1) {
            int64_t tot = 0;
            row_ptr[0] = 0;
            for (int64_t i = 0; i < ns; i++) { tot += cnt[i]; row_ptr[i + 1] = (int)tot; }
            if (tot > cap) { free(cnt); if (method == 1) grid_free(&g); return -tot - 1; }
        }
#pragma omp parallel for schedule(dynamic, 256)
        for (int64_t i = 0; i < ns; i++) {
            const float *q = src + i * ss;
            po_cand *list = unbounded ? NULL : scratch + (size_t)i * m;
            int c = 0;
            int64_t wpos = (unbounded && pass == 1) ? row_ptr[i] : 0;
#define PO_VISIT(J)                                                             \
    do {                                                                        \
        int64_t j_ = (J);                                                       \
        float dd = dist2f(q, tgt + j_ * ts);                                    \
        if (dd < r2) {                                                          \
            if (unbounded) {                                                    \
                if (pass == 1) { col[wpos] = (int)j_; if (d2) d2[wpos] = dd; wpos++; } \
                c++;                                                            \
            } else {                                                            \
                po_cand cc = {dd, (int)j_};                                     \
                c = topm_insert(list, c, m, cc);                                \
            }                                                                   \
        }                                                                       \
    } while (0)
            if (method == 0) {
                for (int64_t j = 0; j < nt; j++) PO_VISIT(j);
            } else {
                int cc3[3], ok = 1;
                for (int a = 0; a < 3; a++) {
                    cc3[a] = cell_of(&g, q[a], a);
                    if (cc3[a] == -2) ok = 0;
                }
                if (ok)
                    for (int dz = -1; dz <= 1; dz++) {
                        int cz = cc3[2] + dz;
                        if (cz < 0 || cz >= g.n[2]) continue;
                        for (int dy = -1; dy <= 1; dy++) {
                            int cy = cc3[1] + dy;
                            if (cy < 0 || cy >= g.n[1]) continue;
                            int x0 = cc3[0] - 1, x1 = cc3[0] + 1;
                            if (x0 < 0) x0 = 0;
                            if (x1 >= g.n[0]) x1 = g.n[0] - 1;
                            if (x0 > x1) continue;
                            int64_t base = ((int64_t)cz * g.n[1] + cy) * g.n[0];
                            int b = g.cell_start[base + x0], e = g.cell_start[base + x1 + 1];
                            for (int p = b; p < e; p++) PO_VISIT(g.order[p]);
                        }
                    }
            }
#undef PO_VISIT
            cnt[i] = c;
            if (unbounded && pass == 1 && c > 1) {
                /* grid visits cells out of index order: restore ascending columns */
                po_cand *tmp = (po_cand *)malloc(sizeof(po_cand) * (size_t)c);
                for (int k = 0; k < c; k++) { tmp[k].idx = col[row_ptr[i] + k]; tmp[k].d2 = d2 ? d2[row_ptr[i] + k] : 0; }
                qsort(tmp, (size_t)c, sizeof(po_cand), cmp_cand_idx);
                for (int k = 0; k < c; k++) { col[row_ptr[i] + k] = tmp[k].idx; if (d2) d2[row_ptr[i] + k] = tmp[k].d2; }
                free(tmp);
            }
        }
    }
    int64_t nnz;
    if (!unbounded) {
        int64_t tot = 0;
        row_ptr[0] = 0;
        for (int64_t i = 0; i < ns; i++) { tot += cnt[i]; row_ptr[i + 1] = (int)tot; }
        if (tot > cap) { free(cnt); free(scratch); if (method == 1) grid_free(&g); return -tot - 1; }
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < ns; i++) {
            po_cand *list = scratch + (size_t)i * m;
            int c = cnt[i];
            /* setFromTriplets ordering: ascending column (cc:82-83) */
            qsort(list, (size_t)c, sizeof(po_cand), cmp_cand_idx);
            for (int k = 0; k < c; k++) { col[row_ptr[i] + k] = list[k].idx; if (d2) d2[row_ptr[i] + k] = list[k].d2; }
        }
        nnz = tot;
        free(scratch);
    } else {
        nnz = row_ptr[ns];
    }
    free(cnt);
    if (method == 1) grid_free(&g);
    return nnz;
}

/* ------------------------------------------------------------------------- */
/* a-6  squared errors in CSR order  (weight_updater_callback.hpp:42-51)      */
/* ------------------------------------------------------------------------- */
void po_squared_errors(const float *src, int ss, const float *tgt, int ts, const int *row_ptr,
                       const int *col, int64_t ns, const double q[4], const double t[3],
                       double *s_out)
{
    for (int64_t i = 0; i < ns; i++)
        for (int k = row_ptr[i]; k < row_ptr[i + 1]; k++) {
            double r[3];
            po_error_term(src + i * ss, tgt + (int64_t)col[k] * ts, q, t, r);
            s_out[k] = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
        }
}

/* ------------------------------------------------------------------------- */
/* a-7  ProbabilisticWeights::updateWeights (probabilistic_weights.hpp:30-105) */
/* ------------------------------------------------------------------------- */
static const double PO_PI = 3.14159265358979323846;

typedef struct { int is_normal; double v, t_exponent, log_norm_constant; int dim; } po_pw;

static po_pw pw_make(double v, int dim)
{
    po_pw p;
    p.dim = dim;
    if (v < INFINITY) { /* :35-41 */
        p.is_normal = 0;
        p.v = v;
        p.t_exponent = -(v + dim) / 2.0;
        p.log_norm_constant = lgamma(v / 2) - lgamma((v + dim) / 2) + (v / 2) * log(PO_PI * v);
    } else { /* :42-45 */
        p.is_normal = 1;
        p.v = v;
        p.t_exponent = 0;
        p.log_norm_constant = (dim / 2.0) * log(2 * PO_PI);
    }
    return p;
}

/* one CSR row (:56-101). s, w point at the row's first entry. */
static void pw_row(const po_pw *p, const double *s, int n, double *w)
{
    double max_lp = -INFINITY; /* :57 */
    for (int k = 0; k < n; k++) {
        double lp;
        if (p->is_normal) lp = -s[k] / 2 + p->log_norm_constant;            /* :69 */
        else lp = p->t_exponent * log1p(s[k] / p->v) - p->log_norm_constant; /* :71-72 */
        if (lp > max_lp) max_lp = lp;                                        /* :77-79 */
        w[k] = lp; /* stash log_probs (:80) */
    }
    double mll = 0;
    for (int k = 0; k < n; k++) mll += exp(w[k] - max_lp); /* :83-85 */
    mll = log(mll) + max_lp;                               /* :86-87 */
    for (int k = 0; k < n; k++) {
        double e = exp(w[k] - mll);
        if (p->is_normal) w[k] = e;                                  /* :93-94 */
        else w[k] = e * ((p->v + p->dim) / (p->v + s[k]));           /* :73, :96-98 */
    }
}

/* weights in CSR order for squared errors in CSR order */
void po_update_weights(const int *row_ptr, int64_t nrows, const double *s, double v, int dim,
                       double *w_out)
{
    po_pw p = pw_make(v, dim);
    for (int64_t i = 0; i < nrows; i++)
        pw_row(&p, s + row_ptr[i], row_ptr[i + 1] - row_ptr[i], w_out + row_ptr[i]);
}

/* ------------------------------------------------------------------------- */
/* weights at theta + moment accumulation (one IRLS half-step)               */
/* ------------------------------------------------------------------------- */
/* = WeightUpdaterCallback::operator() (weight_updater_callback.hpp:36-63)
 * followed by the sufficient statistics of the weighted problem Ceres would
 * assemble (prob_point_cloud_registration_iteration.hpp:37-46). */
void po_accumulate(const float *src, int ss, const float *tgt, int ts, const int *row_ptr,
                   const int *col, int64_t ns, const double q[4], const double t[3], double v,
                   int dim, const double c[3], int threads, double sums[PO_NSUMS])
{
    po_pw p = pw_make(v, dim);
    double R[9];
    po_quat_to_R(q, R);
    int nth = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    nth = omp_get_max_threads();
#else
    (void)threads;
#endif
    double *part = (double *)calloc((size_t)nth * PO_NSUMS, sizeof(double));
#pragma omp parallel
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        double *acc = part + (size_t)tid * PO_NSUMS;
        int capn = 64;
        double *sb = (double *)malloc(sizeof(double) * capn), *wb = (double *)malloc(sizeof(double) * capn);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < ns; i++) {
            int n = row_ptr[i + 1] - row_ptr[i];
            if (n == 0) continue;
            if (n > capn) {
                capn = n * 2;
                sb = (double *)realloc(sb, sizeof(double) * capn);
                wb = (double *)realloc(wb, sizeof(double) * capn);
            }
            const float *xf = src + i * ss;
            double px = xf[0], py = xf[1], pz = xf[2];
            double xr[3] = {(R[0] * px + R[1] * py + R[2] * pz) + t[0],
                            (R[3] * px + R[4] * py + R[5] * pz) + t[1],
                            (R[6] * px + R[7] * py + R[8] * pz) + t[2]};
            for (int k = 0; k < n; k++) {
                const float *yf = tgt + (int64_t)col[row_ptr[i] + k] * ts;
                double r0 = (double)yf[0] - xr[0], r1 = (double)yf[1] - xr[1], r2 = (double)yf[2] - xr[2];
                sb[k] = r0 * r0 + r1 * r1 + r2 * r2;
            }
            pw_row(&p, sb, n, wb);
            double xc[3] = {px - c[0], py - c[1], pz - c[2]};
            double xx = xc[0] * xc[0] + xc[1] * xc[1] + xc[2] * xc[2];
            for (int k = 0; k < n; k++) {
                const float *yf = tgt + (int64_t)col[row_ptr[i] + k] * ts;
                double yc[3] = {(double)yf[0] - c[0], (double)yf[1] - c[1], (double)yf[2] - c[2]};
                double w = wb[k];
                acc[0] += w;
                for (int a = 0; a < 3; a++) {
                    acc[1 + a] += w * xc[a];
                    acc[4 + a] += w * yc[a];
                    for (int b = 0; b < 3; b++) acc[7 + 3 * a + b] += w * xc[a] * yc[b];
                }
                acc[16] += w * sb[k];
                acc[17] += w * xx;
                acc[18] += w * (yc[0] * yc[0] + yc[1] * yc[1] + yc[2] * yc[2]);
            }
        }
        free(sb);
        free(wb);
    }
    for (int a = 0; a < PO_NSUMS; a++) {
        double sacc = 0;
        for (int th = 0; th < nth; th++) sacc += part[(size_t)th * PO_NSUMS + a];
        sums[a] = sacc;
    }
    free(part);
}

/* ------------------------------------------------------------------------- */
/* a-10  pcl::transformPointCloud(cloud, cloud, Affine3d) in place            */
/* ------------------------------------------------------------------------- */
/* src/prob_point_cloud_registration.cc:110-112.  PCL (third-party) restated:
 * per point, each output coordinate = (float)(m0*x + m1*y + m2*z + m3) with the
 * arithmetic in double, summed left to right, result stored as float32.
 * T is the top 3 rows of the 4x4, row-major (12 doubles). */
void po_transform_cloud(float *xyz, int64_t n, int stride, const double T[12], int threads)
{
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        float *p = xyz + i * stride;
        double x = p[0], y = p[1], z = p[2];
        p[0] = (float)(((T[0] * x + T[1] * y) + T[2] * z) + T[3]);
        p[1] = (float)(((T[4] * x + T[5] * y) + T[6] * z) + T[7]);
        p[2] = (float)(((T[8] * x + T[9] * y) + T[10] * z) + T[11]);
    }
}

/* ------------------------------------------------------------------------- */
/* a-4/a-8/a-9  fixed-association solve = ProbPointCloudRegistrationIteration */
/* ------------------------------------------------------------------------- */
/* prob_point_cloud_registration_iteration.hpp:24-67.  The reference runs Ceres
 * LM with the weight callback after every LM iteration; its fixed point for a
 * given association is theta* = argmin sum w(theta*) s(theta).  This restatement
 * reaches the same fixed point by IRLS with the closed-form solve:
 *   state (q,t) starts at params.initial_rotation/translation (:31-34);
 *   weights primed at the initial state (:49)  -> initial_cost;
 *   step: theta_new = kabsch(moments(w(theta_old)));
 *         final_cost = 0.5 sum w(theta_old) s(theta_new)   (weights lag one update)
 *   stop after max_steps, or when (cost_old - final_cost) <= max(f_tol * cost_old, 1e-14 * (Sxx+Syy)/2)
 *   (Ceres function_tolerance, src/prob_point_cloud_registration.cc:97).
 * Outputs: R,t of transformation() (:59-67), initial/final cost, steps taken. */
int po_solve(const float *src, int ss, const float *tgt, int ts, const int *row_ptr, const int *col,
             int64_t ns, const double q0[4], const double t0[3], double v, int dim,
             const double c[3], int max_steps, double f_tol, int threads, double R_out[9],
             double t_out[3], double cost_out[2], int *steps_out)
{
    double q[4] = {q0[0], q0[1], q0[2], q0[3]}, t[3] = {t0[0], t0[1], t0[2]};
    double R[9];
    po_quat_to_R(q, R);
    double sums[PO_NSUMS];
    po_accumulate(src, ss, tgt, ts, row_ptr, col, ns, q, t, v, dim, c, threads, sums);
    double cost_old = 0.5 * sums[16];
    cost_out[0] = cost_old;
    cost_out[1] = cost_old;
    int steps = 0;
    if (max_steps < 1) max_steps = 1;
    for (;;) {
        double Rn[9], tn[3];
        int degenerate = po_kabsch(sums, c, Rn, tn);
        double fc = degenerate ? cost_old : po_cost_from_sums(sums, c, Rn, tn);
        steps++;
        memcpy(R, Rn, sizeof(R));
        memcpy(t, tn, sizeof(t));
        cost_out[1] = fc;
        if (degenerate || steps >= max_steps) break;
        /* a decrease below the rounding floor of the moment-based cost (eps * (Sxx + Syy)) is no decrease */
        if ((cost_old - fc) <= fmax(f_tol * cost_old, 1e-14 * 0.5 * (sums[17] + sums[18]))) break;
        po_R_to_quat(R, q);
        po_accumulate(src, ss, tgt, ts, row_ptr, col, ns, q, t, v, dim, c, threads, sums);
        cost_old = 0.5 * sums[16];
    }
    memcpy(R_out, R, sizeof(R));
    memcpy(t_out, t, sizeof(t));
    *steps_out = steps;
    return 0;
}

/* ------------------------------------------------------------------------- */
/* a-1/a-11  ProbPointCloudRegistration::align + hasConverged                 */
/* ------------------------------------------------------------------------- */
typedef struct {
    int current_iteration;
    double cost_drop;
    int num_unusefull_iter;
} po_conv;

/* src/prob_point_cloud_registration.cc:138-158, including the quirk that
 * cost_drop_ starts at 0 (:20) so the first check counts as "unuseful", and the
 * NaN behaviour (comparison false -> counter reset). n_cost_drop_it is a double
 * in the params struct (prob_point_cloud_registration_params.hpp:11). */
static int has_converged(po_conv *s, int n_iter, double thresh, double n_cost_drop_it)
{
    if (s->current_iteration == n_iter) return 1;
    if (s->cost_drop < thresh) {
        if ((double)s->num_unusefull_iter > n_cost_drop_it) return 1;
        s->num_unusefull_iter++;
    } else {
        s->num_unusefull_iter = 0;
    }
    return 0;
}

/* Full outer loop (src/prob_point_cloud_registration.cc:63-136) on a private
 * copy of the source (the ctor deep-copies it, :22).  source_filter_size == 0
 * path only (filtered copy == source, :31-33).
 * history: room for n_iter * 12 doubles (cumulative [R|t] per outer iteration,
 * T_cum <- T_k * T_cum, :101-107).  costs: n_iter * 2 doubles.  inner_steps:
 * n_iter ints.  src_out (nullable): moved source, float stride ss.
 * rebuild_index_each_iter only changes timing (the reference rebuilds its
 * kd-tree every iteration, :66-67); results are identical.
 * Returns the number of outer iterations performed. */
int po_align(const float *src_in, int64_t ns, int ss, const float *tgt, int64_t nt, int ts,
             double radius, int max_nn, double dof, int dim, int n_iter, double cost_drop_thresh,
             double n_cost_drop_it, const double q0[4], const double t0[3], int inner_max_steps,
             double f_tol, int nn_method, int threads, double *history, double *costs,
             int *inner_steps, float *src_out)
{
    float *src = (float *)malloc(sizeof(float) * (size_t)(ns > 0 ? ns : 1) * ss);
    memcpy(src, src_in, sizeof(float) * (size_t)ns * ss);
    int unbounded = (max_nn <= 0 || (int64_t)max_nn >= nt);
    int64_t cap = unbounded ? 0 : ns * (int64_t)max_nn;
    int *row_ptr = (int *)malloc(sizeof(int) * ((size_t)ns + 1));
    int *col = (int *)malloc(sizeof(int) * (size_t)(cap > 0 ? cap : 1));
    /* fixed origin for the moments: centre of the target bounding box */
    double c[3] = {0, 0, 0};
    if (nt > 0) {
        for (int a = 0; a < 3; a++) {
            float lo = FLT_MAX, hi = -FLT_MAX;
            for (int64_t j = 0; j < nt; j++) {
                float vv = tgt[j * ts + a];
                if (vv < lo) lo = vv;
                if (vv > hi) hi = vv;
            }
            c[a] = 0.5 * ((double)lo + (double)hi);
        }
    }
    po_conv st = {0, 0.0, 0};
    double Tcum[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    while (!has_converged(&st, n_iter, cost_drop_thresh, n_cost_drop_it)) {
        int64_t nnz;
        for (;;) {
            nnz = po_radius_search(src, ns, ss, tgt, nt, ts, radius, max_nn, nn_method, threads,
                                   row_ptr, col, NULL, cap);
            if (nnz >= 0 || nnz == INT64_MIN) break;
            cap = -nnz - 1;
            free(col);
            col = (int *)malloc(sizeof(int) * (size_t)cap);
        }
        double R[9], t[3], cost[2];
        int steps = 0;
        po_solve(src, ss, tgt, ts, row_ptr, col, ns, q0, t0, dof, dim, c, inner_max_steps, f_tol,
                 threads, R, t, cost, &steps);
        double Tk[12] = {R[0], R[1], R[2], t[0], R[3], R[4], R[5], t[1], R[6], R[7], R[8], t[2]};
        /* T_cum <- T_k * T_cum (:101-107) */
        double Tn[12];
        for (int a = 0; a < 3; a++) {
            for (int b = 0; b < 4; b++) {
                double acc = 0;
                for (int k = 0; k < 3; k++) acc += Tk[4 * a + k] * Tcum[4 * k + b];
                if (b == 3) acc += Tk[4 * a + 3];
                Tn[4 * a + b] = acc;
            }
        }
        memcpy(Tcum, Tn, sizeof(Tcum));
        int it = st.current_iteration;
        memcpy(history + (size_t)it * 12, Tcum, sizeof(Tcum));
        costs[2 * it] = cost[0];
        costs[2 * it + 1] = cost[1];
        inner_steps[it] = steps;
        po_transform_cloud(src, ns, ss, Tk, threads);       /* :110-112 */
        st.cost_drop = (cost[0] - cost[1]) / cost[0];       /* :119 */
        st.current_iteration++;                             /* :130 */
    }
    if (src_out) memcpy(src_out, src, sizeof(float) * (size_t)ns * ss);
    free(src);
    free(row_ptr);
    free(col);
    return st.current_iteration;
}

/* calculateMSE (utilities.hpp:16-26): mean Euclidean distance of index-paired
 * points (float coordinates; pcl::euclideanDistance works in float). */
double po_calculate_mse(const float *a, int sa, const float *b, int sb, int64_t n)
{
    double mse = 0;
    for (int64_t i = 0; i < n; i++) {
        float dx = a[i * sa] - b[i * sb], dy = a[i * sa + 1] - b[i * sb + 1], dz = a[i * sa + 2] - b[i * sb + 2];
        mse += sqrtf(dx * dx + dy * dy + dz * dz);
    }
    return mse / (double)n;
}

/* pcl::VoxelGrid<PointXYZ> centroid down-sampling as the reference uses it (src/prob_point_cloud_registration.cc:24-41:
 * setLeafSize(l,l,l), default min_points_per_voxel = 0, no field filter).  Third-party (PCL, unpinned), restated from
 * its published algorithm:
 *   min/max over the finite points; inv = 1/leaf (float); if the voxel count (dx*dy*dz, 64-bit) exceeds INT32_MAX the
 *   input is returned unchanged (PCL warns "Leaf size is too small"); min_b = floor(min*inv), div_b = max_b - min_b + 1;
 *   voxel index = ijk . (1, div_x, div_x*div_y) with ijk = floor(p*inv) - min_b; points grouped by index; output =
 *   float centroid of every occupied voxel, voxels in ascending index.
 * PCL orders the points of one voxel with std::sort (unstable), so its float centroid is only defined up to the order
 * of the additions; here (and in the HIP path, bit for bit) they are added in ascending original index.
 * parity unpinned: the reference has no test for it.
 * Returns the number of output points (out holds up to n); -1 on bad arguments. */
typedef struct { int32_t idx; int32_t pt; } po_vox_pair;
static int po_vox_cmp(const void *a, const void *b)
{
    const po_vox_pair *x = (const po_vox_pair *)a, *y = (const po_vox_pair *)b;
    if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
    return x->pt < y->pt ? -1 : (x->pt > y->pt ? 1 : 0);
}

int64_t po_voxel_filter(const float *in, int sin, int64_t n, float leaf, float *out, int sout)
{
    if (n < 0 || !(leaf > 0) || sin < 3 || sout < 3) return -1;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    int64_t nfin = 0;
    for (int64_t i = 0; i < n; i++) {
        const float *p = in + i * sin;
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
        for (int a = 0; a < 3; a++) {
            if (p[a] < lo[a]) lo[a] = p[a];
            if (p[a] > hi[a]) hi[a] = p[a];
        }
        nfin++;
    }
    if (nfin == 0) return 0;
    const float inv = 1.0f / leaf;
    int64_t d[3];
    for (int a = 0; a < 3; a++) d[a] = (int64_t)((hi[a] - lo[a]) * inv) + 1;
    if (d[0] * d[1] * d[2] > (int64_t)INT32_MAX) {  /* PCL: "Leaf size is too small for the input dataset" */
        for (int64_t i = 0; i < n; i++)
            for (int a = 0; a < 3; a++) out[i * sout + a] = in[i * sin + a];
        return n;
    }
    int32_t min_b[3], div_b[3];
    for (int a = 0; a < 3; a++) {
        min_b[a] = (int32_t)floorf(lo[a] * inv);
        div_b[a] = (int32_t)floorf(hi[a] * inv) - min_b[a] + 1;
    }
    const int32_t mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
    po_vox_pair *pr = (po_vox_pair *)malloc(sizeof(po_vox_pair) * (size_t)(nfin > 0 ? nfin : 1));
    int64_t k = 0;
    for (int64_t i = 0; i < n; i++) {
        const float *p = in + i * sin;
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
        int32_t idx = 0;
        for (int a = 0; a < 3; a++) idx += (int32_t)(floorf(p[a] * inv) - (float)min_b[a]) * mul[a];
        pr[k].idx = idx;
        pr[k].pt = (int32_t)i;
        k++;
    }
    qsort(pr, (size_t)nfin, sizeof(po_vox_pair), po_vox_cmp);
    int64_t nout = 0;
    for (int64_t b = 0; b < nfin;) {
        int64_t e = b;
        float c[3] = {0.f, 0.f, 0.f};
        while (e < nfin && pr[e].idx == pr[b].idx) {
            const float *p = in + (int64_t)pr[e].pt * sin;
            c[0] += p[0];
            c[1] += p[1];
            c[2] += p[2];
            e++;
        }
        const float cnt = (float)(e - b);
        for (int a = 0; a < 3; a++) out[nout * sout + a] = c[a] / cnt;
        nout++;
        b = e;
    }
    free(pr);
    return nout;
}

/* nearestKSearch(k = 1) squared distances as the metrics of utilities.hpp:28-234 use them (third-party FLANN,
 * restated: exact nearest neighbour, float d2 accumulated x, y, z like L2_Simple).  Brute force, O(nq * nt). */
void po_nearest_sq_distances(const float *q, int sq, int64_t nq, const float *t, int st, int64_t nt, float *d2_out,
                             int threads)
{
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nq; i++) {
        const float *a = q + i * sq;
        float best = INFINITY;
        for (int64_t j = 0; j < nt; j++) {
            const float *b = t + j * st;
            float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
            float d = dx * dx;
            d = d + dy * dy;
            d = d + dz * dz;
            if (d < best) best = d;
        }
        d2_out[i] = best;
    }
}

int po_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
