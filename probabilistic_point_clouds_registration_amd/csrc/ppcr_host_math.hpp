// Fixed-size math for the registration hot path (host and device): quaternion <-> rotation, the 3x3 SVD
// and the closed-form weighted rigid solve that replaces the Ceres problem of
// prob_point_cloud_registration_iteration.hpp:36-57 for fixed weights.  Header-only, no deps.
#pragma once
#include <cmath>
#include <cstring>

// the same code serves the host (ppcr_solve_moments, the host-driven inner loop) and the device (the solve that
// reduce_solve_kernel runs right behind the moment fold, so that the next association need not wait for the host)
#if defined(__HIPCC__)
#define PPCR_HD __host__ __device__
#else
#define PPCR_HD
#endif

namespace ppcr {

struct Vec3 {
    double v[3];
    PPCR_HD double &operator[](int i) { return v[i]; }
    PPCR_HD double operator[](int i) const { return v[i]; }
};

struct Mat3 {
    double m[3][3];
    PPCR_HD static Mat3 identity()
    {
        Mat3 r;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) r.m[i][j] = (i == j) ? 1.0 : 0.0;
        return r;
    }
    PPCR_HD Vec3 col(int j) const { return Vec3{{m[0][j], m[1][j], m[2][j]}}; }
    PPCR_HD void set_col(int j, const Vec3 &c)
    {
        for (int i = 0; i < 3; i++) m[i][j] = c[i];
    }
    PPCR_HD double det() const
    {
        return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) -
               m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
               m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    }
};

PPCR_HD inline double dot(const Vec3 &a, const Vec3 &b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
PPCR_HD inline Vec3 cross(const Vec3 &a, const Vec3 &b)
{
    return Vec3{{a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}};
}
PPCR_HD inline Vec3 mul(const Mat3 &A, const Vec3 &x)
{
    Vec3 r;
    for (int i = 0; i < 3; i++) r[i] = A.m[i][0] * x[0] + A.m[i][1] * x[1] + A.m[i][2] * x[2];
    return r;
}
PPCR_HD inline double norm(const Vec3 &a) { return std::sqrt(dot(a, a)); }

// q = (w,x,y,z), any non-zero length (normalised here like Eigen's estimated_rot.normalize(),
// ..._iteration.hpp:62-63)
PPCR_HD inline Mat3 quat_to_rot(const double q[4])
{
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
    Mat3 R;
    R.m[0][0] = 1 - 2 * (y * y + z * z);
    R.m[0][1] = 2 * (x * y - w * z);
    R.m[0][2] = 2 * (x * z + w * y);
    R.m[1][0] = 2 * (x * y + w * z);
    R.m[1][1] = 1 - 2 * (x * x + z * z);
    R.m[1][2] = 2 * (y * z - w * x);
    R.m[2][0] = 2 * (x * z - w * y);
    R.m[2][1] = 2 * (y * z + w * x);
    R.m[2][2] = 1 - 2 * (x * x + y * y);
    return R;
}

PPCR_HD inline void rot_to_quat(const Mat3 &R, double q[4])
{
    const double tr = R.m[0][0] + R.m[1][1] + R.m[2][2];
    // pick the largest of (w,x,y,z) to divide by — the numerically stable branch
    int best = 0;
    double diag[4] = {tr, R.m[0][0], R.m[1][1], R.m[2][2]};
    for (int i = 1; i < 4; i++)
        if (diag[i] > diag[best]) best = i;
    if (best == 0) {
        const double s = 2.0 * std::sqrt(1.0 + tr);
        q[0] = 0.25 * s;
        q[1] = (R.m[2][1] - R.m[1][2]) / s;
        q[2] = (R.m[0][2] - R.m[2][0]) / s;
        q[3] = (R.m[1][0] - R.m[0][1]) / s;
    } else {
        const int i = best - 1, j = (i + 1) % 3, k = (i + 2) % 3;
        const double s = 2.0 * std::sqrt(1.0 + R.m[i][i] - R.m[j][j] - R.m[k][k]);
        q[1 + i] = 0.25 * s;
        q[0] = (R.m[k][j] - R.m[j][k]) / s;
        q[1 + j] = (R.m[j][i] + R.m[i][j]) / s;
        q[1 + k] = (R.m[k][i] + R.m[i][k]) / s;
    }
    if (q[0] < 0)
        for (int i = 0; i < 4; i++) q[i] = -q[i];
}

// One-sided Jacobi SVD: A = U * diag(s) * V^T, singular values sorted descending.
// Columns of U belonging to (numerically) zero singular values are left zero; the caller
// completes them.
PPCR_HD inline void svd3(const Mat3 &A, Mat3 &U, double s[3], Mat3 &V)
{
    Mat3 W = A;
    V = Mat3::identity();
    for (int sweep = 0; sweep < 64; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 3; ++q) {
                const Vec3 cp = W.col(p), cq = W.col(q);
                const double alpha = dot(cp, cp), beta = dot(cq, cq), gamma = dot(cp, cq);
                if (gamma == 0.0 || std::fabs(gamma) <= 1e-16 * std::sqrt(alpha) * std::sqrt(beta))
                    continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double tan_ = std::copysign(1.0, zeta) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + tan_ * tan_), sn = c * tan_;
                for (int r = 0; r < 3; ++r) {
                    const double wp = W.m[r][p], wq = W.m[r][q];
                    W.m[r][p] = c * wp - sn * wq;
                    W.m[r][q] = sn * wp + c * wq;
                    const double vp = V.m[r][p], vq = V.m[r][q];
                    V.m[r][p] = c * vp - sn * vq;
                    V.m[r][q] = sn * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    int order[3] = {0, 1, 2};
    double len[3];
    for (int j = 0; j < 3; j++) len[j] = norm(W.col(j));
    for (int a = 0; a < 2; a++)
        for (int b = a + 1; b < 3; b++)
            if (len[order[b]] > len[order[a]]) {
                const int tmp = order[a];
                order[a] = order[b];
                order[b] = tmp;
            }
    Mat3 Vs;
    for (int j = 0; j < 3; j++) {
        const int o = order[j];
        s[j] = len[o];
        Vec3 u = W.col(o);
        for (int r = 0; r < 3; r++) u[r] = (len[o] > 0) ? u[r] / len[o] : 0.0;
        U.set_col(j, u);
        Vs.set_col(j, V.col(o));
    }
    V = Vs;
}

// moments layout: see PPCR_NSUMS in include/ppcr.h
struct RigidSolve {
    Mat3 R;
    Vec3 t;
    bool degenerate;  // no weight mass: R = I, t = 0
};

PPCR_HD inline RigidSolve solve_rigid_from_moments(const double S[19], const double c[3])
{
    RigidSolve out;
    out.R = Mat3::identity();
    out.t = Vec3{{0, 0, 0}};
    out.degenerate = true;
    const double W = S[0];
    if (!(W > 0) || !std::isfinite(W)) return out;
    out.degenerate = false;
    Vec3 mx, my;
    for (int a = 0; a < 3; a++) {
        mx[a] = S[1 + a] / W;
        my[a] = S[4 + a] / W;
    }
    Mat3 H;  // H[a][b] = sum w (x - mx)_a (y - my)_b
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) H.m[a][b] = S[7 + 3 * a + b] - S[1 + a] * my[b];
    Mat3 U, V;
    double s[3];
    svd3(H, U, s, V);
    if (s[0] > 0) {
        const double tiny = s[0] * 1e-14;
        Vec3 u0 = U.col(0), u1 = U.col(1), u2;
        if (s[1] <= tiny) {  // rank 1: any unit vector orthogonal to u0
            int mi = 0;
            if (std::fabs(u0[1]) < std::fabs(u0[mi])) mi = 1;
            if (std::fabs(u0[2]) < std::fabs(u0[mi])) mi = 2;
            Vec3 e{{0, 0, 0}};
            e[mi] = 1;
            u1 = cross(u0, e);
            const double n1 = norm(u1);
            for (int r = 0; r < 3; r++) u1[r] /= n1;
            U.set_col(1, u1);
        }
        if (s[2] <= tiny || s[1] <= tiny) {
            u2 = cross(u0, u1);
            U.set_col(2, u2);
        }
        // H = U S V^T with H = sum x y^T  =>  R = V diag(1,1,d) U^T maps x onto y
        const double d = (U.det() * V.det() < 0) ? -1.0 : 1.0;
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++)
                out.R.m[a][b] = V.m[a][0] * U.m[b][0] + V.m[a][1] * U.m[b][1] + d * V.m[a][2] * U.m[b][2];
    }
    const Vec3 Rmx = mul(out.R, mx);
    const Vec3 cc{{c[0], c[1], c[2]}};
    const Vec3 Rc = mul(out.R, cc);
    for (int a = 0; a < 3; a++) out.t[a] = (my[a] - Rmx[a]) + c[a] - Rc[a];
    return out;
}

// 0.5 * sum w |y - R x - t|^2 from the moments (Ceres cost convention: cc:119 consumes
// Summary::initial_cost/final_cost which carry the 1/2)
PPCR_HD inline double cost_from_moments(const double S[19], const double c[3], const Mat3 &R, const Vec3 &t)
{
    const double W = S[0];
    const Vec3 cc{{c[0], c[1], c[2]}};
    const Vec3 Rc = mul(R, cc);
    Vec3 tp;  // translation in the frame shifted by c
    for (int a = 0; a < 3; a++) tp[a] = t[a] + Rc[a] - c[a];
    const Vec3 Sx{{S[1], S[2], S[3]}}, Sy{{S[4], S[5], S[6]}};
    const Vec3 RSx = mul(R, Sx);
    double yRx = 0;
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) yRx += R.m[a][b] * S[7 + 3 * b + a];
    const double total = S[18] + S[17] + 2 * dot(tp, RSx) + W * dot(tp, tp) - 2 * yRx - 2 * dot(tp, Sy);
    return 0.5 * total;
}

// T_out = A * B for [R|t] 3x4 row-major rigid transforms
PPCR_HD inline void compose(const double A[12], const double B[12], double out[12])
{
    double r[12];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 4; b++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += A[4 * a + k] * B[4 * k + b];
            if (b == 3) acc += A[4 * a + 3];
            r[4 * a + b] = acc;
        }
    for (int k = 0; k < 12; k++) out[k] = r[k];
}

PPCR_HD inline void pack_T(const Mat3 &R, const Vec3 &t, double T[12])
{
    for (int a = 0; a < 3; a++) {
        for (int b = 0; b < 3; b++) T[4 * a + b] = R.m[a][b];
        T[4 * a + 3] = t[a];
    }
}

}  // namespace ppcr
