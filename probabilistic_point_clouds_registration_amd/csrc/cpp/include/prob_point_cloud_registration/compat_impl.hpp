// out-of-line parts of compat.hpp
#pragma once
#include <algorithm>
#include <sstream>
#include <string>

namespace Eigen {

inline Matrix3d Quaterniond::toRotationMatrix() const
{
    Matrix3d R;
    const double w = qw, x = qx, y = qy, z = qz;
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

inline Quaterniond::Quaterniond(const Matrix3d &R)
{
    const double tr = R.m[0][0] + R.m[1][1] + R.m[2][2];
    if (tr > 0) {
        const double s = 2.0 * std::sqrt(1.0 + tr);
        qw = 0.25 * s;
        qx = (R.m[2][1] - R.m[1][2]) / s;
        qy = (R.m[0][2] - R.m[2][0]) / s;
        qz = (R.m[1][0] - R.m[0][1]) / s;
    } else {
        int i = 0;
        if (R.m[1][1] > R.m[0][0]) i = 1;
        if (R.m[2][2] > R.m[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        const double s = 2.0 * std::sqrt(1.0 + R.m[i][i] - R.m[j][j] - R.m[k][k]);
        double q[3];
        q[i] = 0.25 * s;
        q[j] = (R.m[j][i] + R.m[i][j]) / s;
        q[k] = (R.m[k][i] + R.m[i][k]) / s;
        qw = (R.m[k][j] - R.m[j][k]) / s;
        qx = q[0], qy = q[1], qz = q[2];
    }
}

inline Vector3d Matrix3d::eulerAngles(int a0, int a1, int a2) const
{
    // only the (0,1,2) order the reference's report uses (src/prob_point_cloud_registration.cc:123):
    // R = Rx(a) Ry(b) Rz(c) with a in [0, pi] (Eigen's convention)
    (void)a0, (void)a1, (void)a2;
    const double pi = 3.14159265358979323846;
    Vector3d r;
    r.v[0] = std::atan2(-m[1][2], m[2][2]);
    const double c2 = std::sqrt(m[0][0] * m[0][0] + m[0][1] * m[0][1]);
    if (r.v[0] < 0) {  // map into [0, pi] like Eigen does
        r.v[0] += pi;
        r.v[1] = std::atan2(m[0][2], -c2);
    } else {
        r.v[1] = std::atan2(m[0][2], c2);
    }
    const double s1 = std::sin(r.v[0]), c1 = std::cos(r.v[0]);
    r.v[2] = std::atan2(c1 * m[1][0] + s1 * m[2][0], c1 * m[1][1] + s1 * m[2][1]);
    return r;
}

template <class T, int O>
template <class It>
void SparseMatrix<T, O>::setFromTriplets(It first, It last)
{
    std::vector<Triplet<T>> tr(first, last);
    std::stable_sort(tr.begin(), tr.end(), [](const Triplet<T> &a, const Triplet<T> &b) {
        return a.row() != b.row() ? a.row() < b.row() : a.col() < b.col();
    });
    outer_.assign(static_cast<std::size_t>(rows_) + 1, 0);
    inner_.clear();
    values_.clear();
    for (std::size_t k = 0; k < tr.size(); k++) {
        if (!inner_.empty() && k > 0 && tr[k].row() == tr[k - 1].row() && tr[k].col() == tr[k - 1].col()) {
            values_.back() += tr[k].value();  // Eigen sums duplicates
            continue;
        }
        inner_.push_back(tr[k].col());
        values_.push_back(tr[k].value());
        outer_[static_cast<std::size_t>(tr[k].row()) + 1]++;
    }
    for (long r = 0; r < rows_; r++) outer_[r + 1] += outer_[r];
}

}  // namespace Eigen

namespace ceres {
inline std::string Solver::Summary::FullReport() const
{
    std::ostringstream os;
    os << "closed-form IRLS (HIP): initial_cost " << initial_cost << " final_cost " << final_cost << " steps "
       << num_successful_steps;
    return os.str();
}
}  // namespace ceres
