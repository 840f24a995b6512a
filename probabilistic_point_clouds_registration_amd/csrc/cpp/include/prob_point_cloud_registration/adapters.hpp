// What this implementation needs from the third-party types beyond the reference's own use of them — conversions
// between the C ABI's plain arrays (include/ppcr.h: row-major [R|t] as 12 doubles, CSR as three arrays) and
// Eigen::Affine3d / Eigen::SparseMatrix — as FREE FUNCTIONS written against members that exist in the real libraries
// (linear(), translation(), operator()(r, c), setFromTriplets, resize ...).  The stand-ins of compat.hpp carry no member
// of their own for this; with PPCR_NO_COMPAT_TYPES the same code compiles against real Eigen.
#pragma once
#include <cstddef>
#include <vector>

#include "prob_point_cloud_registration/compat.hpp"

namespace prob_point_cloud_registration {

// row-major top three rows of the 4x4 (the C ABI's T[12]) -> Affine3d and back
inline Eigen::Affine3d affineFromRows(const double T[12])
{
    Eigen::Affine3d a = Eigen::Affine3d::Identity();
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) a.linear()(r, c) = T[4 * r + c];
        a.translation()(r) = T[4 * r + 3];
    }
    return a;
}
inline void affineToRows(const Eigen::Affine3d &a, double T[12])
{
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) T[4 * r + c] = a.linear()(r, c);
        T[4 * r + 3] = a.translation()(r);
    }
}

// a row-major sparse matrix from raw CSR arrays (explicit zeros stay entries, as setFromTriplets keeps them)
inline Eigen::SparseMatrix<double, Eigen::RowMajor> sparseFromCsr(long rows, long cols, const int *outer, const int *inner,
                                                                  const double *values)
{
    std::vector<Eigen::Triplet<double>> entries;
    entries.reserve(static_cast<std::size_t>(outer[rows]));
    for (long r = 0; r < rows; r++)
        for (int k = outer[r]; k < outer[r + 1]; k++) entries.push_back(Eigen::Triplet<double>(static_cast<int>(r), inner[k], values[k]));
    Eigen::SparseMatrix<double, Eigen::RowMajor> m(rows, cols);
    m.setFromTriplets(entries.begin(), entries.end());
    m.makeCompressed();
    return m;
}

}  // namespace prob_point_cloud_registration
