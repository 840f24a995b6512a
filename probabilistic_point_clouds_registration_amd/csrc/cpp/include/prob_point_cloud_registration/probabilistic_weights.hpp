// ProbabilisticWeights — same constructor and updateWeights signature as the reference
// (probabilistic_weights.hpp:30,48-49); the arithmetic runs in the HIP kernel weights_from_errors_kernel
// through ppcr_update_weights().  Soft assignment per source row: softmax of the t-distribution (or, for
// v = +inf, Gaussian) log-likelihoods of the squared errors, times the t "expected weight" (v+d)/(v+s).
#pragma once
#include <cassert>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "ppcr.h"
#include "prob_point_cloud_registration/adapters.hpp"

namespace prob_point_cloud_registration {

class ProbabilisticWeights {
public:
    ProbabilisticWeights(double v, int dimension, int max_neighbours, int device_id = 0)
        : v_(v), dimension_(dimension), max_neighbours_(max_neighbours), device_id_(device_id)
    {
        assert(dimension > 0);  // same preconditions the reference asserts (:33-34)
        assert(v > 0.0);
    }

    // squared_errors are in the row-major storage order of data_association; the result has the same
    // sparsity pattern with the weights as values
    Eigen::SparseMatrix<double, Eigen::RowMajor> updateWeights(
        Eigen::SparseMatrix<double, Eigen::RowMajor> data_association, std::vector<double> squared_errors) const
    {
        const long rows = data_association.rows(), nnz = data_association.nonZeros();
        if (static_cast<long>(squared_errors.size()) < nnz) throw std::invalid_argument("updateWeights: too few squared errors");
        std::vector<double> w(static_cast<std::size_t>(nnz));
        const int rc = ppcr_update_weights(device_id_, data_association.outerIndexPtr(), rows, squared_errors.data(), v_,
                                           dimension_, w.data());
        if (rc != PPCR_OK) throw std::runtime_error(std::string("ppcr_update_weights: ") + ppcr_last_error(nullptr));
        return sparseFromCsr(rows, data_association.cols(), data_association.outerIndexPtr(), data_association.innerIndexPtr(), w.data());
    }

    bool isNormal() const { return !(v_ < std::numeric_limits<double>::infinity()); }
    // additions of this implementation: the model's two constructor arguments (WeightUpdaterCallback's device route)
    double dof() const { return v_; }
    int dimension() const { return dimension_; }

private:
    double v_;
    int dimension_;
    int max_neighbours_;  // only sized a reserve() in the reference
    int device_id_;
};

}  // namespace prob_point_cloud_registration
