// WeightUpdaterCallback — the reference's ceres::IterationCallback (weight_updater_callback.hpp:15-64) with the same
// constructor and operator()(const ceres::IterationSummary &): squared residual of every ErrorTerm at the current
// (rotation, translation), ProbabilisticWeights::updateWeights over the association's rows, every weight pushed back
// into its ErrorTerm by the same running index.
//
// Where the arithmetic runs.  The registration loop of this library never constructs ErrorTerms (K2 / K23 evaluate the
// same residuals for every stored pair on the GPU); this class exists for user code written against the reference's
// header.  Its operator() takes one of two routes with identical results:
//   * device (K2, ppcr_weights): when every ErrorTerm of a row holds the same source point — how the reference builds
//     them, one per nonzero (i, j) from source[i] and target[j], ..._iteration.hpp:37-46 — the terms' points are
//     uploaded ONCE (source = one point per row, target = one point per term, association = the rows of
//     *data_association with column k for term k) and every call is ppcr_weights(q, t) on that handle: residuals
//     (error_term.hpp:21-37) and weights (probabilistic_weights.hpp:48-105) in one kernel;
//   * host residuals + ProbabilisticWeights::updateWeights (itself a device call, ppcr_update_weights) otherwise,
//     and whenever no GPU handle can be created for the first route's set-up (updateWeights then reports the error).
// The weight_updater argument decides the model on both routes (dof / dimension are read from it).
#pragma once
#include <cstddef>
#include <memory>
#include <vector>

#include "ppcr.h"
#include "prob_point_cloud_registration/compat.hpp"
#include "prob_point_cloud_registration/device.hpp"
#include "prob_point_cloud_registration/error_term.hpp"
#include "prob_point_cloud_registration/prob_point_cloud_registration_params.hpp"
#include "prob_point_cloud_registration/probabilistic_weights.hpp"

namespace prob_point_cloud_registration {

class WeightUpdaterCallback : public ceres::IterationCallback {
public:
    WeightUpdaterCallback(Eigen::SparseMatrix<double, Eigen::RowMajor> *data_association,
                          ProbPointCloudRegistrationParams *params, std::vector<ErrorTerm *> *error_terms,
                          ProbabilisticWeights *weight_updater, double rotation[4], double translation[3])
        : data_association_(data_association), params_(params), error_terms_(error_terms), weight_updater_(weight_updater),
          rotation_(rotation), translation_(translation)
    {
    }

    ceres::CallbackReturnType operator()(const ceres::IterationSummary &) override
    {
        const std::size_t n_terms = error_terms_->size();
        // The reference re-reads *error_terms_ and *data_association_ on every call.  The device route holds a snapshot of
        // both (the terms' points, the rows): it is only taken again while the caller's objects are the ones snapshotted —
        // same ErrorTerm objects in the same order (their points never change after construction, error_term.hpp:13-20),
        // same compressed row structure — and rebuilt from scratch otherwise.
        if (route_ == kDevice && !snapshot_is_current()) {
            device_.reset();
            route_ = kUndecided;
        }
        if (route_ == kHostResiduals && n_terms != host_route_terms_) route_ = kUndecided;  // (other terms: decide again)
        if (route_ == kUndecided) prepare_device_route();
        if (route_ == kDevice) {
            weights_.resize(n_terms);
            device_->check(ppcr_weights(device_->get(), rotation_, translation_, weights_.data(), nullptr), "ppcr_weights");
            for (std::size_t k = 0; k < n_terms; ++k) error_terms_->at(k)->updateWeight(weights_[k]);
            return ceres::SOLVER_CONTINUE;
        }
        std::vector<double> squared_errors;
        squared_errors.reserve(n_terms);
        for (std::size_t k = 0; k < n_terms; ++k) {
            double r[3];
            (*(error_terms_->at(k)))(rotation_, translation_, r);
            squared_errors.push_back(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        }
        const Eigen::SparseMatrix<double, Eigen::RowMajor> w = weight_updater_->updateWeights(*data_association_, squared_errors);
        std::size_t k = 0;
        for (long i = 0; i < w.outerSize(); ++i)
            for (Eigen::SparseMatrix<double, Eigen::RowMajor>::InnerIterator it(w, i); it; ++it) error_terms_->at(k++)->updateWeight(it.value());
        return ceres::SOLVER_CONTINUE;
    }

    // which route operator() takes (decided at the first call): true = ppcr_weights on a handle holding the terms' points
    bool onDevice() const { return route_ == kDevice; }

private:
    enum Route { kUndecided, kDevice, kHostResiduals };

    bool snapshot_is_current() const
    {
        const long rows = data_association_->rows();
        if (!data_association_->isCompressed() || static_cast<std::size_t>(rows) + 1 != outer_snapshot_.size()) return false;
        if (error_terms_->size() != terms_snapshot_.size() || static_cast<std::size_t>(data_association_->nonZeros()) != terms_snapshot_.size())
            return false;
        const int *outer = data_association_->outerIndexPtr();
        for (std::size_t i = 0; i < outer_snapshot_.size(); ++i)
            if (outer[i] != outer_snapshot_[i]) return false;
        for (std::size_t k = 0; k < terms_snapshot_.size(); ++k)
            if ((*error_terms_)[k] != terms_snapshot_[k]) return false;
        return true;
    }

    void prepare_device_route()
    {
        route_ = kHostResiduals;
        host_route_terms_ = error_terms_->size();
        const long rows = data_association_->rows();
        const long nnz = data_association_->nonZeros();
        if (rows <= 0 || nnz <= 0 || static_cast<std::size_t>(nnz) != error_terms_->size()) return;
        // (outerIndexPtr()[i + 1] ends row i only in compressed storage; an uncompressed matrix keeps the host route)
        if (!data_association_->isCompressed()) return;
        const int *outer = data_association_->outerIndexPtr();
        if (outer[0] != 0 || outer[rows] != nnz) return;
        std::vector<float> src(static_cast<std::size_t>(rows) * 3, 0.f), tgt(static_cast<std::size_t>(nnz) * 3);
        std::vector<int> col(static_cast<std::size_t>(nnz));
        for (long i = 0; i < rows; ++i)
            for (int k = outer[i]; k < outer[i + 1]; ++k) {
                const ErrorTerm *e = error_terms_->at(static_cast<std::size_t>(k));
                const double *x = e->source(), *y = e->target();
                for (int a = 0; a < 3; a++) {
                    // (the terms were built from float points: the narrowing is exact, and checked)
                    const float xf = static_cast<float>(x[a]), yf = static_cast<float>(y[a]);
                    if (static_cast<double>(xf) != x[a] || static_cast<double>(yf) != y[a]) return;
                    if (k == outer[i]) src[3 * i + a] = xf;
                    else if (src[3 * i + a] != xf) return;  // a row whose terms hold different source points
                    tgt[3 * static_cast<std::size_t>(k) + a] = yf;
                }
                col[static_cast<std::size_t>(k)] = k;
            }
        try {
            std::unique_ptr<DeviceContext> dev(new DeviceContext(params_ ? params_->device_id : 0));
            ppcr_ctx *c = dev->get();
            // max_neighbours = 0: the caller-made rows are taken as they are, whatever their length
            dev->check(ppcr_set_params(c, 1.0, 0, weight_updater_->dof(), weight_updater_->dimension()), "ppcr_set_params");
            dev->check(ppcr_set_target(c, tgt.data(), nnz, 12), "ppcr_set_target");
            dev->check(ppcr_set_source(c, src.data(), rows, 12), "ppcr_set_source");
            dev->check(ppcr_set_association(c, outer, col.data(), rows), "ppcr_set_association");
            device_ = std::move(dev);
            outer_snapshot_.assign(outer, outer + rows + 1);
            terms_snapshot_.assign(error_terms_->begin(), error_terms_->end());
            route_ = kDevice;
        } catch (const DeviceError &) {
            // no handle: the host-residual route reports through updateWeights
        }
    }

    Eigen::SparseMatrix<double, Eigen::RowMajor> *data_association_;
    ProbPointCloudRegistrationParams *params_;
    std::vector<ErrorTerm *> *error_terms_;
    ProbabilisticWeights *weight_updater_;
    double *rotation_;
    double *translation_;
    Route route_ = kUndecided;
    std::unique_ptr<DeviceContext> device_;
    std::vector<double> weights_;
    std::size_t host_route_terms_ = 0;               // the term count the host route was chosen for
    std::vector<int> outer_snapshot_;                // what the handle holds: the rows ...
    std::vector<const ErrorTerm *> terms_snapshot_;  // ... and the terms whose points were uploaded, in order
};

}  // namespace prob_point_cloud_registration
