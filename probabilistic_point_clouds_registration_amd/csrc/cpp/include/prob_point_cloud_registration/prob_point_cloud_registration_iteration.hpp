// ProbPointCloudRegistrationIteration — one fixed-association solve, same constructor / solve /
// transformation surface as the reference (prob_point_cloud_registration_iteration.hpp:21-78).
// The reference builds a Ceres problem (one ErrorTerm + AutoDiff cost + loss wrapper per nonzero) and
// lets LM + a weight callback reach the fixed point; here the association is installed on the GPU
// (ppcr_set_association) and ppcr_solve runs IRLS with the closed-form weighted rigid solve to the same
// fixed point.  Options honoured: function_tolerance, max_num_iterations.
#pragma once
#include <memory>

#include "prob_point_cloud_registration/adapters.hpp"
#include "prob_point_cloud_registration/device.hpp"
#include "prob_point_cloud_registration/error_term.hpp"
#include "prob_point_cloud_registration/prob_point_cloud_registration_params.hpp"
#include "prob_point_cloud_registration/probabilistic_weights.hpp"
#include "prob_point_cloud_registration/weight_updater_callback.hpp"  // (as the reference's header does, :14)

#define DIMENSIONS 3

namespace prob_point_cloud_registration {

class ProbPointCloudRegistrationIteration {
public:
    ProbPointCloudRegistrationIteration(const pcl::PointCloud<pcl::PointXYZ> &source_cloud,
                                        const pcl::PointCloud<pcl::PointXYZ> &target_cloud,
                                        const Eigen::SparseMatrix<double, Eigen::RowMajor> &data_association,
                                        ProbPointCloudRegistrationParams parameters)
        : parameters_(parameters), device_(new DeviceContext(parameters.device_id))
    {
        ppcr_ctx *c = device_->get();
        device_->check(ppcr_set_params(c, parameters_.radius, parameters_.max_neighbours, parameters_.dof, DIMENSIONS),
                       "ppcr_set_params");
        device_->check(ppcr_set_target(c, target_cloud.size() ? &target_cloud[0].x : nullptr,
                                       static_cast<int64_t>(target_cloud.size()), sizeof(pcl::PointXYZ)),
                       "ppcr_set_target");
        device_->check(ppcr_set_source(c, source_cloud.size() ? &source_cloud[0].x : nullptr,
                                       static_cast<int64_t>(source_cloud.size()), sizeof(pcl::PointXYZ)),
                       "ppcr_set_source");
        device_->check(ppcr_set_association(c, data_association.outerIndexPtr(), data_association.innerIndexPtr(),
                                            data_association.rows()),
                       "ppcr_set_association");
        for (int i = 0; i < 4; i++) rotation_[i] = parameters_.initial_rotation[i];
        for (int i = 0; i < 3; i++) translation_[i] = parameters_.initial_translation[i];
        for (int i = 0; i < 12; i++) T_[i] = (i % 5 == 0) ? 1.0 : 0.0;
        solved_ = false;
    }

    void solve(ceres::Solver::Options options, ceres::Solver::Summary *summary)
    {
        const int max_steps = options.max_num_iterations > 100000 ? 100000 : options.max_num_iterations;
        double cost[2] = {0, 0};
        int steps = 0;
        device_->check(ppcr_solve(device_->get(), rotation_, translation_, max_steps, options.function_tolerance, T_, cost,
                                  &steps),
                       "ppcr_solve");
        solved_ = true;
        if (summary) {
            summary->initial_cost = cost[0];
            summary->final_cost = cost[1];
            summary->num_successful_steps = steps;
        }
    }

    // estimated transform; before solve(): the (normalised) initial rotation / translation of the params
    Eigen::Affine3d transformation()
    {
        if (solved_) return affineFromRows(T_);
        Eigen::Quaterniond q(rotation_[0], rotation_[1], rotation_[2], rotation_[3]);
        q.normalize();
        Eigen::Affine3d a = Eigen::Affine3d::Identity();
        a.linear() = q.toRotationMatrix();
        a.translation() = Eigen::Vector3d(translation_[0], translation_[1], translation_[2]);
        return a;
    }

    DeviceContext &device() { return *device_; }

private:
    ProbPointCloudRegistrationParams parameters_;
    std::unique_ptr<DeviceContext> device_;
    double rotation_[4];
    double translation_[3];
    double T_[12];
    bool solved_;
};

}  // namespace prob_point_cloud_registration
