// Outer registration driver on top of the C ABI (include/ppcr.h).  Mirrors the behaviour of the
// reference's src/prob_point_cloud_registration.cc (constructor filtering, align loop, hasConverged,
// report rows) with the loop body executed by ppcr_iterate() on the GPU.
#include "prob_point_cloud_registration/prob_point_cloud_registration.h"

#include <cmath>
#include <cstdint>
#include <iostream>
#include <map>
#include <tuple>

#include "prob_point_cloud_registration/utilities.hpp"

namespace prob_point_cloud_registration {

// pcl::VoxelGrid centroid down-sampling on the device (ppcr_voxel_filter): one point per occupied voxel, voxels in
// ascending index.  (Parity with PCL itself is unpinned: the reference has no test for it and PCL is not
// available here; the definition is pinned against the CPU restatement in the test tree.)
void voxelGridFilter(const pcl::PointCloud<pcl::PointXYZ> &in, double leaf, pcl::PointCloud<pcl::PointXYZ> &out,
                     int device_id)
{
    pcl::PointCloud<pcl::PointXYZ> result;
    if (!(leaf > 0) || in.empty()) {
        result = in;
        out = result;
        return;
    }
    result.points.resize(in.size());
    int64_t n_out = 0;
    const int rc = ppcr_voxel_filter(device_id, &in[0].x, static_cast<int64_t>(in.size()), sizeof(pcl::PointXYZ),
                                     static_cast<float>(leaf), &result.points[0].x, sizeof(pcl::PointXYZ), &n_out);
    if (rc != PPCR_OK) throw DeviceError(rc, std::string("ppcr_voxel_filter: ") + ppcr_last_error(nullptr));
    result.points.resize(static_cast<size_t>(n_out));
    out = result;
}

ProbPointCloudRegistration::ProbPointCloudRegistration(pcl::PointCloud<pcl::PointXYZ>::Ptr source_cloud,
                                                       pcl::PointCloud<pcl::PointXYZ>::Ptr target_cloud,
                                                       ProbPointCloudRegistrationParams parameters)
    : parameters_(parameters),
      target_cloud_(target_cloud),
      ground_truth_(false),
      filtered_(false),
      mse_ground_truth_(0),
      mse_prev_it_(0),
      cost_drop_(0),
      num_unusefull_iter_(0),
      current_iteration_(0),
      output_stream_(parameters.verbose)
{
    // the source is deep-copied, the caller's target is shared (and filtered in place when asked)
    source_cloud_ = std::make_shared<pcl::PointCloud<pcl::PointXYZ>>(*source_cloud);
    filtered_source_cloud_ = std::make_shared<pcl::PointCloud<pcl::PointXYZ>>();
    if (parameters_.source_filter_size > 0) {
        output_stream_ << "Filtering source point cloud with leaf of size " << parameters_.source_filter_size << "\n";
        voxelGridFilter(*source_cloud_, parameters_.source_filter_size, *filtered_source_cloud_, parameters_.device_id);
        filtered_ = true;
    } else {
        *filtered_source_cloud_ = *source_cloud_;
    }
    if (parameters_.target_filter_size > 0) {
        output_stream_ << "Filtering target point cloud with leaf of size " << parameters_.target_filter_size << "\n";
        voxelGridFilter(*target_cloud_, parameters_.target_filter_size, *target_cloud_, parameters_.device_id);
    }
    if (parameters_.summary) {
        report_ << "iter, n_success_steps, initial_cost, final_cost, tx, ty, tz, roll, pitch, yaw, mse_prev_iter, mse_gtruth"
                << std::endl;
    }
    device_.reset(new DeviceContext(parameters_.device_id));
    ppcr_ctx *c = device_->get();
    device_->check(ppcr_set_params(c, parameters_.radius, parameters_.max_neighbours, parameters_.dof, DIMENSIONS),
                   "ppcr_set_params");
    device_->check(ppcr_set_target(c, target_cloud_->size() ? &(*target_cloud_)[0].x : nullptr,
                                   static_cast<int64_t>(target_cloud_->size()), sizeof(pcl::PointXYZ)),
                   "ppcr_set_target");
    // the association runs on the (possibly filtered) copy; when the two differ the full copy rides along on the
    // device as the handle's companion: moved by every iteration, looked at by the reports, read back once
    device_->check(ppcr_set_source(c, filtered_source_cloud_->size() ? &(*filtered_source_cloud_)[0].x : nullptr,
                                   static_cast<int64_t>(filtered_source_cloud_->size()), sizeof(pcl::PointXYZ)),
                   "ppcr_set_source");
    if (filtered_)
        device_->check(ppcr_set_companion(c, source_cloud_->size() ? &(*source_cloud_)[0].x : nullptr,
                                          static_cast<int64_t>(source_cloud_->size()), sizeof(pcl::PointXYZ)),
                       "ppcr_set_companion");
    if (parameters_.summary) device_->check(ppcr_mse_previous(c, nullptr), "ppcr_mse_previous");  // prev = source (cc:51)
}

ProbPointCloudRegistration::ProbPointCloudRegistration(pcl::PointCloud<pcl::PointXYZ>::Ptr source_cloud,
                                                       pcl::PointCloud<pcl::PointXYZ>::Ptr target_cloud,
                                                       ProbPointCloudRegistrationParams parameters,
                                                       pcl::PointCloud<pcl::PointXYZ>::Ptr ground_truth_cloud)
    : ProbPointCloudRegistration(source_cloud, target_cloud, parameters)
{
    ground_truth_cloud_ = std::make_shared<pcl::PointCloud<pcl::PointXYZ>>(*ground_truth_cloud);
    ground_truth_ = true;
    device_->check(ppcr_set_ground_truth(device_->get(), ground_truth_cloud_->size() ? &(*ground_truth_cloud_)[0].x : nullptr,
                                         static_cast<int64_t>(ground_truth_cloud_->size()), sizeof(pcl::PointXYZ)),
                   "ppcr_set_ground_truth");
    device_->check(ppcr_mse_ground_truth(device_->get(), &mse_ground_truth_), "ppcr_mse_ground_truth");
    output_stream_ << "Initial MSE w.r.t. ground truth: " << mse_ground_truth_ << "\n";
}

ProbPointCloudRegistration::~ProbPointCloudRegistration() = default;

// the caller-visible full-resolution source, read back from the device (companion when the source was filtered)
void ProbPointCloudRegistration::fetchSource()
{
    if (source_cloud_->empty()) return;
    if (filtered_)
        device_->check(ppcr_get_companion(device_->get(), &(*source_cloud_)[0].x, sizeof(pcl::PointXYZ)), "ppcr_get_companion");
    else
        device_->check(ppcr_get_source(device_->get(), &(*source_cloud_)[0].x, sizeof(pcl::PointXYZ)), "ppcr_get_source");
}

void ProbPointCloudRegistration::align()
{
    while (!hasConverged()) {
        double Tk[12], cost[2];
        int steps = 0;
        device_->check(ppcr_iterate(device_->get(), parameters_.initial_rotation, parameters_.initial_translation,
                                    parameters_.inner_max_steps, 10e-6 /* function_tolerance of the reference */, Tk, cost,
                                    &steps),
                       "ppcr_iterate");
        const Eigen::Affine3d incremental = Eigen::Affine3d::from_rows(Tk);
        Eigen::Affine3d current_trans = incremental;
        if (current_iteration_ > 0) current_trans = incremental * transformation_history_.back();
        transformation_history_.push_back(current_trans);
        output_stream_ << "iteration " << current_iteration_ << ": initial_cost " << cost[0] << " final_cost " << cost[1]
                       << " inner steps " << steps << "\n";
        // both copies of the source were moved on the device by ppcr_iterate; the reports are device reductions
        if (ground_truth_) {
            device_->check(ppcr_mse_ground_truth(device_->get(), &mse_ground_truth_), "ppcr_mse_ground_truth");
            output_stream_ << "MSE w.r.t. ground truth: " << mse_ground_truth_ << "\n";
        }
        cost_drop_ = (cost[0] - cost[1]) / cost[0];
        if (parameters_.summary) {
            device_->check(ppcr_mse_previous(device_->get(), &mse_prev_it_), "ppcr_mse_previous");
            const Eigen::Vector3d rpy = current_trans.rotation().eulerAngles(0, 1, 2);
            report_ << current_iteration_ << ", " << steps << ", " << cost[0] << ", " << cost[1] << ", "
                    << current_trans.translation().x() << ", " << current_trans.translation().y() << ", "
                    << current_trans.translation().z() << ", " << pcl::rad2deg(rpy(0, 0)) << ", " << pcl::rad2deg(rpy(1, 0))
                    << ", " << pcl::rad2deg(rpy(2, 0)) << ", " << mse_prev_it_ << ", " << mse_ground_truth_ << std::endl;
        }
        current_iteration_++;
    }
    fetchSource();
    if (ground_truth_) {
        device_->check(ppcr_mse_ground_truth(device_->get(), &mse_ground_truth_), "ppcr_mse_ground_truth");
        std::cout << "MSE w.r.t. ground truth: " << mse_ground_truth_ << std::endl;
    }
}

bool ProbPointCloudRegistration::hasConverged()
{
    if (current_iteration_ == parameters_.n_iter) {
        output_stream_ << "Terminating because maximum number of iterations has been reached ( " << current_iteration_
                       << " iter)\n";
        return true;
    }
    if (cost_drop_ < parameters_.cost_drop_thresh) {
        if (num_unusefull_iter_ > parameters_.n_cost_drop_it) {
            output_stream_ << "Terminating because cost drop has been under " << parameters_.cost_drop_thresh * 100
                           << " % for more than " << parameters_.n_cost_drop_it << " iterations\n";
            return true;
        }
        num_unusefull_iter_++;
    } else {
        num_unusefull_iter_ = 0;
    }
    return false;
}

}  // namespace prob_point_cloud_registration
