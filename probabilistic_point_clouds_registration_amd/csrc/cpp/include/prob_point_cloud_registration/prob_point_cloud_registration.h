// ProbPointCloudRegistration — the outer registration driver.  Public surface = the reference's
// (prob_point_cloud_registration.h:18-45 there): two constructors, align(), hasConverged(), transformation(),
// transformation_history(), report().  Everything behind it is this library's own: the state lives in a
// private implementation object (src/prob_point_cloud_registration.cc) that owns one device handle of the C ABI;
// align() is one call into the device-paced loop (ppcr_align_report) and the stopping rule is ppcr_stop_rule (ppcr.h).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "prob_point_cloud_registration/compat.hpp"
#include "prob_point_cloud_registration/device.hpp"
#include "prob_point_cloud_registration/output_stream.hpp"
#include "prob_point_cloud_registration/prob_point_cloud_registration_iteration.hpp"
#include "prob_point_cloud_registration/prob_point_cloud_registration_params.hpp"

namespace prob_point_cloud_registration {

// centroid-per-voxel down-sampling (the pcl::VoxelGrid step of the reference's constructor) on the device,
// runs once before the loop
void voxelGridFilter(const pcl::PointCloud<pcl::PointXYZ> &in, double leaf, pcl::PointCloud<pcl::PointXYZ> &out,
                     int device_id = 0);

class ProbPointCloudRegistration {
public:
    ProbPointCloudRegistration(pcl::PointCloud<pcl::PointXYZ>::Ptr source_cloud,
                               pcl::PointCloud<pcl::PointXYZ>::Ptr target_cloud,
                               ProbPointCloudRegistrationParams parameters);
    ProbPointCloudRegistration(pcl::PointCloud<pcl::PointXYZ>::Ptr source_cloud,
                               pcl::PointCloud<pcl::PointXYZ>::Ptr target_cloud,
                               ProbPointCloudRegistrationParams parameters,
                               pcl::PointCloud<pcl::PointXYZ>::Ptr ground_truth_cloud);
    ~ProbPointCloudRegistration();
    ProbPointCloudRegistration(const ProbPointCloudRegistration &) = delete;
    ProbPointCloudRegistration &operator=(const ProbPointCloudRegistration &) = delete;

    void align();
    bool hasConverged();
    Eigen::Affine3d transformation();
    std::vector<Eigen::Affine3d> transformation_history();
    std::string report();

private:
    struct State;
    std::unique_ptr<State> state_;
};

}  // namespace prob_point_cloud_registration
