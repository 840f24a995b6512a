// ProbPointCloudRegistration — the outer registration driver with the reference's public surface
// (prob_point_cloud_registration.h:18-64): constructors, align(), hasConverged(), transformation(),
// transformation_history(), report().  The loop body of align() runs on the GPU (ppcr_iterate).
#pragma once
#include <memory>
#include <sstream>
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
    void align();
    bool hasConverged();
    inline Eigen::Affine3d transformation() { return transformation_history_.back(); }
    inline std::vector<Eigen::Affine3d> transformation_history() { return transformation_history_; }
    inline std::string report() { return report_.str(); }

private:
    void fetchSource();  // device -> source_cloud_ (once, after the loop)

    ProbPointCloudRegistrationParams parameters_;
    pcl::PointCloud<pcl::PointXYZ>::Ptr target_cloud_;
    pcl::PointCloud<pcl::PointXYZ>::Ptr source_cloud_;
    pcl::PointCloud<pcl::PointXYZ>::Ptr filtered_source_cloud_;
    pcl::PointCloud<pcl::PointXYZ>::Ptr prev_source_cloud_;
    pcl::PointCloud<pcl::PointXYZ>::Ptr ground_truth_cloud_;
    bool ground_truth_;
    bool filtered_;
    double mse_ground_truth_;
    double mse_prev_it_;
    double cost_drop_;
    int num_unusefull_iter_;
    int current_iteration_;
    OutputStream output_stream_;
    std::vector<Eigen::Affine3d> transformation_history_;
    std::stringstream report_;
    std::unique_ptr<DeviceContext> device_;
};

}  // namespace prob_point_cloud_registration
