// API-subset mock (see ../../README.md)
#pragma once
#include <Eigen/Geometry>
#include <pcl/point_cloud.h>
namespace pcl {
template <class PointT>
void transformPointCloud(const PointCloud<PointT> &cloud_in, PointCloud<PointT> &cloud_out, const Eigen::Affine3d &transform,
                         bool copy_all_fields = true);
}
