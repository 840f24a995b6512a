// calculateMSE (utilities.hpp:16-26 of the reference): mean Euclidean DISTANCE of index-paired points,
// float arithmetic per pair like pcl::euclideanDistance, double accumulation.
#pragma once
#include <cassert>
#include <cmath>

#include "prob_point_cloud_registration/compat.hpp"

namespace prob_point_cloud_registration {

inline double calculateMSE(pcl::PointCloud<pcl::PointXYZ>::Ptr cloud1, pcl::PointCloud<pcl::PointXYZ>::Ptr cloud2)
{
    assert(cloud1->size() == cloud2->size());
    double mse = 0;
    for (std::size_t i = 0; i < cloud1->size(); i++) {
        const float dx = cloud1->at(i).x - cloud2->at(i).x, dy = cloud1->at(i).y - cloud2->at(i).y,
                    dz = cloud1->at(i).z - cloud2->at(i).z;
        mse += std::sqrt(dx * dx + dy * dy + dz * dz);
    }
    return mse / static_cast<double>(cloud1->size());
}

// in-place/out-of-place pcl::transformPointCloud(cloud_in, cloud_out, Affine3d): f64 math, f32 store
inline void transformPointCloud(const pcl::PointCloud<pcl::PointXYZ> &in, pcl::PointCloud<pcl::PointXYZ> &out,
                                const Eigen::Affine3d &T)
{
    if (&in != &out) out.points.resize(in.size());
    for (std::size_t i = 0; i < in.size(); i++) {
        const double x = in[i].x, y = in[i].y, z = in[i].z;
        pcl::PointXYZ p = in[i];
        p.x = static_cast<float>(((T.R.m[0][0] * x + T.R.m[0][1] * y) + T.R.m[0][2] * z) + T.t.v[0]);
        p.y = static_cast<float>(((T.R.m[1][0] * x + T.R.m[1][1] * y) + T.R.m[1][2] * z) + T.t.v[1]);
        p.z = static_cast<float>(((T.R.m[2][0] * x + T.R.m[2][1] * y) + T.R.m[2][2] * z) + T.t.v[2]);
        out[i] = p;
    }
}

}  // namespace prob_point_cloud_registration

namespace pcl {
using prob_point_cloud_registration::transformPointCloud;
}
