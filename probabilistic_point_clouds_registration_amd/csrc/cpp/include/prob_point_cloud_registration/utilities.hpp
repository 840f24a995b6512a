// calculateMSE (utilities.hpp:16-26 of the reference): mean Euclidean DISTANCE of index-paired points,
// float arithmetic per pair like pcl::euclideanDistance, double accumulation.
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "ppcr.h"
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

// ---- closest-point evaluation metrics (utilities.hpp:28-234 of the reference) ---------------------------------------
// All of them are statistics of d_i = SQUARED distance from point i of cloud1 to its nearest neighbour in cloud2
// (pcl::KdTreeFLANN::nearestKSearch, k = 1, returns squared distances).  The search runs on the device
// (ppcr_nearest_sq_distances); the statistics keep the reference's exact index conventions, including its
// "median" (element (n+1)/2 of the sorted list for odd n, mean of elements n/2 and n/2+1 for even n: one past the
// textbook median).  Like the reference they need cloud2 non-empty and, for the medians, n > 2.
namespace detail {
inline std::vector<float> closestSquaredDistances(const pcl::PointCloud<pcl::PointXYZ> &cloud1,
                                                  const pcl::PointCloud<pcl::PointXYZ> &cloud2, int device_id)
{
    std::vector<float> d(cloud1.size());
    const int rc = ppcr_nearest_sq_distances(device_id, cloud1.empty() ? nullptr : &cloud1[0].x,
                                             static_cast<int64_t>(cloud1.size()), sizeof(pcl::PointXYZ),
                                             cloud2.empty() ? nullptr : &cloud2[0].x, static_cast<int64_t>(cloud2.size()),
                                             sizeof(pcl::PointXYZ), d.data());
    if (rc != PPCR_OK) throw std::runtime_error(std::string("ppcr_nearest_sq_distances: ") + ppcr_last_error(nullptr));
    return d;
}
template <class V>
double referenceMedian(const V &sorted)  // the reference's convention (see above); needs size() > 2
{
    const std::size_t n = sorted.size();
    if (n % 2 != 0) return sorted.at((n + 1) / 2);
    return (sorted.at(n / 2) + sorted.at(n / 2 + 1)) / 2.0;
}
}  // namespace detail

inline double averageClosestDistance(pcl::PointCloud<pcl::PointXYZ>::Ptr cloud1, pcl::PointCloud<pcl::PointXYZ>::Ptr cloud2,
                                     int device_id = 0)
{
    const std::vector<float> d = detail::closestSquaredDistances(*cloud1, *cloud2, device_id);
    double sum = 0;
    for (float v : d) sum += v;
    return sum / static_cast<double>(cloud1->size());
}

inline double sumSquaredError(pcl::PointCloud<pcl::PointXYZ>::Ptr cloud1, pcl::PointCloud<pcl::PointXYZ>::Ptr cloud2,
                              int device_id = 0)
{
    const std::vector<float> d = detail::closestSquaredDistances(*cloud1, *cloud2, device_id);
    double sum = 0;
    for (float v : d) sum += v;
    return sum;
}

// sum of the d_i within [median / factor, median * factor]; DBL_MAX when fewer than 10 survive
inline double robustSumSquaredError(pcl::PointCloud<pcl::PointXYZ>::Ptr cloud1, pcl::PointCloud<pcl::PointXYZ>::Ptr cloud2,
                                    double factor, int device_id = 0, int *num_filtered_out = nullptr)
{
    const std::vector<float> d = detail::closestSquaredDistances(*cloud1, *cloud2, device_id);
    std::vector<double> all(d.begin(), d.end());
    std::sort(all.begin(), all.end());
    const double median = detail::referenceMedian(all);
    double sum = 0;
    int num_filtered = 0;
    for (double v : all)
        if (v <= median * factor && v >= median / factor) {
            sum += v;
            num_filtered++;
        }
    if (num_filtered_out) *num_filtered_out = num_filtered;
    if (num_filtered < 10) return std::numeric_limits<double>::max();
    return sum;
}
inline double robustSumSquaredError(pcl::PointCloud<pcl::PointXYZ>::Ptr cloud1, pcl::PointCloud<pcl::PointXYZ>::Ptr cloud2)
{
    return robustSumSquaredError(cloud1, cloud2, 3.0);
}

inline double robustAveragedSumSquaredError(pcl::PointCloud<pcl::PointXYZ>::Ptr cloud1,
                                            pcl::PointCloud<pcl::PointXYZ>::Ptr cloud2, int device_id = 0)
{
    int num_filtered = 0;
    const double sum = robustSumSquaredError(cloud1, cloud2, 3.0, device_id, &num_filtered);
    if (num_filtered < 10) return std::numeric_limits<double>::max();
    return sum / num_filtered;
}

inline double medianClosestDistance(pcl::PointCloud<pcl::PointXYZ>::Ptr cloud1, pcl::PointCloud<pcl::PointXYZ>::Ptr cloud2,
                                    int device_id = 0)
{
    std::vector<float> d = detail::closestSquaredDistances(*cloud1, *cloud2, device_id);
    std::sort(d.begin(), d.end());
    return detail::referenceMedian(d);  // float elements, like the reference's vector<float>
}

inline double robustMedianClosestDistance(pcl::PointCloud<pcl::PointXYZ>::Ptr cloud1,
                                          pcl::PointCloud<pcl::PointXYZ>::Ptr cloud2, int device_id = 0)
{
    std::vector<float> d = detail::closestSquaredDistances(*cloud1, *cloud2, device_id);
    std::sort(d.begin(), d.end());
    const double median = detail::referenceMedian(d);
    std::vector<float> filtered;
    for (float v : d)
        if (v <= median * 3 && v >= median / 3.0) filtered.push_back(v);
    return detail::referenceMedian(filtered) / filtered.size();  // the reference divides the median by the count
}

// median of the association's stored values (the float d2 of every correspondence) with the reference's index
// convention (utilities.hpp:236-250: one past the textbook median, see detail::referenceMedian); needs size() > 2
inline double medianDistance(std::vector<Eigen::Triplet<double>> tripletList)
{
    std::vector<double> values;
    values.reserve(tripletList.size());
    for (const auto &t : tripletList) values.push_back(t.value());
    std::sort(values.begin(), values.end());
    return detail::referenceMedian(values);
}

// q = Rz(yaw) * Ry(pitch) * Rx(roll) (utilities.hpp:252-263)
inline Eigen::Quaterniond euler2Quaternion(const double roll, const double pitch, const double yaw)
{
    const Eigen::AngleAxisd about_x(roll, Eigen::Vector3d::UnitX()), about_y(pitch, Eigen::Vector3d::UnitY()),
        about_z(yaw, Eigen::Vector3d::UnitZ());
    return about_z * about_y * about_x;
}

#ifndef PPCR_NO_COMPAT_TYPES
// in-place/out-of-place pcl::transformPointCloud(cloud_in, cloud_out, Affine3d): f64 math, f32 store
// (stand-in builds only: with the real libraries pcl/common/transforms.h provides it)
inline void transformPointCloud(const pcl::PointCloud<pcl::PointXYZ> &in, pcl::PointCloud<pcl::PointXYZ> &out,
                                const Eigen::Affine3d &T)
{
    if (&in != &out) out.points.resize(in.size());
    for (std::size_t i = 0; i < in.size(); i++) {
        const double x = in[i].x, y = in[i].y, z = in[i].z;
        pcl::PointXYZ p = in[i];
        const Eigen::Matrix3d &R = T.linear();
        const Eigen::Vector3d &t = T.translation();
        p.x = static_cast<float>(((R(0, 0) * x + R(0, 1) * y) + R(0, 2) * z) + t(0));
        p.y = static_cast<float>(((R(1, 0) * x + R(1, 1) * y) + R(1, 2) * z) + t(1));
        p.z = static_cast<float>(((R(2, 0) * x + R(2, 1) * y) + R(2, 2) * z) + t(2));
        out[i] = p;
    }
}
#endif

}  // namespace prob_point_cloud_registration

#ifndef PPCR_NO_COMPAT_TYPES
namespace pcl {
using prob_point_cloud_registration::transformPointCloud;
}
#endif
