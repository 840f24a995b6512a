// PCD reader/writer for x y z float clouds (the on-disk format of the reference CLI:
// pcl::io::loadPCDFile / savePCDFile at src/prob_point_cloud_registration_ex.cc:113,123,132,164).
// Supported: DATA ascii and DATA binary, any field list that contains float32 x, y, z.
// Not supported: binary_compressed (LZF) — loadPCDFile returns -1 with a message on stderr.
#pragma once
#include <string>

#include "prob_point_cloud_registration/compat.hpp"

namespace prob_point_cloud_registration {
namespace io {

// returns 0 on success, -1 on failure (like pcl::io::loadPCDFile)
int loadPCDFile(const std::string &file_name, pcl::PointCloud<pcl::PointXYZ> &cloud);
// ASCII by default, like pcl::io::savePCDFile(name, cloud)
int savePCDFile(const std::string &file_name, const pcl::PointCloud<pcl::PointXYZ> &cloud, bool binary_mode = false);

}  // namespace io
}  // namespace prob_point_cloud_registration
