// PCD reader/writer for x y z float clouds (the on-disk format of the reference CLI:
// pcl::io::loadPCDFile / savePCDFile at src/prob_point_cloud_registration_ex.cc:113,123,132,164).
// Supported: DATA ascii, binary and binary_compressed (LZF), any field list that contains float32 x, y, z.
// Anything malformed — truncated data, a header that promises more than the file holds, a corrupt LZF stream — makes
// loadPCDFile return -1 with a message on stderr (tests/test_sanitizers.py runs the reader under ASan + UBSan).
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
