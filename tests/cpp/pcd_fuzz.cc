// Sanitizer driver for the PCD reader (csrc/cpp/src/pcd_io.cc): built by tests/test_sanitizers.py with
// -fsanitize=address,undefined and run over a corpus of valid and malformed files.  For every path on the command line:
// load it; a file that loads is written back (binary) and loaded again.  Prints "<rc> <points> <path>" per file; the
// process only exits non-zero when a round trip changes a cloud — crashes and sanitizer reports abort it.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "prob_point_cloud_registration/pcd_io.hpp"

using prob_point_cloud_registration::io::loadPCDFile;
using prob_point_cloud_registration::io::savePCDFile;

int main(int argc, char **argv)
{
    int bad = 0;
    for (int k = 1; k < argc; k++) {
        pcl::PointCloud<pcl::PointXYZ> cloud;
        const int rc = loadPCDFile(argv[k], cloud);
        std::printf("%d %zu %s\n", rc, cloud.size(), argv[k]);
        if (rc != 0) continue;
        const std::string copy = std::string(argv[k]) + ".roundtrip";
        pcl::PointCloud<pcl::PointXYZ> again;
        if (savePCDFile(copy, cloud, true) != 0 || loadPCDFile(copy, again) != 0 || again.size() != cloud.size()) {
            std::printf("ROUNDTRIP FAILED %s\n", argv[k]);
            bad++;
            continue;
        }
        for (std::size_t i = 0; i < cloud.size(); i++)
            if (std::memcmp(&cloud[i].x, &again[i].x, 12) != 0) {
                std::printf("ROUNDTRIP DIFFERS %s at %zu\n", argv[k], i);
                bad++;
                break;
            }
        std::remove(copy.c_str());
    }
    return bad ? 1 : 0;
}
