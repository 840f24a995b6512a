// API-subset mock (see ../README.md)
#pragma once
namespace pcl {
struct alignas(16) PointXYZ {
    float x, y, z;
    float data_pad_;  // (real: the union's fourth float)
    PointXYZ();
    PointXYZ(float x_, float y_, float z_);
};
}  // namespace pcl
