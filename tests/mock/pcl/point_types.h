// tests/mock: the two PCL point types the hot calls receive (see README.md)
#pragma once
#include <cstdint>
namespace pcl {
struct PointXYZRGB { float x, y, z; uint8_t b, g, r, a; };
struct Normal { float normal_x, normal_y, normal_z, curvature; };
}  // namespace pcl
