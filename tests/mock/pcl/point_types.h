// tests/mock: the two PCL point types the hot calls receive (see README.md), with PCL's memory layout: both are
// 16-byte aligned and padded to 32 bytes (x y z pad | b g r a | pad..., normal_x normal_y normal_z pad | curvature | pad...)
#pragma once
#include <cstdint>
namespace pcl {
struct alignas(16) PointXYZRGB { float x, y, z, data3; uint8_t b, g, r, a; };
struct alignas(16) Normal { float normal_x, normal_y, normal_z, data_n3; float curvature; };
static_assert(sizeof(PointXYZRGB) == 32 && sizeof(Normal) == 32, "PCL pads both point types to 32 bytes");
}  // namespace pcl
