// tests/mock: sensor_msgs::CameraInfo's K field (see README.md)
#pragma once
#include <memory>
namespace sensor_msgs {
struct CameraInfo { double K[9]; };
typedef std::shared_ptr<CameraInfo const> CameraInfoConstPtr;
}  // namespace sensor_msgs
