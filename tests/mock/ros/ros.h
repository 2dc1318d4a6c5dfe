// tests/mock: ros::Subscriber as far as camera_info_cb uses it (see README.md)
#pragma once
namespace ros {
class Subscriber {
public:
    void shutdown() {}
};
}  // namespace ros
