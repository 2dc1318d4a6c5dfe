// tests/mock: pcl::PointCloud as far as hotpath.hpp reads it (see README.md)
#pragma once
#include <cstdint>
#include <memory>
#include <vector>
namespace pcl {
template <typename PointT> class PointCloud {
public:
    typedef std::shared_ptr<PointCloud<PointT> > Ptr;            // boost::shared_ptr in PCL 1.7
    typedef std::shared_ptr<const PointCloud<PointT> > ConstPtr;
    std::vector<PointT> points;
    uint32_t width, height;
    PointCloud() : width(0), height(0) {}
    const PointT& at(int column, int row) const { return points[(size_t)row * width + column]; }
};
}  // namespace pcl
