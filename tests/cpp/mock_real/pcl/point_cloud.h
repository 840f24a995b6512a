// API-subset mock (see ../README.md)
#pragma once
#include <cstddef>
#include <memory>
#include <vector>
namespace pcl {
template <class PointT>
class PointCloud {
public:
    using Ptr = std::shared_ptr<PointCloud<PointT>>;
    using ConstPtr = std::shared_ptr<const PointCloud<PointT>>;
    std::vector<PointT> points;
    std::size_t size() const;
    bool empty() const;
    void reserve(std::size_t n);
    void resize(std::size_t n);
    void clear();
    void push_back(const PointT &p);
    PointT &operator[](std::size_t i);
    const PointT &operator[](std::size_t i) const;
    PointT &at(std::size_t i);
    const PointT &at(std::size_t i) const;
    typename std::vector<PointT>::iterator begin();
    typename std::vector<PointT>::iterator end();
    typename std::vector<PointT>::const_iterator begin() const;
    typename std::vector<PointT>::const_iterator end() const;
};
}  // namespace pcl
