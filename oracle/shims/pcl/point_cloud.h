// Stand-in for <pcl/point_cloud.h> — TEST INFRASTRUCTURE ONLY (oracle build).
#ifndef ORACLE_SHIM_PCL_POINT_CLOUD_H
#define ORACLE_SHIM_PCL_POINT_CLOUD_H
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
#include <pcl/point_types.h>

namespace pcl
{
struct PCLHeader
{
  std::uint32_t seq = 0;
  std::uint64_t stamp = 0;
  std::string frame_id;
};

template <typename PointT>
class PointCloud
{
public:
  using Ptr = std::shared_ptr<PointCloud<PointT>>;
  using ConstPtr = std::shared_ptr<const PointCloud<PointT>>;
  using VectorType = std::vector<PointT>;
  using iterator = typename VectorType::iterator;
  using const_iterator = typename VectorType::const_iterator;

  PCLHeader header;
  VectorType points;
  std::uint32_t width = 0;
  std::uint32_t height = 0;
  bool is_dense = true;

  std::size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  iterator begin() { return points.begin(); }
  iterator end() { return points.end(); }
  const_iterator begin() const { return points.begin(); }
  const_iterator end() const { return points.end(); }
  PointT& operator[](std::size_t i) { return points[i]; }
  const PointT& operator[](std::size_t i) const { return points[i]; }
  void reserve(std::size_t n) { points.reserve(n); }
  void clear()
  {
    points.clear();
    width = height = 0;
  }
  void push_back(const PointT& p)
  {
    points.push_back(p);
    width = static_cast<std::uint32_t>(points.size());
    height = 1;
  }
  iterator erase(iterator first, iterator last)
  {
    iterator it = points.erase(first, last);
    width = static_cast<std::uint32_t>(points.size());
    height = 1;
    return it;
  }
  PointCloud& operator+=(const PointCloud& rhs)
  {
    points.insert(points.end(), rhs.points.begin(), rhs.points.end());
    width = static_cast<std::uint32_t>(points.size());
    height = 1;
    return *this;
  }
  Ptr makeShared() const { return Ptr(new PointCloud<PointT>(*this)); }
};
}  // namespace pcl
#endif
