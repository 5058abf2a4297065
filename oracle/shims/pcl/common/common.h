// Stand-in for <pcl/common/common.h> — TEST INFRASTRUCTURE ONLY (oracle build).
// pcl::getMinMax3D: component-wise min/max of the x,y,z of every (finite) point.
#ifndef ORACLE_SHIM_PCL_COMMON_COMMON_H
#define ORACLE_SHIM_PCL_COMMON_COMMON_H
#include <cfloat>
#include <cmath>
#include <Eigen/Core>
#include <pcl/point_cloud.h>

namespace pcl
{
template <typename PointT>
inline void getMinMax3D(const PointCloud<PointT>& cloud, Eigen::Vector4f& min_pt, Eigen::Vector4f& max_pt)
{
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  float mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (const auto& p : cloud.points)
  {
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z))
      continue;
    const float v[3] = {p.x, p.y, p.z};
    for (int i = 0; i < 3; ++i)
    {
      if (v[i] < mn[i]) mn[i] = v[i];
      if (v[i] > mx[i]) mx[i] = v[i];
    }
  }
  for (int i = 0; i < 3; ++i)
  {
    min_pt[i] = mn[i];
    max_pt[i] = mx[i];
  }
  min_pt[3] = max_pt[3] = 0.0f;
}
}  // namespace pcl
#endif
