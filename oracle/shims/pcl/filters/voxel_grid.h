// Stand-in for <pcl/filters/voxel_grid.h> — TEST INFRASTRUCTURE ONLY (oracle build).
//
// PCL is not vendored by the reference (CMakeLists.txt:26-31 only asks for PCL >= 1.8) and is not installed here, so
// pcl::VoxelGrid<PointT>::applyFilter is RESTATED from its published algorithm (PCL 1.8-1.12, filters/impl/voxel_grid.hpp),
// for the only configuration the node uses — setInputCloud + setLeafSize + filter (src/mcl_3dl.cpp:147-151, 363-367,
// 1155-1158): no filter field, downsample_all_data_ = true, min_points_per_voxel_ = 0, save_leaf_layout_ = false.
//   1. getMinMax3D over the finite points.
//   2. dx,dy,dz = int64((max - min) * inverse_leaf_size) + 1; if dx*dy*dz > INT32_MAX: warning, output = input.
//   3. min_b = floor(min * inverse_leaf_size), max_b likewise, div_b = max_b - min_b + 1, divb_mul = (1, div_b0, div_b0*div_b1).
//   4. every finite point: ijk = int(floor(p * inverse_leaf_size) - float(min_b)); idx = ijk . divb_mul.
//   5. sort by idx; one output point per distinct idx, in ascending idx order: the centroid (pcl::CentroidPoint): xyz =
//      float sum / count, intensity likewise, label = the most frequent label, the smallest on a tie (AccumulatorLabel
//      walks a std::map with a strict `>`).
// Parity unpinned at this boundary: PCL's std::sort is not stable, so the order in which the points of one leaf are
// summed — and with it the last bit of a centroid — is unspecified upstream; this restatement (and the GPU path) sums in
// input order.
#ifndef ORACLE_SHIM_PCL_FILTERS_VOXEL_GRID_H
#define ORACLE_SHIM_PCL_FILTERS_VOXEL_GRID_H
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <map>
#include <vector>

#include <pcl/point_cloud.h>

namespace pcl
{
template <typename PointT>
class VoxelGrid
{
public:
  void setInputCloud(const typename PointCloud<PointT>::ConstPtr& cloud)
  {
    input_ = cloud;
  }
  void setLeafSize(float lx, float ly, float lz)
  {
    leaf_[0] = lx;
    leaf_[1] = ly;
    leaf_[2] = lz;
    for (int a = 0; a < 3; ++a)
      inv_[a] = 1.0f / leaf_[a];  // Eigen::Array4f::Ones() / leaf_size_.array()
  }
  void filter(PointCloud<PointT>& output)
  {
    output.header = input_->header;
    output.points.clear();
    output.is_dense = true;
    const auto finite = [](const PointT& p) { return std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z); };
    float mn[3] = { std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max() };
    float mx[3] = { -mn[0], -mn[1], -mn[2] };
    std::size_t n_finite = 0;
    for (const PointT& p : input_->points)
    {
      if (!finite(p))
        continue;
      ++n_finite;
      const float c[3] = { p.x, p.y, p.z };
      for (int a = 0; a < 3; ++a)
      {
        mn[a] = std::min(mn[a], c[a]);
        mx[a] = std::max(mx[a], c[a]);
      }
    }
    if (n_finite == 0)
    {
      output.width = output.height = 0;
      return;
    }
    std::int64_t d[3];
    int min_b[3], max_b[3];
    for (int a = 0; a < 3; ++a)
    {
      d[a] = static_cast<std::int64_t>((mx[a] - mn[a]) * inv_[a]) + 1;
      min_b[a] = static_cast<int>(std::floor(mn[a] * inv_[a]));
      max_b[a] = static_cast<int>(std::floor(mx[a] * inv_[a]));
    }
    if (d[0] * d[1] * d[2] > static_cast<std::int64_t>(std::numeric_limits<std::int32_t>::max()))
    {
      output = *input_;  // "Leaf size is too small for the input dataset. Integer indices would overflow."
      return;
    }
    const int mul[3] = { 1, max_b[0] - min_b[0] + 1, (max_b[0] - min_b[0] + 1) * (max_b[1] - min_b[1] + 1) };
    std::vector<std::pair<unsigned, unsigned>> index;
    index.reserve(n_finite);
    for (std::size_t i = 0; i < input_->points.size(); ++i)
    {
      const PointT& p = input_->points[i];
      if (!finite(p))
        continue;
      const int i0 = static_cast<int>(std::floor(p.x * inv_[0]) - static_cast<float>(min_b[0]));
      const int i1 = static_cast<int>(std::floor(p.y * inv_[1]) - static_cast<float>(min_b[1]));
      const int i2 = static_cast<int>(std::floor(p.z * inv_[2]) - static_cast<float>(min_b[2]));
      index.emplace_back(static_cast<unsigned>(i0 * mul[0] + i1 * mul[1] + i2 * mul[2]), static_cast<unsigned>(i));
    }
    std::stable_sort(index.begin(), index.end(),
                     [](const std::pair<unsigned, unsigned>& a, const std::pair<unsigned, unsigned>& b) { return a.first < b.first; });
    std::size_t k = 0;
    while (k < index.size())
    {
      std::size_t e = k;
      float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
      std::map<std::uint32_t, std::size_t> labels;
      while (e < index.size() && index[e].first == index[k].first)
      {
        const PointT& p = input_->points[index[e].second];
        sx += p.x;
        sy += p.y;
        sz += p.z;
        si += p.intensity;
        ++labels[p.label];
        ++e;
      }
      const float cnt = static_cast<float>(e - k);
      PointT c;
      c.x = sx / cnt;
      c.y = sy / cnt;
      c.z = sz / cnt;
      c.intensity = si / cnt;
      std::size_t best = 0;
      for (const auto& l : labels)
        if (l.second > best)
        {
          best = l.second;
          c.label = l.first;
        }
      output.points.push_back(c);
      k = e;
    }
    output.width = static_cast<std::uint32_t>(output.points.size());
    output.height = 1;
  }

private:
  typename PointCloud<PointT>::ConstPtr input_;
  float leaf_[3] = { 0, 0, 0 }, inv_[3] = { 0, 0, 0 };
};
}  // namespace pcl
#endif
