// Stand-in for <pcl/kdtree/kdtree_flann.h> — TEST INFRASTRUCTURE ONLY (oracle build).
//
// PCL / FLANN are third-party code that is NOT under /root/reference and not installed here
// (reference pin: find_package(PCL 1.8.0 ...) CMakeLists.txt:26-31; ROS Noetic => PCL 1.10 /
// FLANN 1.9.1). This file restates the *published semantics* of
//   pcl::KdTreeFLANN<PointT>::radiusSearch(p, radius, k_indices, k_sqr_distances, max_nn)
// on top of flann::L2_Simple<float>:
//   * points and query are first vectorised by the PointRepresentation (x,y,z, each multiplied
//     by its rescale value in float),
//   * d2 = ((0 + dx*dx) + dy*dy) + dz*dz accumulated in float in that order (L2_Simple),
//   * a point is a neighbour iff d2 < (float)(radius*radius)   (strict, KNNRadiusResultSet),
//   * the max_nn nearest neighbours are returned ascending by d2 (max_nn==0 => all).
// Deviation (stated in DESIGN.md, "parity unpinned at the PCL/FLANN boundary"): setEpsilon() is
// accepted and IGNORED — the search is exact (eps = 0). Ties in d2 resolve to the lower index.
// The spatial index is a uniform grid (ours), not FLANN's kd-tree; results do not depend on it.
// For MEASURING what the ignored epsilon is worth, kdtree_flann_epsilon_mode() = 1 routes max_nn = 1 searches with
// eps > 0 through a restated FLANN KDTreeSingleIndex with (1 + eps) pruning (flann_single_index.h): a sensitivity probe
// for tests/test_flann_eps_sensitivity.py, never the oracle's definition.
#ifndef ORACLE_SHIM_PCL_KDTREE_KDTREE_FLANN_H
#define ORACLE_SHIM_PCL_KDTREE_KDTREE_FLANN_H
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <memory>
#include <utility>
#include <vector>
#include <pcl/kdtree/flann_single_index.h>
#include <pcl/kdtree/kdtree.h>
#include <pcl/point_cloud.h>

namespace pcl
{
// 0 (default): exact search whatever setEpsilon() was given; 1: honour it through the restated FLANN single index
inline int& kdtree_flann_epsilon_mode()
{
  static int mode = 0;
  return mode;
}

template <typename PointT>
class KdTreeFLANN
{
public:
  using Ptr = std::shared_ptr<KdTreeFLANN<PointT>>;
  using ConstPtr = std::shared_ptr<const KdTreeFLANN<PointT>>;
  using PointCloudConstPtr = typename PointCloud<PointT>::ConstPtr;
  using PointRepresentationConstPtr = typename PointRepresentation<PointT>::ConstPtr;

  KdTreeFLANN()
    : point_representation_(new DefaultPointRepresentation<PointT>)
  {
  }
  void setEpsilon(float eps) { epsilon_ = eps; }
  float getEpsilon() const { return epsilon_; }
  void setPointRepresentation(const PointRepresentationConstPtr& rep) { point_representation_ = rep; }
  PointCloudConstPtr getInputCloud() const { return input_; }

  void setInputCloud(const PointCloudConstPtr& cloud)
  {
    input_ = cloud;
    const std::size_t n = cloud->points.size();
    std::vector<float> raw(3 * n);
    float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    for (std::size_t i = 0; i < n; ++i)
    {
      float v[3];
      point_representation_->vectorize(cloud->points[i], v);
      for (int a = 0; a < 3; ++a)
      {
        raw[3 * i + a] = v[a];
        if (i == 0 || v[a] < mn[a]) mn[a] = v[a];
        if (i == 0 || v[a] > mx[a]) mx[a] = v[a];
      }
    }
    cell_ = 0.25f;
    for (;;)
    {
      double total = 1;
      for (int a = 0; a < 3; ++a)
      {
        dim_[a] = static_cast<int>(std::floor((mx[a] - mn[a]) / cell_)) + 1;
        total *= dim_[a];
      }
      if (total <= 64.0 * 1024 * 1024)
        break;
      cell_ *= 2;
    }
    for (int a = 0; a < 3; ++a)
      origin_[a] = mn[a];
    const std::size_t ncell = static_cast<std::size_t>(dim_[0]) * dim_[1] * dim_[2];
    std::vector<std::uint32_t> cell_of(n);
    cell_start_.assign(ncell + 1, 0);
    for (std::size_t i = 0; i < n; ++i)
    {
      cell_of[i] = cellIndex(&raw[3 * i]);
      ++cell_start_[cell_of[i] + 1];
    }
    for (std::size_t c = 0; c < ncell; ++c)
      cell_start_[c + 1] += cell_start_[c];
    std::vector<std::uint32_t> fill(cell_start_.begin(), cell_start_.end() - 1);
    pts_.resize(3 * n);
    ids_.resize(n);
    raw_ = raw;
    approx_built_ = false;
    for (std::size_t i = 0; i < n; ++i)  // stable: ascending original index inside a cell
    {
      const std::uint32_t dst = fill[cell_of[i]]++;
      pts_[3 * dst + 0] = raw[3 * i + 0];
      pts_[3 * dst + 1] = raw[3 * i + 1];
      pts_[3 * dst + 2] = raw[3 * i + 2];
      ids_[dst] = static_cast<int>(i);
    }
  }

  int radiusSearch(const PointT& point, double radius, std::vector<int>& k_indices,
                   std::vector<float>& k_sqr_distances, unsigned int max_nn = 0) const
  {
    k_indices.clear();
    k_sqr_distances.clear();
    if (!input_ || ids_.empty())
      return 0;
    float q[3];
    point_representation_->vectorize(point, q);
    const float r2 = static_cast<float>(radius * radius);
    if (max_nn == 1 && epsilon_ > 0.0f && kdtree_flann_epsilon_mode() == 1)
    {
      if (!approx_built_)
      {
        approx_.build(raw_.data(), raw_.size() / 3, 15);  // KDTreeSingleIndexParams(15), kdtree_flann.hpp
        approx_built_ = true;
      }
      int id;
      float d2;
      if (!approx_.nearestWithin(q, r2, epsilon_, &id, &d2))
        return 0;
      k_indices.push_back(id);
      k_sqr_distances.push_back(d2);
      return 1;
    }
    int lo[3], hi[3];
    for (int a = 0; a < 3; ++a)
    {
      // generous cell window (one extra cell each side guards the float rounding of the binning)
      const double l = std::floor((static_cast<double>(q[a]) - radius - origin_[a]) / cell_) - 1;
      const double h = std::floor((static_cast<double>(q[a]) + radius - origin_[a]) / cell_) + 1;
      if (h < 0 || l > dim_[a] - 1)
        return 0;
      lo[a] = static_cast<int>(std::max(l, 0.0));
      hi[a] = static_cast<int>(std::min(h, static_cast<double>(dim_[a] - 1)));
    }
    if (max_nn == 1)
    {
      float best = r2;
      int best_id = -1;
      for (int z = lo[2]; z <= hi[2]; ++z)
        for (int y = lo[1]; y <= hi[1]; ++y)
        {
          const std::size_t row = (static_cast<std::size_t>(z) * dim_[1] + y) * dim_[0];
          const std::uint32_t b = cell_start_[row + lo[0]], e = cell_start_[row + hi[0] + 1];
          for (std::uint32_t i = b; i < e; ++i)
          {
            const float d2 = dist2(q, &pts_[3 * i]);
            if (d2 < best || (d2 == best && best_id >= 0 && ids_[i] < best_id))
            {
              best = d2;
              best_id = ids_[i];
            }
          }
        }
      if (best_id < 0)
        return 0;
      k_indices.push_back(best_id);
      k_sqr_distances.push_back(best);
      return 1;
    }
    std::vector<std::pair<float, int>> found;
    for (int z = lo[2]; z <= hi[2]; ++z)
      for (int y = lo[1]; y <= hi[1]; ++y)
      {
        const std::size_t row = (static_cast<std::size_t>(z) * dim_[1] + y) * dim_[0];
        const std::uint32_t b = cell_start_[row + lo[0]], e = cell_start_[row + hi[0] + 1];
        for (std::uint32_t i = b; i < e; ++i)
        {
          const float d2 = dist2(q, &pts_[3 * i]);
          if (d2 < r2)
            found.emplace_back(d2, ids_[i]);
        }
      }
    std::sort(found.begin(), found.end());
    if (max_nn != 0 && found.size() > max_nn)
      found.resize(max_nn);
    for (const auto& f : found)
    {
      k_indices.push_back(f.second);
      k_sqr_distances.push_back(f.first);
    }
    return static_cast<int>(found.size());
  }

private:
  // flann::L2_Simple<float>: result += diff*diff, dimension by dimension, all float.
  static inline float dist2(const float* a, const float* b)
  {
    float result = 0.0f;
    for (int i = 0; i < 3; ++i)
    {
      const float diff = a[i] - b[i];
      result += diff * diff;
    }
    return result;
  }
  std::uint32_t cellIndex(const float* v) const
  {
    int c[3];
    for (int a = 0; a < 3; ++a)
    {
      c[a] = static_cast<int>(std::floor((v[a] - origin_[a]) / cell_));
      c[a] = std::min(std::max(c[a], 0), dim_[a] - 1);
    }
    return static_cast<std::uint32_t>((static_cast<std::size_t>(c[2]) * dim_[1] + c[1]) * dim_[0] + c[0]);
  }

  PointCloudConstPtr input_;
  PointRepresentationConstPtr point_representation_;
  float epsilon_ = 0.0f;
  float cell_ = 0.25f;
  float origin_[3] = {0, 0, 0};
  int dim_[3] = {1, 1, 1};
  std::vector<std::uint32_t> cell_start_;
  std::vector<float> pts_;
  std::vector<int> ids_;
  std::vector<float> raw_;  // vectorised points in input order (the restated FLANN index is built over them on demand)
  mutable flann_restated::KDTreeSingleIndex approx_;
  mutable bool approx_built_ = false;
};
}  // namespace pcl
#endif
