// TEST INFRASTRUCTURE ONLY (oracle build) — a restatement of FLANN's KDTreeSingleIndex (flann 1.9.1,
// src/cpp/flann/algorithms/kdtree_single_index.h; the same algorithm as its port nanoflann) as pcl::KdTreeFLANN drives it:
// KDTreeSingleIndexParams(15) (pcl/kdtree/impl/kdtree_flann.hpp), flann::L2_Simple<float>, radiusSearch with
// SearchParams(-1, eps) and max_neighbors = 1 -> KNNRadiusResultSet of capacity one.
//
// FLANN is not vendored by the reference and not installed here, so this is written from the published algorithm, NOT
// compiled from FLANN's sources: it exists to MEASURE how far an eps-approximate search (the node sets
// eps = map_grid_min / 16, src/mcl_3dl.cpp:1328) can move a likelihood score away from the exact search that both oracles
// and the GPU engine implement (tests/test_flann_eps_sensitivity.py). It is not an oracle.
//
//   build    divideTree: leaves of <= 15 points; middleSplit_: among the dimensions whose bounding-box span is within
//            1e-5 of the largest one the one with the largest exact spread; cut at the middle of the box clamped to the
//            points' range; planeSplit; index balanced towards count / 2
//   search   searchLevel: nearer child first, the other one only if mindist^2 * (1 + eps) <= worst distance so far
#ifndef ORACLE_SHIM_FLANN_SINGLE_INDEX_H
#define ORACLE_SHIM_FLANN_SINGLE_INDEX_H
#include <cstddef>
#include <utility>
#include <vector>

namespace flann_restated
{
class KDTreeSingleIndex
{
public:
  void build(const float* pts3, std::size_t n, int leaf_max_size = 15)
  {
    pts_ = pts3;
    n_ = n;
    leaf_max_ = leaf_max_size;
    vind_.resize(n);
    for (std::size_t i = 0; i < n; ++i)
      vind_[i] = static_cast<int>(i);
    nodes_.clear();
    nodes_.reserve(2 * n / leaf_max_size + 16);
    if (n == 0)
      return;
    for (int d = 0; d < 3; ++d)
      root_bbox_[d].low = root_bbox_[d].high = pts3[d];
    for (std::size_t k = 1; k < n; ++k)
      for (int d = 0; d < 3; ++d)
      {
        if (pts3[3 * k + d] < root_bbox_[d].low) root_bbox_[d].low = pts3[3 * k + d];
        if (pts3[3 * k + d] > root_bbox_[d].high) root_bbox_[d].high = pts3[3 * k + d];
      }
    Interval bbox[3] = { root_bbox_[0], root_bbox_[1], root_bbox_[2] };
    root_ = divideTree(0, static_cast<int>(n), bbox);
  }

  // nearest point with d2 < r2 (strict) under eps-approximate pruning; returns false if none. L2_Simple float arithmetic.
  bool nearestWithin(const float* q, float r2, float eps, int* index, float* d2) const
  {
    if (n_ == 0)
      return false;
    Result res{ r2, -1 };
    float dists[3] = { 0, 0, 0 };
    float distsq = 0;
    for (int i = 0; i < 3; ++i)  // computeInitialDistances
    {
      if (q[i] < root_bbox_[i].low)
      {
        dists[i] = (q[i] - root_bbox_[i].low) * (q[i] - root_bbox_[i].low);
        distsq += dists[i];
      }
      if (q[i] > root_bbox_[i].high)
      {
        dists[i] = (q[i] - root_bbox_[i].high) * (q[i] - root_bbox_[i].high);
        distsq += dists[i];
      }
    }
    searchLevel(res, q, root_, distsq, dists, 1.0f + eps);
    if (res.index < 0)
      return false;
    *index = res.index;
    *d2 = res.worst;
    return true;
  }

private:
  struct Interval
  {
    float low, high;
  };
  struct Node
  {
    int left, right;  // leaf: [left, right) into vind_
    int divfeat;      // -1 = leaf
    float divlow, divhigh;
    int child1, child2;
  };
  struct Result  // KNNRadiusResultSet with capacity 1: worstDist() = radius until something is found, then its distance
  {
    float worst;
    int index;
  };

  float coord(int idx, int d) const
  {
    return pts_[3 * static_cast<std::size_t>(idx) + d];
  }
  void computeMinMax(const int* ind, int count, int dim, float& min_elem, float& max_elem) const
  {
    min_elem = max_elem = coord(ind[0], dim);
    for (int i = 1; i < count; ++i)
    {
      const float v = coord(ind[i], dim);
      if (v < min_elem) min_elem = v;
      if (v > max_elem) max_elem = v;
    }
  }
  void planeSplit(int* ind, int count, int cutfeat, float cutval, int& lim1, int& lim2) const
  {
    int left = 0, right = count - 1;
    for (;;)
    {
      while (left <= right && coord(ind[left], cutfeat) < cutval) ++left;
      while (right && left <= right && coord(ind[right], cutfeat) >= cutval) --right;
      if (left > right || !right) break;
      std::swap(ind[left], ind[right]);
      ++left;
      --right;
    }
    lim1 = left;
    right = count - 1;
    for (;;)
    {
      while (left <= right && coord(ind[left], cutfeat) <= cutval) ++left;
      while (right && left <= right && coord(ind[right], cutfeat) > cutval) --right;
      if (left > right || !right) break;
      std::swap(ind[left], ind[right]);
      ++left;
      --right;
    }
    lim2 = left;
  }
  void middleSplit(int* ind, int count, int& index, int& cutfeat, float& cutval, const Interval* bbox) const
  {
    const float EPS = 0.00001f;
    float max_span = bbox[0].high - bbox[0].low;
    for (int i = 1; i < 3; ++i)
    {
      const float span = bbox[i].high - bbox[i].low;
      if (span > max_span) max_span = span;
    }
    float max_spread = -1;
    cutfeat = 0;
    for (int i = 0; i < 3; ++i)
    {
      const float span = bbox[i].high - bbox[i].low;
      if (span > (1 - EPS) * max_span)
      {
        float mn, mx;
        computeMinMax(ind, count, i, mn, mx);
        const float spread = mx - mn;
        if (spread > max_spread)
        {
          cutfeat = i;
          max_spread = spread;
        }
      }
    }
    const float split_val = (bbox[cutfeat].low + bbox[cutfeat].high) / 2;
    float mn, mx;
    computeMinMax(ind, count, cutfeat, mn, mx);
    cutval = split_val < mn ? mn : (split_val > mx ? mx : split_val);
    int lim1, lim2;
    planeSplit(ind, count, cutfeat, cutval, lim1, lim2);
    index = lim1 > count / 2 ? lim1 : (lim2 < count / 2 ? lim2 : count / 2);
  }
  int divideTree(int left, int right, Interval* bbox)
  {
    const int id = static_cast<int>(nodes_.size());
    nodes_.push_back(Node());
    if (right - left <= leaf_max_)
    {
      Node& nd = nodes_[id];
      nd.divfeat = -1;
      nd.left = left;
      nd.right = right;
      nd.child1 = nd.child2 = -1;
      for (int d = 0; d < 3; ++d)
        bbox[d].low = bbox[d].high = coord(vind_[left], d);
      for (int k = left + 1; k < right; ++k)
        for (int d = 0; d < 3; ++d)
        {
          const float v = coord(vind_[k], d);
          if (v < bbox[d].low) bbox[d].low = v;
          if (v > bbox[d].high) bbox[d].high = v;
        }
      return id;
    }
    int idx, cutfeat;
    float cutval;
    middleSplit(&vind_[left], right - left, idx, cutfeat, cutval, bbox);
    Interval lb[3] = { bbox[0], bbox[1], bbox[2] }, rb[3] = { bbox[0], bbox[1], bbox[2] };
    lb[cutfeat].high = cutval;
    const int c1 = divideTree(left, left + idx, lb);
    rb[cutfeat].low = cutval;
    const int c2 = divideTree(left + idx, right, rb);
    Node& nd = nodes_[id];
    nd.divfeat = cutfeat;
    nd.child1 = c1;
    nd.child2 = c2;
    nd.divlow = lb[cutfeat].high;
    nd.divhigh = rb[cutfeat].low;
    for (int d = 0; d < 3; ++d)
    {
      bbox[d].low = lb[d].low < rb[d].low ? lb[d].low : rb[d].low;
      bbox[d].high = lb[d].high > rb[d].high ? lb[d].high : rb[d].high;
    }
    return id;
  }
  void searchLevel(Result& res, const float* q, int node, float mindistsq, float* dists, float epsError) const
  {
    const Node& nd = nodes_[node];
    if (nd.divfeat < 0)
    {
      const float worst = res.worst;
      for (int i = nd.left; i < nd.right; ++i)
      {
        const int index = vind_[i];
        float dist = 0.0f;  // flann::L2_Simple: result += diff * diff, float
        for (int d = 0; d < 3; ++d)
        {
          const float diff = q[d] - coord(index, d);
          dist += diff * diff;
        }
        if (dist < worst && dist < res.worst)  // (addPoint: dist >= worst_distance_ is dropped)
        {
          res.worst = dist;
          res.index = index;
        }
      }
      return;
    }
    const int idx = nd.divfeat;
    const float val = q[idx], diff1 = val - nd.divlow, diff2 = val - nd.divhigh;
    int best, other;
    float cut_dist;
    if (diff1 + diff2 < 0)
    {
      best = nd.child1;
      other = nd.child2;
      cut_dist = (val - nd.divhigh) * (val - nd.divhigh);
    }
    else
    {
      best = nd.child2;
      other = nd.child1;
      cut_dist = (val - nd.divlow) * (val - nd.divlow);
    }
    searchLevel(res, q, best, mindistsq, dists, epsError);
    const float dst = dists[idx];
    mindistsq = mindistsq + cut_dist - dst;
    dists[idx] = cut_dist;
    if (mindistsq * epsError <= res.worst)
      searchLevel(res, q, other, mindistsq, dists, epsError);
    dists[idx] = dst;
  }

  const float* pts_ = nullptr;
  std::size_t n_ = 0;
  int leaf_max_ = 15;
  std::vector<int> vind_;
  std::vector<Node> nodes_;
  int root_ = 0;
  Interval root_bbox_[3];
};
}  // namespace flann_restated
#endif
