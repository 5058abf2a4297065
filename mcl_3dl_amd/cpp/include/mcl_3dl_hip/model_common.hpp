// mcl_3dl_hip/model_common.hpp — helpers shared by the two drop-in model translation units.
#ifndef MCL_3DL_HIP_MODEL_COMMON_HPP
#define MCL_3DL_HIP_MODEL_COMMON_HPP

#include <cstdint>
#include <cstring>
#include <vector>

#include <mcl_3dl/chunked_kdtree.h>
#include <mcl_3dl/lidar_measurement_model_base.h>
#include <mcl_3dl/state_6dof.h>
#include <mcl_3dl_hip/engine.hpp>

namespace mcl_3dl
{
namespace hip
{
using PointType = LidarMeasurementModelBase::PointType;
using Cloud = pcl::PointCloud<PointType>;

// ChunkedKdtree keeps its PointRepresentation protected (chunked_kdtree.h:280). A pointer-to-member formed in a derived
// class reaches it without touching the reference header.
struct KdtreePeek : public ChunkedKdtree<PointType>
{
  static pcl::PointRepresentation<PointType>::ConstPtr representation(const ChunkedKdtree<PointType>& k)
  {
    return k.*(&KdtreePeek::point_rep_);
  }
};

// Uploads the map when the kd-tree's input cloud (or its header stamp, the reference's own cache key,
// raycast_using_dda.h:164-171) changes. dist_weight is recovered from the kd-tree's PointRepresentation by vectorising
// (1,1,1): out[i] = alpha[i] (src/mcl_3dl.cpp:1270 setRescaleValues).
inline void syncMap(Engine& e, const ChunkedKdtree<PointType>& kdtree)
{
  const Cloud::ConstPtr& map = kdtree.getInputCloud();
  if (!map || map->points.empty())
    throw std::runtime_error("mcl3dl_hip: the kd-tree has no input cloud");
  float w[3] = { 1.f, 1.f, 1.f };
  bool has_w = false;
  const auto rep = KdtreePeek::representation(kdtree);
  if (rep)
  {
    PointType one;
    one.x = one.y = one.z = 1.0f;
    float out[8] = { 1.f, 1.f, 1.f, 0, 0, 0, 0, 0 };
    rep->vectorize(one, out);
    has_w = true;
    for (int i = 0; i < 3; ++i)
      w[i] = out[i];
  }
  const bool same = e.map_cloud == map.get() && e.map_stamp == map->header.stamp && e.map_size == map->points.size() &&
                    e.has_weight == has_w && std::memcmp(e.dist_weight, w, sizeof(w)) == 0;
  if (same)
    return;
  std::vector<float> xyz(3 * map->points.size());
  std::vector<std::uint32_t> label(map->points.size());
  float mn[3] = { map->points[0].x, map->points[0].y, map->points[0].z };
  for (std::size_t i = 0; i < map->points.size(); ++i)
  {
    xyz[3 * i + 0] = map->points[i].x;
    xyz[3 * i + 1] = map->points[i].y;
    xyz[3 * i + 2] = map->points[i].z;
    label[i] = map->points[i].label;
    for (int a = 0; a < 3; ++a)
      mn[a] = xyz[3 * i + a] < mn[a] ? xyz[3 * i + a] : mn[a];  // pcl::getMinMax3D (raycast_using_dda.h:175)
  }
  std::memcpy(e.map_min, mn, sizeof(mn));
  e.check(mcl3dl_hip_group_set_map(e.group(), xyz.data(), label.data(), label.size(), map->header.stamp,
                                   has_w ? w : nullptr));
  e.map_cloud = map.get();
  e.map_stamp = map->header.stamp;
  e.map_size = map->points.size();
  e.has_weight = has_w;
  std::memcpy(e.dist_weight, w, sizeof(w));
}

inline void packPose(const State6DOF& s, float* out7)
{
  out7[0] = s.pos_.x_;
  out7[1] = s.pos_.y_;
  out7[2] = s.pos_.z_;
  out7[3] = s.rot_.x_;
  out7[4] = s.rot_.y_;
  out7[5] = s.rot_.z_;
  out7[6] = s.rot_.w_;
}

// Poses of the whole published batch, or of `s` alone outside pf::measure (tests, debug markers). Returns the index of
// `s` inside the batch.
inline std::size_t gatherPoses(const State6DOF& s, std::vector<float>& poses, std::uint64_t* epoch)
{
  const BatchDescriptor& b = currentBatch();
  const char* first = static_cast<const char*>(b.first_state);
  const char* self = reinterpret_cast<const char*>(&s);
  if (b.count > 0 && self >= first && self < first + b.count * b.stride && (self - first) % b.stride == 0)
  {
    poses.resize(7 * b.count);
    for (std::size_t i = 0; i < b.count; ++i)
      packPose(*reinterpret_cast<const State6DOF*>(first + i * b.stride), &poses[7 * i]);
    *epoch = b.epoch;
    return static_cast<std::size_t>(self - first) / b.stride;
  }
  poses.resize(7);
  packPose(s, poses.data());
  *epoch = 0;
  return 0;
}

inline void packCloud(const Cloud& pc, std::vector<float>& xyz, std::vector<std::uint32_t>* label)
{
  xyz.resize(3 * pc.points.size());
  if (label)
    label->resize(pc.points.size());
  for (std::size_t i = 0; i < pc.points.size(); ++i)
  {
    xyz[3 * i + 0] = pc.points[i].x;
    xyz[3 * i + 1] = pc.points[i].y;
    xyz[3 * i + 2] = pc.points[i].z;
    if (label)
      (*label)[i] = pc.points[i].label;
  }
}

// The clip predicate of both models' filter() (src/lidar_measurement_model_likelihood.cpp:84-93,
// src/lidar_measurement_model_beam.cpp:103-112): keep points with clip_near^2 <= x^2+y^2 <= clip_far^2 and
// clip_z_min <= z <= clip_z_max.
inline Cloud::Ptr clipCloud(const Cloud& pc, const float clip_near_sq, const float clip_far_sq, const float z_min,
                            const float z_max)
{
  Cloud::Ptr out(new Cloud);
  out->header = pc.header;
  out->points.reserve(pc.points.size());
  for (const PointType& p : pc.points)
  {
    const float range_sq = p.x * p.x + p.y * p.y;
    const bool rejected = range_sq > clip_far_sq || range_sq < clip_near_sq || p.z < z_min || z_max < p.z;
    if (!rejected)
      out->points.push_back(p);
  }
  out->width = 1;
  out->height = out->points.size();
  return out;
}

// setGlobalLocalizationStatus of both models (likelihood.cpp:63-77, beam.cpp:82-96)
inline std::size_t pointsPerParticle(const std::size_t num_default, const std::size_t num_global,
                                     const std::size_t num_particles, const std::size_t current_num_particles)
{
  if (current_num_particles <= num_particles)
    return num_default;
  const std::size_t scaled = num_default * num_particles / current_num_particles;
  return scaled < num_global ? num_global : scaled;
}
}  // namespace hip
}  // namespace mcl_3dl

#endif  // MCL_3DL_HIP_MODEL_COMMON_HPP
