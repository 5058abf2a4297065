// mcl_3dl_hip/batched_model.hpp — what the two GPU-backed LiDAR models share: the clip-and-sample filter(), the
// points-per-particle policy during global localisation, and the per-update result cache that turns N per-particle
// measure() calls into one batched launch (protocol: mcl_3dl_hip/engine.hpp, BatchDescriptor).
#ifndef MCL_3DL_HIP_BATCHED_MODEL_HPP
#define MCL_3DL_HIP_BATCHED_MODEL_HPP

#include <cstddef>
#include <cstdint>
#include <vector>

#include <mcl_3dl/lidar_measurement_model_base.h>
#include <mcl_3dl/point_cloud_random_sampler.h>
#include <mcl_3dl_hip/model_common.hpp>

namespace mcl_3dl
{
namespace hip
{
class BatchedLidarModel : public LidarMeasurementModelBase
{
public:
  // reference: both models' setGlobalLocalizationStatus (src/lidar_measurement_model_likelihood.cpp:63-77,
  // src/lidar_measurement_model_beam.cpp:82-96)
  void setGlobalLocalizationStatus(const size_t num_particles, const size_t current_num_particles) override
  {
    points_now_ = pointsPerParticle(points_default_, points_global_, num_particles, current_num_particles);
  }

  // reference: both models' filter() (likelihood.cpp:79-103, beam.cpp:98-122): clip, then sampler.sample(num_points_)
  Cloud::Ptr filter(const Cloud::ConstPtr& pc, const PointCloudRandomSampler<PointType>& sampler) const override
  {
    const Cloud::Ptr clipped = clipCloud(*pc, clip_.near_sq, clip_.far_sq, clip_.z_min, clip_.z_max);
    return sampler.sample(clipped, points_now_);
  }

protected:
  struct Clip
  {
    float near_sq = 0.f, far_sq = 0.f, z_min = 0.f, z_max = 0.f;
  };
  void configureFilter(const std::size_t points_default, const std::size_t points_global, const float clip_near,
                       const float clip_far, const float clip_z_min, const float clip_z_max)
  {
    points_default_ = points_now_ = points_default;
    points_global_ = points_global;
    clip_.near_sq = clip_near * clip_near;
    clip_.far_sq = clip_far * clip_far;
    clip_.z_min = clip_z_min;
    clip_.z_max = clip_z_max;
    results_ = Results();
  }

  // Results of the last batched launch, valid for one (pf::measure epoch, scan cloud) pair.
  struct Results
  {
    std::uint64_t epoch = 0;
    const void* cloud = nullptr;
    std::vector<float> likelihood, quality;
  };
  // Where `s` sits in the published batch and whether this model's cached results answer it. The cache is tested FIRST:
  // inside pf::measure this runs once per particle, so nothing here may cost more than a few comparisons (packing the
  // poses of the whole batch happens in refreshPoses(), once per epoch).
  struct Slot
  {
    std::size_t index = 0, count = 1;
    std::uint64_t epoch = 0;
    bool refresh = true;
  };
  Slot lookup(const State6DOF& s, const void* cloud) const
  {
    Slot slot;
    const BatchDescriptor& b = currentBatch();
    const char* first = static_cast<const char*>(b.first_state);
    const char* self = reinterpret_cast<const char*>(&s);
    if (b.count > 0 && self >= first && self < first + b.count * b.stride && (self - first) % b.stride == 0)
    {
      slot.index = static_cast<std::size_t>(self - first) / b.stride;
      slot.count = b.count;
      slot.epoch = b.epoch;
    }
    slot.refresh = !(slot.epoch != 0 && results_.epoch == slot.epoch && results_.cloud == cloud &&
                     results_.likelihood.size() == slot.count);
    if (slot.refresh)
    {
      results_.epoch = slot.epoch;
      results_.cloud = cloud;
      results_.likelihood.assign(slot.count, 0.f);
      results_.quality.assign(slot.count, 0.f);
    }
    return slot;
  }

  // Makes sure the engine holds the poses of this epoch: packed and uploaded by whichever model asks first (the node
  // calls "beam" before "likelihood", src/mcl_3dl.cpp:409), reused by the other. Outside pf::measure (epoch 0) the single
  // state is sent every time.
  static void refreshPoses(Engine& e, const State6DOF& s, const Slot& slot)
  {
    if (slot.epoch != 0 && e.pose_epoch == slot.epoch && e.pose_count == slot.count)
      return;
    std::vector<float>& poses = e.pose_scratch;
    std::uint64_t epoch = 0;
    gatherPoses(s, poses, &epoch);
    e.check(mcl3dl_hip_group_upload_poses(e.group(), poses.data(), poses.size() / 7));
    e.pose_epoch = epoch;
    e.pose_count = poses.size() / 7;
  }

  std::size_t points_default_ = 0, points_global_ = 0, points_now_ = 0;
  Clip clip_;
  mutable Results results_;
};
}  // namespace hip
}  // namespace mcl_3dl

#endif  // MCL_3DL_HIP_BATCHED_MODEL_HPP
