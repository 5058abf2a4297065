// Drop-in replacement for the reference header of the same path
// (include/mcl_3dl/lidar_measurement_models/lidar_measurement_model_beam.h): same class name and public surface
// (enum BeamStatus, getBeamStatus, getSinTotalRef, getFilterLabelMax are used by the node's debug markers,
// src/mcl_3dl.cpp:471-478).  The raycaster is the DDA one, resident on the GPU; use_raycast_using_dda_ = false is refused.
#ifndef MCL_3DL_HIP_LIDAR_MEASUREMENT_MODEL_BEAM_H
#define MCL_3DL_HIP_LIDAR_MEASUREMENT_MODEL_BEAM_H

#include <cstdint>
#include <memory>
#include <vector>

#include <pcl/point_types.h>
#include <pcl_ros/point_cloud.h>

#include <mcl_3dl/chunked_kdtree.h>
#include <mcl_3dl/lidar_measurement_model_base.h>
#include <mcl_3dl/parameters.h>
#include <mcl_3dl/pf.h>
#include <mcl_3dl/point_cloud_random_sampler.h>
#include <mcl_3dl/raycast.h>
#include <mcl_3dl/vec3.h>

namespace mcl_3dl
{
class LidarMeasurementModelBeam : public LidarMeasurementModelBase
{
public:
  enum class BeamStatus
  {
    SHORT,
    HIT,
    LONG,
    TOTAL_REFLECTION
  };
  explicit LidarMeasurementModelBeam(const std::shared_ptr<LidarMeasurementModelBeamParameters>& params);

  inline float getMaxSearchRange() const
  {
    return search_range_;
  }
  inline float getSinTotalRef() const
  {
    return sin_total_ref_;
  }
  inline uint32_t getFilterLabelMax() const
  {
    return params_->filter_label_max_;
  }
  void refreshParameters() final;
  void setGlobalLocalizationStatus(const size_t num_particles, const size_t current_num_particles);
  pcl::PointCloud<PointType>::Ptr filter(
      const pcl::PointCloud<PointType>::ConstPtr& pc,
      const PointCloudRandomSampler<PointType>& sampler) const;
  LidarMeasurementResult measure(
      ChunkedKdtree<PointType>::Ptr& kdtree,
      const pcl::PointCloud<PointType>::ConstPtr& pc,
      const std::vector<Vec3>& origins,
      const State6DOF& s) const;
  BeamStatus getBeamStatus(
      ChunkedKdtree<PointType>::Ptr& kdtree,
      const Vec3& beam_begin, const Vec3& beam_end,
      typename mcl_3dl::Raycast<PointType>::CastResult& result) const;

private:
  void pushParameters() const;

  std::shared_ptr<LidarMeasurementModelBeamParameters> params_;
  size_t num_points_;
  float clip_far_sq_;
  float clip_near_sq_;
  float search_range_;
  float sin_total_ref_;

  struct Cache
  {
    std::uint64_t epoch = 0;
    const void* cloud = nullptr;
    std::vector<float> likelihood;
  };
  mutable Cache cache_;
};
}  // namespace mcl_3dl

#endif  // MCL_3DL_HIP_LIDAR_MEASUREMENT_MODEL_BEAM_H
