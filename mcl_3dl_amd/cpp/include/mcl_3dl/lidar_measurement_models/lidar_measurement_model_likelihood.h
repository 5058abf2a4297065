// Drop-in replacement for the reference header of the same path
// (include/mcl_3dl/lidar_measurement_models/lidar_measurement_model_likelihood.h): same class name, same public surface,
// same LidarMeasurementModelBase parent — measure() is answered by the gfx950 engine through the C ABI.
#ifndef MCL_3DL_HIP_LIDAR_MEASUREMENT_MODEL_LIKELIHOOD_H
#define MCL_3DL_HIP_LIDAR_MEASUREMENT_MODEL_LIKELIHOOD_H

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
#include <mcl_3dl/vec3.h>

namespace mcl_3dl
{
class LidarMeasurementModelLikelihood : public LidarMeasurementModelBase
{
public:
  explicit LidarMeasurementModelLikelihood(
      const std::shared_ptr<LidarMeasurementModelLikelihoodParameters>& params);

  inline float getMaxSearchRange() const
  {
    return params_->match_dist_min_;
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

private:
  std::shared_ptr<LidarMeasurementModelLikelihoodParameters> params_;
  size_t num_points_;
  float clip_far_sq_;
  float clip_near_sq_;

  // results of the last batched launch, valid for one (pf::measure epoch, scan cloud) pair
  struct Cache
  {
    std::uint64_t epoch = 0;
    const void* cloud = nullptr;
    std::vector<float> likelihood;
    std::vector<float> quality;
  };
  mutable Cache cache_;
};
}  // namespace mcl_3dl

#endif  // MCL_3DL_HIP_LIDAR_MEASUREMENT_MODEL_LIKELIHOOD_H
