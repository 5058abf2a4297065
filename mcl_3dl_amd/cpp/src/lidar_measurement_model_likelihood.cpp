// GPU-backed mcl_3dl::LidarMeasurementModelLikelihood (drop-in for src/lidar_measurement_model_likelihood.cpp).
#include <memory>
#include <vector>

#include <mcl_3dl/lidar_measurement_models/lidar_measurement_model_likelihood.h>
#include <mcl_3dl_hip/model_common.hpp>

namespace mcl_3dl
{
LidarMeasurementModelLikelihood::LidarMeasurementModelLikelihood(
    const std::shared_ptr<LidarMeasurementModelLikelihoodParameters>& params)
  : params_(params ? params : std::make_shared<LidarMeasurementModelLikelihoodParameters>())
{
  refreshParameters();
}

// reference: src/lidar_measurement_model_likelihood.cpp:56-61
void LidarMeasurementModelLikelihood::refreshParameters()
{
  num_points_ = params_->num_points_default_;
  clip_near_sq_ = params_->clip_near_ * params_->clip_near_;
  clip_far_sq_ = params_->clip_far_ * params_->clip_far_;
  cache_ = Cache();
}

void LidarMeasurementModelLikelihood::setGlobalLocalizationStatus(const size_t num_particles,
                                                                  const size_t current_num_particles)
{
  num_points_ = hip::pointsPerParticle(params_->num_points_default_, params_->num_points_global_, num_particles,
                                       current_num_particles);
}

// reference: src/lidar_measurement_model_likelihood.cpp:79-103 (clip, then sampler.sample(num_points_))
pcl::PointCloud<LidarMeasurementModelBase::PointType>::Ptr LidarMeasurementModelLikelihood::filter(
    const pcl::PointCloud<PointType>::ConstPtr& pc, const PointCloudRandomSampler<PointType>& sampler) const
{
  const hip::Cloud::Ptr clipped =
      hip::clipCloud(*pc, clip_near_sq_, clip_far_sq_, params_->clip_z_min_, params_->clip_z_max_);
  return sampler.sample(clipped, num_points_);
}

// reference: src/lidar_measurement_model_likelihood.cpp:105-139.  Same inputs, same (likelihood, quality) result; inside
// pf::measure the first call evaluates every particle of the batch in one launch.
LidarMeasurementResult LidarMeasurementModelLikelihood::measure(ChunkedKdtree<PointType>::Ptr& kdtree,
                                                                const pcl::PointCloud<PointType>::ConstPtr& pc,
                                                                const std::vector<Vec3>& /*origins*/,
                                                                const State6DOF& s) const
{
  if (!pc || pc->size() == 0)
    return LidarMeasurementResult(1, 0);

  std::vector<float> poses;
  std::uint64_t epoch = 0;
  const std::size_t index = hip::gatherPoses(s, poses, &epoch);
  const bool cached = epoch != 0 && cache_.epoch == epoch && cache_.cloud == pc.get() &&
                      cache_.likelihood.size() == poses.size() / 7;
  if (!cached)
  {
    hip::Engine& e = hip::Engine::shared();
    hip::syncMap(e, *kdtree);
    e.check(mcl3dl_hip_set_likelihood_params(e.get(), params_->match_dist_min_, params_->match_dist_flat_,
                                             params_->match_weight_));
    std::vector<float> scan;
    hip::packCloud(*pc, scan, nullptr);
    const std::size_t n_p = poses.size() / 7;
    cache_.likelihood.assign(n_p, 0.f);
    cache_.quality.assign(n_p, 0.f);
    e.check(mcl3dl_hip_measure_batch(e.get(), poses.data(), n_p, scan.data(), pc->size(), nullptr, nullptr, 0,
                                     nullptr, 0, cache_.likelihood.data(), cache_.quality.data(), nullptr));
    cache_.epoch = epoch;
    cache_.cloud = pc.get();
  }
  return LidarMeasurementResult(cache_.likelihood[index], cache_.quality[index]);
}
}  // namespace mcl_3dl
