// GPU-backed mcl_3dl::LidarMeasurementModelLikelihood (drop-in for src/lidar_measurement_model_likelihood.cpp).
#include <mcl_3dl/lidar_measurement_models/lidar_measurement_model_likelihood.h>

namespace mcl_3dl
{
LidarMeasurementModelLikelihood::LidarMeasurementModelLikelihood(const std::shared_ptr<Params>& params)
  : params_(params ? params : std::make_shared<Params>())
{
  refreshParameters();
}

// reference: src/lidar_measurement_model_likelihood.cpp:56-61 — the search radius / flat distance / weight themselves
// are pushed to the engine at every batched launch, so a parameter object mutated by dynamic_reconfigure is always seen.
void LidarMeasurementModelLikelihood::refreshParameters()
{
  const std::shared_ptr<Params> params = params_;
  configureFilter(hip::Engine::LIKELIHOOD,
                  [params]
                  {
                    hip::Engine& e = hip::Engine::shared();
                    e.check(mcl3dl_hip_group_set_likelihood_params(e.group(), params->match_dist_min_,
                                                                   params->match_dist_flat_, params->match_weight_));
                  },
                  params_->num_points_default_, params_->num_points_global_, params_->clip_near_, params_->clip_far_,
                  params_->clip_z_min_, params_->clip_z_max_);
}

// reference: src/lidar_measurement_model_likelihood.cpp:105-139. Same inputs, same (likelihood, quality); inside
// pf::measure the first call evaluates every particle of the batch in one launch.
LidarMeasurementResult LidarMeasurementModelLikelihood::measure(ChunkedKdtree<PointType>::Ptr& kdtree,
                                                                const hip::Cloud::ConstPtr& pc,
                                                                const std::vector<Vec3>& origins,
                                                                const State6DOF& s) const
{
  if (!pc || pc->size() == 0)
    return LidarMeasurementResult(1, 0);  // :111-114

  const Slot slot = lookup(s, pc.get());
  if (slot.refresh)
    evaluate(*kdtree, *pc, origins, s, slot);
  awaitResult(slot.index);
  const hip::Engine::Results& r = results();
  return LidarMeasurementResult(r.likelihood[slot.index], r.quality[slot.index]);
}
}  // namespace mcl_3dl
