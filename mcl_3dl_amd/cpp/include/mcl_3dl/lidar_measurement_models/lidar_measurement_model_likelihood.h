// Drop-in for the reference header of the same path: class mcl_3dl::LidarMeasurementModelLikelihood with the reference's
// public surface, answered by the gfx950 engine through the C ABI (mcl3dl_hip_measure_batch).
#ifndef MCL_3DL_HIP_LIDAR_MEASUREMENT_MODEL_LIKELIHOOD_H
#define MCL_3DL_HIP_LIDAR_MEASUREMENT_MODEL_LIKELIHOOD_H

#include <memory>
#include <vector>

#include <mcl_3dl/parameters.h>
#include <mcl_3dl/pf.h>
#include <mcl_3dl_hip/batched_model.hpp>

namespace mcl_3dl
{
class LidarMeasurementModelLikelihood final : public hip::BatchedLidarModel
{
  using Params = LidarMeasurementModelLikelihoodParameters;

public:
  explicit LidarMeasurementModelLikelihood(const std::shared_ptr<Params>& params);
  float getMaxSearchRange() const override { return params_->match_dist_min_; }
  void refreshParameters() override;
  LidarMeasurementResult measure(ChunkedKdtree<PointType>::Ptr& kdtree, const hip::Cloud::ConstPtr& pc,
                                 const std::vector<Vec3>& origins, const State6DOF& s) const override;

private:
  std::shared_ptr<Params> params_;
};
}  // namespace mcl_3dl

#endif  // MCL_3DL_HIP_LIDAR_MEASUREMENT_MODEL_LIKELIHOOD_H
