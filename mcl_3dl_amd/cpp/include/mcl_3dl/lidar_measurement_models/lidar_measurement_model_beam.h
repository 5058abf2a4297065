// Drop-in for the reference header of the same path: class mcl_3dl::LidarMeasurementModelBeam with the reference's public
// surface — enum BeamStatus, getBeamStatus, getSinTotalRef, getFilterLabelMax are what the node's debug markers use
// (src/mcl_3dl.cpp:471-478). The raycaster is the DDA one, resident on the GPU; use_raycast_using_dda_ = false is refused.
#ifndef MCL_3DL_HIP_LIDAR_MEASUREMENT_MODEL_BEAM_H
#define MCL_3DL_HIP_LIDAR_MEASUREMENT_MODEL_BEAM_H

#include <cstdint>
#include <memory>
#include <vector>

#include <mcl_3dl/parameters.h>
#include <mcl_3dl/pf.h>
#include <mcl_3dl/raycast.h>
#include <mcl_3dl_hip/batched_model.hpp>

namespace mcl_3dl
{
class LidarMeasurementModelBeam final : public hip::BatchedLidarModel
{
  using Params = LidarMeasurementModelBeamParameters;
  using CastResult = Raycast<PointType>::CastResult;

public:
  enum class BeamStatus  // same enumerators, same order as the reference
  {
    SHORT,
    HIT,
    LONG,
    TOTAL_REFLECTION
  };

  explicit LidarMeasurementModelBeam(const std::shared_ptr<Params>& params);
  float getMaxSearchRange() const override { return search_range_; }
  float getSinTotalRef() const { return sin_total_ref_; }
  uint32_t getFilterLabelMax() const { return params_->filter_label_max_; }
  void refreshParameters() override;
  LidarMeasurementResult measure(ChunkedKdtree<PointType>::Ptr& kdtree, const hip::Cloud::ConstPtr& pc,
                                 const std::vector<Vec3>& origins, const State6DOF& s) const override;
  BeamStatus getBeamStatus(ChunkedKdtree<PointType>::Ptr& kdtree, const Vec3& beam_begin, const Vec3& beam_end,
                           CastResult& result) const;

private:
  void pushParameters() const;
  static void pushBeamParameters(const Params& p);

  std::shared_ptr<Params> params_;
  float search_range_ = 0.f;
  float sin_total_ref_ = 0.f;
};
}  // namespace mcl_3dl

#endif  // MCL_3DL_HIP_LIDAR_MEASUREMENT_MODEL_BEAM_H
