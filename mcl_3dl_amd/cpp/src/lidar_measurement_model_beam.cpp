// GPU-backed mcl_3dl::LidarMeasurementModelBeam (drop-in for src/lidar_measurement_model_beam.cpp, DDA raycaster).
#include <algorithm>
#include <cmath>
#include <stdexcept>

#include <mcl_3dl/lidar_measurement_models/lidar_measurement_model_beam.h>

namespace mcl_3dl
{
LidarMeasurementModelBeam::LidarMeasurementModelBeam(const std::shared_ptr<Params>& params)
  : params_(params ? params : std::make_shared<Params>())
{
  refreshParameters();
}

// reference: src/lidar_measurement_model_beam.cpp:58-80. The derived constants that decide results (hit_range^2,
// beam_likelihood_, the DDA caster) are re-derived inside the engine by mcl3dl_hip_set_beam_params with the reference's
// expressions; only what the public getters expose is kept here.
void LidarMeasurementModelBeam::refreshParameters()
{
  if (!params_->use_raycast_using_dda_)
    throw std::runtime_error("mcl3dl_hip: the GPU beam model implements RaycastUsingDDA only; set "
                             "beam/use_raycast_using_dda to true");
  search_range_ = std::max({ params_->map_grid_x_, params_->map_grid_y_, params_->map_grid_z_ }) * 4;
  sin_total_ref_ = sinf(params_->ang_total_ref_);
  const std::shared_ptr<Params> params = params_;
  configureFilter(hip::Engine::BEAM, [params] { pushBeamParameters(*params); }, params_->num_points_default_,
                  params_->num_points_global_, params_->clip_near_, params_->clip_far_, params_->clip_z_min_,
                  params_->clip_z_max_);
}

void LidarMeasurementModelBeam::pushBeamParameters(const Params& p)
{
  hip::Engine& e = hip::Engine::shared();
  e.check(mcl3dl_hip_group_set_beam_params(e.group(), p.map_grid_x_, p.map_grid_y_, p.map_grid_z_, p.dda_grid_size_,
                                           p.ray_angle_half_, p.hit_range_, p.beam_likelihood_min_,
                                           static_cast<std::uint32_t>(p.num_points_default_), p.ang_total_ref_,
                                           p.filter_label_max_, p.add_penalty_short_only_mode_ ? 1 : 0));
}

void LidarMeasurementModelBeam::pushParameters() const
{
  pushBeamParameters(*params_);
}

// reference: src/lidar_measurement_model_beam.cpp:124-155
LidarMeasurementResult LidarMeasurementModelBeam::measure(ChunkedKdtree<PointType>::Ptr& kdtree,
                                                          const hip::Cloud::ConstPtr& pc,
                                                          const std::vector<Vec3>& origins, const State6DOF& s) const
{
  if (!pc || pc->size() == 0)
    return LidarMeasurementResult(1, 0);  // :130-133

  const Slot slot = lookup(s, pc.get());
  if (slot.refresh)
    evaluate(*kdtree, *pc, origins, s, slot);
  awaitResult(slot.index);
  return LidarMeasurementResult(results().likelihood[slot.index], 1.0);
}

// reference: src/lidar_measurement_model_beam.cpp:157-192. result.point_ points into the kd-tree's input cloud like the
// reference's; result.pos_ is the centre of the voxel the ray collided in (raycast_using_dda.h:150-156 returns
// fromIndex(current_index_), :219-223 — the node draws the collision marker there, src/mcl_3dl.cpp:489-491). The collided
// voxel is the collided point's voxel: toIndex (:205-217: float difference, double division, truncation), then the
// centre in double, rounded to float by Vec3's constructor.
LidarMeasurementModelBeam::BeamStatus LidarMeasurementModelBeam::getBeamStatus(ChunkedKdtree<PointType>::Ptr& kdtree,
                                                                               const Vec3& lidar_pos,
                                                                               const Vec3& scan_pos,
                                                                               CastResult& result) const
{
  hip::Engine& e = hip::Engine::shared();
  hip::syncMap(e, *kdtree);
  pushParameters();
  const float b[3] = { lidar_pos.x_, lidar_pos.y_, lidar_pos.z_ };
  const float en[3] = { scan_pos.x_, scan_pos.y_, scan_pos.z_ };
  std::int32_t status = 2, hit = -1;
  e.checkContext(mcl3dl_hip_beam_status(e.get(), b, en, 1, &status, &hit));
  if (hit >= 0)
  {
    const PointType* p = &kdtree->getInputCloud()->points[hit];
    const double grid = params_->dda_grid_size_;
    const float c[3] = { p->x, p->y, p->z };
    float centre[3];
    for (int a = 0; a < 3; ++a)
    {
      const int index = static_cast<int>((c[a] - e.map_min[a]) / grid);
      centre[a] = static_cast<float>((index + 0.5) * grid + e.map_min[a]);
    }
    result = CastResult(Vec3(centre[0], centre[1], centre[2]), true, 1.0, p);
  }
  return static_cast<BeamStatus>(status);
}
}  // namespace mcl_3dl
