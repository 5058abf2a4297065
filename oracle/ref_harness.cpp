// oracle/ref_harness.cpp — TEST INFRASTRUCTURE ONLY. Never linked into the product.
//
// Builds the REAL reference (at-wat/mcl_3dl v0.7.0) hot path into oracle/_ref/libmcl3dl_ref.so:
//   /root/reference/src/lidar_measurement_model_likelihood.cpp   (compiled unmodified)
//   /root/reference/src/lidar_measurement_model_beam.cpp         (compiled unmodified)
//   /root/reference/include/mcl_3dl/{pf,vec3,quat,state_6dof,chunked_kdtree,raycast,raycasts/*}.h
// against the stand-in PCL/Eigen/ROS headers in oracle/shims/ (PCL/FLANN/Eigen/ROS are not
// installed here; see oracle/shims/pcl/kdtree/kdtree_flann.h for the one piece of third-party
// arithmetic that is restated rather than compiled).
//
// This file only adds a plain-C ABI on top so that tests (ctypes) can drive the reference classes
// with flat arrays. It contains no algorithm of its own except the 20-line restatement of the node's
// measure lambda (src/mcl_3dl.cpp:398-426), which cannot be compiled because it lives inside the ROS node.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <mcl_3dl/chunked_kdtree.h>
#include <mcl_3dl/lidar_measurement_model_base.h>
#include <mcl_3dl/lidar_measurement_models/lidar_measurement_model_beam.h>
#include <mcl_3dl/lidar_measurement_models/lidar_measurement_model_likelihood.h>
#include <mcl_3dl/pf.h>
#include <mcl_3dl/point_cloud_random_sampler.h>
#include <mcl_3dl/point_types.h>
#include <mcl_3dl/quat.h>
#include <mcl_3dl/raycasts/raycast_using_dda.h>
#include <mcl_3dl/state_6dof.h>
#include <mcl_3dl/vec3.h>

#include <pcl/filters/voxel_grid.h>  // oracle/shims: restated (PCL is not vendored by the reference)

// The reference's sampler seeds its private engine from std::random_device; the tests need the very same class with a
// known seed, so the header is read with its private section open (every header it includes is already included above).
#include <random>
#define private public
#include <mcl_3dl/point_cloud_random_samplers/point_cloud_uniform_sampler.h>
#undef private

#ifdef _OPENMP
#include <omp.h>
#endif

namespace
{
using PointType = mcl_3dl::LidarMeasurementModelBase::PointType;
using Cloud = pcl::PointCloud<PointType>;
using Kdtree = mcl_3dl::ChunkedKdtree<PointType>;

// The node's point representation (src/mcl_3dl.cpp:109-126): xyz only, rescaled by dist_weight (:1270).
class XyzRepresentation : public pcl::PointRepresentation<PointType>
{
public:
  XyzRepresentation()
  {
    nr_dimensions_ = 3;
    trivial_ = true;
  }
  void copyToFloatArray(const PointType& p, float* out) const override
  {
    out[0] = p.x;
    out[1] = p.y;
    out[2] = p.z;
  }
};

struct Ref
{
  Kdtree::Ptr kdtree;
  std::shared_ptr<XyzRepresentation> rep;
  Cloud::Ptr map;
  std::shared_ptr<mcl_3dl::LidarMeasurementModelLikelihoodParameters> lik_params;
  std::shared_ptr<mcl_3dl::LidarMeasurementModelBeamParameters> beam_params;
  std::shared_ptr<mcl_3dl::LidarMeasurementModelLikelihood> lik;
  std::vector<std::shared_ptr<mcl_3dl::LidarMeasurementModelBeam>> beam;  // one per thread (raycaster_ is mutable state)
};

Cloud::Ptr makeCloud(const float* xyz, const uint32_t* label, size_t n)
{
  Cloud::Ptr pc(new Cloud);
  pc->points.resize(n);
  for (size_t i = 0; i < n; ++i)
  {
    PointType p;
    p.x = xyz[3 * i + 0];
    p.y = xyz[3 * i + 1];
    p.z = xyz[3 * i + 2];
    p.label = label ? label[i] : 0;
    pc->points[i] = p;
  }
  pc->width = n;
  pc->height = 1;
  return pc;
}

mcl_3dl::State6DOF makeState(const float* pose7)
{
  return mcl_3dl::State6DOF(mcl_3dl::Vec3(pose7[0], pose7[1], pose7[2]),
                            mcl_3dl::Quat(pose7[3], pose7[4], pose7[5], pose7[6]));
}

int maxThreads()
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void ensureBeamModels(Ref* r, int n)
{
  while (static_cast<int>(r->beam.size()) < n)
    r->beam.emplace_back(new mcl_3dl::LidarMeasurementModelBeam(r->beam_params));
}
}  // namespace

extern "C"
{
void* ref_create(float chunk_length, float max_search_radius)
{
  Ref* r = new Ref;
  r->kdtree.reset(new Kdtree(chunk_length, max_search_radius));
  r->rep.reset(new XyzRepresentation);
  r->lik_params.reset(new mcl_3dl::LidarMeasurementModelLikelihoodParameters);
  r->beam_params.reset(new mcl_3dl::LidarMeasurementModelBeamParameters);
  r->lik.reset(new mcl_3dl::LidarMeasurementModelLikelihood(r->lik_params));
  r->beam.emplace_back(new mcl_3dl::LidarMeasurementModelBeam(r->beam_params));
  return r;
}

void ref_destroy(void* h)
{
  delete static_cast<Ref*>(h);
}

int ref_max_threads()
{
  return maxThreads();
}

// 1: the kd-tree shim honours setEpsilon() through the restated FLANN KDTreeSingleIndex (a sensitivity probe, see
// shims/pcl/kdtree/kdtree_flann.h); 0 (default): exact search — the definition every parity test is measured against
void ref_set_flann_epsilon_mode(int mode)
{
  pcl::kdtree_flann_epsilon_mode() = mode;
}

// src/mcl_3dl.cpp:1270,1327-1329 (setRescaleValues / setEpsilon / setPointRepresentation) + :1369 setInputCloud.
void ref_set_map(void* h, const float* xyz, const uint32_t* label, size_t n, uint64_t stamp,
                 const float* dist_weight, float epsilon)
{
  Ref* r = static_cast<Ref*>(h);
  r->map = makeCloud(xyz, label, n);
  r->map->header.stamp = stamp;
  if (dist_weight)
  {
    r->rep->setRescaleValues(dist_weight);
    r->kdtree->setPointRepresentation(r->rep);
  }
  if (epsilon >= 0)
    r->kdtree->setEpsilon(epsilon);
  r->kdtree->setInputCloud(r->map);
}

void ref_set_likelihood_params(void* h, float match_dist_min, float match_dist_flat, float match_weight,
                               uint32_t num_points, uint32_t num_points_global, float clip_near, float clip_far,
                               float clip_z_min, float clip_z_max)
{
  Ref* r = static_cast<Ref*>(h);
  auto& p = *r->lik_params;
  p.match_dist_min_ = match_dist_min;
  p.match_dist_flat_ = match_dist_flat;
  p.match_weight_ = match_weight;
  p.num_points_default_ = num_points;
  p.num_points_global_ = num_points_global;
  p.clip_near_ = clip_near;
  p.clip_far_ = clip_far;
  p.clip_z_min_ = clip_z_min;
  p.clip_z_max_ = clip_z_max;
  r->lik->refreshParameters();
}

void ref_set_beam_params(void* h, float map_grid_x, float map_grid_y, float map_grid_z, float dda_grid_size,
                         float ray_angle_half, float hit_range, float beam_likelihood_min, uint32_t num_points,
                         uint32_t num_points_global, float ang_total_ref, uint32_t filter_label_max,
                         int add_penalty_short_only_mode, int use_raycast_using_dda, float clip_near,
                         float clip_far, float clip_z_min, float clip_z_max)
{
  Ref* r = static_cast<Ref*>(h);
  auto& p = *r->beam_params;
  p.map_grid_x_ = map_grid_x;
  p.map_grid_y_ = map_grid_y;
  p.map_grid_z_ = map_grid_z;
  p.dda_grid_size_ = dda_grid_size;
  p.ray_angle_half_ = ray_angle_half;
  p.hit_range_ = hit_range;
  p.beam_likelihood_min_ = beam_likelihood_min;
  p.num_points_default_ = num_points;
  p.num_points_global_ = num_points_global;
  p.ang_total_ref_ = ang_total_ref;
  p.filter_label_max_ = filter_label_max;
  p.add_penalty_short_only_mode_ = add_penalty_short_only_mode != 0;
  p.use_raycast_using_dda_ = use_raycast_using_dda != 0;
  p.clip_near_ = clip_near;
  p.clip_far_ = clip_far;
  p.clip_z_min_ = clip_z_min;
  p.clip_z_max_ = clip_z_max;
  for (auto& b : r->beam)
    b->refreshParameters();  // re-creates the raycaster (lidar_measurement_model_beam.cpp:58-80)
}

void ref_set_global_localization_status(void* h, size_t num_particles, size_t current_num_particles,
                                        uint64_t* lik_points_dummy)
{
  Ref* r = static_cast<Ref*>(h);
  r->lik->setGlobalLocalizationStatus(num_particles, current_num_particles);
  for (auto& b : r->beam)
    b->setGlobalLocalizationStatus(num_particles, current_num_particles);
  (void)lik_points_dummy;
}

float ref_beam_max_search_range(void* h)
{
  return static_cast<Ref*>(h)->beam[0]->getMaxSearchRange();
}

// ChunkedKdtree::radiusSearch (include/mcl_3dl/chunked_kdtree.h:217-237), max 1 neighbour.
// returns found (0/1) per query.
void ref_radius_search(void* h, const float* q_xyz, size_t n, float radius, int* found, int* id, float* sqdist)
{
  Ref* r = static_cast<Ref*>(h);
  std::vector<int> ids(1);
  std::vector<float> sq(1);
  for (size_t i = 0; i < n; ++i)
  {
    PointType p;
    p.x = q_xyz[3 * i + 0];
    p.y = q_xyz[3 * i + 1];
    p.z = q_xyz[3 * i + 2];
    ids.assign(1, -1);
    sq.assign(1, 0.f);
    const int ret = r->kdtree->radiusSearch(p, radius, ids, sq, 1);
    found[i] = ret ? 1 : 0;
    id[i] = ret ? ids[0] : -1;
    sqdist[i] = ret ? sq[0] : -1.f;
  }
}

// State6DOF::transform (include/mcl_3dl/state_6dof.h:214-225) applied to a flat point list.
void ref_transform(const float* pose7, const float* xyz_in, size_t n, float* xyz_out)
{
  Cloud pc;
  pc.points.resize(n);
  for (size_t i = 0; i < n; ++i)
  {
    pc.points[i].x = xyz_in[3 * i + 0];
    pc.points[i].y = xyz_in[3 * i + 1];
    pc.points[i].z = xyz_in[3 * i + 2];
  }
  makeState(pose7).transform(pc);
  for (size_t i = 0; i < n; ++i)
  {
    xyz_out[3 * i + 0] = pc.points[i].x;
    xyz_out[3 * i + 1] = pc.points[i].y;
    xyz_out[3 * i + 2] = pc.points[i].z;
  }
}

// Quat::operator*(Vec3) (include/mcl_3dl/quat.h:139-143), quaternion given as (x,y,z,w), NOT normalised.
void ref_quat_rotate(const float* q4, const float* v3, float* out3)
{
  const mcl_3dl::Vec3 o = mcl_3dl::Quat(q4[0], q4[1], q4[2], q4[3]) * mcl_3dl::Vec3(v3[0], v3[1], v3[2]);
  out3[0] = o.x_;
  out3[1] = o.y_;
  out3[2] = o.z_;
}

// LidarMeasurementModelLikelihood::measure (src/lidar_measurement_model_likelihood.cpp:105-139) for a batch of poses.
// threads<=1: the reference's own single-threaded execution. threads>1: OpenMP over particles.
double ref_likelihood_measure(void* h, const float* poses, size_t n_p, const float* scan_xyz,
                              const uint32_t* scan_label, size_t n_s, float* out_lik, float* out_quality, int threads)
{
  Ref* r = static_cast<Ref*>(h);
  Cloud::ConstPtr pc = makeCloud(scan_xyz, scan_label, n_s);
  const std::vector<mcl_3dl::Vec3> origins;
  const auto t0 = std::chrono::steady_clock::now();
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 1 ? threads : 1)
#endif
  for (long i = 0; i < static_cast<long>(n_p); ++i)
  {
    Kdtree::Ptr kd = r->kdtree;
    const mcl_3dl::LidarMeasurementResult res = r->lik->measure(kd, pc, origins, makeState(poses + 7 * i));
    out_lik[i] = res.likelihood;
    out_quality[i] = res.quality;
  }
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// LidarMeasurementModelBeam::measure (src/lidar_measurement_model_beam.cpp:124-155) for a batch of poses.
double ref_beam_measure(void* h, const float* poses, size_t n_p, const float* scan_xyz, const uint32_t* scan_label,
                        size_t n_b, const float* origins_xyz, size_t n_o, float* out_lik, float* out_quality,
                        int threads)
{
  Ref* r = static_cast<Ref*>(h);
  Cloud::ConstPtr pc = makeCloud(scan_xyz, scan_label, n_b);
  std::vector<mcl_3dl::Vec3> origins;
  for (size_t i = 0; i < n_o; ++i)
    origins.emplace_back(origins_xyz[3 * i], origins_xyz[3 * i + 1], origins_xyz[3 * i + 2]);
  const int nt = threads > 1 ? threads : 1;
  ensureBeamModels(r, nt);
  // build every raycaster's DDA grid outside the timed region (reference builds it lazily once per map stamp)
  if (n_p > 0 && n_b > 0)
    for (int t = 0; t < nt; ++t)
    {
      Kdtree::Ptr kd = r->kdtree;
      mcl_3dl::Raycast<PointType>::CastResult cr;
      r->beam[t]->getBeamStatus(kd, mcl_3dl::Vec3(0, 0, 0), mcl_3dl::Vec3(0, 0, 0), cr);
    }
  const auto t0 = std::chrono::steady_clock::now();
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
#endif
  for (long i = 0; i < static_cast<long>(n_p); ++i)
  {
#ifdef _OPENMP
    const int t = omp_get_thread_num();
#else
    const int t = 0;
#endif
    Kdtree::Ptr kd = r->kdtree;
    const mcl_3dl::LidarMeasurementResult res = r->beam[t]->measure(kd, pc, origins, makeState(poses + 7 * i));
    out_lik[i] = res.likelihood;
    out_quality[i] = res.quality;
  }
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// LidarMeasurementModelBeam::getBeamStatus (src/lidar_measurement_model_beam.cpp:157-192) per ray.
// status: 0 SHORT, 1 HIT, 2 LONG, 3 TOTAL_REFLECTION (enum order of lidar_measurement_model_beam.h:64-70).
// hit_index[i] = index into the map cloud of the collided point (or -1).
void ref_beam_status(void* h, const float* begin_xyz, const float* end_xyz, size_t n, int* status, int* hit_index)
{
  Ref* r = static_cast<Ref*>(h);
  for (size_t i = 0; i < n; ++i)
  {
    Kdtree::Ptr kd = r->kdtree;
    mcl_3dl::Raycast<PointType>::CastResult cr;
    const auto s = r->beam[0]->getBeamStatus(
        kd, mcl_3dl::Vec3(begin_xyz[3 * i], begin_xyz[3 * i + 1], begin_xyz[3 * i + 2]),
        mcl_3dl::Vec3(end_xyz[3 * i], end_xyz[3 * i + 1], end_xyz[3 * i + 2]), cr);
    status[i] = static_cast<int>(s);
    if (hit_index)
      hit_index[i] = (cr.point_ && s != mcl_3dl::LidarMeasurementModelBeam::BeamStatus::LONG) ?
                         static_cast<int>(cr.point_ - &r->map->points[0]) :
                         -1;
  }
}

// RaycastUsingDDA waypoints (include/mcl_3dl/raycasts/raycast_using_dda.h:66-159): the voxel-centre sequence a ray
// visits until the first collision (inclusive) or exhaustion, exactly as test/src/test_raycast_dda.cpp:157-183 collects it.
// returns number of waypoints written (<= max_out); *collided = 1 if the walk ended on a collision.
int ref_dda_waypoints(void* h, double map_grid_x, double map_grid_y, double map_grid_z, double dda_grid_size,
                      double ray_angle_half, double hit_tolerance, const float* begin3, const float* end3,
                      float* out_xyz, int max_out, int* collided, int* hit_index, int stop_at_collision)
{
  Ref* r = static_cast<Ref*>(h);
  mcl_3dl::RaycastUsingDDA<PointType> rc(map_grid_x, map_grid_y, map_grid_z, dda_grid_size, ray_angle_half,
                                         hit_tolerance);
  rc.setRay(r->kdtree, mcl_3dl::Vec3(begin3[0], begin3[1], begin3[2]), mcl_3dl::Vec3(end3[0], end3[1], end3[2]));
  mcl_3dl::Raycast<PointType>::CastResult cr;
  int n = 0;
  *collided = 0;
  if (hit_index)
    *hit_index = -1;
  while (rc.getNextCastResult(cr))
  {
    if (n < max_out)
    {
      out_xyz[3 * n + 0] = cr.pos_.x_;
      out_xyz[3 * n + 1] = cr.pos_.y_;
      out_xyz[3 * n + 2] = cr.pos_.z_;
    }
    ++n;
    if (cr.collision_)
    {
      if (!*collided && hit_index)
        *hit_index = static_cast<int>(cr.point_ - &r->map->points[0]);
      *collided = 1;
      if (stop_at_collision)
        break;
    }
  }
  return n;
}

// pf::ParticleFilter::measure (include/mcl_3dl/pf.h:252-279) driven with a precomputed per-particle likelihood.
// weights are updated in place; returns 1 if the "sum <= 0 -> restore" branch was taken.
int ref_pf_measure(float* weight_inout, const float* likelihood, size_t n, float* entropy)
{
  using PF = mcl_3dl::pf::ParticleFilter<mcl_3dl::State6DOF, float, mcl_3dl::ParticleWeightedMeanQuat,
                                         std::default_random_engine>;
  PF pf(static_cast<int>(n), 12345);
  size_t i = 0;
  for (auto it = pf.begin(); it != pf.end(); ++it, ++i)
    it->probability_ = weight_inout[i];
  size_t k = 0;
  pf.measure([&](const mcl_3dl::State6DOF&) -> float { return likelihood[k++]; });
  // the reference's own decision variable (pf.h:255-261): float, sequential
  float sum = 0;
  for (size_t j = 0; j < n; ++j)
  {
    float w = weight_inout[j];
    w *= likelihood[j];
    sum += w;
  }
  const int restored = !(sum > 0.0);
  i = 0;
  for (auto it = pf.begin(); it != pf.end(); ++it, ++i)
    weight_inout[i] = it->probability_;
  *entropy = restored ? NAN : pf.getEntropy();
  return restored;
}

// The node's whole measurement update for the LiDAR models (src/mcl_3dl.cpp:398-426 + pf.h:252-279):
//   per particle: likelihood = 1 * beam.likelihood * likelihood.likelihood  (std::map key order: "beam" < "likelihood")
//                 match_ratio_min/max from the likelihood model's quality
//                 * NormalLikelihood(odom_err_integ_lin.norm())     (nd.h:41-58: a*expf(-x*x/sq2))
//   then pf::measure's multiply / sum / normalise / entropy.
// use_beam / use_lik select which models are in the map (0 => that model contributes (1,0) like an empty cloud does).
// out_* may be NULL. Returns seconds spent inside pf->measure().
double ref_measure_update(void* h, const float* poses, const float* odom_err_integ_lin /*n_p*3 or NULL*/,
                          float* weight_inout, size_t n_p, const float* scan_lik_xyz, size_t n_s,
                          const float* scan_beam_xyz, const uint32_t* scan_beam_label, size_t n_b,
                          const float* origins_xyz, size_t n_o, float odom_err_integ_lin_sigma, float* out_lik,
                          float* out_beam, float* out_quality, float* entropy, float* match_ratio_min_out,
                          float* match_ratio_max_out, int* restored_out)
{
  Ref* r = static_cast<Ref*>(h);
  using PF = mcl_3dl::pf::ParticleFilter<mcl_3dl::State6DOF, float, mcl_3dl::ParticleWeightedMeanQuat,
                                         std::default_random_engine>;
  PF pf(static_cast<int>(n_p), 12345);
  {
    size_t i = 0;
    for (auto it = pf.begin(); it != pf.end(); ++it, ++i)
    {
      it->state_ = makeState(poses + 7 * i);
      if (odom_err_integ_lin)
        it->state_.odom_err_integ_lin_ = mcl_3dl::Vec3(odom_err_integ_lin[3 * i], odom_err_integ_lin[3 * i + 1],
                                                       odom_err_integ_lin[3 * i + 2]);
      it->probability_ = weight_inout[i];
    }
  }
  std::map<std::string, mcl_3dl::LidarMeasurementModelBase::Ptr> lidar_measurements;
  std::map<std::string, Cloud::ConstPtr> pc_locals;
  lidar_measurements["likelihood"] = r->lik;
  lidar_measurements["beam"] = r->beam[0];
  pc_locals["likelihood"] = makeCloud(scan_lik_xyz, nullptr, n_s);
  pc_locals["beam"] = makeCloud(scan_beam_xyz, scan_beam_label, n_b);
  std::vector<mcl_3dl::Vec3> origins;
  for (size_t i = 0; i < n_o; ++i)
    origins.emplace_back(origins_xyz[3 * i], origins_xyz[3 * i + 1], origins_xyz[3 * i + 2]);

  float match_ratio_min = 1.0;
  float match_ratio_max = 0.0;
  // NormalLikelihood<float> (include/mcl_3dl/nd.h:41-58) restated: nd.h itself needs Eigen/LU for its N-D sibling.
  const float sigma = odom_err_integ_lin_sigma;
  const float nd_a = 1.0 / std::sqrt(2.0 * M_PI * sigma * sigma);
  const float nd_sq2 = sigma * sigma * 2.0;
  size_t idx = 0;
  std::vector<float> returned(n_p, 0.f);
  Kdtree::Ptr kdtree = r->kdtree;
  const auto measure_func = [&](const mcl_3dl::State6DOF& s) -> float
  {
    float likelihood = 1;
    std::map<std::string, float> qualities;
    for (const auto& lm : lidar_measurements)
    {
      const mcl_3dl::LidarMeasurementResult result = lm.second->measure(kdtree, pc_locals[lm.first], origins, s);
      likelihood *= result.likelihood;
      qualities[lm.first] = result.quality;
      if (lm.first == "likelihood" && out_lik)
        out_lik[idx] = result.likelihood;
      if (lm.first == "beam" && out_beam)
        out_beam[idx] = result.likelihood;
    }
    if (out_quality)
      out_quality[idx] = qualities["likelihood"];
    if (match_ratio_min > qualities["likelihood"])
      match_ratio_min = qualities["likelihood"];
    if (match_ratio_max < qualities["likelihood"])
      match_ratio_max = qualities["likelihood"];
    const float x = s.odom_err_integ_lin_.norm();
    const float odom_error = nd_a * expf(-x * x / nd_sq2);
    returned[idx] = likelihood * odom_error;
    return returned[idx++];
  };
  const auto t0 = std::chrono::steady_clock::now();
  pf.measure(measure_func);
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

  // the restore branch is decided by the reference's float sequential sum of w*L (pf.h:255-261)
  float sum = 0;
  for (size_t i = 0; i < n_p; ++i)
  {
    float w = weight_inout[i];
    w *= returned[i];
    sum += w;
  }
  const bool all_same = !(sum > 0.0);
  {
    size_t i = 0;
    for (auto it = pf.begin(); it != pf.end(); ++it, ++i)
      weight_inout[i] = it->probability_;
  }
  if (restored_out)
    *restored_out = all_same ? 1 : 0;
  if (entropy)
    *entropy = all_same ? NAN : pf.getEntropy();
  if (match_ratio_min_out)
    *match_ratio_min_out = match_ratio_min;
  if (match_ratio_max_out)
    *match_ratio_max_out = match_ratio_max;
  return dt;
}

// pf::ParticleFilter::expectationBiased (include/mcl_3dl/pf.h:294-303) with ParticleWeightedMeanQuat
// (include/mcl_3dl/state_6dof.h:316-355), plus max() / maxBiased() (pf.h:361-390).  bias == NULL -> probability_bias_ = 1.
void ref_expectation(const float* poses, const float* weights, const float* bias, size_t n, float* out_mean7,
                     int* out_max_index, int* out_max_biased_index)
{
  using PF = mcl_3dl::pf::ParticleFilter<mcl_3dl::State6DOF, float, mcl_3dl::ParticleWeightedMeanQuat,
                                         std::default_random_engine>;
  PF pf(static_cast<int>(n), 12345);
  size_t i = 0;
  for (auto it = pf.begin(); it != pf.end(); ++it, ++i)
  {
    it->state_ = makeState(poses + 7 * i);
    it->state_.odom_err_integ_lin_.x_ = static_cast<float>(i);  // tag: which particle max() returned
    it->probability_ = weights[i];
    it->probability_bias_ = bias ? bias[i] : 1.0f;
  }
  const mcl_3dl::State6DOF e = pf.expectationBiased();
  out_mean7[0] = e.pos_.x_;
  out_mean7[1] = e.pos_.y_;
  out_mean7[2] = e.pos_.z_;
  out_mean7[3] = e.rot_.x_;
  out_mean7[4] = e.rot_.y_;
  out_mean7[5] = e.rot_.z_;
  out_mean7[6] = e.rot_.w_;
  *out_max_index = static_cast<int>(pf.max().odom_err_integ_lin_.x_);
  *out_max_biased_index = static_cast<int>(pf.maxBiased().odom_err_integ_lin_.x_);
}

// pf::ParticleFilter::covariance(1.0, 1.0) (pf.h:304-360) with State6DOF::covElement (state_6dof.h:162-184); also returns
// the expectation(1.0) it is centred on (pf.h:280-293, including its "stop once the running sum exceeds pass_ratio" rule).
void ref_covariance(const float* poses, const float* weights, size_t n, float* out_cov36, float* out_mean7)
{
  using PF = mcl_3dl::pf::ParticleFilter<mcl_3dl::State6DOF, float, mcl_3dl::ParticleWeightedMeanQuat,
                                         std::default_random_engine>;
  PF pf(static_cast<int>(n), 12345);
  size_t i = 0;
  for (auto it = pf.begin(); it != pf.end(); ++it, ++i)
  {
    it->state_ = makeState(poses + 7 * i);
    it->probability_ = weights[i];
    it->probability_bias_ = 1.0f;
  }
  const mcl_3dl::State6DOF e = pf.expectation();
  out_mean7[0] = e.pos_.x_;
  out_mean7[1] = e.pos_.y_;
  out_mean7[2] = e.pos_.z_;
  out_mean7[3] = e.rot_.x_;
  out_mean7[4] = e.rot_.y_;
  out_mean7[5] = e.rot_.z_;
  out_mean7[6] = e.rot_.w_;
  std::vector<mcl_3dl::State6DOF> cov = pf.covariance(1.0, 1.0);
  for (size_t j = 0; j < 6; ++j)
    for (size_t k = 0; k < 6; ++k)
      out_cov36[6 * j + k] = cov[j][k];
}

namespace
{
using PF6 = mcl_3dl::pf::ParticleFilter<mcl_3dl::State6DOF, float, mcl_3dl::ParticleWeightedMeanQuat,
                                        std::default_random_engine>;
void loadStates13(PF6& pf, const float* state13, const float* weight)
{
  size_t i = 0;
  for (auto it = pf.begin(); it != pf.end(); ++it, ++i)
  {
    for (size_t k = 0; k < 13; ++k)
      it->state_[k] = state13[13 * i + k];
    it->probability_ = weight[i];
  }
}
void storeStates13(PF6& pf, float* state13, float* weight)
{
  size_t i = 0;
  for (auto it = pf.begin(); it != pf.end(); ++it, ++i)
  {
    for (size_t k = 0; k < 13; ++k)
      state13[13 * i + k] = it->state_[k];
    if (weight)
      weight[i] = it->probability_;
  }
}
mcl_3dl::State6DOF sigmaState(const float* sigma6)
{
  return mcl_3dl::State6DOF(mcl_3dl::Vec3(sigma6[0], sigma6[1], sigma6[2]), mcl_3dl::Vec3(sigma6[3], sigma6[4], sigma6[5]));
}
}  // namespace

// pf::ParticleFilter::resample(sigma) (include/mcl_3dl/pf.h:187-225) with the filter's engine seeded with `seed`, on
// 13-dof State6DOF vectors (state_6dof.h:80-149: pos, rot, odom_err_integ_lin, odom_err_integ_ang).
void ref_resample(const float* state13, const float* weight, size_t n, unsigned seed, const float* sigma6,
                  float* out_state13, float* out_weight)
{
  PF6 pf(static_cast<int>(n), seed);
  loadStates13(pf, state13, weight);
  pf.resample(sigmaState(sigma6));
  storeStates13(pf, out_state13, out_weight);
}

// The random numbers that call consumes, in order, reproduced on an identically seeded engine: initial_p =
// uniform_real_distribution<float>(0, pstep)(engine) (pf.h:203), then one State6DOF::generateNoise per duplicated
// particle (pf.h:216, state_6dof.h:226-247, noise_generators/diagonal_noise_generator.h:66-80).
void ref_resample_draws(unsigned seed, float pstep, const float* sigma6, size_t n_dup, float* out_initial_p,
                        float* out_noise13)
{
  std::default_random_engine engine(seed);
  *out_initial_p = std::uniform_real_distribution<float>(0.0, pstep)(engine);
  const mcl_3dl::DiagonalNoiseGenerator<float> gen(mcl_3dl::State6DOF(), sigmaState(sigma6));
  for (size_t d = 0; d < n_dup; ++d)
  {
    mcl_3dl::State6DOF noise = mcl_3dl::State6DOF::generateNoise<mcl_3dl::State6DOF>(engine, gen);
    for (size_t k = 0; k < 13; ++k)
      out_noise13[13 * d + k] = noise[k];
  }
}

// pf::ParticleFilter::resizeParticle(num) (pf.h:399-436)
void ref_resize(const float* state13, const float* weight, size_t n, size_t n_out, float* out_state13, float* out_weight)
{
  PF6 pf(static_cast<int>(n), 12345);
  loadStates13(pf, state13, weight);
  pf.resizeParticle(n_out);
  storeStates13(pf, out_state13, out_weight);
}
// ---- the steps either side of the measurement update ------------------------------------------
namespace
{
size_t storeCloud(const Cloud& pc, float* out_xyz, uint32_t* out_label, size_t cap)
{
  const size_t n = pc.points.size();
  if (out_xyz && n <= cap)
    for (size_t i = 0; i < n; ++i)
    {
      out_xyz[3 * i + 0] = pc.points[i].x;
      out_xyz[3 * i + 1] = pc.points[i].y;
      out_xyz[3 * i + 2] = pc.points[i].z;
      if (out_label)
        out_label[i] = pc.points[i].label;
    }
  return n;
}

// hands the clipped cloud through unchanged: filter()'s clip step alone
class PassThroughSampler : public mcl_3dl::PointCloudRandomSampler<PointType>
{
public:
  mutable size_t asked = 0;
  Cloud::Ptr sample(const Cloud::ConstPtr& pc, const size_t num) const final
  {
    asked = num;
    return Cloud::Ptr(new Cloud(*pc));
  }
};

mcl_3dl::LidarMeasurementModelBase* modelOf(Ref* r, int model)
{
  if (model == 0)
    return r->lik.get();
  return r->beam[0].get();
}
}  // namespace

// pcl::VoxelGrid as the node uses it (src/mcl_3dl.cpp:147-151, 363-367, 1155-1158) — the restated shim, see its header
size_t ref_voxel_grid(const float* xyz, const uint32_t* label, size_t n, const float* leaf3, float* out_xyz,
                      uint32_t* out_label, size_t cap)
{
  Cloud::Ptr in = makeCloud(xyz, label, n);
  Cloud out;
  pcl::VoxelGrid<PointType> ds;
  ds.setInputCloud(in);
  ds.setLeafSize(leaf3[0], leaf3[1], leaf3[2]);
  ds.filter(out);
  return storeCloud(out, out_xyz, out_label, cap);
}

// the clip step of the reference's own filter() (likelihood.cpp:79-103 / beam.cpp:98-122) and the num_points_ it asks
// the sampler for
size_t ref_clip(void* h, int model, const float* xyz, const uint32_t* label, size_t n, float* out_xyz, uint32_t* out_label,
                size_t cap, size_t* num_points)
{
  Ref* r = static_cast<Ref*>(h);
  PassThroughSampler sampler;
  const Cloud::Ptr out = modelOf(r, model)->filter(makeCloud(xyz, label, n), sampler);
  if (num_points)
    *num_points = sampler.asked;
  return storeCloud(*out, out_xyz, out_label, cap);
}

// the reference's filter() with the reference's PointCloudUniformSampler, its engine re-seeded with `seed`
size_t ref_filter_uniform(void* h, int model, const float* xyz, const uint32_t* label, size_t n, unsigned seed,
                          float* out_xyz, uint32_t* out_label, size_t cap)
{
  Ref* r = static_cast<Ref*>(h);
  mcl_3dl::PointCloudUniformSampler<PointType> sampler;
  sampler.engine_.reset(new std::default_random_engine(seed));
  const Cloud::Ptr out = modelOf(r, model)->filter(makeCloud(xyz, label, n), sampler);
  return storeCloud(*out, out_xyz, out_label, cap);
}

// the indices that sampler draws from a clipped cloud of n_clipped points (point_cloud_uniform_sampler.h:66-71): what a
// caller of mcl3dl_hip_scan_finish does with its own engine
void ref_uniform_indices(unsigned seed, size_t n_clipped, size_t num, uint32_t* out)
{
  std::default_random_engine engine(seed);
  if (n_clipped == 0)
    return;
  std::uniform_int_distribution<size_t> ud(0, n_clipped - 1);
  for (size_t i = 0; i < num; ++i)
    out[i] = static_cast<uint32_t>(ud(engine));
}

// src/mcl_3dl.cpp:771-789 restated on the reference's own kd-tree and State6DOF::transform:
// cls[i] = 2 unmatched (radiusSearch(p, unmatch_output_dist) finds nothing), 1 matched (sqdist < match_output_dist^2), 0 neither
void ref_match_split(void* h, const float* pose7, const float* xyz, size_t n, double unmatch_dist, double match_dist,
                     uint8_t* cls, float* out_xyz)
{
  Ref* r = static_cast<Ref*>(h);
  Cloud::Ptr pc_local = makeCloud(xyz, nullptr, n);
  const mcl_3dl::State6DOF e = makeState(pose7);
  e.transform(*pc_local);
  std::vector<int> id(1);
  std::vector<float> sqdist(1);
  const double match_dist_sq = match_dist * match_dist;
  for (size_t i = 0; i < n; ++i)
  {
    const PointType& p = pc_local->points[i];
    cls[i] = 0;
    if (!r->kdtree->radiusSearch(p, unmatch_dist, id, sqdist, 1))
      cls[i] = 2;
    else if (sqdist[0] < match_dist_sq)
      cls[i] = 1;
    out_xyz[3 * i + 0] = p.x;
    out_xyz[3 * i + 1] = p.y;
    out_xyz[3 * i + 2] = p.z;
  }
}
}  // extern "C"
