// tests/cpp/adapter_demo.cpp — TEST INFRASTRUCTURE: a miniature of the reference node's call site
// (src/mcl_3dl.cpp:1315-1329 set-up, :377-426 measurement update) compiled against the DROP-IN headers of
// mcl_3dl_amd/cpp/include (same class names as the reference) + the reference's remaining headers + stand-in PCL.
// It proves that code written against the reference's plugin surface runs unmodified on the GPU engine:
//   scene file in -> LidarMeasurementModel{Likelihood,Beam} + pf::ParticleFilter::measure -> result file out.
// tests/test_gpu_adapter.py compares the result file with the CPU oracle.
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
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
#include <mcl_3dl/state_6dof.h>

using PointType = mcl_3dl::LidarMeasurementModelBase::PointType;
using Cloud = pcl::PointCloud<PointType>;

// src/mcl_3dl.cpp:109-126
class MyPointRepresentation : public pcl::PointRepresentation<PointType>
{
  using pcl::PointRepresentation<PointType>::nr_dimensions_;

public:
  MyPointRepresentation()
  {
    nr_dimensions_ = 3;
    trivial_ = true;
  }
  virtual void copyToFloatArray(const PointType& p, float* out) const
  {
    out[0] = p.x;
    out[1] = p.y;
    out[2] = p.z;
  }
};

// The scene file holds the clouds AFTER sampling; the node's filter() step (src/mcl_3dl.cpp:378-383) still runs — with
// clips wide open and this sampler, which hands the clipped cloud through — because the drop-in models learn about the
// update's clouds there.
class PassThroughSampler : public mcl_3dl::PointCloudRandomSampler<PointType>
{
public:
  Cloud::Ptr sample(const Cloud::ConstPtr& pc, const size_t) const final
  {
    Cloud::Ptr out(new Cloud);
    *out = *pc;
    return out;
  }
};

template <typename T>
static std::vector<T> readVec(FILE* f, size_t n)
{
  std::vector<T> v(n);
  if (n && fread(v.data(), sizeof(T), n, f) != n)
  {
    fprintf(stderr, "short read\n");
    exit(2);
  }
  return v;
}

static Cloud::Ptr makeCloud(const std::vector<float>& xyz, const std::vector<uint32_t>& label)
{
  Cloud::Ptr pc(new Cloud);
  for (size_t i = 0; i < xyz.size() / 3; ++i)
  {
    PointType p;
    p.x = xyz[3 * i];
    p.y = xyz[3 * i + 1];
    p.z = xyz[3 * i + 2];
    p.label = label.empty() ? 0 : label[i];
    pc->push_back(p);
  }
  return pc;
}

int main(int argc, char** argv)
{
  if (argc != 3 && argc != 4)
  {
    fprintf(stderr, "usage: adapter_demo scene.bin result.bin [timed repetitions]\n");
    return 2;
  }
  const int reps = argc == 4 ? atoi(argv[3]) : 0;
  FILE* f = fopen(argv[1], "rb");
  if (!f)
    return 2;
  const auto hdr = readVec<uint64_t>(f, 8);  // n_map n_p n_s n_b n_o beam_num_points short_only filter_label_max
  const auto fl = readVec<float>(f, 5);      // dist_weight[3], has_weight, odom sigma
  const size_t n_map = hdr[0], n_p = hdr[1], n_s = hdr[2], n_b = hdr[3], n_o = hdr[4];
  const auto map_xyz = readVec<float>(f, 3 * n_map);
  const auto map_label = readVec<uint32_t>(f, n_map);
  const auto poses = readVec<float>(f, 7 * n_p);
  const auto odom_err = readVec<float>(f, 3 * n_p);
  const auto weights = readVec<float>(f, n_p);
  const auto scan_lik = readVec<float>(f, 3 * n_s);
  const auto scan_beam = readVec<float>(f, 3 * n_b);
  const auto beam_label = readVec<uint32_t>(f, n_b);
  const auto origins_f = readVec<float>(f, 3 * n_o);
  fclose(f);

  // ---- set-up as in MCL3dlNode::configure, src/mcl_3dl.cpp:1270-1329 ------------------------------------------
  auto lik_params = std::make_shared<mcl_3dl::LidarMeasurementModelLikelihoodParameters>();
  auto beam_params = std::make_shared<mcl_3dl::LidarMeasurementModelBeamParameters>();
  beam_params->num_points_default_ = hdr[5];
  beam_params->add_penalty_short_only_mode_ = hdr[6] != 0;
  beam_params->filter_label_max_ = static_cast<uint32_t>(hdr[7]);
  beam_params->use_raycast_using_dda_ = true;
  lik_params->clip_near_ = beam_params->clip_near_ = 0.0f;
  lik_params->clip_far_ = beam_params->clip_far_ = 1.0e9f;
  lik_params->clip_z_min_ = beam_params->clip_z_min_ = -1.0e9f;
  lik_params->clip_z_max_ = beam_params->clip_z_max_ = 1.0e9f;
  std::shared_ptr<MyPointRepresentation> point_rep(new MyPointRepresentation);
  if (fl[3] != 0.f)
    point_rep->setRescaleValues(fl.data());

  std::map<std::string, mcl_3dl::LidarMeasurementModelBase::Ptr> lidar_measurements_;
  lidar_measurements_["likelihood"] =
      mcl_3dl::LidarMeasurementModelBase::Ptr(new mcl_3dl::LidarMeasurementModelLikelihood(lik_params));
  lidar_measurements_["beam"] =
      mcl_3dl::LidarMeasurementModelBase::Ptr(new mcl_3dl::LidarMeasurementModelBeam(beam_params));
  float max_search_radius = 0;
  for (const auto& lm : lidar_measurements_)
    max_search_radius = std::max(max_search_radius, lm.second->getMaxSearchRange());
  mcl_3dl::ChunkedKdtree<PointType>::Ptr kdtree_(new mcl_3dl::ChunkedKdtree<PointType>(20.0, max_search_radius));
  kdtree_->setEpsilon(0.1 / 16);
  if (fl[3] != 0.f)
    kdtree_->setPointRepresentation(point_rep);
  Cloud::Ptr map = makeCloud(map_xyz, map_label);
  map->header.stamp = 42;
  kdtree_->setInputCloud(map);

  std::shared_ptr<mcl_3dl::pf::ParticleFilter<mcl_3dl::State6DOF, float, mcl_3dl::ParticleWeightedMeanQuat,
                                              std::default_random_engine>>
      pf_(new mcl_3dl::pf::ParticleFilter<mcl_3dl::State6DOF, float, mcl_3dl::ParticleWeightedMeanQuat,
                                          std::default_random_engine>(static_cast<int>(n_p), 12345));
  {
    size_t i = 0;
    for (auto it = pf_->begin(); it != pf_->end(); ++it, ++i)
    {
      it->state_ = mcl_3dl::State6DOF(mcl_3dl::Vec3(poses[7 * i], poses[7 * i + 1], poses[7 * i + 2]),
                                      mcl_3dl::Quat(poses[7 * i + 3], poses[7 * i + 4], poses[7 * i + 5], poses[7 * i + 6]));
      it->state_.odom_err_integ_lin_ = mcl_3dl::Vec3(odom_err[3 * i], odom_err[3 * i + 1], odom_err[3 * i + 2]);
      it->probability_ = weights[i];
    }
  }

  // ---- the measurement update, src/mcl_3dl.cpp:377-426 ---------------------------------------------------------
  // :377-383: every model filters the accumulated cloud into its own pc_local (here: the scene's per-model clouds)
  std::map<std::string, Cloud::ConstPtr> pc_raw, pc_locals;
  pc_raw["likelihood"] = makeCloud(scan_lik, {});
  pc_raw["beam"] = makeCloud(scan_beam, beam_label);
  const PassThroughSampler sampler;
  for (auto& lm : lidar_measurements_)
  {
    lm.second->setGlobalLocalizationStatus(n_p, n_p);
    pc_locals[lm.first] = lm.second->filter(pc_raw[lm.first], sampler);
  }
  std::vector<mcl_3dl::Vec3> origins;
  for (size_t i = 0; i < n_o; ++i)
    origins.emplace_back(origins_f[3 * i], origins_f[3 * i + 1], origins_f[3 * i + 2]);

  std::vector<float> out_lik(n_p), out_beam(n_p), out_quality(n_p);
  size_t idx = 0;
  float match_ratio_min = 1.0;
  float match_ratio_max = 0.0;
  const float sigma = fl[4];
  const float nd_a = 1.0 / std::sqrt(2.0 * M_PI * sigma * sigma);  // NormalLikelihood<float>, nd.h:44-48
  const float nd_sq2 = sigma * sigma * 2.0;
  const auto measure_func = [&](const mcl_3dl::State6DOF& s) -> float
  {
    float likelihood = 1;
    std::map<std::string, float> qualities;
    for (const auto& lm : lidar_measurements_)
    {
      const mcl_3dl::LidarMeasurementResult result = lm.second->measure(kdtree_, pc_locals[lm.first], origins, s);
      likelihood *= result.likelihood;
      qualities[lm.first] = result.quality;
      (lm.first == "beam" ? out_beam : out_lik)[idx] = result.likelihood;
    }
    out_quality[idx] = qualities["likelihood"];
    if (match_ratio_min > qualities["likelihood"])
      match_ratio_min = qualities["likelihood"];
    if (match_ratio_max < qualities["likelihood"])
      match_ratio_max = qualities["likelihood"];
    ++idx;
    const float x = s.odom_err_integ_lin_.norm();
    return likelihood * (nd_a * expf(-x * x / nd_sq2));
  };
  pf_->measure(measure_func);

  // Optional timing of this very statement (bench.py "route_a"): the node-visible latency of one measurement update
  // through the per-particle virtuals. Weights are reset before every repetition (resampling leaves them uniform).
  if (reps > 0)
  {
    std::vector<float> posterior(n_p);
    {
      size_t i = 0;
      for (auto it = pf_->begin(); it != pf_->end(); ++it, ++i)
        posterior[i] = it->probability_;
    }
    double total_ms = 0;
    mcl_3dl::hip::Engine& eng = mcl_3dl::hip::Engine::shared();
    // clock ramp: a GPU that has just compiled a map needs ~0.1 s of load before its timings mean anything (bench.py's
    // `prewarm`); five cold repetitions read 0.31 ms for a batched call whose kernels take 0.24
    for (const auto t_warm = std::chrono::steady_clock::now();
         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_warm).count() < 150.0;)
    {
      size_t i = 0;
      for (auto it = pf_->begin(); it != pf_->end(); ++it, ++i)
        it->probability_ = weights[i];
      idx = 0;
      pf_->measure(measure_func);
    }
    for (int r = 0; r < reps + 1; ++r)
    {
      if (r == 1)
        eng.profile = mcl_3dl::hip::Engine::Profile();
      size_t i = 0;
      for (auto it = pf_->begin(); it != pf_->end(); ++it, ++i)
        it->probability_ = weights[i];
      idx = 0;
      const auto t0 = std::chrono::steady_clock::now();
      pf_->measure(measure_func);
      const auto t1 = std::chrono::steady_clock::now();
      if (r > 0)  // the first repetition is a warm-up
        total_ms += std::chrono::duration<double, std::milli>(t1 - t0).count();
    }
    printf("route_a_ms_per_update %.6f\n", total_ms / reps);
    // of which inside the engine-side adapter (the rest is the reference's pf.h loop: particle copy, 2 N virtual calls,
    // weight product, normalisation, entropy)
    // (measure_batch = the call that enqueues the batch; waiting = blocked on a slice that had not arrived yet)
    printf("route_a_breakdown_us pose_gather_upload %.1f cloud_pack %.1f measure_batch %.1f batched_calls_per_update %.2f "
           "waiting %.1f\n",
           eng.profile.poses_us / reps, eng.profile.pack_us / reps, eng.profile.batch_us / reps,
           static_cast<double>(eng.profile.launches) / reps, eng.profile.wait_us / reps);
    size_t i = 0, n_diff = 0, first = 0;
    for (auto it = pf_->begin(); it != pf_->end(); ++it, ++i)
      if (it->probability_ != posterior[i] && n_diff++ == 0)
        first = i;
    if (n_diff)
    {
      fprintf(stderr, "repetition changed the posterior of %zu particle(s), first %zu\n", n_diff, first);
      return 3;
    }
  }

  // a call outside pf::measure (the debug-marker path, src/mcl_3dl.cpp:471-478): batch of one
  auto beam = std::dynamic_pointer_cast<mcl_3dl::LidarMeasurementModelBeam>(lidar_measurements_["beam"]);
  mcl_3dl::Raycast<PointType>::CastResult cr;
  const mcl_3dl::State6DOF s0 = pf_->getParticle(0);
  const auto single = lidar_measurements_["likelihood"]->measure(kdtree_, pc_locals["likelihood"], origins, s0);
  const int status = n_b ? static_cast<int>(beam->getBeamStatus(
                               kdtree_, s0.pos_, s0.pos_ + mcl_3dl::Vec3(3.0, 0.5, -0.2), cr)) : -1;

  // two live models of one kind (a node never builds them): whoever is used owns the engine's slots of the kind,
  // and a dying model clears them only while they are its own
  {
    mcl_3dl::LidarMeasurementModelBase::Ptr second(new mcl_3dl::LidarMeasurementModelLikelihood(lik_params));
    second->setGlobalLocalizationStatus(n_p, n_p);
    const auto r2 = second->measure(kdtree_, pc_locals["likelihood"], origins, s0);
    const auto r1 = lidar_measurements_["likelihood"]->measure(kdtree_, pc_locals["likelihood"], origins, s0);
    second.reset();  // the first model used last: it owns the slots, the dying one must not clear them
    const auto r3 = lidar_measurements_["likelihood"]->measure(kdtree_, pc_locals["likelihood"], origins, s0);
    mcl_3dl::LidarMeasurementModelBase::Ptr third(new mcl_3dl::LidarMeasurementModelLikelihood(lik_params));
    third->setGlobalLocalizationStatus(n_p, n_p);
    third.reset();   // registered last, never used again: its death leaves the kind unowned, the first model claims it back
    const auto r4 = lidar_measurements_["likelihood"]->measure(kdtree_, pc_locals["likelihood"], origins, s0);
    for (const auto& r : { r1, r2, r3, r4 })
      if (r.likelihood != single.likelihood || r.quality != single.quality)
      {
        fprintf(stderr, "two live likelihood models: %.9g / %.9g instead of %.9g / %.9g\n", r.likelihood, r.quality,
                single.likelihood, single.quality);
        return 4;
      }
  }

  FILE* g = fopen(argv[2], "wb");
  std::vector<float> w(n_p);
  {
    size_t i = 0;
    for (auto it = pf_->begin(); it != pf_->end(); ++it, ++i)
      w[i] = it->probability_;
  }
  const float tail[6] = { pf_->getEntropy(), match_ratio_min, match_ratio_max, single.likelihood, single.quality,
                          static_cast<float>(status) };
  fwrite(w.data(), 4, n_p, g);
  fwrite(out_lik.data(), 4, n_p, g);
  fwrite(out_beam.data(), 4, n_p, g);
  fwrite(out_quality.data(), 4, n_p, g);
  fwrite(tail, 4, 6, g);
  // the likelihood cloud as the node holds it after filter() (MCL3DL_HIP_ENGINE_ORDER=1: in the engine's scan order)
  for (const PointType& q : pc_locals["likelihood"]->points)
  {
    const float xyz[3] = { q.x, q.y, q.z };
    fwrite(xyz, 4, 3, g);
  }
  fclose(g);
  printf("adapter_demo: %zu particles, entropy %.6f, match ratio [%.4f, %.4f]\n", n_p, tail[0], tail[1], tail[2]);
  return 0;
}
