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
  ~BatchedLidarModel()  // (the reference base class declares no virtual destructor; the node holds models in shared_ptrs made from the derived type)
  {
    if (registered_)
    {
      Engine& e = Engine::shared();
      if (e.owner[kind_] == this)  // (a newer model of the same kind keeps its registration)
      {
        e.push_params[kind_] = nullptr;
        e.filtered[kind_] = Engine::FilteredScan();
        e.results[kind_] = Engine::Results();
        e.owner[kind_] = nullptr;
      }
    }
  }

  // reference: both models' setGlobalLocalizationStatus (src/lidar_measurement_model_likelihood.cpp:63-77,
  // src/lidar_measurement_model_beam.cpp:82-96)
  void setGlobalLocalizationStatus(const size_t num_particles, const size_t current_num_particles) override
  {
    points_now_ = pointsPerParticle(points_default_, points_global_, num_particles, current_num_particles);
  }

  // reference: both models' filter() (likelihood.cpp:79-103, beam.cpp:98-122): clip, then sampler.sample(num_points_).
  // The sampled cloud is also packed for the engine right here, so that the first measure() of the coming pf::measure —
  // whichever model it reaches — can evaluate BOTH models' scans in one launch (engine.hpp, "one launch for BOTH models").
  Cloud::Ptr filter(const Cloud::ConstPtr& pc, const PointCloudRandomSampler<PointType>& sampler) const override
  {
    const Cloud::Ptr clipped = clipCloud(*pc, clip_.near_sq, clip_.far_sq, clip_.z_min, clip_.z_max);
    const Cloud::Ptr sampled = sampler.sample(clipped, points_now_);
    if (registered_ && sampled)
    {
      claim();
      Engine& e = Engine::shared();
      Engine::FilteredScan& f = e.filtered[kind_];
      const double t0 = Engine::nowUs();
      packCloud(*sampled, f.xyz, &f.label);
      if (e.engine_order && kind_ == Engine::LIKELIHOOD && !sampled->points.empty())
      {
        // The order of a sampled cloud is the order the sampler drew it in (point_cloud_uniform_sampler.h:66-71) and means
        // nothing; the node holds the cloud in the ENGINE's order instead, so that the float recurrence of
        // likelihood.cpp:124-134 over the node's cloud is the one the likelihood kernel runs (strict_order = 3).
        std::vector<std::uint32_t> order(sampled->points.size());
        if (mcl3dl_hip_scan_order_host(f.xyz.data(), order.size(), order.data()) != 0)
          throw std::runtime_error("mcl3dl_hip: mcl3dl_hip_scan_order_host rejected the sampled cloud");
        Cloud::VectorType pts(sampled->points.size());
        for (std::size_t k = 0; k < order.size(); ++k)
          pts[k] = sampled->points[order[k]];
        sampled->points.swap(pts);
        packCloud(*sampled, f.xyz, &f.label);
      }
      e.profile.pack_us += Engine::nowUs() - t0;
      f.cloud = sampled.get();
      f.from_filter = true;
      last_filtered_ = sampled;  // alive until the next filter(): its address cannot be handed to another cloud meanwhile
    }
    return sampled;
  }

protected:
  struct Clip
  {
    float near_sq = 0.f, far_sq = 0.f, z_min = 0.f, z_max = 0.f;
  };
  // `push` sends this model's parameters to the engine (called ahead of every launch that evaluates this model's scan)
  void configureFilter(const Engine::Kind kind, std::function<void()> push, const std::size_t points_default,
                       const std::size_t points_global, const float clip_near, const float clip_far, const float clip_z_min,
                       const float clip_z_max)
  {
    kind_ = kind;
    points_default_ = points_now_ = points_default;
    points_global_ = points_global;
    clip_.near_sq = clip_near * clip_near;
    clip_.far_sq = clip_far * clip_far;
    clip_.z_min = clip_z_min;
    clip_.z_max = clip_z_max;
    push_ = std::move(push);
    registered_ = true;
    claim();
  }

  // The engine's slots of this kind belong to one model instance at a time. Two live models of one kind (a node never builds
  // them, a test may) take turns: whoever is used claims the slots and drops the other's cached results.
  void claim() const
  {
    Engine& e = Engine::shared();
    if (e.owner[kind_] == this)
      return;
    e.endBatch();
    e.owner[kind_] = this;
    e.push_params[kind_] = push_;
    e.results[kind_] = Engine::Results();
    e.filtered[kind_] = Engine::FilteredScan();
  }

  // Where `s` sits in the published batch and whether the cached results of this model answer it. The cache is tested
  // FIRST: inside pf::measure this runs once per particle, so nothing here may cost more than a few comparisons (packing
  // the poses of the whole batch happens in refreshPoses(), once per epoch).
  struct Slot
  {
    std::size_t index = 0, count = 1;
    std::uint64_t epoch = 0;
    bool refresh = true;
  };
  Slot lookup(const State6DOF& s, const void* cloud) const
  {
    if (Engine::shared().owner[kind_] != this)
      claim();
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
    const Engine::Results& r = results();
    slot.refresh = !(slot.epoch != 0 && r.epoch == slot.epoch && r.cloud == cloud && r.likelihood.size() == slot.count);
    return slot;
  }
  Engine::Results& results() const
  {
    return Engine::shared().results[kind_];
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

  // The batched launch behind a measure() call that the cache could not answer: this model's scan `pc` and — inside
  // pf::measure, when the other model's filter() result is waiting — the other model's scan too. Fills results() (and the
  // other model's results).
  void evaluate(ChunkedKdtree<PointType>& kdtree, const Cloud& pc, const std::vector<Vec3>& origins, const State6DOF& s,
                const Slot& slot) const
  {
    Engine& e = Engine::shared();
    e.endBatch();  // (a batch of the other model still in flight writes into its result vectors)
    syncMap(e, kdtree);
    if (!e.push_params[kind_])
      throw std::runtime_error("mcl3dl_hip: LiDAR model used before its parameters were configured");
    e.push_params[kind_]();
    const double t0 = Engine::nowUs();
    refreshPoses(e, s, slot);
    const double t1 = Engine::nowUs();
    e.profile.poses_us += t1 - t0;
    Engine::FilteredScan& mine = e.filtered[kind_];
    if (mine.cloud != &pc || mine.xyz.size() != 3 * pc.points.size())
    {
      packCloud(pc, mine.xyz, &mine.label);  // a cloud filter() has not seen (tests, the debug-marker path)
      mine.cloud = &pc;
      mine.from_filter = false;
      e.profile.pack_us += Engine::nowUs() - t1;
    }
    const Engine::Kind other_kind = kind_ == Engine::LIKELIHOOD ? Engine::BEAM : Engine::LIKELIHOOD;
    Engine::FilteredScan& oth = e.filtered[other_kind];
    const bool both = slot.epoch != 0 && oth.from_filter && oth.cloud && !oth.xyz.empty() && e.push_params[other_kind] &&
                      e.results[other_kind].epoch != slot.epoch;
    if (both)
      e.push_params[other_kind]();
    const Engine::FilteredScan* lik = kind_ == Engine::LIKELIHOOD ? &mine : (both ? &oth : nullptr);
    const Engine::FilteredScan* beam = kind_ == Engine::BEAM ? &mine : (both ? &oth : nullptr);
    for (int k = 0; k < 2; ++k)
    {
      if (k != kind_ && !both)
        continue;
      Engine::Results& r = e.results[k];
      r.epoch = slot.epoch;
      r.cloud = e.filtered[k].cloud;
      r.likelihood.assign(slot.count, 0.f);
      r.quality.assign(slot.count, k == Engine::BEAM ? 1.f : 0.f);  // beam.cpp:154: quality 1
    }
    std::vector<float>& org = e.origin_scratch;
    org.resize(3 * origins.size());
    for (std::size_t i = 0; i < origins.size(); ++i)
    {
      org[3 * i + 0] = origins[i].x_;
      org[3 * i + 1] = origins[i].y_;
      org[3 * i + 2] = origins[i].z_;
    }
    const double t2 = Engine::nowUs();
    ++e.profile.launches;
    struct Stop
    {
      Engine& e;
      double t;
      ~Stop()
      {
        e.profile.batch_us += Engine::nowUs() - t;
      }
    } stop{ e, t2 };
    // results arrive in particle order while the later particles are still on the GPU: the reference's per-particle loop
    // (pf.h:255-260 with the lambda of src/mcl_3dl.cpp:399-426, ~80 ns of host work per particle) starts on the first slice
    e.check(mcl3dl_hip_group_measure_batch_begin(
        e.group(), nullptr, slot.count, lik ? lik->xyz.data() : nullptr, lik ? lik->xyz.size() / 3 : 0,
        beam ? beam->xyz.data() : nullptr, beam ? beam->label.data() : nullptr, beam ? beam->xyz.size() / 3 : 0,
        beam ? org.data() : nullptr, beam ? origins.size() : 0, lik ? e.results[Engine::LIKELIHOOD].likelihood.data() : nullptr,
        lik ? e.results[Engine::LIKELIHOOD].quality.data() : nullptr, beam ? e.results[Engine::BEAM].likelihood.data() : nullptr,
        0));
    e.batch_open = true;
    e.batch_ready = 0;
  }

  // blocks until the results of particle `index` have arrived (one comparison once its slice is in)
  static void awaitResult(const std::size_t index)
  {
    Engine& e = Engine::shared();
    if (e.batch_open && index >= e.batch_ready)
      e.waitBatch(index);
  }

  std::size_t points_default_ = 0, points_global_ = 0, points_now_ = 0;
  Clip clip_;
  std::function<void()> push_;
  Engine::Kind kind_ = Engine::LIKELIHOOD;
  bool registered_ = false;
  mutable Cloud::ConstPtr last_filtered_;
};
}  // namespace hip
}  // namespace mcl_3dl

#endif  // MCL_3DL_HIP_BATCHED_MODEL_HPP
