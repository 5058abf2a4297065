// mcl_3dl_hip/engine.hpp — C++ host layer over the C ABI (include/mcl3dl_hip.h) shared by the two drop-in LiDAR
// model classes.  Header-only except for the context singleton in src/engine.cpp.
//
//  * Engine            RAII handle on one mcl3dl_hip_group (1..N GPUs, particles sharded over them inside the library);
//                      C-ABI errors become std::runtime_error.
//  * BatchDescriptor   what the batch-aware pf::ParticleFilter::measure publishes before running the reference's
//                      per-particle loop (include/mcl_3dl/pf.h:252-260), so a model's per-particle measure() can
//                      evaluate ALL particles on the GPU at the first call and answer the rest from the result buffer.
#ifndef MCL_3DL_HIP_ENGINE_HPP
#define MCL_3DL_HIP_ENGINE_HPP

#include <chrono>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include <mcl3dl_hip.h>

namespace mcl_3dl
{
namespace hip
{
class Engine
{
public:
  // One device group (include/mcl3dl_hip.h "device groups"): `devices` lists the GPUs the particles are sharded over.
  explicit Engine(const std::vector<int>& devices = { 0 })
  {
    const int rc = mcl3dl_hip_group_create(&group_, devices.data(), static_cast<int>(devices.size()));
    if (rc != 0 || !group_)
      throw std::runtime_error("mcl3dl_hip_group_create failed (" + std::to_string(rc) +
                               "): no usable gfx950 device; this engine has no CPU fallback");
  }
  ~Engine()
  {
    mcl3dl_hip_group_destroy(group_);
  }
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;

  mcl3dl_hip_group* group() const
  {
    return group_;
  }
  // the first device's context: single-ray queries (getBeamStatus) run there
  mcl3dl_hip_ctx* get() const
  {
    return mcl3dl_hip_group_context(group_, 0);
  }
  int size() const
  {
    return mcl3dl_hip_group_size(group_);
  }
  // C-ABI errors become std::runtime_error (the reference's models only ever throw std::runtime_error,
  // chunked_kdtree.h:224-225)
  void check(const int rc) const
  {
    if (rc != 0)
      throw std::runtime_error(std::string("mcl3dl_hip: ") + mcl3dl_hip_group_last_error(group_));
  }
  void checkContext(const int rc) const
  {
    if (rc != 0)
      throw std::runtime_error(std::string("mcl3dl_hip: ") + mcl3dl_hip_last_error(get()));
  }

  // process-wide context used by the drop-in model classes (the node builds one likelihood and one beam model that
  // share the map: src/mcl_3dl.cpp:1315-1318)
  static Engine& shared();
  // the same engine if shared() has built it already, otherwise nullptr (never creates one)
  static Engine*& live()
  {
    static Engine* e = nullptr;
    return e;
  }

  // MCL3DL_HIP_ENGINE_ORDER=1 (src/engine.cpp): sampled likelihood clouds leave filter() in the engine's scan order
  bool engine_order = false;

  // ---- the batch in flight (mcl3dl_hip_group_measure_batch_begin): results arrive in particle order while the GPU works on
  // the later particles; batch_ready = number of leading particles whose results are in `results`
  bool batch_open = false;
  std::size_t batch_ready = 0;
  void waitBatch(const std::size_t index)
  {
    std::size_t n = 0;
    const double t0 = nowUs();
    check(mcl3dl_hip_group_measure_batch_wait(group_, index, &n));
    profile.wait_us += nowUs() - t0;
    batch_ready = n;
  }
  void endBatch()
  {
    if (!batch_open)
      return;
    batch_open = false;
    const double t0 = nowUs();
    check(mcl3dl_hip_group_measure_batch_end(group_));
    profile.wait_us += nowUs() - t0;
  }

  // ---- map bookkeeping shared by both models: upload when the cloud or its stamp changes ------------------------
  const void* map_cloud = nullptr;
  std::uint64_t map_stamp = 0;
  std::size_t map_size = 0;
  float dist_weight[3] = { 1.f, 1.f, 1.f };
  bool has_weight = false;
  float map_min[3] = { 0.f, 0.f, 0.f };  // pcl::getMinMax3D's minimum of the map (RaycastUsingDDA::min_p_): voxel centres
  // ---- poses of the current pf::measure epoch, uploaded once and shared by both models ---------------------------
  std::uint64_t pose_epoch = 0;
  std::size_t pose_count = 0;
  std::vector<float> pose_scratch;
  // ---- one launch for BOTH models. The node filters every model's cloud first (src/mcl_3dl.cpp:378-383) and only then
  // runs pf_->measure, whose lambda asks "beam", then "likelihood", for every particle (:409). filter() therefore leaves
  // its result here, packed; the first measure() of an epoch — whichever model it reaches — evaluates both scans in one
  // batched call and both models answer their remaining calls from `results`.
  enum Kind
  {
    LIKELIHOOD = 0,
    BEAM = 1
  };
  struct FilteredScan
  {
    const void* cloud = nullptr;  // identity of the cloud filter() returned (the model keeps it alive: no address reuse)
    bool from_filter = false;     // false: packed by measure() itself for a cloud filter() has not seen
    std::vector<float> xyz;
    std::vector<std::uint32_t> label;
  };
  struct Results
  {
    std::uint64_t epoch = 0;
    const void* cloud = nullptr;
    std::vector<float> likelihood, quality;
  };
  FilteredScan filtered[2];
  Results results[2];
  std::function<void()> push_params[2];  // each model's parameters -> the engine (set by the model, cleared when it dies)
  // which model instance the slots of a kind (push_params, filtered, results) belong to: the one that configured itself last.
  // Another live instance of the same kind takes them over again on its next use (and the result cache of the kind is
  // invalidated on every change of owner); a dying instance clears them only while they are still its own.
  const void* owner[2] = { nullptr, nullptr };
  std::vector<float> origin_scratch;
  // ---- where an update through the per-particle virtuals spends its host time (microseconds, accumulated; read and reset
  // by whoever wants a breakdown: tests/cpp/adapter_demo.cpp prints it next to the total)
  struct Profile
  {
    double pack_us = 0, poses_us = 0, batch_us = 0, wait_us = 0;
    std::uint64_t launches = 0;
  } profile;
  static double nowUs()
  {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }

private:
  mcl3dl_hip_group* group_ = nullptr;
};

// Published by pf::ParticleFilter::measure (mcl_3dl/pf.h in this directory tree) for the duration of one update.
struct BatchDescriptor
{
  const void* first_state = nullptr;  // &particles_[0].state_
  std::size_t stride = 0;             // sizeof(Particle<State6DOF, float>)
  std::size_t count = 0;
  std::uint64_t epoch = 0;  // changes every measure() call: invalidates the models' result caches
};

inline BatchDescriptor& currentBatch()
{
  static thread_local BatchDescriptor d;
  return d;
}

class BatchScope
{
public:
  BatchScope(const void* first_state, const std::size_t stride, const std::size_t count)
  {
    static thread_local std::uint64_t counter = 0;
    BatchDescriptor& d = currentBatch();
    saved_ = d;
    d.first_state = first_state;
    d.stride = stride;
    d.count = count;
    d.epoch = ++counter;
  }
  ~BatchScope()
  {
    currentBatch() = saved_;
    // nothing of the batch stays in flight once pf::measure returns (also when the measure lambda threw half-way)
    if (Engine* e = Engine::live())
    {
      try
      {
        e->endBatch();
      }
      catch (const std::exception&)
      {
      }
    }
  }

private:
  BatchDescriptor saved_;
};
}  // namespace hip
}  // namespace mcl_3dl

#endif  // MCL_3DL_HIP_ENGINE_HPP
