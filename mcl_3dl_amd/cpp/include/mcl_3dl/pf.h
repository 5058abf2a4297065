// mcl_3dl/pf.h (drop-in shim) — put mcl_3dl_amd/cpp/include AHEAD of the reference's include directory.
//
// The reference's pf::ParticleFilter is used untouched (it is pulled in with #include_next under a different class
// name); this shim only derives from it to hide measure() with a version that publishes the particle batch
// (first state, stride, count) before running the reference's own loop (include/mcl_3dl/pf.h:252-279).  The GPU-backed
// LiDAR models read that descriptor inside their per-particle measure() (SURVEY.md §8b "batch-prepass protocol"), so
// src/mcl_3dl.cpp compiles and behaves unchanged: weights are still multiplied, summed and normalised by the reference's
// code, in the reference's float order; only the per-particle likelihoods now come from one GPU launch per model.
#ifndef MCL_3DL_HIP_PF_SHIM_H
#define MCL_3DL_HIP_PF_SHIM_H

#define ParticleFilter ParticleFilterCpu
#include_next <mcl_3dl/pf.h>
#undef ParticleFilter

#include <functional>
#include <random>

#include <mcl_3dl_hip/engine.hpp>

namespace mcl_3dl
{
namespace pf
{
template <typename T, typename FLT_TYPE = float, typename MEAN = ParticleWeightedMean<T, FLT_TYPE>,
          typename RANDOM_ENGINE = std::default_random_engine>
class ParticleFilter : public ParticleFilterCpu<T, FLT_TYPE, MEAN, RANDOM_ENGINE>
{
  using Base = ParticleFilterCpu<T, FLT_TYPE, MEAN, RANDOM_ENGINE>;

public:
  using Base::Base;

  void measure(std::function<FLT_TYPE(const T&)> likelihood)
  {
    if (this->particles_.empty())
    {
      Base::measure(likelihood);
      return;
    }
    const mcl_3dl::hip::BatchScope scope(&this->particles_[0].state_, sizeof(this->particles_[0]),
                                         this->particles_.size());
    Base::measure(likelihood);
  }
};
}  // namespace pf
}  // namespace mcl_3dl

#endif  // MCL_3DL_HIP_PF_SHIM_H
