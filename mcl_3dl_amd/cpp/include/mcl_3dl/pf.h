// mcl_3dl/pf.h (drop-in shim) — put mcl_3dl_amd/cpp/include AHEAD of the reference's include directory.
//
// The reference's pf::ParticleFilter is used untouched (it is pulled in with #include_next under a different class
// name); this shim only derives from it to hide measure() with a version that publishes the particle batch
// (first state, stride, count) around the reference's loop (include/mcl_3dl/pf.h:252-279).  The GPU-backed LiDAR models read
// that descriptor inside their per-particle measure() (SURVEY.md §8b "batch-prepass protocol"), so src/mcl_3dl.cpp compiles
// and behaves unchanged: weights are multiplied, summed and normalised in the reference's statements and float order; only
// the per-particle likelihoods now come from one GPU launch.
//
// measure() restates pf.h:252-279 line by line with ONE difference that cannot be observed: the backup the restore rule
// needs (`auto particles_prev = particles_`, :254 — 112 bytes per particle copied on every update) is reduced to the
// probabilities. The callback takes the state by const reference (:252), and nothing else of a particle is written by the
// loop, so restoring the probabilities restores the vector (:274-278).
#ifndef MCL_3DL_HIP_PF_SHIM_H
#define MCL_3DL_HIP_PF_SHIM_H

#define ParticleFilter ParticleFilterCpu
#include_next <mcl_3dl/pf.h>
#undef ParticleFilter

#include <cmath>
#include <functional>
#include <random>
#include <vector>

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
    auto& particles = this->particles_;
    prev_probability_.resize(particles.size());
    for (std::size_t i = 0; i < particles.size(); ++i)
      prev_probability_[i] = particles[i].probability_;  // pf.h:254 (what the loop below can change of it)
    FLT_TYPE sum = 0;
    for (auto& p : particles)
    {
      p.probability_ *= likelihood(p.state_);  // pf.h:258
      sum += p.probability_;
    }
    if (sum > 0.0)
    {
      this->entropy_ = 0;
      for (auto& p : particles)
      {
        p.probability_ /= sum;
        if (p.probability_ > 0)
        {
          this->entropy_ += p.probability_ * std::log(p.probability_);
        }
      }
      this->entropy_ *= -1;
    }
    else
    {
      for (std::size_t i = 0; i < particles.size(); ++i)
        particles[i].probability_ = prev_probability_[i];  // pf.h:276
    }
  }

private:
  std::vector<FLT_TYPE> prev_probability_;
};
}  // namespace pf
}  // namespace mcl_3dl

#endif  // MCL_3DL_HIP_PF_SHIM_H
