// update_kernels.h — the whole measurement update in ONE launch for the reference's own operating range (N_p <= a few
// thousand particles, N_s <= 2048 likelihood points, N_b <= 256 beam points; parameters.h:68,98 default to 64 x 96 + 3):
// there the update is launch-bound — three to five launches of ~4 us around ~10 us of work.
//
//   work-group p  = particle p: likelihood score (lik_particle — the code of likelihood_kernel), beam score (cast_ray —
//                   the code of beam_kernel; penalty count -> power table -> clamp), results written to out_lik / _ratio / _beam
//   256 particles = one "virtual block" of pf::measure's split form (pf_partial_kernel's 256-thread blocks): the LAST of
//                   its work-groups to finish (an arrival ticket) computes that block's partial {sum w, sum w ln w, max
//                   ratio, -min ratio} with the split form's association
//   the LAST virtual block to finish runs pf_reduce_kernel's 64-lane reduction and pf_apply_kernel's normalisation
//
// Nobody waits for anybody: a ticket is an atomic counter, the work-group that draws the last number does the next stage.
// Arrivals are counted through an 8-ary TREE of tickets: atomics on one address from many CUs are served one after the
// other by the memory side (~0.25 us each, measured: 256 arrivals on one counter cost 60 us), eight per node cost 2 us a level.
// Hand-offs between work-groups follow the guide's fence-free form for inter-workgroup visibility on gfx950: every payload
// word is written with an agent-scope (sc1, write-through) store and read with an agent-scope (sc1, past the L1) load, the
// stores are drained (s_waitcnt vmcnt(0)) before the ticket is drawn. (A release fence per work-group — buffer_wbl2, a
// write-back of the XCD's L2 — was measured first: 4096 work-groups x 96 points went from 20 to 110 us.) Same arithmetic, same association as likelihood_kernel + beam_kernel (+ finalize) +
// pf_partial / reduce / apply: bit-identical results (tests/test_gpu_update_small.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "beam_kernels.h"
#include "likelihood_kernels.h"
#include "pf_kernels.h"

#pragma clang fp contract(off)

namespace mcl3dl
{
struct UpdateSmallArgs
{
  const float* pose7;
  int n_p;
  // likelihood-field model
  const float4* scan_lik;
  int n_s;
  LikGrid g;
  RecGrid rg;
  LikParams prm;
  int coop;
  // beam model (n_b == 0: score 1)
  const float4* scan_beam;
  int n_b;
  const float4* origins;
  DdaGrid dg;
  BeamParams bp;
  const float* pow_table;
  // pf::measure
  float* w;            // in / out
  const float* extra;  // may be null
  int use_beam;        // 1: the weight product includes the beam score (pf_partial_kernel's `beam` pointer non-null)
  float* out_lik;
  float* out_ratio;
  float* out_beam;
  float* w_new;
  double* vb_partials;  // [4 x number of virtual blocks]
  unsigned* tickets;    // [37 per virtual block + ticket_tree_size(number of virtual blocks)], zero between launches
  double* packed;       // [4]
  float* stats4;
  PfEmit emit;          // page-locked host copies of the results (pf_kernels.h), each may be null
  int conformant;       // 1: every arrival is an acq_rel read-modify-write at agent scope (see last_arrival)
  const uint32_t* perm; // device scan index -> index in the caller's array: the likelihood terms are added in THAT order, as
                        // floats (lik_particle's row in dynamic LDS); null = fixed-order fp64 tree
  int float_order_w;    // 1: pf::measure's `sum += p.probability_` (pf.h:255-260) as the float recurrence over the particles
};

__device__ __forceinline__ void store_agent(float* p, float v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float load_agent(const float* p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_agent(double* p, double v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_agent(const double* p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// thread 0 draws a ticket once every wavefront's stores are drained; every thread learns whether it was the last of `expected`.
// Two forms of the hand-off:
//   default     what the guide gives as the fence-free form on gfx950: the payload was written with agent-scope (sc1,
//               write-through) stores, every wavefront waits for its own stores (s_waitcnt vmcnt(0)), then a RELAXED ticket
//               increment; the last arrival reads the payload with agent-scope (sc1) loads. Correct on this ISA, but the
//               ordering between payload and ticket rests on the waitcnt, not on the language's memory model.
//   conformant  (option "update_small_conformant") the ticket increment is an ACQ_REL read-modify-write at agent scope,
//               issued by thread 0 behind a work-group barrier: the barrier orders every thread's payload stores before it
//               (release is cumulative), the last arrival's increment acquires every earlier one. This is the form the
//               HIP / LLVM memory model defines; it costs a write-back of the XCD's L2 per arrival (measured in round 3:
//               4096 work-groups x 96 points 20 -> 110 us), which is why it is a switch for a field problem, not the default.
//               tests/test_gpu_soak.py runs both.
__device__ __forceinline__ bool last_arrival(unsigned* ticket, unsigned expected, int conformant)
{
  __shared__ int s_last;
  if (conformant)
  {
    __syncthreads();
    if (threadIdx.x == 0)
    {
      const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      s_last = (t + 1u == expected) ? 1 : 0;
    }
    __syncthreads();
    return s_last != 0;
  }
  // every wavefront drains its own (write-through) stores before the barrier — a barrier does not wait for memory
  // operations in flight — so that what lane 0 releases below includes them
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // (also: a previous call's s_last has been read)
  if (threadIdx.x == 0)
  {
    const unsigned t = atomicAdd(ticket, 1u);
    s_last = (t + 1u == expected) ? 1 : 0;
  }
  __syncthreads();
  return s_last != 0;
}

constexpr int US_FAN = 8;
// counters one tree over `width` leaves needs (levels of ceil(width / 8) nodes down to one)
__host__ __device__ inline int ticket_tree_size(int width)
{
  int n = 0;
  while (width > 1)
  {
    width = (width + US_FAN - 1) / US_FAN;
    n += width;
  }
  return n;
}

// true for exactly one work-group among the `width` that call this with their index and the same tree: the one whose
// arrival completes the root. A work-group leaves as soon as it is not the last arrival at a node.
__device__ __forceinline__ bool last_of_tree(unsigned* tree, int idx, int width, int conformant)
{
  while (width > 1)
  {
    const int node = idx / US_FAN, nodes = (width + US_FAN - 1) / US_FAN;
    const int children = min(US_FAN, width - node * US_FAN);
    if (!last_arrival(tree + node, static_cast<unsigned>(children), conformant))
      return false;
    tree += nodes;
    idx = node;
    width = nodes;
  }
  return true;
}

template <int BLOCK, int MODE>
__global__ __launch_bounds__(BLOCK) void update_small_kernel(UpdateSmallArgs a)
{
  constexpr int NW = BLOCK / 64;
  const int p = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* ps = a.pose7 + 7 * static_cast<size_t>(p);
  const Vec3f pos = { ps[0], ps[1], ps[2] };
  const Quat raw = { ps[3], ps[4], ps[5], ps[6] };
  const Quat rot = qnormalized(raw);  // state_6dof.h:217
  // ---- likelihood-field model (likelihood.cpp:105-139; empty scan -> (1, 0), :111-114)
  float lik = 1.0f, ratio = 0.0f;
  if (a.n_s > 0)
  {
    double sum = 0.0;
    unsigned num = 0, unused = 0;
    lik_particle<BLOCK, MODE, false>(pos, rot, a.scan_lik, a.n_s, a.g, a.rg, a.prm, a.coop, sum, num, unused, a.perm,
                                     a.perm ? dyn_row : nullptr);
    lik = static_cast<float>(sum);
    ratio = static_cast<float>(num) / static_cast<float>(a.n_s);  // :136
  }
  // ---- beam model (beam.cpp:124-155): one lane per ray, the non-prepared form of beam_kernel
  float beam = 1.0f;
  if (a.n_b > 0)
  {
    __shared__ unsigned s_pen;
    if (threadIdx.x == 0)
      s_pen = 0u;
    __syncthreads();
    for (int base = 0; base < a.n_b; base += BLOCK)
    {
      const int i = base + static_cast<int>(threadIdx.x);
      bool penalised = false;
      if (i < a.n_b)
      {
        const float4 v = a.scan_beam[i];
        const Vec3f end = vadd(qrot(rot, Vec3f{ v.x, v.y, v.z }), pos);  // beam.cpp:139 (transform)
        const float4 og = a.origins[__float_as_uint(v.w)];
        const Vec3f begin = vadd(pos, qrot(raw, Vec3f{ og.x, og.y, og.z }));  // beam.cpp:145: s.pos_ + s.rot_ * origin
        int hit;
        unsigned s0 = 0, s1 = 0, s2 = 0;
        const int status = cast_ray<false, false, false>(a.dg, a.bp, begin, end, &hit, s0, s1, s2);
        penalised = (status == 0) || (!a.bp.short_only && (status == 2));  // beam.cpp:146
      }
      const unsigned long long m = __ballot(penalised);
      if (m != 0ull && lane == 0)
        atomicAdd(&s_pen, static_cast<unsigned>(__popcll(m)));
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
      float s = a.pow_table[s_pen];  // beam_likelihood_^k by k float multiplications (beam.cpp:148)
      if (s < a.bp.beam_likelihood_min)
        s = a.bp.beam_likelihood_min;  // :151-152
      beam = s;
    }
  }
  if (threadIdx.x == 0)
  {
    store_agent(a.out_lik + p, lik);
    store_agent(a.out_ratio + p, ratio);
    store_agent(a.out_beam + p, beam);
    if (a.emit.lik)
      a.emit.lik[p] = lik;
    if (a.emit.ratio)
      a.emit.ratio[p] = ratio;
    if (a.emit.beam)
      a.emit.beam[p] = beam;
  }
  // ---- pf::measure (pf.h:252-279). Stage 1: the last work-group of each 256-particle virtual block.
  const int nvb = (a.n_p + PF_BLOCK - 1) / PF_BLOCK;
  const int vb = p / PF_BLOCK;
  const int in_vb = min(PF_BLOCK, a.n_p - vb * PF_BLOCK);
  constexpr int VB_TREE = 32 + 4 + 1;  // ticket_tree_size(256)
  if (!last_of_tree(a.tickets + vb * VB_TREE, p - vb * PF_BLOCK, in_vb, a.conformant))
    return;
  __shared__ double sh[4][4];
  for (int chunk = wave; chunk < 4; chunk += NW)  // pf_partial_kernel: wavefront `chunk` of the block, one element per lane
  {
    double s = 0.0, t = 0.0, rmax = 0.0, rneg = -1.0;  // match_ratio_max = 0, match_ratio_min = 1 (mcl_3dl.cpp:398-399)
    const int i = vb * PF_BLOCK + chunk * 64 + lane;
    if (i < a.n_p)
    {
      float l = 1.0f;
      if (a.use_beam)
        l *= load_agent(a.out_beam + i);
      l *= load_agent(a.out_lik + i);
      if (a.extra)
        l = l * a.extra[i];
      const float wn = a.w[i] * l;  // pf.h:258
      store_agent(a.w_new + i, wn);
      s += static_cast<double>(wn);
      if (wn > 0.0f)
        t += static_cast<double>(wn) * log(static_cast<double>(wn));
      const double r = static_cast<double>(load_agent(a.out_ratio + i));
      rmax = r > rmax ? r : rmax;
      rneg = -r > rneg ? -r : rneg;
    }
    s = wave_sum(s);
    t = wave_sum(t);
    rmax = wave_max(rmax);
    rneg = wave_max(rneg);
    if (lane == 0)
    {
      sh[0][chunk] = s;
      sh[1][chunk] = t;
      sh[2][chunk] = rmax;
      sh[3][chunk] = rneg;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    double x = 0, y = 0, c = sh[2][0], d = sh[3][0];
    for (int k = 0; k < 4; ++k)
    {
      x += sh[0][k];
      y += sh[1][k];
      c = sh[2][k] > c ? sh[2][k] : c;
      d = sh[3][k] > d ? sh[3][k] : d;
    }
    store_agent(a.vb_partials + 4 * vb + 0, x);
    store_agent(a.vb_partials + 4 * vb + 1, y);
    store_agent(a.vb_partials + 4 * vb + 2, c);
    store_agent(a.vb_partials + 4 * vb + 3, d);
  }
  // ---- Stage 2: the last virtual block to finish
  if (!last_of_tree(a.tickets + nvb * VB_TREE, vb, nvb, a.conformant))
    return;
  __shared__ double tot[4];
  if (wave == 0)
  {
    // pf_reduce_kernel: 64 lanes stride the block partials, wavefront reduction
    double x = 0, y = 0, c = 0.0, d = -1.0;
    for (int k = lane; k < nvb; k += 64)
    {
      x += load_agent(a.vb_partials + 4 * k + 0);
      y += load_agent(a.vb_partials + 4 * k + 1);
      const double ck = load_agent(a.vb_partials + 4 * k + 2), dk = load_agent(a.vb_partials + 4 * k + 3);
      c = ck > c ? ck : c;
      d = dk > d ? dk : d;
    }
    x = wave_sum(x);
    y = wave_sum(y);
    c = wave_max(c);
    d = wave_max(d);
    if (lane == 0)
    {
      tot[0] = x;
      tot[1] = y;
      tot[2] = c;
      tot[3] = d;
      a.packed[0] = x;
      a.packed[1] = y;
      a.packed[2] = c;
      a.packed[3] = d;
    }
  }
  __syncthreads();
  if (a.float_order_w)
  {
    // the reference's own sum (pf.h:255-260): float, sequentially, in particle order — the un-normalised weights staged in
    // the dynamic LDS row (free again: every particle's likelihood is done), the first wavefront runs the recurrence
    // (float_chain.h). The weights are then divided by exactly the reference's float.
    const int n4 = chain_row_floats(a.n_p);
    for (int i = threadIdx.x; i < n4; i += BLOCK)
      dyn_row[i] = i < a.n_p ? load_agent(a.w_new + i) : 0.0f;
    __syncthreads();
    if (wave == 0)
    {
      const float sum_w = seq_sum_wave(dyn_row, a.n_p, lane);
      if (lane == 0)
      {
        tot[0] = static_cast<double>(sum_w);
        a.packed[0] = static_cast<double>(sum_w);
      }
    }
    __syncthreads();
  }
  // pf_apply_kernel
  const double S = tot[0];
  const float sum_f = static_cast<float>(S);
  const bool alive = sum_f > 0.0f;
  if (alive)
    for (int i = threadIdx.x; i < a.n_p; i += BLOCK)
    {
      const float wv = load_agent(a.w_new + i) / sum_f;
      a.w[i] = wv;
      if (a.emit.w)
        a.emit.w[i] = wv;
    }
  else if (a.emit.w)
    for (int i = threadIdx.x; i < a.n_p; i += BLOCK)
      a.emit.w[i] = a.w[i];
  if (threadIdx.x == 0)
  {
    const float st[4] = { alive ? static_cast<float>(log(S) - tot[1] / S) : __builtin_nanf(""), static_cast<float>(-tot[3]),
                          static_cast<float>(tot[2]), alive ? 0.0f : 1.0f };
    for (int k = 0; k < 4; ++k)
    {
      if (a.stats4)
        a.stats4[k] = st[k];
      if (a.emit.stats4)
        a.emit.stats4[k] = st[k];
    }
  }
  for (int k = threadIdx.x; k < nvb * VB_TREE + ticket_tree_size(nvb); k += BLOCK)
    a.tickets[k] = 0u;  // ready for the next launch
}
// ---------------------------------------------------------------------------------------------------------------------------------
// Both models of a large update in ONE launch (round 6): the tiled likelihood kernel's work-groups and the beam kernel's,
// INTERLEAVED over the block index — rounds of `beam8` x 8 beam work-groups followed by `tiled8` x 8 tiled ones, the two counts in the
// ratio of the two kernels' grids, so both run out together and every CU hosts wavefronts of both all the way: the beam model's
// traversal waits on dependent loads (VALU issue 0.52 alone), the tiled kernel on the L2's request rate (0.61 alone), and what one
// leaves idle the other uses. (Two streams do not give this: the second kernel's work-groups only get the slots the first one's
// free at its end.) Rounds are multiples of eight blocks, so a tiled work-group keeps the XCD its index is congruent to
// (likelihood_tiled_body's tile -> XCD mapping). Each work-group runs exactly the code of its own kernel: same bits.
struct LikBeamArgs
{
  // tiled likelihood kernel (likelihood_tiled_body<16, 2, 8, true, DEFER, false>)
  const float* pose7;
  int n_p;
  const float4* scan;
  int n_s, n_tiles, n_groups;
  LikGrid g;
  RecGrid rg;
  LikParams prm;
  double* partial_sum;
  unsigned* partial_cnt;
  const uint32_t* scan_perm;
  float* strict_terms;
  int strict_skew4;
  // beam kernel (beam_body<false, OVERLAY>)
  const float4* scan_beam;
  int n_b;
  const float4* origins;
  long long n_rays;
  DdaGrid dg;
  BeamParams bp;
  unsigned* penalty;
  const BeamOrigin* prepared;
  int n_o;
  // the interleave
  uint32_t beam8, tiled8;
  uint32_t n_beam_blocks, n_tiled_blocks;
  // CHAIN: the float recurrence inside the tiled kernel (strict_order 3); a work-group's producer keeps its lower block index
  LikChain ch;
};

template <int G, bool DEFER, bool OVERLAY, bool CHAIN = false>
__global__ __launch_bounds__(256, 8) void lik_beam_kernel(LikBeamArgs a)
{
  const uint32_t nb8 = 8u * a.beam8, round = nb8 + 8u * a.tiled8;
  const uint32_t k = blockIdx.x / round, r = blockIdx.x - k * round;
  if (r < nb8)
  {
    const uint32_t bi = k * nb8 + r;
    if (bi < a.n_beam_blocks)
      beam_body<false, OVERLAY>(static_cast<long long>(bi), a.pose7, a.scan_beam, a.n_b, a.origins, a.n_rays, a.dg, a.bp, a.penalty,
                                static_cast<RayStats*>(nullptr), a.prepared, a.n_o);
  }
  else
  {
    const uint32_t ti = k * 8u * a.tiled8 + (r - nb8);
    if (ti < a.n_tiled_blocks)
      likelihood_tiled_body<G, 2, 8, true, DEFER, CHAIN>(ti, a.pose7, a.n_p, a.scan, a.n_s, a.n_tiles, a.n_groups, a.g, a.rg, a.prm,
                                                         a.partial_sum, a.partial_cnt, a.scan_perm, a.strict_terms, a.strict_skew4,
                                                         a.ch);
  }
}

// The same interleave for the PER-PARTICLE likelihood kernel (likelihood_kernel<256, 2>: one work-group per particle, scans of 129
// points up to where the tiled kernel takes over, and every scan below 2048 particles in the default mode — the caller-order
// rows) and the beam kernel: both are bound by dependent loads at a fraction of the chip's wavefront slots, so run side by side
// they take about what the longer one takes alone. Dynamic LDS: the caller-order term row of lik_particle (the beam work-groups
// carry it unused).
struct LikParticleBeamArgs
{
  const float* pose7;
  int n_p;
  const float4* scan;
  int n_s;
  LikGrid g;
  RecGrid rg;
  LikParams prm;
  float* out_lik;
  float* out_ratio;
  int coop;
  const uint32_t* perm;
  const float4* scan_beam;
  int n_b;
  const float4* origins;
  long long n_rays;
  DdaGrid dg;
  BeamParams bp;
  unsigned* penalty;
  const BeamOrigin* prepared;
  int n_o;
  uint32_t beam8, lik8;  // per round: beam8 x 8 beam work-groups, then lik8 x 8 particles
  uint32_t n_beam_blocks;
};

// BLOCK = the likelihood kernel's work-group size (launch_measure: 64 threads up to 128 points, 256 beyond) = rays per beam work-group
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void lik_particle_beam_kernel(LikParticleBeamArgs a)
{
  const uint32_t nb8 = 8u * a.beam8, round = nb8 + 8u * a.lik8;
  const uint32_t k = blockIdx.x / round, r = blockIdx.x - k * round;
  if (r < nb8)
  {
    const uint32_t bi = k * nb8 + r;
    if (bi < a.n_beam_blocks)
      beam_body<false, false, BLOCK>(static_cast<long long>(bi), a.pose7, a.scan_beam, a.n_b, a.origins, a.n_rays, a.dg, a.bp, a.penalty,
                                     static_cast<RayStats*>(nullptr), a.prepared, a.n_o);
  }
  else
  {
    const uint32_t p = k * 8u * a.lik8 + (r - nb8);
    if (p < static_cast<uint32_t>(a.n_p))
      likelihood_particle_body<BLOCK, 2, false>(static_cast<int>(p), a.pose7, a.scan, a.n_s, a.g, a.rg, a.prm, a.out_lik, a.out_ratio,
                                              static_cast<double*>(nullptr), a.coop, a.perm);
  }
}
}  // namespace mcl3dl
