// pf_kernels.h — pf::ParticleFilter::measure (R1/R2) and the "next" rows: expectation / max / covariance, resampling.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.h"
#include "float_chain.h"

#pragma clang fp contract(off)

namespace mcl3dl
{
// ---------------------------------------------------------------------------------------------------------
// pf::ParticleFilter::measure, include/mcl_3dl/pf.h:252-279  (+ the lambda's product, src/mcl_3dl.cpp:407-424)
// ---------------------------------------------------------------------------------------------------------
constexpr int PF_BLOCK = 256;

// The results of an update written a second time, into page-locked (device-mapped) host memory, by the kernel that produces
// the final weights — a host-buffer update then ends without a D2H copy (SURVEY.md 8d's region: "D2H of weights"). Any
// pointer may be null.
struct PfEmit
{
  float* stats4;
  float* w;
  float* lik;
  float* ratio;
  float* beam;
};

// The beam model's last step (beam_kernels.h: beam_finalize_kernel — score = table[penalised rays], clamped from below, beam.cpp:
// 146-152) for the kernel that needs the score next, instead of a launch of its own in front of it (round 6): penalty != null —
// the score is formed from the count, written to beam_out, and the counter is ZEROED behind the read (the next update's beam kernel
// counts from 0 without a launch that clears: mcl3dl_hip_ctx::penalty_clean_n).
struct BeamCounts
{
  unsigned* penalty;     // [n] penalised rays per particle
  const float* pow_table;
  float beam_likelihood_min;
  float* beam_out;       // [n]
};
__device__ inline float beam_score_from_count(const BeamCounts& bc, int i)
{
  float sb = bc.pow_table[bc.penalty[i]];
  bc.penalty[i] = 0u;
  if (sb < bc.beam_likelihood_min)
    sb = bc.beam_likelihood_min;
  bc.beam_out[i] = sb;
  return sb;
}

// w_new = w * (((1 * beam) * lik) * extra); per-block partials {sum w, sum w ln w, max ratio, -min ratio}.
__global__ __launch_bounds__(PF_BLOCK) void pf_partial_kernel(const float* __restrict__ w, const float* __restrict__ lik,
                                                              const float* __restrict__ beam,
                                                              const float* __restrict__ extra,
                                                              const float* __restrict__ ratio, int n,
                                                              float* __restrict__ w_new,
                                                              double* __restrict__ block_partials, BeamCounts bc = BeamCounts{})
{
  double s = 0.0, t = 0.0, rmax = 0.0, rneg = -1.0;  // match_ratio_max = 0, match_ratio_min = 1 (mcl_3dl.cpp:398-399)
  for (int i = blockIdx.x * PF_BLOCK + threadIdx.x; i < n; i += gridDim.x * PF_BLOCK)
  {
    float l = 1.0f;
    if (bc.penalty)
      l *= beam_score_from_count(bc, i);
    else if (beam)
      l *= beam[i];
    l *= lik[i];
    if (extra)
      l = l * extra[i];
    const float wn = w[i] * l;  // pf.h:258
    w_new[i] = wn;
    s += static_cast<double>(wn);
    if (wn > 0.0f)
      t += static_cast<double>(wn) * log(static_cast<double>(wn));
    if (ratio)
    {
      const double r = static_cast<double>(ratio[i]);
      rmax = r > rmax ? r : rmax;
      rneg = -r > rneg ? -r : rneg;
    }
  }
  __shared__ double sh[4][PF_BLOCK / 64];
  s = wave_sum(s);
  t = wave_sum(t);
  rmax = wave_max(rmax);
  rneg = wave_max(rneg);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
  {
    sh[0][wave] = s;
    sh[1][wave] = t;
    sh[2][wave] = rmax;
    sh[3][wave] = rneg;
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    double a = 0, b = 0, c = sh[2][0], d = sh[3][0];
    for (int k = 0; k < PF_BLOCK / 64; ++k)
    {
      a += sh[0][k];
      b += sh[1][k];
      c = sh[2][k] > c ? sh[2][k] : c;
      d = sh[3][k] > d ? sh[3][k] : d;
    }
    block_partials[4 * blockIdx.x + 0] = a;
    block_partials[4 * blockIdx.x + 1] = b;
    block_partials[4 * blockIdx.x + 2] = c;
    block_partials[4 * blockIdx.x + 3] = d;
  }
}

// lik_finalize_kernel + pf_partial_kernel as ONE launch (round 6; one GPU, the tiled likelihood kernel in front, at most
// 1024 x 256 particles — one per thread of pf_partial_kernel's grid). A work-group owns 64 consecutive particles (one wavefront
// of pf_partial_kernel's 256-thread blocks) with lik_finalize_kernel's own parallelism: its four wavefronts add up the eight
// tile slices of the per-tile partials (wavefront v: slices v and v + 4, every eighth tile in increasing order), the first one
// combines them in order — lik_finalize_kernel's association — and goes on with pf_partial_kernel's product, fp64 terms and
// wavefront reductions. What leaves is the WAVEFRONT partial {sum w, sum w ln w, max ratio, -min ratio} of the 64 particles;
// pf_reduce_kernel (waves = 1) forms pf_partial_kernel's block partials from four of them in that kernel's order (0 + w0 + w1 +
// w2 + w3; a missing wavefront counts {0, 0, 0, -1}, what a wavefront without particles reduces to). Same arithmetic in the same
// association as the two launches: the same bits (tests/test_gpu_pf_fused.py).
struct LikTiles
{
  const double* psum;    // [n_tiles][n]
  const unsigned* pcnt;  // [n_tiles][n]
  int n_tiles, n_s;
  float* lik_out;        // [n]
  float* ratio_out;      // [n]
  float* beam_fill;      // [n] set to 1 (an update without beam points), or null
  BeamCounts bc;         // the beam model's last step as well (penalty != null)
};
__global__ __launch_bounds__(256) void lik_pf_partial_kernel(LikTiles lt, const float* __restrict__ w,
                                                             const float* __restrict__ beam, const float* __restrict__ extra,
                                                             int n, float* __restrict__ w_new,
                                                             double* __restrict__ wave_partials)
{
  __shared__ double s_a[8][64];
  __shared__ unsigned s_n[8][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + lane;
  double a0 = 0.0, a1 = 0.0;
  unsigned n0 = 0, n1 = 0;
  if (p < n)
  {
    const double* ps = lt.psum + p;
    const unsigned* pc = lt.pcnt + p;
    int tl = wave;
    for (; tl + 4 < lt.n_tiles; tl += 8)  // slice `wave` and slice `wave + 4` side by side: two independent chains of loads
    {
      const double x0 = ps[static_cast<size_t>(tl) * n], x1 = ps[static_cast<size_t>(tl + 4) * n];
      const unsigned c0 = pc[static_cast<size_t>(tl) * n], c1 = pc[static_cast<size_t>(tl + 4) * n];
      a0 += x0;
      a1 += x1;
      n0 += c0;
      n1 += c1;
    }
    if (tl < lt.n_tiles)
    {
      a0 += ps[static_cast<size_t>(tl) * n];
      n0 += pc[static_cast<size_t>(tl) * n];
    }
  }
  s_a[wave][lane] = a0;
  s_a[wave + 4][lane] = a1;
  s_n[wave][lane] = n0;
  s_n[wave + 4][lane] = n1;
  __syncthreads();
  if (wave != 0)
    return;
  double s = 0.0, t = 0.0, rmax = 0.0, rneg = -1.0;
  if (p < n)
  {
    double a = a0;
    unsigned cnt = n0;
#pragma unroll
    for (int k = 1; k < 8; ++k)
    {
      a += s_a[k][lane];
      cnt += s_n[k][lane];
    }
    const float lk = static_cast<float>(a);
    const float r32 = static_cast<float>(cnt) / static_cast<float>(lt.n_s);
    lt.lik_out[p] = lk;
    lt.ratio_out[p] = r32;
    if (lt.beam_fill)
      lt.beam_fill[p] = 1.0f;
    float l = 1.0f;
    if (lt.bc.penalty)
      l *= beam_score_from_count(lt.bc, p);
    else if (beam)  // (1 * 1 for an update without beam points: what pf_partial_kernel reads back from the array filled with ones)
      l *= lt.beam_fill ? 1.0f : beam[p];
    l *= lk;
    if (extra)
      l = l * extra[p];
    const float wn = w[p] * l;  // pf.h:258
    w_new[p] = wn;
    s += static_cast<double>(wn);  // (0.0 + x, as pf_partial_kernel's loop forms it: the sign of a zero included)
    if (wn > 0.0f)
      t += static_cast<double>(wn) * log(static_cast<double>(wn));
    const double r = static_cast<double>(r32);
    rmax = r > rmax ? r : rmax;
    rneg = -r > rneg ? -r : rneg;
  }
  s = wave_sum(s);
  t = wave_sum(t);
  rmax = wave_max(rmax);
  rneg = wave_max(rneg);
  if (lane == 0)
  {
    wave_partials[4 * blockIdx.x + 0] = s;
    wave_partials[4 * blockIdx.x + 1] = t;
    wave_partials[4 * blockIdx.x + 2] = rmax;
    wave_partials[4 * blockIdx.x + 3] = rneg;
  }
}

// Fixed-order reduction of the block partials (deterministic run to run). The result is written in the layout the
// update's single all-reduce(SUM) needs (mcl_3dl_amd/distributed.py): [0] sum w, [1] sum w ln w, then per rank r the pair
// [2+2r] max ratio, [3+2r] -min ratio — this rank fills its own pair and zeroes the others, so that after the SUM every
// rank holds every rank's pair. world == 1 degenerates to the plain 4 doubles.
// n_waves > 0: `block_partials` holds lik_pf_partial_kernel's n_waves WAVEFRONT partials instead (block k = wavefronts 4 k .. 4 k + 3,
// added up here the way pf_partial_kernel's thread 0 does).
__device__ inline void pf_reduce_lanes(const double* __restrict__ block_partials, int n_blocks, int n_waves, int lane, double& a,
                                       double& b, double& c, double& d)
{
  a = 0;
  b = 0;
  c = 0.0;
  d = -1.0;
  if (n_waves > 0)
  {
    for (int k = lane; k < n_blocks; k += 64)
    {
      const double* wp = block_partials + 16 * static_cast<size_t>(k);
      const bool h1 = 4 * k + 1 < n_waves, h2 = 4 * k + 2 < n_waves, h3 = 4 * k + 3 < n_waves;
      double ba = 0, bb = 0, bc = wp[2], bd = wp[3];
      ba += wp[0];
      bb += wp[1];
      const double a1 = h1 ? wp[4] : 0.0, b1 = h1 ? wp[5] : 0.0, c1 = h1 ? wp[6] : 0.0, d1 = h1 ? wp[7] : -1.0;
      const double a2 = h2 ? wp[8] : 0.0, b2 = h2 ? wp[9] : 0.0, c2 = h2 ? wp[10] : 0.0, d2 = h2 ? wp[11] : -1.0;
      const double a3 = h3 ? wp[12] : 0.0, b3 = h3 ? wp[13] : 0.0, c3 = h3 ? wp[14] : 0.0, d3 = h3 ? wp[15] : -1.0;
      ba += a1;
      bb += b1;
      bc = c1 > bc ? c1 : bc;
      bd = d1 > bd ? d1 : bd;
      ba += a2;
      bb += b2;
      bc = c2 > bc ? c2 : bc;
      bd = d2 > bd ? d2 : bd;
      ba += a3;
      bb += b3;
      bc = c3 > bc ? c3 : bc;
      bd = d3 > bd ? d3 : bd;
      a += ba;
      b += bb;
      c = bc > c ? bc : c;
      d = bd > d ? bd : d;
    }
  }
  else
  for (int k = lane; k < n_blocks; k += 64)
  {
    a += block_partials[4 * k + 0];
    b += block_partials[4 * k + 1];
    c = block_partials[4 * k + 2] > c ? block_partials[4 * k + 2] : c;
    d = block_partials[4 * k + 3] > d ? block_partials[4 * k + 3] : d;
  }
  a = wave_sum(a);
  b = wave_sum(b);
  c = wave_max(c);
  d = wave_max(d);
}

__global__ __launch_bounds__(64) void pf_reduce_kernel(const double* __restrict__ block_partials, int n_blocks, int rank,
                                                       int world, double* __restrict__ packed, int n_waves = 0)
{
  double a, b, c, d;
  pf_reduce_lanes(block_partials, n_blocks, n_waves, threadIdx.x, a, b, c, d);
  if (threadIdx.x == 0)
  {
    packed[0] = a;
    packed[1] = b;
    for (int r = 0; r < world; ++r)
    {
      packed[2 + 2 * r] = (r == rank) ? c : 0.0;
      packed[3 + 2 * r] = (r == rank) ? d : 0.0;
    }
  }
}

// Normalise (pf.h:262-272) or restore (pf.h:274-278); entropy = ln S - T/S == -sum (w/S) ln (w/S).
// `packed` is the (all-reduced) vector described above.
// emit (+ the device arrays its lik / ratio / beam copies come from): see PfEmit.
// partials != null (one GPU, world == 1): EVERY work-group runs pf_reduce_kernel's reduction itself (first wavefront, the same
// association) instead of reading `packed` behind a launch of its own; work-group 0 also leaves the four sums in packed_w.
__global__ __launch_bounds__(PF_BLOCK) void pf_apply_kernel(float* __restrict__ w, const float* __restrict__ w_new,
                                                            int n, int world, const double* __restrict__ packed_in,
                                                            float* __restrict__ stats4, PfEmit emit = PfEmit{},
                                                            const float* __restrict__ lik = nullptr,
                                                            const float* __restrict__ ratio = nullptr,
                                                            const float* __restrict__ beam = nullptr,
                                                            const double* __restrict__ partials = nullptr, int n_blocks = 0,
                                                            int n_waves = 0, double* __restrict__ packed_w = nullptr)
{
  __shared__ double s_tot[4];
  const double* packed = packed_in;
  if (partials)
  {
    if (threadIdx.x < 64)
    {
      double a, b, c, d;
      pf_reduce_lanes(partials, n_blocks, n_waves, threadIdx.x, a, b, c, d);
      if (threadIdx.x == 0)
      {
        s_tot[0] = a;
        s_tot[1] = b;
        s_tot[2] = c;
        s_tot[3] = d;
        if (blockIdx.x == 0 && packed_w)
        {
          packed_w[0] = a;
          packed_w[1] = b;
          packed_w[2] = c;
          packed_w[3] = d;
        }
      }
    }
    __syncthreads();
    packed = s_tot;
  }
  const double S = packed[0];
  const float sum_f = static_cast<float>(S);
  const bool alive = sum_f > 0.0f;
  if (alive)
  {
    for (int i = blockIdx.x * PF_BLOCK + threadIdx.x; i < n; i += gridDim.x * PF_BLOCK)
    {
      const float wv = w_new[i] / sum_f;
      w[i] = wv;
      if (emit.w)
        emit.w[i] = wv;
    }
  }
  else if (emit.w)
  {
    for (int i = blockIdx.x * PF_BLOCK + threadIdx.x; i < n; i += gridDim.x * PF_BLOCK)
      emit.w[i] = w[i];
  }
  if (emit.lik || emit.ratio || emit.beam)
    for (int i = blockIdx.x * PF_BLOCK + threadIdx.x; i < n; i += gridDim.x * PF_BLOCK)
    {
      if (emit.lik)
        emit.lik[i] = lik[i];
      if (emit.ratio)
        emit.ratio[i] = ratio[i];
      if (emit.beam)
        emit.beam[i] = beam[i];
    }
  if (blockIdx.x == 0 && threadIdx.x == 0 && (stats4 || emit.stats4))
  {
    // every rank's slot holds a value in [0,1] resp. [-1,0] (0 in both for a rank whose shard saw no ratios is impossible:
    // an empty shard reports max 0 / -min -1); the maxima over the slots are the global max ratio and -min ratio
    double rmax = packed[2], rneg = packed[3];
    for (int r = 1; r < world; ++r)
    {
      rmax = packed[2 + 2 * r] > rmax ? packed[2 + 2 * r] : rmax;
      rneg = packed[3 + 2 * r] > rneg ? packed[3 + 2 * r] : rneg;
    }
    const float st[4] = { alive ? static_cast<float>(log(S) - packed[1] / S) : __builtin_nanf(""), static_cast<float>(-rneg),
                          static_cast<float>(rmax), alive ? 0.0f : 1.0f };
    for (int k = 0; k < 4; ++k)
    {
      if (stats4)
        stats4[k] = st[k];
      if (emit.stats4)
        emit.stats4[k] = st[k];
    }
  }
}

// pf_partial_kernel -> pf_reduce_kernel -> pf_apply_kernel for ONE GPU and at most PF_FUSED_MAX particles, as a single
// work-group: the reference's real operating range (64 .. a few thousand particles) is launch-bound — three launches of
// a few microseconds each around ~1 us of work. The arithmetic is the three kernels' own, in the same association (the
// 256-thread "blocks" of pf_partial_kernel become quarters of this 1024-thread group that walk the same elements, the
// 64-lane reduce runs in the first wavefront), so weights, entropy and ratio bounds are bit-identical to the split form.
// Measured and NOT kept (round 3, commit 5b571a1): pf_partial_kernel's grid with the reduce and the apply run by the last
// work-group behind an arrival-ticket tree, for 1024 < n <= 16 384 — bit-identical, and the group cost 24 us instead of
// 12.5 us at 4096 particles (49 at 16 384): three launches enqueued back to back cost 4.2 us each, less than two ticket
// levels plus one work-group's sc1 loads of every weight (profiles/r03z_pf_one_launch_ab.txt).
constexpr int PF_FUSED_MAX = 4096;

__global__ __launch_bounds__(1024) void pf_fused_kernel(float* __restrict__ w, const float* __restrict__ lik,
                                                        const float* __restrict__ beam, const float* __restrict__ extra,
                                                        const float* __restrict__ ratio, int n, float* __restrict__ w_new,
                                                        double* __restrict__ packed, float* __restrict__ stats4,
                                                        PfEmit emit = PfEmit{}, int float_order = 0, BeamCounts bc = BeamCounts{})
{
  // float_order: pf::measure's `sum += p.probability_` (pf.h:255-260) as the reference runs it — float, sequentially, in
  // particle order (float_chain.h) — instead of the fp64 tree: the weights are divided by exactly the reference's float
  __shared__ __attribute__((aligned(16))) float s_w[PF_FUSED_MAX + 4];
  __shared__ double sh[4][16];          // per wavefront of the group
  __shared__ double part[4][16];        // per virtual block (n <= 4096 -> at most 16 of them)
  __shared__ double tot[4];
  const int nb = (n + PF_BLOCK - 1) / PF_BLOCK;  // pf_blocks(n) for n <= 4096
  const int q = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int vb0 = 0; vb0 < nb; vb0 += 4)
  {
    const int vb = vb0 + q;  // this quarter's virtual block
    double s = 0.0, t = 0.0, rmax = 0.0, rneg = -1.0;
    const int i = vb * PF_BLOCK + tid;  // stride nb * PF_BLOCK >= n: one element per thread, like pf_partial_kernel
    if (vb < nb && i < n)
    {
      float l = 1.0f;
      if (bc.penalty)  // (beam == bc.beam_out: the apply loop below re-reads what THIS thread writes here)
        l *= beam_score_from_count(bc, i);
      else if (beam)
        l *= beam[i];
      l *= lik[i];
      if (extra)
        l = l * extra[i];
      const float wn = w[i] * l;
      w_new[i] = wn;
      if (float_order)
        s_w[i] = wn;
      s += static_cast<double>(wn);
      if (wn > 0.0f)
        t += static_cast<double>(wn) * log(static_cast<double>(wn));
      if (ratio)
      {
        const double r = static_cast<double>(ratio[i]);
        rmax = r > rmax ? r : rmax;
        rneg = -r > rneg ? -r : rneg;
      }
    }
    s = wave_sum(s);
    t = wave_sum(t);
    rmax = wave_max(rmax);
    rneg = wave_max(rneg);
    if (lane == 0)
    {
      sh[0][wave] = s;
      sh[1][wave] = t;
      sh[2][wave] = rmax;
      sh[3][wave] = rneg;
    }
    __syncthreads();
    if (tid == 0 && vb < nb)
    {
      double a = 0, b = 0, c = sh[2][4 * q], d = sh[3][4 * q];
      for (int k = 0; k < PF_BLOCK / 64; ++k)
      {
        a += sh[0][4 * q + k];
        b += sh[1][4 * q + k];
        c = sh[2][4 * q + k] > c ? sh[2][4 * q + k] : c;
        d = sh[3][4 * q + k] > d ? sh[3][4 * q + k] : d;
      }
      part[0][vb] = a;
      part[1][vb] = b;
      part[2][vb] = c;
      part[3][vb] = d;
    }
    __syncthreads();
  }
  if (wave == 0)
  {
    // pf_reduce_kernel: 64 lanes stride the block partials, wavefront reduction
    double a = 0, b = 0, c = 0.0, d = -1.0;
    for (int k = lane; k < nb; k += 64)
    {
      a += part[0][k];
      b += part[1][k];
      c = part[2][k] > c ? part[2][k] : c;
      d = part[3][k] > d ? part[3][k] : d;
    }
    a = wave_sum(a);
    b = wave_sum(b);
    c = wave_max(c);
    d = wave_max(d);
    if (float_order)
    {
      // (s_w[0 .. n) was written before the loop's last barrier; the padding of the last float4 here, by this wavefront)
      if (lane < chain_row_floats(n) - n)
        s_w[n + lane] = 0.0f;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      a = static_cast<double>(seq_sum_wave(s_w, n, lane));
    }
    if (lane == 0)
    {
      tot[0] = a;
      tot[1] = b;
      tot[2] = c;
      tot[3] = d;
      packed[0] = a;
      packed[1] = b;
      packed[2] = c;
      packed[3] = d;
    }
  }
  __syncthreads();
  // pf_apply_kernel
  const double S = tot[0];
  const float sum_f = static_cast<float>(S);
  const bool alive = sum_f > 0.0f;
  for (int i = threadIdx.x; i < n; i += 1024)
  {
    float wv;
    if (alive)
    {
      wv = w_new[i] / sum_f;
      w[i] = wv;
    }
    else
      wv = w[i];
    if (emit.w)
      emit.w[i] = wv;
    if (emit.lik)
      emit.lik[i] = lik[i];
    if (emit.ratio)
      emit.ratio[i] = ratio ? ratio[i] : 0.f;
    if (emit.beam)
      emit.beam[i] = beam ? beam[i] : 1.f;
  }
  if (threadIdx.x == 0 && (stats4 || emit.stats4))
  {
    const float st[4] = { alive ? static_cast<float>(log(S) - tot[1] / S) : __builtin_nanf(""), static_cast<float>(-tot[3]),
                          static_cast<float>(tot[2]), alive ? 0.0f : 1.0f };
    for (int k = 0; k < 4; ++k)
    {
      if (stats4)
        stats4[k] = st[k];
      if (emit.stats4)
        emit.stats4[k] = st[k];
    }
  }
}

// per-particle results of the two models -> page-locked host memory (mcl3dl_hip_measure_batch without a D2H copy)
__global__ __launch_bounds__(PF_BLOCK) void emit3_kernel(PfEmit emit, const float* __restrict__ lik, const float* __restrict__ ratio,
                                                         const float* __restrict__ beam, int n)
{
  for (int i = blockIdx.x * PF_BLOCK + threadIdx.x; i < n; i += gridDim.x * PF_BLOCK)
  {
    if (emit.lik)
      emit.lik[i] = lik[i];
    if (emit.ratio)
      emit.ratio[i] = ratio[i];
    if (emit.beam)
      emit.beam[i] = beam[i];
  }
}


__device__ __forceinline__ float pf_weight_product(float w, float lik, float beam, bool has_beam, const float* __restrict__ extra,
                                                   int i)
{
  float l = 1.0f;
  if (has_beam)
    l *= beam;
  l *= lik;
  if (extra)
    l = l * extra[i];
  return w * l;  // pf.h:258
}

// ---------------------------------------------------------------------------------------------------------
// "Next" row: the reductions that follow pf::measure in the node (src/mcl_3dl.cpp:451-452,706-709):
// pf::expectationBiased / max / maxBiased (include/mcl_3dl/pf.h:294-303,361-390) with ParticleWeightedMeanQuat
// (include/mcl_3dl/state_6dof.h:316-355), and pf::covariance (pf.h:304-360) with State6DOF::covElement (:162-184).
// Per-particle products are the reference's float expressions; the sums are fp64 trees (reference: float sequential).
// ---------------------------------------------------------------------------------------------------------
constexpr int MOM_N = 10;  // p_sum, pos[3], front[3], up[3]

struct ArgMax
{
  float v;
  int i;
};
__device__ inline ArgMax argmax_better(ArgMax a, ArgMax b)
{
  // pf.h:365-372: `if (max_probability < p.probability_)` -> the FIRST maximum wins
  return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}

__global__ __launch_bounds__(PF_BLOCK) void pf_moments_kernel(const float* __restrict__ pose7,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ bias, int n,
                                                              double* __restrict__ block_mom /*[grid][MOM_N]*/,
                                                              ArgMax* __restrict__ block_arg /*[grid][2]*/)
{
  double m[MOM_N];
#pragma unroll
  for (int k = 0; k < MOM_N; ++k)
    m[k] = 0.0;
  ArgMax am = { -1.0f, 0x7fffffff }, ab = { -1.0f, 0x7fffffff };
  bool first = true;
  for (int i = blockIdx.x * PF_BLOCK + threadIdx.x; i < n; i += gridDim.x * PF_BLOCK)
  {
    const float* ps = pose7 + 7 * static_cast<size_t>(i);
    const float prob = w[i] * (bias ? bias[i] : 1.0f);  // pf.h:300
    const Quat rot = { ps[3], ps[4], ps[5], ps[6] };
    const Vec3f front = vscale(qrot(rot, Vec3f{ 1.0f, 0.0f, 0.0f }), prob);  // state_6dof.h:337-338
    const Vec3f up = vscale(qrot(rot, Vec3f{ 0.0f, 0.0f, 1.0f }), prob);
    m[0] += static_cast<double>(prob);
    m[1] += static_cast<double>(ps[0] * prob);  // e_.pos_ += e1.pos_ * prob, :335
    m[2] += static_cast<double>(ps[1] * prob);
    m[3] += static_cast<double>(ps[2] * prob);
    m[4] += static_cast<double>(front.x);
    m[5] += static_cast<double>(front.y);
    m[6] += static_cast<double>(front.z);
    m[7] += static_cast<double>(up.x);
    m[8] += static_cast<double>(up.y);
    m[9] += static_cast<double>(up.z);
    const ArgMax cm = { w[i], i }, cb = { prob, i };
    am = first ? cm : argmax_better(am, cm);
    ab = first ? cb : argmax_better(ab, cb);
    first = false;
  }
  __shared__ double sh[MOM_N][PF_BLOCK / 64];
  __shared__ ArgMax sa[2][PF_BLOCK / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < MOM_N; ++k)
  {
    const double s = wave_sum(m[k]);
    if (lane == 0)
      sh[k][wave] = s;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
  {
    ArgMax o1 = { __shfl_down(am.v, off, 64), __shfl_down(am.i, off, 64) };
    ArgMax o2 = { __shfl_down(ab.v, off, 64), __shfl_down(ab.i, off, 64) };
    am = argmax_better(am, o1);
    ab = argmax_better(ab, o2);
  }
  if (lane == 0)
  {
    sa[0][wave] = am;
    sa[1][wave] = ab;
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    for (int k = 0; k < MOM_N; ++k)
    {
      double s = 0;
      for (int q = 0; q < PF_BLOCK / 64; ++q)
        s += sh[k][q];
      block_mom[MOM_N * blockIdx.x + k] = s;
    }
    ArgMax a = sa[0][0], b = sa[1][0];
    for (int q = 1; q < PF_BLOCK / 64; ++q)
    {
      a = argmax_better(a, sa[0][q]);
      b = argmax_better(b, sa[1][q]);
    }
    block_arg[2 * blockIdx.x + 0] = a;
    block_arg[2 * blockIdx.x + 1] = b;
  }
}

__global__ __launch_bounds__(64) void pf_moments_reduce_kernel(const double* __restrict__ block_mom,
                                                               const ArgMax* __restrict__ block_arg, int n_blocks,
                                                               double* __restrict__ out_mom /*[MOM_N]*/,
                                                               int* __restrict__ out_arg /*[2]*/,
                                                               double* __restrict__ out16 /*or null: shard record*/)
{
  if (threadIdx.x < MOM_N)
  {
    double s = 0;
    for (int b = 0; b < n_blocks; ++b)  // fixed order
      s += block_mom[MOM_N * b + threadIdx.x];
    out_mom[threadIdx.x] = s;
    if (out16)
      out16[threadIdx.x] = s;
  }
  if (threadIdx.x == 32 || threadIdx.x == 33)
  {
    const int which = threadIdx.x - 32;
    ArgMax a = block_arg[which];
    for (int b = 1; b < n_blocks; ++b)
      a = argmax_better(a, block_arg[2 * b + which]);
    out_arg[which] = a.i;
    if (out16)
    {
      // {10: max weight, 11: its index in this shard, 12: max biased weight, 13: its index, 14-15: 0}
      out16[MOM_N + 2 * which] = static_cast<double>(a.v);
      out16[MOM_N + 2 * which + 1] = static_cast<double>(a.i);
      out16[14 + which] = 0.0;
    }
  }
}

// Quat::getRPY, include/mcl_3dl/quat.h:188-203 (float storage, double intermediates; device atan2f / asinf)
__host__ __device__ inline Vec3f quat_get_rpy(Quat q)
{
  const float ysq = q.y * q.y;
  const float t0 = static_cast<float>(-2.0 * (ysq + q.z * q.z) + 1.0);
  const float t1 = static_cast<float>(+2.0 * (q.x * q.y + q.w * q.z));
  const double t2d = -2.0 * (q.x * q.z - q.w * q.y);
  const float t2 = static_cast<float>(t2d > 1.0 ? 1.0 : (t2d < -1.0 ? -1.0 : t2d));
  const float t3 = static_cast<float>(+2.0 * (q.y * q.z + q.w * q.x));
  const float t4 = static_cast<float>(-2.0 * (q.x * q.x + ysq) + 1.0);
  return { atan2f(t3, t4), asinf(t2), atan2f(t1, t0) };
}

constexpr int COV_N = 22;  // 21 upper-triangular sums + p_sum

// subset == nullptr: particles 0..n-1; else the n indices the caller drew (pf.h:322-336 shuffles them with its own RNG)
__global__ __launch_bounds__(PF_BLOCK) void pf_covariance_kernel(const float* __restrict__ pose7,
                                                                 const float* __restrict__ w,
                                                                 const uint32_t* __restrict__ subset, int n,
                                                                 float e0, float e1, float e2, Vec3f exp_rpy,
                                                                 double* __restrict__ block_cov /*[grid][COV_N]*/)
{
  double acc[COV_N];
#pragma unroll
  for (int k = 0; k < COV_N; ++k)
    acc[k] = 0.0;
  for (int t = blockIdx.x * PF_BLOCK + threadIdx.x; t < n; t += gridDim.x * PF_BLOCK)
  {
    const size_t i = subset ? subset[t] : static_cast<size_t>(t);
    const float* ps = pose7 + 7 * i;
    const float prob = w[i];
    const Vec3f rpy = quat_get_rpy(Quat{ ps[3], ps[4], ps[5], ps[6] });
    float d[6] = { ps[0] - e0, ps[1] - e1, ps[2] - e2, rpy.x - exp_rpy.x, rpy.y - exp_rpy.y, rpy.z - exp_rpy.z };
#pragma unroll
    for (int a = 3; a < 6; ++a)  // covElement, state_6dof.h:175-179
    {
      while (d[a] > M_PI)
        d[a] = static_cast<float>(d[a] - 2 * M_PI);
      while (d[a] < -M_PI)
        d[a] = static_cast<float>(d[a] + 2 * M_PI);
    }
    int idx = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int k = j; k < 6; ++k)
      {
        float val = 1.0f;
        val *= d[j];
        val *= d[k];
        acc[idx++] += static_cast<double>(val * prob);  // pf.h:347
      }
    acc[21] += static_cast<double>(prob);
  }
  __shared__ double sh[COV_N][PF_BLOCK / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < COV_N; ++k)
  {
    const double s = wave_sum(acc[k]);
    if (lane == 0)
      sh[k][wave] = s;
  }
  __syncthreads();
  if (threadIdx.x < COV_N)
  {
    double s = 0;
    for (int q = 0; q < PF_BLOCK / 64; ++q)
      s += sh[threadIdx.x][q];
    block_cov[COV_N * blockIdx.x + threadIdx.x] = s;
  }
}

__global__ __launch_bounds__(64) void pf_covariance_reduce_kernel(const double* __restrict__ block_cov, int n_blocks,
                                                                  double* __restrict__ out_cov /*[COV_N]*/)
{
  if (threadIdx.x < COV_N)
  {
    double s = 0;
    for (int b = 0; b < n_blocks; ++b)
      s += block_cov[COV_N * b + threadIdx.x];
    out_cov[threadIdx.x] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------
// "Next" row: pf::ParticleFilter::resample / resizeParticle (include/mcl_3dl/pf.h:187-225, 399-436).
// The serial, order-defining parts (float prefix sums, libstdc++'s std::sort of the tie groups, the it/it_prev walk)
// stay on the host in mcl3dl_hip.hip; the device does the n_out independent std::lower_bound searches and the
// gather of the 13-dof states with State6DOF::operator+ / normalize() for the duplicated ones.
// ---------------------------------------------------------------------------------------------------------
__global__ void resample_lower_bound_kernel(const float* __restrict__ keys, int n, const float* __restrict__ pscan,
                                            float pstep, float initial_p, int n_out, uint32_t* __restrict__ it_out,
                                            uint32_t* __restrict__ last_valid)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out)
    return;
  // resample: pscan = pstep * i + initial_p with i a size_t converted to float (pf.h:209); resizeParticle: the host's
  // running sum
  const float p = pscan ? pscan[i] : pstep * static_cast<float>(static_cast<unsigned long long>(i)) + initial_p;
  int lo = 0, len = n;  // std::lower_bound with Particle::operator< (pf.h:104-107): first key with !(key < p)
  while (len > 0)
  {
    const int half = len >> 1;
    if (keys[lo + half] < p)
    {
      lo += half + 1;
      len -= half + 1;
    }
    else
      len = half;
  }
  it_out[i] = static_cast<uint32_t>(lo);
  if (lo < n)
    atomicMax(last_valid, static_cast<uint32_t>(lo));  // = it_prev once the scan position runs past the last key
}

// The it / it_prev walk of pf.h:204-223 (resample) and :414-434 (resizeParticle) for non-decreasing search results:
// slot i copies particles_dup_[it[i]]; it is a duplicate (gets noise) when it[i] equals the previous slot's result
// (it_prev starts at begin(), so slot 0 is a duplicate when it lands on element 0); once it[i] == n every remaining slot
// copies the last element found before the end and nothing else changes (`continue`, :212-216).
__global__ void resample_walk_kernel(const uint32_t* __restrict__ it, int n, const uint32_t* __restrict__ order /*or null*/,
                                     int mode, int n_out, uint32_t* __restrict__ source, uint32_t* __restrict__ dup_flag)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out)
    return;
  const uint32_t cur = it[i];
  uint32_t pick, dup = 0u;
  if (cur >= static_cast<uint32_t>(n))
    pick = it[n_out];  // last search result below n (0 = begin() if there was none)
  else
  {
    pick = cur;
    const uint32_t prev = i ? it[i - 1] : 0u;
    dup = (mode == 0 && cur == prev) ? 1u : 0u;
  }
  source[i] = order ? order[pick] : pick;
  dup_flag[i] = dup;
}

// noise slot of every duplicated output slot = its rank among the duplicates (exclusive scan of the flags)
__global__ void resample_slot_kernel(const uint32_t* __restrict__ scanned, int n_out, uint32_t* __restrict__ slot_inout,
                                     uint8_t* __restrict__ dup8)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out)
    return;
  const uint32_t flag = slot_inout[i];  // the un-scanned flag was parked here
  slot_inout[i] = flag ? scanned[i] : 0xffffffffu;
  if (dup8)
    dup8[i] = static_cast<uint8_t>(flag);
}

// slot i receives the state of particle source[i]; duplicated picks get `state + noise` (State6DOF::operator+,
// state_6dof.h:248-260: components 0-2 and 7-12 add, rot = noise.rot * state.rot) followed by normalize() (:150-153).
__global__ void resample_apply_kernel(const float* __restrict__ state_in, const uint32_t* __restrict__ source,
                                      const uint32_t* __restrict__ noise_slot /* 0xffffffff = not duplicated */,
                                      const float* __restrict__ noise13, int n_out, float* __restrict__ state_out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out)
    return;
  const float* s = state_in + 13 * static_cast<size_t>(source[i]);
  float* o = state_out + 13 * static_cast<size_t>(i);
  const uint32_t slot = noise_slot[i];
  if (slot == 0xffffffffu)
  {
#pragma unroll
    for (int k = 0; k < 13; ++k)
      o[k] = s[k];
    return;
  }
  const float* a = noise13 + 13 * static_cast<size_t>(slot);
#pragma unroll
  for (int k = 0; k < 3; ++k)
    o[k] = s[k] + a[k];
#pragma unroll
  for (int k = 7; k < 13; ++k)
    o[k] = s[k] + a[k];
  const Quat r = qnormalized(qmul(Quat{ a[3], a[4], a[5], a[6] }, Quat{ s[3], s[4], s[5], s[6] }));
  o[3] = r.x;
  o[4] = r.y;
  o[5] = r.z;
  o[6] = r.w;
}
}  // namespace mcl3dl
