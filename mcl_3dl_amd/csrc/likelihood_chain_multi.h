// likelihood_chain_multi.h — strict_order = 3 (the reference's float recurrence inside the likelihood kernel,
// likelihood_kernels.h: LikChain) for FEW particles on LONG scans.
//
// The chain of a particle is n_tiles hand-offs long in likelihood_tiled_kernel<..., CHAIN> — ~2.2 us each, 0.14 ms for a
// 16 384-point scan whatever the particle count — and a hop is mostly the hand-off (a store and a polled load across XCDs,
// ~1.6 us), not the 256 dependent adds (~0.6 us). Below ~3000 particles the evaluation is shorter than the chain and the hops
// are what the launch costs (1024 particles: 0.152 ms against 0.080 ms with fp64 sums, profiles/r05f_chain_midsize.txt). Here
// a work-group evaluates PPL consecutive tiles for its G particles — every term stays in LDS, PPL x G x 260 floats — and only
// THEN takes the running sums over, adds its PPL x 256 terms and hands on: a quarter of the hand-offs for PPL = 4, the same
// adds in the same order, so the same bits. Same (super-tile, group) -> work-group mapping as the one-tile form (rows of eight
// super-tiles, one per XCD; a producer always has a lower block index), same hand-off words and tags (tag0 + super-tile).
// Measured (profiles/r05r_chain_multi.txt, 16 384 points): 64 particles 0.122 -> 0.096 ms, 1024 particles 0.139 -> 0.122 ms (fp64
// sums: 0.072), 1024 x 65 536 points 0.480 -> 0.358 ms; SLOWER from 2048 particles (groups of four particles, a longer start-up),
// so the host uses it up to chain_multi_max = 1536 particles.
// Only the default kernel family (packed 64-byte records, quad-cooperative fetch, deferred overflow rounds); everything else
// and every launch with enough particles stays with likelihood_tiled_kernel<..., CHAIN>.
#pragma once
#include <hip/hip_runtime.h>

#include "likelihood_kernels.h"

#pragma clang fp contract(off)

namespace mcl3dl
{
template <int G, int PPL>
__global__ __launch_bounds__(256, 8) void likelihood_chain_multi_kernel(const float* __restrict__ pose7, int n_p,
                                                                        const float4* __restrict__ scan, int n_s, int n_tiles,
                                                                        int n_super, int n_groups, RecGrid rg, LikParams prm,
                                                                        LikChain ch)
{
  constexpr int LD = 260;  // (rows padded: the chain's lanes read different rows at the same column, 16 bytes at a time)
  __shared__ float s_pose[G][8];
  __shared__ __attribute__((aligned(16))) float s_term[PPL][G][LD];
  __shared__ unsigned s_cnt[PPL][G][4];
  __shared__ DeferQueue s_q;
  const uint32_t xcd = blockIdx.x & 7u, seq = blockIdx.x >> 3, ng = static_cast<uint32_t>(n_groups);
  const uint32_t row = seq / ng;
  const int st = static_cast<int>(row * 8u + xcd);
  const int group = static_cast<int>(seq - row * ng);
  if (st >= n_super)
    return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t < G)
  {
    const int p = group * G + t;
    float valid = 0.f;
    if (p < n_p)
    {
      const float* ps = pose7 + 7 * static_cast<size_t>(p);
      const Quat r = qnormalized(Quat{ ps[3], ps[4], ps[5], ps[6] });
      s_pose[t][0] = ps[0];
      s_pose[t][1] = ps[1];
      s_pose[t][2] = ps[2];
      s_pose[t][3] = r.x;
      s_pose[t][4] = r.y;
      s_pose[t][5] = r.z;
      s_pose[t][6] = r.w;
      valid = 1.f;
    }
    s_pose[t][7] = valid;
  }
  __syncthreads();
  const int n_valid = min(G, n_p - group * G);
  const int first_tile = st * PPL;
  const int n_sub = min(PPL, n_tiles - first_tile);
  for (int m = 0; m < n_sub; ++m)
  {
    const int i = (first_tile + m) * 256 + t;
    const bool have_point = i < n_s;
    const float4 v = have_point ? scan[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    // (the evaluation of likelihood_tiled_kernel<G, 2, 8, true, true>: every lane stays active through the cooperative fetch,
    // overflow rounds are queued per wavefront and run densely; the queue is empty again before the next tile's point replaces v)
    uint32_t qn = 0;
    for (int k = 0; k < n_valid; ++k)
    {
      const Vec3f pos = { s_pose[k][0], s_pose[k][1], s_pose[k][2] };
      const Quat rot = { s_pose[k][3], s_pose[k][4], s_pose[k][5], s_pose[k][6] };
      bool matched, over;
      float best;
      uint32_t mine;
      unsigned long long over_m;
      const float term = eval_coop_first(rg, prm, pos, rot, v, have_point, lane, matched, over, best, mine, over_m);
      s_term[m][k][t] = over ? best : term;
      const unsigned long long mm = __builtin_amdgcn_ballot_w64(matched);
      if (lane == 0)
        s_cnt[m][k][wave] = static_cast<unsigned>(__popcll(mm));
      const unsigned long long om = __builtin_amdgcn_ballot_w64(over);
      if (om != 0ull)
      {
        const uint32_t n_new = static_cast<uint32_t>(__popcll(om));
        if (qn + n_new > static_cast<uint32_t>(DEFER_QCAP))
        {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          defer_drain(rg, prm, s_pose, s_term[m], s_cnt[m], s_q, wave, lane, 0u, qn, v);
          qn = 0;
        }
        if (over)
        {
          const uint32_t slot = qn + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(om >> 32),
                                                               __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(om), 0u));
          s_q.word[wave][slot] = mine;
          s_q.who[wave][slot] = static_cast<uint16_t>((static_cast<uint32_t>(k) << 6) | static_cast<uint32_t>(lane));
        }
        qn += n_new;
        if (qn >= 64u)
        {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          defer_drain(rg, prm, s_pose, s_term[m], s_cnt[m], s_q, wave, lane, qn - 64u, 64u, v);
          qn -= 64u;
        }
      }
    }
    if (qn != 0u)
    {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      defer_drain(rg, prm, s_pose, s_term[m], s_cnt[m], s_q, wave, lane, 0u, qn, v);
    }
  }
  __syncthreads();
  if (wave != 0 || lane >= n_valid)
    return;
  // ---- the chain: one lane per particle
  const int k = lane;
  const size_t p = static_cast<size_t>(group) * G + k;
  unsigned long long* cw = ch.carry + 2 * p;
  float s = 0.0f;
  uint32_t cnt = 0;
  if (st > 0)
  {
    const uint32_t want = ch.tag0 + static_cast<uint32_t>(st) - 1u;
    unsigned long long a = 0, b = 0;
    int polls = 0;
    bool got = false;
    while (!got)
    {
      a = __hip_atomic_load(cw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      b = __hip_atomic_load(cw + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      got = static_cast<uint32_t>(a >> 32) == want && static_cast<uint32_t>(b >> 32) == want;
      if (!got)
      {
        if (++polls > CHAIN_POLL_MAX)
        {
          *ch.err = 1u;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    s = __uint_as_float(static_cast<uint32_t>(a));
    cnt = static_cast<uint32_t>(b);
  }
  for (int m = 0; m < n_sub; ++m)
  {
    const float4* rowp = reinterpret_cast<const float4*>(&s_term[m][k][0]);
#pragma unroll 8
    for (int j = 0; j < 64; ++j)
    {
      const float4 t4 = rowp[j];
      s = s + t4.x;
      s = s + t4.y;
      s = s + t4.z;
      s = s + t4.w;
    }
    cnt += s_cnt[m][k][0] + s_cnt[m][k][1] + s_cnt[m][k][2] + s_cnt[m][k][3];
  }
  if (st == n_super - 1)
  {
    ch.out_lik[p] = s;
    if (ch.out_ratio)
      ch.out_ratio[p] = static_cast<float>(cnt) / static_cast<float>(n_s);
    if (ch.also_fill)
      ch.also_fill[p] = 1.0f;
  }
  else
  {
    const uint32_t tag = ch.tag0 + static_cast<uint32_t>(st);
    __hip_atomic_store(cw, chain_pack(__float_as_uint(s), tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(cw + 1, chain_pack(cnt, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
}  // namespace mcl3dl
