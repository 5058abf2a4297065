// stage_kernels.h — the head of a HOST-BUFFER measurement update (mcl3dl_hip_measure_update, SURVEY.md §8d's timed region):
// the caller's arrays are taken over by ONE launch and no DMA copy (the tail — results written back into page-locked memory by
// the kernel that normalises the weights — is pf_kernels.h:PfEmit):
//
//   scan_stage_kernel   reads the caller's arrays where they lie in page-locked host memory (or in a device mirror of that
//                       block after one copy) and leaves everything the update kernels need in device memory:
//                         work-group 0   the likelihood scan: xyz -> float4, min corner, Morton keys (cloud_keys.h), stable LSD radix
//                                        sort of (key, index) with the pairs exchanged through LDS, ordered points + the
//                                        permutation written by the last pass                 (api_core.inl:order_scan)
//                         work-group 1   the beam scan the same way, keyed by squared range from its origin
//                         work-groups 2+ poses, prior weights and the odometry factor copied to their device arrays
//                       Same keys, same stable order as host_cloud.h:device_order_scans (pack + min / max, key + count, three
//                       count / scatter pairs: seven launches) and as the host ordering: bit-identical results on every path.
//   stage_pack_kernel   scans of more than 2048 points: the same take-over without the ordering (the chip-wide sort follows)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cloud_kernels.h"
#include "pf_kernels.h"
#include "sort_kernels.h"

#pragma clang fp contract(off)

namespace mcl3dl
{
// One work-group orders up to 2048 points (measured with 16 rounds per thread, 149 KB of LDS: a 16 384-point scan took 83 us —
// ~100 VALU instructions per 64 elements and pass on ONE CU — against ~35 us for the seven launches of the chip-wide sort:
// profiles/r04a_time8d_C2.json; larger scans go through stage_pack_kernel + device_order_scans)
constexpr int ST_MAX_ROUNDS = 2;
constexpr int ST_MAX_POINTS = RS_THREADS * ST_MAX_ROUNDS;

struct StageArgs
{
  // poses / prior weights / odometry factor: in_* host-visible (or device mirror), d_* device arrays; null in_* = no copy
  const float* in_pose;
  const float* in_w;
  const float* in_extra;
  float* d_pose;
  float* d_w;
  float* d_extra;
  int n_p;
  // likelihood scan
  const float* in_lik_xyz;
  int n_s;
  float4* raw_lik;              // [n_s] the scan as float4 in the caller's order (device)
  float* mm6;                   // {min xyz, max xyz} of its finite points (device, 6 floats)
  unsigned long long* mm_cnt;   // number of finite points
  float4* out_lik;              // ordered, w = 0
  uint32_t* out_perm;           // out_lik[k] = raw_lik[out_perm[k]]
  // beam scan
  const float* in_beam_xyz;
  const uint32_t* in_beam_origin;  // may be null: origin 0
  int n_b;
  float4* raw_beam;
  float4* out_beam;             // ordered by range, w = origin id
  const float* in_origins;      // n_o x 3
  int n_o;
  float4* d_origins;
  int* d_err;                   // set to 2 when a beam point names an origin that does not exist
  int presorted;                // option scan_presorted: the likelihood scan is installed in the caller's order as it is
};

template <int ROUNDS>
struct StageLds
{
  uint32_t kx[RS_THREADS * ROUNDS];
  uint32_t vx[RS_THREADS * ROUNDS];
  uint32_t cnt[RS_WAVES][256];
  uint32_t dbase[256];
  uint32_t wsum[4];
  float mm[6][RS_WAVES];
  unsigned long long fin[RS_WAVES];
  float mm6[6];
};

// One work-group of RS_THREADS threads: n <= RS_THREADS * ROUNDS points (xyz [+ w word]) -> `raw` (float4, caller's order)
// and `out` (stable ascending order of the key), `perm` (optional).
template <int ROUNDS, int KEYMODE>
__device__ __forceinline__ void stage_order(const float* __restrict__ in_xyz, const uint32_t* __restrict__ in_w, int n,
                                            float4* __restrict__ raw, float4* __restrict__ out, uint32_t* __restrict__ perm,
                                            float* __restrict__ mm6_out, unsigned long long* __restrict__ cnt_out,
                                            const float4* __restrict__ origins, uint32_t n_o, int* __restrict__ err,
                                            StageLds<ROUNDS>& s, bool presorted = false)
{
  static_assert(KEYMODE == RS_KEY_MORTON || KEYMODE == RS_KEY_RANGE, "scan keys only");
  const int rounds = (n + RS_THREADS - 1) / RS_THREADS;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t valid = 0;
  float mn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, mx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
  unsigned cnt32 = 0;
  // ---- the points as float4 in the caller's order (+ min / max of the finite ones for the Morton key)
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r)
  {
    const int idx = (w * rounds + r) * 64 + lane;
    if (r < rounds && idx < n)
    {
      valid |= 1u << r;
      const float4 p = make_float4(in_xyz[3 * idx], in_xyz[3 * idx + 1], in_xyz[3 * idx + 2],
                                   __uint_as_float(in_w ? in_w[idx] : 0u));
      raw[idx] = p;
      if (KEYMODE == RS_KEY_MORTON)
        minmax_accumulate(p, mn, mx, cnt32);
    }
  }
  if (KEYMODE == RS_KEY_MORTON)
  {
    unsigned long long cnt = cnt32;
    wave_minmax(mn, mx, cnt);
    if (lane == 0)
    {
      for (int a = 0; a < 3; ++a)
      {
        s.mm[a][w] = mn[a];
        s.mm[3 + a][w] = mx[a];
      }
      s.fin[w] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
      float r6[6] = { 3.0e38f, 3.0e38f, 3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f };
      unsigned long long c = 0;
      for (int k = 0; k < RS_WAVES; ++k)
      {
        for (int a = 0; a < 3; ++a)
        {
          r6[a] = fminf(r6[a], s.mm[a][k]);
          r6[3 + a] = fmaxf(r6[3 + a], s.mm[3 + a][k]);
        }
        c += s.fin[k];
      }
      for (int a = 0; a < 6; ++a)
      {
        s.mm6[a] = r6[a];
        mm6_out[a] = r6[a];
      }
      *cnt_out = c;
    }
  }
  __syncthreads();  // raw[] (written by this work-group) and s.mm6 are visible to every thread of it
  if (KEYMODE == RS_KEY_MORTON && presorted)
  {
    // the caller holds its scan in the engine's order already (mcl3dl_hip_scan_order_host): installed as it is
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
      if ((valid >> r) & 1u)
      {
        const int idx = (w * rounds + r) * 64 + lane;
        float4 q = raw[idx];
        q.w = 0.f;
        out[idx] = q;
        if (perm)
          perm[idx] = static_cast<uint32_t>(idx);
      }
    return;
  }
  // ---- keys
  float mmr[6] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
  if (KEYMODE == RS_KEY_MORTON)
  {
#pragma unroll
    for (int a = 0; a < 6; ++a)
      mmr[a] = s.mm6[a];
  }
  uint32_t key[ROUNDS], val[ROUNDS], dst[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r)
  {
    const int idx = (w * rounds + r) * 64 + lane;
    key[r] = 0xffffffffu;
    val[r] = static_cast<uint32_t>(idx);
    dst[r] = 0;
    if ((valid >> r) & 1u)
    {
      const float4 p = raw[idx];
      key[r] = KEYMODE == RS_KEY_MORTON ? morton_scan_key(p, mmr) : range_scan_key(p, origins, n_o, err);
    }
  }
  constexpr int END_BIT = KEYMODE == RS_KEY_MORTON ? MCL3DL_MORTON_BITS : 32;
  constexpr int N_PASS = (END_BIT + 7) / 8;
  constexpr uint32_t MASK = END_BIT >= 32 ? 0xffffffffu : ((1u << (END_BIT & 31)) - 1u);
  for (int p = 0; p < N_PASS; ++p)
  {
    rs_rank_pass<ROUNDS>(key, valid, rounds, 8 * p, MASK, dst, s.cnt, s.dbase, s.wsum,
                         [](uint32_t total_d, uint32_t* ws) { return rs_scan256(total_d, ws); });
    if (p + 1 == N_PASS)
      break;
    // exchange through LDS: pair r goes to position dst[r]; the thread then picks up the pair AT its own positions
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
      if ((valid >> r) & 1u)
      {
        s.kx[dst[r]] = key[r];
        s.vx[dst[r]] = val[r];
      }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
      if ((valid >> r) & 1u)
      {
        const int idx = (w * rounds + r) * 64 + lane;
        key[r] = s.kx[idx];
        val[r] = s.vx[idx];
      }
    // (kx / vx are written again behind the two barriers of the next pass's ranking)
  }
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r)
    if ((valid >> r) & 1u)
    {
      float4 q = raw[val[r]];
      if (KEYMODE == RS_KEY_MORTON)
        q.w = 0.f;
      out[dst[r]] = q;
      if (perm)
        perm[dst[r]] = val[r];
    }
}

template <int ROUNDS>
__global__ __launch_bounds__(RS_THREADS) void scan_stage_kernel(StageArgs a)
{
  __shared__ StageLds<ROUNDS> s;
  if (blockIdx.x == 0)
  {
    if (a.n_s > 0)
      stage_order<ROUNDS, RS_KEY_MORTON>(a.in_lik_xyz, nullptr, a.n_s, a.raw_lik, a.out_lik, a.out_perm, a.mm6, a.mm_cnt,
                                         nullptr, 0u, nullptr, s, a.presorted != 0);
    return;
  }
  if (blockIdx.x == 1)
  {
    // the origins first (float4, w = 0): the range keys read them
    for (int i = threadIdx.x; i < a.n_o; i += RS_THREADS)
      a.d_origins[i] = make_float4(a.in_origins[3 * i], a.in_origins[3 * i + 1], a.in_origins[3 * i + 2], 0.f);
    __syncthreads();
    if (a.n_b > 0)
      stage_order<ROUNDS, RS_KEY_RANGE>(a.in_beam_xyz, a.in_beam_origin, a.n_b, a.raw_beam, a.out_beam, nullptr, nullptr,
                                        nullptr, a.d_origins, static_cast<uint32_t>(a.n_o), a.d_err, s);
    return;
  }
  // ---- copies: poses (7 floats per particle), prior weights, odometry factor
  const long long n_pose = a.in_pose ? 7ll * a.n_p : 0, n_w = a.in_w ? a.n_p : 0, n_e = a.in_extra ? a.n_p : 0;
  const long long total = n_pose + n_w + n_e;
  const long long stride = static_cast<long long>(gridDim.x - 2) * RS_THREADS;
  for (long long i = static_cast<long long>(blockIdx.x - 2) * RS_THREADS + threadIdx.x; i < total; i += stride)
  {
    if (i < n_pose)
      a.d_pose[i] = a.in_pose[i];
    else if (i < n_pose + n_w)
      a.d_w[i - n_pose] = a.in_w[i - n_pose];
    else
      a.d_extra[i - n_pose - n_w] = a.in_extra[i - n_pose - n_w];
  }
}

// Scans of more than 2 * RS_THREADS points: one CU ranks ~1000 elements per microsecond and pass, so a 16 384-point scan is
// ordered by the multi-work-group sort of sort_kernels.h (host_cloud.h:device_order_scans) — this kernel only brings the
// caller's arrays over: work-groups [0, nb_lik) the likelihood scan as float4 + its min / max (last work-group to arrive
// folds the partials: cloud_kernels.h:block_minmax_finish), [nb_lik, nb_lik + nb_beam) the beam scan (+ origins), the rest
// the poses / weights / odometry factor. 256 threads per work-group, 256 points resp. 1024 floats per work-group and round:
// one PCIe round trip for a 16 384-point scan and 4096 particles, every byte fetched once.
// 256 consecutive points of a packed xyz array (host-visible memory) -> this thread's point. Read as float4s through LDS where
// the array is 16-byte aligned: three 4-byte loads at stride 12 fetch every 64-byte segment of UNCACHED host memory three times —
// a 16 384-point scan crossed PCIe as 590 KB instead of 196 (stage_pack 14 us, profiles/r05o_timeline_8d_C2.txt).
// Every thread of the work-group calls it (two barriers inside).
__device__ inline float4 stage_load_xyz256(const float* __restrict__ in, int first_point, int n_points, float* s_xyz /*[768]*/)
{
  const int t = threadIdx.x;
  const int n_here = min(256, n_points - first_point);
  const float* src = in + 3ll * first_point;
  const int n_f = 3 * n_here;
  if ((reinterpret_cast<uintptr_t>(src) & 15u) == 0)
  {
    const int n4 = n_f >> 2;
    if (t < n4)
      reinterpret_cast<float4*>(s_xyz)[t] = reinterpret_cast<const float4*>(src)[t];
    if (t < (n_f & 3))
      s_xyz[4 * n4 + t] = src[4 * n4 + t];
  }
  else
    for (int k = t; k < n_f; k += 256)
      s_xyz[k] = src[k];
  __syncthreads();
  float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
  if (t < n_here)
    p = make_float4(s_xyz[3 * t], s_xyz[3 * t + 1], s_xyz[3 * t + 2], 0.f);
  __syncthreads();
  return p;
}

// a flat array of n floats copied by work-groups [0, n_blocks) of 256 threads, 16 bytes per thread and load where both ends
// are 16-byte aligned
__device__ inline void stage_copy_floats(const float* __restrict__ in, float* __restrict__ out, long long n, unsigned bid,
                                         unsigned n_blocks)
{
  const long long stride = static_cast<long long>(n_blocks) * 256;
  const long long me = static_cast<long long>(bid) * 256 + threadIdx.x;
  if (((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0)
  {
    const long long n4 = n >> 2;
    for (long long i = me; i < n4; i += stride)
      reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(in)[i];
    if (me < (n & 3))
      out[4 * n4 + me] = in[4 * n4 + me];
    return;
  }
  for (long long i = me; i < n; i += stride)
    out[i] = in[i];
}

__global__ __launch_bounds__(256) void stage_pack_kernel(StageArgs a, MinMaxOut mm, unsigned nb_lik, unsigned nb_beam)
{
  __shared__ __attribute__((aligned(16))) float s_xyz[768];
  const unsigned b = blockIdx.x;
  if (b < nb_lik)
  {
    float mn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, mx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    unsigned cnt = 0;
    for (int first = static_cast<int>(b) * 256; first < a.n_s; first += static_cast<int>(nb_lik) * 256)
    {
      const float4 p = stage_load_xyz256(a.in_lik_xyz, first, a.n_s, s_xyz);
      const int i = first + static_cast<int>(threadIdx.x);
      if (i < a.n_s)
      {
        a.raw_lik[i] = p;
        minmax_accumulate(p, mn, mx, cnt);
      }
    }
    block_minmax_finish(mn, mx, cnt, mm, b, nb_lik);
    return;
  }
  if (b < nb_lik + nb_beam)
  {
    const unsigned bb = b - nb_lik;
    if (bb == 0)
      for (int i = threadIdx.x; i < a.n_o; i += 256)
        a.d_origins[i] = make_float4(a.in_origins[3 * i], a.in_origins[3 * i + 1], a.in_origins[3 * i + 2], 0.f);
    for (int first = static_cast<int>(bb) * 256; first < a.n_b; first += static_cast<int>(nb_beam) * 256)
    {
      float4 p = stage_load_xyz256(a.in_beam_xyz, first, a.n_b, s_xyz);
      const int i = first + static_cast<int>(threadIdx.x);
      if (i < a.n_b)
      {
        p.w = __uint_as_float(a.in_beam_origin ? a.in_beam_origin[i] : 0u);
        a.raw_beam[i] = p;
      }
    }
    return;
  }
  // poses (7 floats per particle), prior weights, odometry factor: three flat arrays, every copy work-group takes its share of each
  const unsigned bc = b - nb_lik - nb_beam, nbc = gridDim.x - nb_lik - nb_beam;
  if (a.in_pose)
    stage_copy_floats(a.in_pose, a.d_pose, 7ll * a.n_p, bc, nbc);
  if (a.in_w)
    stage_copy_floats(a.in_w, a.d_w, a.n_p, bc, nbc);
  if (a.in_extra)
    stage_copy_floats(a.in_extra, a.d_extra, a.n_p, bc, nbc);
}

}  // namespace mcl3dl
