// stage_kernels.h — the head and the tail of a HOST-BUFFER measurement update (mcl3dl_hip_measure_update, SURVEY.md §8d's
// timed region) as one launch each, so that an update of 4096 particles x 16 384 points is three launches and no DMA copy:
//
//   scan_stage_kernel   reads the caller's arrays where they lie in page-locked host memory (or in a device mirror of that
//                       block after one copy) and leaves everything the update kernels need in device memory:
//                         work-group 0   the likelihood scan: xyz -> float4, min corner, 22-bit Morton keys, stable LSD radix
//                                        sort of (key, index) with the pairs exchanged through LDS, ordered points + the
//                                        permutation written by the last pass                 (api_core.inl:order_scan)
//                         work-group 1   the beam scan the same way, keyed by squared range from its origin
//                         work-groups 2+ poses, prior weights and the odometry factor copied to their device arrays
//                       Same keys, same stable order as host_cloud.h:device_order_scans (pack + min / max, key + count, three
//                       count / scatter pairs: seven launches) and as the host ordering: bit-identical results on every path.
//   pf_tail_kernel      lik_finalize_kernel + pf_partial_kernel + pf_reduce_kernel + pf_apply_kernel of one GPU in one launch
//                       of pf_blocks(n) <= 32 work-groups: every work-group adds the per-tile partials of its 256 particles,
//                       forms its weights and its partial sums in the association of the split kernels (same bits), and the
//                       LAST one to arrive (one acq_rel ticket at agent scope: a single-level hand-off) reduces the <= 32
//                       partials, normalises every weight and writes the results — optionally straight into page-locked host
//                       memory, so that no D2H copy follows.
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
                                            StageLds<ROUNDS>& s)
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
  constexpr uint32_t MASK = END_BIT >= 32 ? 0xffffffffu : ((1u << END_BIT) - 1u);
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
                                         nullptr, 0u, nullptr, s);
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
// the poses / weights / odometry factor. 256 threads per work-group.
__global__ __launch_bounds__(256) void stage_pack_kernel(StageArgs a, MinMaxOut mm, unsigned nb_lik, unsigned nb_beam)
{
  const unsigned b = blockIdx.x;
  if (b < nb_lik)
  {
    float mn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, mx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    unsigned cnt = 0;
    for (int i = static_cast<int>(b) * 256 + threadIdx.x; i < a.n_s; i += static_cast<int>(nb_lik) * 256)
    {
      const float4 p = make_float4(a.in_lik_xyz[3 * i], a.in_lik_xyz[3 * i + 1], a.in_lik_xyz[3 * i + 2], 0.f);
      a.raw_lik[i] = p;
      minmax_accumulate(p, mn, mx, cnt);
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
    for (int i = static_cast<int>(bb) * 256 + threadIdx.x; i < a.n_b; i += static_cast<int>(nb_beam) * 256)
      a.raw_beam[i] = make_float4(a.in_beam_xyz[3 * i], a.in_beam_xyz[3 * i + 1], a.in_beam_xyz[3 * i + 2],
                                  __uint_as_float(a.in_beam_origin ? a.in_beam_origin[i] : 0u));
    return;
  }
  const long long n_pose = a.in_pose ? 7ll * a.n_p : 0, n_w = a.in_w ? a.n_p : 0, n_e = a.in_extra ? a.n_p : 0;
  const long long total = n_pose + n_w + n_e;
  const long long stride = static_cast<long long>(gridDim.x - nb_lik - nb_beam) * 256;
  for (long long i = static_cast<long long>(b - nb_lik - nb_beam) * 256 + threadIdx.x; i < total; i += stride)
  {
    if (i < n_pose)
      a.d_pose[i] = a.in_pose[i];
    else if (i < n_pose + n_w)
      a.d_w[i - n_pose] = a.in_w[i - n_pose];
    else
      a.d_extra[i - n_pose - n_w] = a.in_extra[i - n_pose - n_w];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
struct TailArgs
{
  // the tiled likelihood kernel's per-(tile, particle) partials; null = lik[] / ratio[] hold the final values already
  const double* partial_sum;
  const unsigned* partial_cnt;
  int n_tiles, n_s;
  float* lik;
  float* ratio;
  float* beam;
  int beam_fill;   // 1 = an update without beam points: beam[] := 1 (beam.cpp:130-133), written here
  float* w;        // prior weights in, normalised weights out (untouched when every weight became 0: pf.h:274-278)
  const float* extra;
  int n;
  float* w_new;
  double* block_partials;  // [gridDim.x][4]
  unsigned* ticket;        // zero before the launch; left zero
  double* packed;          // [4] the reduced {sum w, sum w ln w, max ratio, -min ratio}
  float* stats4;           // device copy of {entropy, min ratio, max ratio, restored}
  // host-visible (page-locked, device-mapped) result arrays, each may be null
  float* h_stats4;
  float* h_w;
  float* h_lik;
  float* h_ratio;
  float* h_beam;
};

constexpr int PF_TAIL_MAX_BLOCKS = 32;  // n <= 8192: the last work-group normalises every weight itself

__global__ __launch_bounds__(PF_BLOCK) void pf_tail_kernel(TailArgs a)
{
  __shared__ double s_a[8][8][32];
  __shared__ unsigned s_n[8][8][32];
  __shared__ double sh[4][PF_BLOCK / 64];
  __shared__ double s_tot[4];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = blockIdx.x * PF_BLOCK + tid;  // this thread's particle (gridDim.x * PF_BLOCK >= n: one per thread)
  float lik_i = 0.f, ratio_i = 0.f;
  if (a.partial_sum)
  {
    // lik_finalize_kernel: 8 lane-slices per particle each walk every 8th tile, slice 0 adds the 8 sub-sums in order
    const int pl = tid & 31, slice = tid >> 5;
#pragma unroll 1
    for (int sub = 0; sub < 8; ++sub)
    {
      const int p = blockIdx.x * PF_BLOCK + sub * 32 + pl;
      double acc = 0.0;
      unsigned cn = 0;
      if (p < a.n)
        for (int tl = slice; tl < a.n_tiles; tl += 8)
        {
          acc += a.partial_sum[static_cast<size_t>(tl) * a.n + p];
          cn += a.partial_cnt[static_cast<size_t>(tl) * a.n + p];
        }
      s_a[sub][slice][pl] = acc;
      s_n[sub][slice][pl] = cn;
    }
    __syncthreads();
    const int sub = tid >> 5;
    double acc = s_a[sub][0][pl];
    unsigned cn = s_n[sub][0][pl];
#pragma unroll
    for (int k = 1; k < 8; ++k)
    {
      acc += s_a[sub][k][pl];
      cn += s_n[sub][k][pl];
    }
    lik_i = static_cast<float>(acc);
    ratio_i = static_cast<float>(cn) / static_cast<float>(a.n_s);
    if (i < a.n)
    {
      a.lik[i] = lik_i;
      a.ratio[i] = ratio_i;
    }
  }
  else if (i < a.n)
  {
    lik_i = a.lik[i];
    ratio_i = a.ratio[i];
  }
  // pf_partial_kernel
  double sum = 0.0, t = 0.0, rmax = 0.0, rneg = -1.0;
  if (i < a.n)
  {
    const float beam_i = a.beam_fill ? 1.0f : a.beam[i];
    if (a.beam_fill)
      a.beam[i] = 1.0f;
    float l = 1.0f;
    l *= beam_i;
    l *= lik_i;
    if (a.extra)
      l = l * a.extra[i];
    const float wn = a.w[i] * l;  // pf.h:258
    a.w_new[i] = wn;
    sum += static_cast<double>(wn);
    if (wn > 0.0f)
      t += static_cast<double>(wn) * log(static_cast<double>(wn));
    const double r = static_cast<double>(ratio_i);
    rmax = r > rmax ? r : rmax;
    rneg = -r > rneg ? -r : rneg;
    if (a.h_lik)
      a.h_lik[i] = lik_i;
    if (a.h_ratio)
      a.h_ratio[i] = ratio_i;
    if (a.h_beam)
      a.h_beam[i] = beam_i;
  }
  sum = wave_sum(sum);
  t = wave_sum(t);
  rmax = wave_max(rmax);
  rneg = wave_max(rneg);
  if (lane == 0)
  {
    sh[0][wave] = sum;
    sh[1][wave] = t;
    sh[2][wave] = rmax;
    sh[3][wave] = rneg;
  }
  __syncthreads();
  if (tid == 0)
  {
    double pa = 0, pb = 0, pc = sh[2][0], pd = sh[3][0];
    for (int k = 0; k < PF_BLOCK / 64; ++k)
    {
      pa += sh[0][k];
      pb += sh[1][k];
      pc = sh[2][k] > pc ? sh[2][k] : pc;
      pd = sh[3][k] > pd ? sh[3][k] : pd;
    }
    a.block_partials[4 * blockIdx.x + 0] = pa;
    a.block_partials[4 * blockIdx.x + 1] = pb;
    a.block_partials[4 * blockIdx.x + 2] = pc;
    a.block_partials[4 * blockIdx.x + 3] = pd;
    // the hand-off: everything this work-group wrote (ordered before this thread by the barrier above) is released at
    // agent scope with the arrival; the last arrival acquires every other work-group's writes with the same operation
    const unsigned arrived = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (arrived + 1 == gridDim.x) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last)
    return;
  // ---- the last work-group: pf_reduce_kernel (64 lanes stride the partials) ...
  const int nb = static_cast<int>(gridDim.x);
  if (wave == 0)
  {
    double ra = 0, rb = 0, rc = 0.0, rd = -1.0;
    for (int k = lane; k < nb; k += 64)
    {
      ra += a.block_partials[4 * k + 0];
      rb += a.block_partials[4 * k + 1];
      rc = a.block_partials[4 * k + 2] > rc ? a.block_partials[4 * k + 2] : rc;
      rd = a.block_partials[4 * k + 3] > rd ? a.block_partials[4 * k + 3] : rd;
    }
    ra = wave_sum(ra);
    rb = wave_sum(rb);
    rc = wave_max(rc);
    rd = wave_max(rd);
    if (lane == 0)
    {
      s_tot[0] = ra;
      s_tot[1] = rb;
      s_tot[2] = rc;
      s_tot[3] = rd;
      a.packed[0] = ra;
      a.packed[1] = rb;
      a.packed[2] = rc;
      a.packed[3] = rd;
      *a.ticket = 0u;  // for the next launch (kernel boundary orders it)
    }
  }
  __syncthreads();
  // ... and pf_apply_kernel over every particle
  const double S = s_tot[0];
  const float sum_f = static_cast<float>(S);
  const bool alive = sum_f > 0.0f;
  for (int k = tid; k < a.n; k += PF_BLOCK)
  {
    float wv;
    if (alive)
    {
      wv = a.w_new[k] / sum_f;
      a.w[k] = wv;
    }
    else
      wv = a.w[k];
    if (a.h_w)
      a.h_w[k] = wv;
  }
  if (tid == 0)
  {
    const float st0 = alive ? static_cast<float>(log(S) - s_tot[1] / S) : __builtin_nanf("");
    const float st1 = static_cast<float>(-s_tot[3]), st2 = static_cast<float>(s_tot[2]), st3 = alive ? 0.0f : 1.0f;
    if (a.stats4)
    {
      a.stats4[0] = st0;
      a.stats4[1] = st1;
      a.stats4[2] = st2;
      a.stats4[3] = st3;
    }
    if (a.h_stats4)
    {
      a.h_stats4[0] = st0;
      a.h_stats4[1] = st1;
      a.h_stats4[2] = st2;
      a.h_stats4[3] = st3;
    }
  }
}
}  // namespace mcl3dl
