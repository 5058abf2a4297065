// sort_kernels.h — stable LSD radix sort of (key, value) pairs for the per-update cloud path (scan ordering, VoxelGrid),
// written for one gfx950 CU instead of calling a library: the arrays on this path hold 10^2 .. 10^5 elements, where a
// general device sort is a chain of 10-30 launches of a few microseconds each (round 2: hipcub::DeviceRadixSort expanded
// into ~28 merge-sort launches for a 65 536-point cloud).
//
//   * A work-group of 1024 threads (16 wavefronts) owns 1024 R consecutive elements, R per thread IN REGISTERS, arranged so
//     that wavefront w, round r, lane l holds element (w R + r) 64 + l: memory order = (wavefront, round, lane) order, so
//     stability is "earlier wavefront, earlier round, lower lane first".
//   * One pass = one 8-bit digit. Rank of an element = [elements with a smaller digit] + [same digit in earlier work-groups /
//     wavefronts] + [same digit in earlier rounds of its wavefront] + [same digit in lower lanes of its round]. The last term
//     comes from eight ballots (the lanes that share all eight digit bits) + v_mbcnt; the third from a per-wavefront counter row
//     in LDS that the round's highest lane of each digit group bumps (same wavefront: LDS operations complete in order); the
//     rest from a scan over the 16 x 256 counters (and the table of the other work-groups' totals).
//   * n <= 2048: ONE launch of one work-group does everything — derives the keys from the points (Morton / range / VoxelGrid
//     leaf), runs all passes (the exchange between passes goes through a global scratch array: written, __syncthreads(),
//     re-read past the L1) and writes the points in sorted order straight from the last pass (no separate gather). One CU
//     ranks ~1000 elements per microsecond, so larger arrays are spread over the chip:
//   * 2048 < n <= 524 288: n / 1024 (n / 4096 above 65 536) work-groups, two launches per pass: every work-group counts its
//     digits (rs_hist_kernel; the first count rides on the kernel that makes the keys), then every work-group reads the whole
//     table of counts and scatters (rs_pass_kernel). No work-group ever waits for another one inside a launch. (Counting
//     the NEXT pass's digits from inside the scatter — one device-scope atomic per element — was measured first: the atomics
//     alone cost 7 us per pass, more than the extra launch.)
//   * Larger arrays (whole maps) go to rocprim::radix_sort_pairs (host_cloud.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cloud_keys.h"

namespace mcl3dl
{
constexpr int RS_THREADS = 1024;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_ONE_LAUNCH_ROUNDS = 2;                           // the one-launch form: up to 2048 elements
constexpr int RS_ONE_LAUNCH_MAX = RS_THREADS * RS_ONE_LAUNCH_ROUNDS;
constexpr int RS_MAX_ELEMS = 524288;                              // above: rocprim

enum
{
  RS_KEY_ARRAY = 0,   // keys (and values, or the index when vals == nullptr) come from arrays
  RS_KEY_MORTON = 1,  // MCL3DL_MORTON_BITS-bit Morton key of a likelihood scan point (cloud_keys.h, api_core.inl:order_scan)
  RS_KEY_RANGE = 2,   // squared range of a beam point from its origin, as float bits
  RS_KEY_LEAF = 3     // pcl::VoxelGrid leaf index
};

struct RsKeyGen
{
  const uint32_t* keys;
  const uint32_t* vals;
  const float4* pts;
  const float* min3;       // RS_KEY_MORTON: the cloud's {min x, y, z, max x, y, z} (device memory)
  const float4* origins;   // RS_KEY_RANGE
  uint32_t n_o;
  int* error;              // RS_KEY_RANGE: set to 2 when a point names an origin that does not exist
  VoxelGridParams vp;      // RS_KEY_LEAF
};

// What the last pass writes: (key, value) pairs, or — APPLY — the points themselves in sorted order.
struct RsFinal
{
  const float4* src_pts;
  float4* out_pts;
  uint32_t* out_perm;  // may be null
  int zero_w;          // 1: the w component of the output is 0 (likelihood scan), 0: kept (beam scan: origin id)
};

template <int KEYMODE>
__device__ __forceinline__ uint32_t rs_make_key(const RsKeyGen& kg, uint32_t idx)
{
  if (KEYMODE == RS_KEY_ARRAY)
    return kg.keys[idx];
  const float4 p = kg.pts[idx];
  if (KEYMODE == RS_KEY_MORTON)
    return morton_scan_key(p, kg.min3);
  if (KEYMODE == RS_KEY_RANGE)
    return range_scan_key(p, kg.origins, kg.n_o, kg.error);
  return voxel_leaf_key(p, kg.vp);
}

// lanes of this wavefront that hold the same 8-bit digit (0 for a lane without an element)
__device__ __forceinline__ unsigned long long rs_match(uint32_t digit, bool valid)
{
  unsigned long long m = __ballot(valid);
#pragma unroll
  for (int b = 0; b < 8; ++b)
  {
    const bool bit = (digit >> b) & 1u;
    const unsigned long long bal = __ballot(bit);
    m &= bit ? bal : ~bal;
  }
  return valid ? m : 0ull;
}

__device__ __forceinline__ uint32_t rs_lanes_below(unsigned long long m)
{
  return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
}

// exclusive scan of one value per thread over threads 0..255 (the first four wavefronts); every thread of the work-group
// must call it (two barriers inside). `wsum` = 4 words of LDS.
__device__ __forceinline__ uint32_t rs_scan256(uint32_t v, uint32_t* wsum)
{
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1)
  {
    const uint32_t o = __shfl_up(inc, off, 64);
    if (lane >= off)
      inc += o;
  }
  if (w < 4 && lane == 63)
    wsum[w] = inc;
  __syncthreads();
  uint32_t base = 0;
  for (int k = 0; k < 4; ++k)
    if (k < w)
      base += wsum[k];
  __syncthreads();
  return base + inc - v;
}

// Ranks of one pass inside a work-group. On return dst[r] = position of element r among the work-group's elements IF the
// work-group's elements with digit d started at dbase[d]: dst = dbase[d] + (earlier wavefronts) + (earlier rounds) + (lower
// lanes). The caller provides dbase through `digit_base` (called by every thread — it may contain barriers — with the
// work-group's total for digit threadIdx.x; what threads 0..255 return is where their digit starts).
template <int ROUNDS, typename DigitBase>
__device__ __forceinline__ void rs_rank_pass(const uint32_t (&key)[ROUNDS], uint32_t valid, int rounds, int shift, uint32_t mask,
                                             uint32_t (&dst)[ROUNDS], uint32_t (*cnt)[256], uint32_t* dbase, uint32_t* wsum,
                                             DigitBase digit_base)
{
  const int w = threadIdx.x >> 6;
  for (int k = threadIdx.x; k < RS_WAVES * 256; k += RS_THREADS)
    (&cnt[0][0])[k] = 0u;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r)
  {
    if (r < rounds)
    {
      const uint32_t d = ((key[r] & mask) >> shift) & 255u;
      const bool have = (valid >> r) & 1u;
      const unsigned long long m = rs_match(d, have);
      const uint32_t below = rs_lanes_below(m), total = static_cast<uint32_t>(__popcll(m));
      uint32_t base = 0;
      if (have)
        base = cnt[w][d];
      // every lane of a digit group has ISSUED its read before the group's highest lane writes the same word: LDS operations
      // of one wavefront complete in order, and the wave barrier keeps the compiler from moving either across the other
      __builtin_amdgcn_wave_barrier();
      if (have && below + 1 == total)
        cnt[w][d] = base + total;
      dst[r] = base + below;
    }
  }
  __syncthreads();
  uint32_t total_d = 0;
  if (threadIdx.x < 256)
  {
    for (int ww = 0; ww < RS_WAVES; ++ww)
    {
      const uint32_t x = cnt[ww][threadIdx.x];
      cnt[ww][threadIdx.x] = total_d;
      total_d += x;
    }
  }
  const uint32_t start = digit_base(total_d, wsum);
  if (threadIdx.x < 256)
    dbase[threadIdx.x] = start;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r)
    if (r < rounds && ((valid >> r) & 1u))
    {
      const uint32_t d = ((key[r] & mask) >> shift) & 255u;
      dst[r] += dbase[d] + cnt[w][d];
    }
}

template <int ROUNDS>
__device__ __forceinline__ void rs_apply(const RsFinal& fin, const uint32_t (&val)[ROUNDS], const uint32_t (&dst)[ROUNDS],
                                         uint32_t valid)
{
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r)
    if ((valid >> r) & 1u)
    {
      float4 q = fin.src_pts[val[r]];
      if (fin.zero_w)
        q.w = 0.f;
      fin.out_pts[dst[r]] = q;
      if (fin.out_perm)
        fin.out_perm[dst[r]] = val[r];
    }
}

// ---- n <= RS_ONE_LAUNCH_MAX: the whole sort in one launch of one work-group --------------------------------------------
// scratch: keys_x / vals_x (n entries each). Plain pairs land in keys_out / vals_out.
template <int KEYMODE, bool APPLY>
__global__ __launch_bounds__(RS_THREADS) void rs_sort_block_kernel(RsKeyGen kg, RsFinal fin, uint32_t* __restrict__ keys_out,
                                                                   uint32_t* __restrict__ vals_out,
                                                                   uint32_t* __restrict__ keys_x, uint32_t* __restrict__ vals_x,
                                                                   int n, int n_pass, uint32_t mask)
{
  constexpr int ROUNDS = RS_ONE_LAUNCH_ROUNDS;
  __shared__ uint32_t cnt[RS_WAVES][256];
  __shared__ uint32_t dbase[256];
  __shared__ uint32_t wsum[4];
  const int rounds = (n + RS_THREADS - 1) / RS_THREADS;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t key[ROUNDS], val[ROUNDS], dst[ROUNDS];
  uint32_t valid = 0;  // bit r: this thread holds an element in round r
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r)
  {
    const int idx = (w * rounds + r) * 64 + lane;
    const bool have = r < rounds && idx < n;
    valid |= have ? (1u << r) : 0u;
    key[r] = 0xffffffffu;
    val[r] = 0;
    dst[r] = 0;
    if (have)
    {
      key[r] = rs_make_key<KEYMODE>(kg, static_cast<uint32_t>(idx));
      val[r] = (KEYMODE == RS_KEY_ARRAY && kg.vals) ? kg.vals[idx] : static_cast<uint32_t>(idx);
    }
  }
  for (int p = 0; p < n_pass; ++p)
  {
    rs_rank_pass<ROUNDS>(key, valid, rounds, 8 * p, mask, dst, cnt, dbase, wsum,
                         [](uint32_t total_d, uint32_t* ws) { return rs_scan256(total_d, ws); });
    const bool last = p + 1 == n_pass;
    if (last && APPLY)
    {
      rs_apply<ROUNDS>(fin, val, dst, valid);
      return;
    }
    // pass p writes the array the LAST pass must leave the result in when (n_pass - 1 - p) is even
    const bool to_out = ((n_pass - 1 - p) & 1) == 0;
    uint32_t* kd = to_out ? keys_out : keys_x;
    uint32_t* vd = to_out ? vals_out : vals_x;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
      if ((valid >> r) & 1u)
      {
        kd[dst[r]] = key[r];
        vd[dst[r]] = val[r];
      }
    if (last)
      return;
    __syncthreads();  // work-group scope: the stores above are visible to the loads below ...
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
      if ((valid >> r) & 1u)
      {
        const int idx = (w * rounds + r) * 64 + lane;
        // ... which go past this CU's L1 (an earlier pass read the same addresses: no stale line can answer)
        key[r] = __hip_atomic_load(kd + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        val[r] = __hip_atomic_load(vd + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    // (this array is written again two passes on, behind the barriers of the next pass's ranking)
  }
}

// ---- RS_ONE_LAUNCH_MAX < n <= RS_MAX_ELEMS: one launch per pass --------------------------------------------------------
// A work-group owns `elems` (= 1024 x ROUNDS of the pass kernels) consecutive elements.
// table[p][b][d] = number of elements with digit d (of pass p) that work-group b owns at the start of pass p.
// Launch 0: keys + values into arrays, table[0] counted.
template <int KEYMODE>
__global__ __launch_bounds__(RS_THREADS) void rs_keygen_count_kernel(RsKeyGen kg, uint32_t* __restrict__ keys,
                                                                     uint32_t* __restrict__ vals, uint32_t* __restrict__ table,
                                                                     int n, int elems, uint32_t mask)
{
  __shared__ uint32_t hist[256];
  const int b = blockIdx.x;
  if (threadIdx.x < 256)
    hist[threadIdx.x] = 0u;
  __syncthreads();
  for (int k = threadIdx.x; k < elems; k += RS_THREADS)
  {
    const long long idx = static_cast<long long>(b) * elems + k;
    if (idx < n)
    {
      const uint32_t key = rs_make_key<KEYMODE>(kg, static_cast<uint32_t>(idx));
      keys[idx] = key;
      vals[idx] = (KEYMODE == RS_KEY_ARRAY && kg.vals) ? kg.vals[idx] : static_cast<uint32_t>(idx);
      atomicAdd(&hist[key & mask & 255u], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x < 256)
  {
    table[static_cast<size_t>(b) * 256 + threadIdx.x] = hist[threadIdx.x];
  }
}

// table[pass][b][d] for a later pass: the digits of the elements work-group b owns now
__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const uint32_t* __restrict__ keys, uint32_t* __restrict__ table, int n,
                                                             int elems, int pass, uint32_t mask)
{
  __shared__ uint32_t hist[256];
  const int nb = gridDim.x, b = blockIdx.x;
  if (threadIdx.x < 256)
    hist[threadIdx.x] = 0u;
  __syncthreads();
  for (int k = threadIdx.x; k < elems; k += RS_THREADS)
  {
    const long long idx = static_cast<long long>(b) * elems + k;
    if (idx < n)
      atomicAdd(&hist[((keys[idx] & mask) >> (8 * pass)) & 255u], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 256)
    table[(static_cast<size_t>(pass) * nb + b) * 256 + threadIdx.x] = hist[threadIdx.x];
}

template <bool APPLY, int ROUNDS>
__global__ __launch_bounds__(RS_THREADS) void rs_pass_kernel(const uint32_t* __restrict__ keys_src,
                                                             const uint32_t* __restrict__ vals_src,
                                                             uint32_t* __restrict__ keys_dst, uint32_t* __restrict__ vals_dst,
                                                             RsFinal fin, const uint32_t* __restrict__ table, int n, int pass, int n_pass,
                                                             uint32_t mask)
{
  constexpr int ELEMS = RS_THREADS * ROUNDS;
  __shared__ uint32_t cnt[RS_WAVES][256];
  __shared__ uint32_t dbase[256];
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t s_part[4][256];
  const int nb = gridDim.x, b = blockIdx.x;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool last = pass + 1 == n_pass;
  uint32_t key[ROUNDS], val[ROUNDS], dst[ROUNDS];
  uint32_t valid = 0;
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r)
  {
    const long long idx = static_cast<long long>(b) * ELEMS + (w * ROUNDS + r) * 64 + lane;
    const bool have = idx < n;
    valid |= have ? (1u << r) : 0u;
    key[r] = have ? keys_src[idx] : 0xffffffffu;
    val[r] = have ? vals_src[idx] : 0u;
    dst[r] = 0;
  }
  const uint32_t* tab = table + static_cast<size_t>(pass) * nb * 256;
  rs_rank_pass<ROUNDS>(key, valid, ROUNDS, 8 * pass, mask, dst, cnt, dbase, wsum, [&](uint32_t total_d, uint32_t* ws) {
    // where this work-group's elements with digit d go: behind every smaller digit of every work-group and behind digit
    // d of the work-groups ahead of this one. Four threads per digit each add up a quarter of the work-groups' rows.
    (void)total_d;  // == tab[b][d]
    const int d = threadIdx.x & 255, q = threadIdx.x >> 8;
    uint32_t all = 0, ahead = 0;
    for (int bb = q; bb < nb; bb += 4)
    {
      const uint32_t x = tab[static_cast<size_t>(bb) * 256 + d];
      all += x;
      ahead += bb < b ? x : 0u;
    }
    s_part[q][d] = all;
    __syncthreads();
    if (q == 0)
      all = s_part[0][d] + s_part[1][d] + s_part[2][d] + s_part[3][d];
    __syncthreads();
    s_part[q][d] = ahead;
    __syncthreads();
    if (q == 0)
      ahead = s_part[0][d] + s_part[1][d] + s_part[2][d] + s_part[3][d];
    return rs_scan256(all, ws) + ahead;   // (threads 256.. scan garbage: only threads 0..255 are used)
  });
  if (last && APPLY)
  {
    rs_apply<ROUNDS>(fin, val, dst, valid);
    return;
  }
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r)
    if ((valid >> r) & 1u)
    {
      keys_dst[dst[r]] = key[r];
      vals_dst[dst[r]] = val[r];
    }
}

// (Rounds 4-5 also carried a one-launch-per-pass form in which every work-group counted every work-group's digits itself, and
// a one-launch sort of 16-bit keys with the whole key distribution in LDS: both correct, both bound by LDS atomics that retire
// one lane a clock, neither faster than the launches above — profiles/r04d_time8d_C2.json, r05p_sort16_one_launch.txt. Gone in
// round 6 with their options.)

// ---- the two ends of a sort that goes to rocprim (more than RS_MAX_ELEMS elements) -------------------------
template <int KEYMODE>
__global__ void rs_keygen_kernel(RsKeyGen kg, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, long long n)
{
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  keys[i] = rs_make_key<KEYMODE>(kg, static_cast<uint32_t>(i));
  vals[i] = (KEYMODE == RS_KEY_ARRAY && kg.vals) ? kg.vals[i] : static_cast<uint32_t>(i);
}

__global__ void rs_apply_kernel(RsFinal fin, const uint32_t* __restrict__ vals, long long n)
{
  const long long k = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k >= n)
    return;
  float4 q = fin.src_pts[vals[k]];
  if (fin.zero_w)
    q.w = 0.f;
  fin.out_pts[k] = q;
  if (fin.out_perm)
    fin.out_perm[k] = vals[k];
}
}  // namespace mcl3dl
