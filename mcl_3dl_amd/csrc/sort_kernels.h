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
//   * (option, off) 2048 < n <= 32 768 with keys of at most 16 bits: ONE launch, every work-group holds the whole key
//     distribution in LDS (rs_sort16_kernel below; no gain: LDS atomics).
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

// ---- RS_ONE_LAUNCH_MAX < n <= RS_FULL_MAX: ONE launch per pass, no counting launches ------------------------------------
// What a pass needs from the other work-groups is how many elements with each digit every one of them owns — a property of
// the keys alone. For arrays this small every work-group can count that ITSELF: it reads all n keys (64 KB at 16 384; pass 0
// derives them from the points) and histograms them per owning work-group into LDS (n LDS atomics, ~2 us) — less than the
// ~5 us a separate counting launch costs between two dependent kernels. A 16 384-point scan is ordered by three launches
// instead of seven (keygen + count, then count / scatter pairs). Same ranks, same stable order as rs_pass_kernel.
constexpr int RS_FULL_MAX_BLOCKS = 32;
constexpr int RS_FULL_MAX = RS_THREADS * RS_FULL_MAX_BLOCKS;

template <int KEYMODE, bool FIRST, bool APPLY>
__global__ __launch_bounds__(RS_THREADS) void rs_pass_full_kernel(RsKeyGen kg, const uint32_t* __restrict__ keys_src,
                                                                  const uint32_t* __restrict__ vals_src,
                                                                  uint32_t* __restrict__ keys_dst, uint32_t* __restrict__ vals_dst,
                                                                  RsFinal fin, int n, int pass, int n_pass, uint32_t mask)
{
  __shared__ uint32_t cnt[RS_WAVES][256];
  __shared__ uint32_t dbase[256];
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t tab[RS_FULL_MAX_BLOCKS][256];
  __shared__ uint32_t s_part[4][256];
  const int nb = gridDim.x, b = blockIdx.x;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int shift = 8 * pass;
  for (int k = threadIdx.x; k < nb * 256; k += RS_THREADS)
    (&tab[0][0])[k] = 0u;
  __syncthreads();
  // every work-group's digit counts, from all n keys (element idx belongs to work-group idx / RS_THREADS)
  uint32_t key[1] = { 0xffffffffu }, val[1] = { 0u }, dst[1] = { 0u };
  const int own = b * RS_THREADS + w * 64 + lane;
  for (int idx = threadIdx.x; idx < n; idx += RS_THREADS)
  {
    const uint32_t k = FIRST ? rs_make_key<KEYMODE>(kg, static_cast<uint32_t>(idx)) : keys_src[idx];
    atomicAdd(&tab[idx / RS_THREADS][((k & mask) >> shift) & 255u], 1u);
    if (idx == own)
      key[0] = k;
  }
  const bool have = own < n;
  const uint32_t valid = have ? 1u : 0u;
  if (have)
    val[0] = FIRST ? ((KEYMODE == RS_KEY_ARRAY && kg.vals) ? kg.vals[own] : static_cast<uint32_t>(own)) : vals_src[own];
  // (rs_rank_pass starts with a barrier: tab is complete behind it)
  rs_rank_pass<1>(key, valid, 1, shift, mask, dst, cnt, dbase, wsum, [&](uint32_t total_d, uint32_t* ws) {
    (void)total_d;  // == tab[b][d]
    const int d = threadIdx.x & 255, q = threadIdx.x >> 8;
    uint32_t all = 0, ahead = 0;
    for (int bb = q; bb < nb; bb += 4)
    {
      const uint32_t x = tab[bb][d];
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
    return rs_scan256(all, ws) + ahead;
  });
  if (APPLY && pass + 1 == n_pass)
  {
    rs_apply<1>(fin, val, dst, valid);
    return;
  }
  if (have)
  {
    keys_dst[dst[0]] = key[0];
    vals_dst[dst[0]] = val[0];
  }
}

// ---- RS_ONE_LAUNCH_MAX < n <= RS16_MAX, keys of at most 16 bits (the Morton key of a likelihood scan): ONE launch -------
// Two launches per 8-bit pass is four or five dependent launches for a 16 384-point scan — 25 us of the host-buffer update's
// 39 us head, most of it the 4 - 5 us between dependent launches on an otherwise empty queue (profiles/r05o_timeline_8d_C2.txt).
// With 16 key bits the whole distribution fits in LDS: 65 536 sixteen-bit counters = 128 KB of the CU's 160. Every work-group
// (1024 consecutive elements each) counts ALL n keys itself and needs nothing from any other work-group:
//   position of element i  =  [elements with a smaller key]  +  [elements j < i with the same key]
//   * the elements of the work-groups AHEAD are counted first (LDS atomics, any order), so that when the work-group turns to its
//     own 1024 elements the counter of a key holds the second term's share of all earlier work-groups; its own elements are
//     then added wavefront by wavefront in index order (sixteen barriers), the lanes of one wavefront ordered by sixteen ballots
//     (the lanes that share all sixteen key bits) + v_mbcnt — the second term is complete;
//   * then the elements of the work-groups BEHIND, and an exclusive prefix over the 65 536 counters in place (every wavefront
//     4096 counters, 64 words per round, carried from round to round; the sixteen wavefront totals are added at look-up).
// Same stable order as the LSD passes. n <= 32 768 keeps every count and every prefix inside sixteen bits.
// MEASURED (profiles/r05p_sort16_one_launch.txt, profiles/sort16_phases.hip): 15.6 us for 16 384 elements, 7.7 of them the
// 15 360 LDS atomics of the two counting loops — LDS atomics retire one lane per clock — so the launch costs what the passes
// cost in a live update (0.2755 against 0.2742 ms); kept as an option ("sort_one_launch"), off.
constexpr int RS16_MAX = 32768;

// lanes of this wavefront that hold the same BITS-bit key (0 for a lane without an element)
template <int BITS>
__device__ __forceinline__ unsigned long long rs_match_bits(uint32_t key, bool valid)
{
  unsigned long long m = __ballot(valid);
#pragma unroll
  for (int b = 0; b < BITS; ++b)
  {
    const bool bit = (key >> b) & 1u;
    const unsigned long long bal = __ballot(bit);
    m &= bit ? bal : ~bal;
  }
  return valid ? m : 0ull;
}

// what a key needs besides its point, fetched once per thread (the Morton key's corner and shift are the same for every point)
template <int KEYMODE>
struct Rs16KeyCtx
{
  float mn[3];
  uint32_t drop;
  __device__ __forceinline__ void init(const RsKeyGen& kg)
  {
    if (KEYMODE == RS_KEY_MORTON)
    {
      mn[0] = kg.min3[0];
      mn[1] = kg.min3[1];
      mn[2] = kg.min3[2];
      drop = morton_key_drop(kg.min3);
    }
  }
  __device__ __forceinline__ uint32_t key(const RsKeyGen& kg, uint32_t idx, const float4& p) const
  {
    if (KEYMODE == RS_KEY_ARRAY)
      return kg.keys[idx];
    if (KEYMODE == RS_KEY_MORTON)
    {
      const float c[3] = { p.x, p.y, p.z };
      uint32_t q[3];
#pragma unroll
      for (int a = 0; a < 3; ++a)
      {
        const float f = (c[a] - mn[a]) * 4.0f;   // (morton_scan_key, cloud_keys.h: the same expression)
        q[a] = (f >= 0.f) ? (f < 1023.f ? static_cast<uint32_t>(f) : 1023u) : 0u;
      }
      return (spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2)) >> drop;
    }
    if (KEYMODE == RS_KEY_RANGE)
      return range_scan_key(p, kg.origins, kg.n_o, kg.error);
    return voxel_leaf_key(p, kg.vp);
  }
};

// counter of key k: half (k & 1) of word (k >> 1); words are placed so that the 64 lanes of a wavefront, each walking its own
// run of 32 consecutive words (the prefix below), hit 64 different banks
__device__ __forceinline__ uint32_t rs16_word(uint32_t j)
{
  return j ^ ((j >> 6) & 31u);
}

template <int KEYMODE, bool APPLY>
__global__ __launch_bounds__(RS_THREADS) void rs_sort16_kernel(RsKeyGen kg, RsFinal fin, uint32_t* __restrict__ keys_out,
                                                               uint32_t* __restrict__ vals_out, int n, uint32_t mask)
{
  __shared__ __attribute__((aligned(16))) uint32_t bins[32768];
  __shared__ uint32_t wtot[RS_WAVES];
  constexpr int BATCH = 8;  // loads in flight per thread while counting the other work-groups' elements
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  const int own0 = static_cast<int>(blockIdx.x) * RS_THREADS;
  const uint32_t m16 = mask & 0xffffu;
  Rs16KeyCtx<KEYMODE> kc;
  kc.init(kg);
  uint4* b4 = reinterpret_cast<uint4*>(bins);
  for (int k = t; k < 8192; k += RS_THREADS)
    b4[k] = make_uint4(0u, 0u, 0u, 0u);
  // this thread's own element
  const int own = own0 + t;
  const bool have = own < n;
  uint32_t kraw = 0, val = 0;
  if (have)
  {
    const float4 p = KEYMODE == RS_KEY_ARRAY ? make_float4(0.f, 0.f, 0.f, 0.f) : kg.pts[own];
    kraw = kc.key(kg, static_cast<uint32_t>(own), p);
    val = (KEYMODE == RS_KEY_ARRAY && kg.vals) ? kg.vals[own] : static_cast<uint32_t>(own);
  }
  const uint32_t k16 = kraw & m16;
  const uint32_t sh = (k16 & 1u) * 16u;
  const uint32_t own_word = rs16_word(k16 >> 1);
  // counts the elements [first, last) (whole chunks of 1024), BATCH loads in flight per thread
  const auto count_range = [&](int first, int last)
  {
    for (int base = first + t; base < last; base += BATCH * RS_THREADS)
    {
      float4 p[BATCH];
      uint32_t ka[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; ++u)
      {
        const int idx = base + u * RS_THREADS;
        if (idx < last)
        {
          if (KEYMODE == RS_KEY_ARRAY)
            ka[u] = kg.keys[idx];
          else
            p[u] = kg.pts[idx];
        }
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u)
      {
        const int idx = base + u * RS_THREADS;
        if (idx < last)
        {
          const uint32_t k = (KEYMODE == RS_KEY_ARRAY ? ka[u] : kc.key(kg, static_cast<uint32_t>(idx), p[u])) & m16;
          atomicAdd(&bins[rs16_word(k >> 1)], 1u << ((k & 1u) * 16u));
        }
      }
    }
  };
  __syncthreads();
  // ---- the work-groups ahead
  count_range(0, own0);
  // ---- this work-group's own elements, in index order
  const unsigned long long m = rs_match_bits<16>(k16, have);
  const uint32_t below = rs_lanes_below(m), total = static_cast<uint32_t>(__popcll(m));
  uint32_t tie = 0;
  __syncthreads();
  for (int ww = 0; ww < RS_WAVES; ++ww)
  {
    if (w == ww)
    {
      uint32_t base = 0;
      if (have)
        base = (bins[own_word] >> sh) & 0xffffu;
      // every lane of a key group has ISSUED its read before the group's highest lane adds to the same word (LDS operations
      // of one wavefront complete in order; the wave barrier keeps the compiler from moving either across the other)
      __builtin_amdgcn_wave_barrier();
      if (have && below + 1 == total)
        atomicAdd(&bins[own_word], total << sh);
      tie = base + below;
    }
    __syncthreads();
  }
  // ---- the work-groups behind
  count_range(min(own0 + RS_THREADS, n), n);
  __syncthreads();
  // ---- exclusive prefix in place: lane l of wavefront w owns the words [2048 w + 32 l, + 32) (64 counters), values relative to
  // the wavefront's first counter
  {
    const uint32_t first = 2048u * static_cast<uint32_t>(w) + 32u * static_cast<uint32_t>(lane);
    uint32_t sum = 0;
#pragma unroll 8
    for (int r = 0; r < 32; ++r)
    {
      const uint32_t x = bins[rs16_word(first + r)];
      sum += (x & 0xffffu) + (x >> 16);
    }
    uint32_t inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
    {
      const uint32_t o = __shfl_up(inc, off, 64);
      if (lane >= off)
        inc += o;
    }
    uint32_t run = inc - sum;
#pragma unroll 8
    for (int r = 0; r < 32; ++r)
    {
      const uint32_t x = bins[rs16_word(first + r)];
      const uint32_t lo = x & 0xffffu, hi = x >> 16;
      bins[rs16_word(first + r)] = run | ((run + lo) << 16);
      run += lo + hi;
    }
    if (lane == 63)
      wtot[w] = inc;
  }
  __syncthreads();
  if (!have)
    return;
  uint32_t pos = ((bins[own_word] >> sh) & 0xffffu) + tie;
  for (uint32_t ww = 0; ww < (k16 >> 12); ++ww)
    pos += wtot[ww];
  if (APPLY)
  {
    float4 q = fin.src_pts[val];
    if (fin.zero_w)
      q.w = 0.f;
    fin.out_pts[pos] = q;
    if (fin.out_perm)
      fin.out_perm[pos] = val;
  }
  else
  {
    keys_out[pos] = kraw;
    vals_out[pos] = val;
  }
}

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
