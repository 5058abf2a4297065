// cloud_kernels.h — point-cloud preparation on the device: the steps either side of the
// measurement update that the reference runs through PCL on one core.
//
//   cloud_decode_kernel      sensor_msgs/PointCloud2 bytes -> {x, y, z, label} (mcl_3dl::fromROSMsg, point_conversion.h:64-92)
//   vg_*                     pcl::VoxelGrid<PointXYZIL>::filter as the node configures it (setLeafSize only:
//                            src/mcl_3dl.cpp:363-367 scan, :1155-1158 map, :147-151 map update) — centroid per occupied leaf
//   clip_flag_kernel         the clip predicate of both models' filter() (likelihood.cpp:84-93, beam.cpp:103-112)
//   compact / gather         std::remove_if + erase (order kept) and the sampler's `output->push_back(pc->points[i])`
//   order_*                  the scan ordering upload_scan does on the host (Morton order / range order), same keys,
//                            same stable order, so that both entry paths give bit-identical results
//   match_split_kernel       the matched / unmatched classification of src/mcl_3dl.cpp:776-789
//
// Points are float4 {x, y, z, label bits}. PointXYZIL's intensity is not carried: nothing on the measurement path reads
// it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cloud_keys.h"
#include "device_math.h"
#include "map_structs.h"
#pragma clang fp contract(off)

namespace mcl3dl
{

__device__ inline uint32_t load_u32_unaligned(const uint8_t* p)
{
  return static_cast<uint32_t>(p[0]) | (static_cast<uint32_t>(p[1]) << 8) | (static_cast<uint32_t>(p[2]) << 16) |
         (static_cast<uint32_t>(p[3]) << 24);
}

// One thread per point of a little-endian PointCloud2 buffer: x / y / z are FLOAT32 fields at the given byte offsets,
// label a UINT32 field (off_label < 0: the message has none -> 0, what pcl::fromROSMsg leaves in PointXYZL::label).
__global__ void cloud_decode_kernel(const uint8_t* __restrict__ data, long long n, uint32_t point_step, int off_x,
                                    int off_y, int off_z, int off_label, float4* __restrict__ out)
{
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const uint8_t* p = data + static_cast<size_t>(i) * point_step;
  const uint32_t lab = off_label >= 0 ? load_u32_unaligned(p + off_label) : 0u;
  out[i] = make_float4(__uint_as_float(load_u32_unaligned(p + off_x)), __uint_as_float(load_u32_unaligned(p + off_y)),
                       __uint_as_float(load_u32_unaligned(p + off_z)), __uint_as_float(lab));
}

// xyz + label arrays (the plain C-ABI form) -> float4
__global__ void cloud_pack_kernel(const float* __restrict__ xyz, const uint32_t* __restrict__ label, long long n,
                                  float4* __restrict__ out)
{
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  out[i] = make_float4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], __uint_as_float(label ? label[i] : 0u));
}

__global__ void cloud_unpack_kernel(const float4* __restrict__ in, long long n, float* __restrict__ xyz,
                                    uint32_t* __restrict__ label)
{
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const float4 p = in[i];
  xyz[3 * i] = p.x;
  xyz[3 * i + 1] = p.y;
  xyz[3 * i + 2] = p.z;
  if (label)
    label[i] = __float_as_uint(p.w);
}

// ---- min / max over the finite points (pcl::getMinMax3D), fused into the kernel that produces the points ------------------
// Every thread brings its running min / max / finite count; the work-group reduces them, publishes one partial, and the
// LAST work-group to arrive (atomic ticket, release / acquire fences at agent scope — nobody waits for anybody) folds the
// partials into out6 = {min xyz, max xyz} and out_cnt, then resets the ticket for the next launch.
struct MinMaxOut
{
  float* block_out;            // [6 x gridDim.x]
  unsigned* block_cnt;         // [gridDim.x]
  unsigned* ticket;            // zero before the first launch; left zero
  float* out6;
  unsigned long long* out_cnt;
};

__device__ inline void minmax_accumulate(const float4 p, float (&mn)[3], float (&mx)[3], unsigned& cnt)
{
  if (!finite3(p))
    return;
  ++cnt;
  mn[0] = fminf(mn[0], p.x);
  mn[1] = fminf(mn[1], p.y);
  mn[2] = fminf(mn[2], p.z);
  mx[0] = fmaxf(mx[0], p.x);
  mx[1] = fmaxf(mx[1], p.y);
  mx[2] = fmaxf(mx[2], p.z);
}

__device__ inline void wave_minmax(float (&mn)[3], float (&mx)[3], unsigned long long& cnt)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
  {
#pragma unroll
    for (int a = 0; a < 3; ++a)
    {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64));
    }
    cnt += __shfl_xor(cnt, off, 64);
  }
}

// called by every thread of the work-group (blockDim.x a multiple of 64, at most 1024)
// bid / n_blocks: this work-group's slot and the number of work-groups that take part (default: the whole grid)
__device__ inline void block_minmax_finish(float (&mn)[3], float (&mx)[3], unsigned cnt32, const MinMaxOut& o, unsigned bid,
                                           unsigned n_blocks)
{
  __shared__ float s_mm[6][16];
  __shared__ unsigned long long s_cnt[16];
  __shared__ int s_last;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  unsigned long long cnt = cnt32;
  wave_minmax(mn, mx, cnt);
  if (lane == 0)
  {
    for (int a = 0; a < 3; ++a)
    {
      s_mm[a][w] = mn[a];
      s_mm[3 + a][w] = mx[a];
    }
    s_cnt[w] = cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    float r[6];
    unsigned long long c = 0;
    for (int a = 0; a < 3; ++a)
    {
      r[a] = 3.0e38f;
      r[3 + a] = -3.0e38f;
    }
    for (int k = 0; k < nw; ++k)
    {
      for (int a = 0; a < 3; ++a)
      {
        r[a] = fminf(r[a], s_mm[a][k]);
        r[3 + a] = fmaxf(r[3 + a], s_mm[3 + a][k]);
      }
      c += s_cnt[k];
    }
    for (int a = 0; a < 6; ++a)
      o.block_out[6 * bid + a] = r[a];
    o.block_cnt[bid] = static_cast<unsigned>(c);
    __threadfence();  // release: the partial is out of this XCD's L2 before the ticket moves
    const unsigned t = atomicAdd(o.ticket, 1u);
    s_last = (t + 1 == n_blocks) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last || w != 0)
    return;
  __threadfence();  // acquire: the other work-groups' partials
  float fmn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, fmx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
  unsigned long long total = 0;
  for (unsigned b = lane; b < n_blocks; b += 64)
  {
    for (int a = 0; a < 3; ++a)
    {
      const uint32_t lo = __hip_atomic_load(reinterpret_cast<const uint32_t*>(o.block_out) + 6 * b + a, __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t hi = __hip_atomic_load(reinterpret_cast<const uint32_t*>(o.block_out) + 6 * b + 3 + a, __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT);
      fmn[a] = fminf(fmn[a], __uint_as_float(lo));
      fmx[a] = fmaxf(fmx[a], __uint_as_float(hi));
    }
    total += __hip_atomic_load(o.block_cnt + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  wave_minmax(fmn, fmx, total);
  if (lane == 0)
  {
    for (int a = 0; a < 3; ++a)
    {
      o.out6[a] = fmn[a];
      o.out6[3 + a] = fmx[a];
    }
    *o.out_cnt = total;
    *o.ticket = 0u;
  }
}

__device__ inline void block_minmax_finish(float (&mn)[3], float (&mx)[3], unsigned cnt32, const MinMaxOut& o)
{
  block_minmax_finish(mn, mx, cnt32, o, blockIdx.x, gridDim.x);
}

// xyz + label arrays -> float4 cloud + its min / max (one launch)
__global__ __launch_bounds__(256) void cloud_pack_minmax_kernel(const float* __restrict__ xyz, const uint32_t* __restrict__ label,
                                                                long long n, float4* __restrict__ out, MinMaxOut mm)
{
  float mn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, mx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
  unsigned cnt = 0;
  for (long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<long long>(gridDim.x) * 256)
  {
    const float4 p = make_float4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], __uint_as_float(label ? label[i] : 0u));
    out[i] = p;
    minmax_accumulate(p, mn, mx, cnt);
  }
  block_minmax_finish(mn, mx, cnt, mm);
}

// PointCloud2 bytes -> float4 cloud + its min / max (one launch)
__global__ __launch_bounds__(256) void cloud_decode_minmax_kernel(const uint8_t* __restrict__ data, long long n, uint32_t point_step,
                                                                  int off_x, int off_y, int off_z, int off_label,
                                                                  float4* __restrict__ out, MinMaxOut mm)
{
  float mn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, mx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
  unsigned cnt = 0;
  for (long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<long long>(gridDim.x) * 256)
  {
    const uint8_t* p = data + static_cast<size_t>(i) * point_step;
    const uint32_t lab = off_label >= 0 ? load_u32_unaligned(p + off_label) : 0u;
    const float4 q = make_float4(__uint_as_float(load_u32_unaligned(p + off_x)), __uint_as_float(load_u32_unaligned(p + off_y)),
                                 __uint_as_float(load_u32_unaligned(p + off_z)), __uint_as_float(lab));
    out[i] = q;
    minmax_accumulate(q, mn, mx, cnt);
  }
  block_minmax_finish(mn, mx, cnt, mm);
}

// min / max of a device cloud that is already there
__global__ __launch_bounds__(256) void cloud_minmax_kernel(const float4* __restrict__ pts, long long n, MinMaxOut mm)
{
  float mn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, mx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
  unsigned cnt = 0;
  for (long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<long long>(gridDim.x) * 256)
    minmax_accumulate(pts[i], mn, mx, cnt);
  block_minmax_finish(mn, mx, cnt, mm);
}

// ---- ordered compaction in two launches --------------------------------------------------------------------------------
// A work-group of 256 threads owns CP_BLOCK consecutive elements, four per thread (a 65 536-point cloud = 64 work-groups:
// these are latency-bound passes, spread them). Launch 1 counts what each work-group keeps; launch 2 recomputes the
// predicate, adds up the counts of the work-groups ahead of it (no waiting: they are the previous launch's output) and
// writes its survivors in order.
constexpr int CP_THREADS = 256;
constexpr int CP_BLOCK = 4 * CP_THREADS;

// exclusive scan of one value per thread over the work-group (up to 1024 threads); `total` = sum over the work-group
__device__ inline uint32_t block_scan_excl(uint32_t v, uint32_t& total)
{
  __shared__ uint32_t s_w[16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1)
  {
    const uint32_t o = __shfl_up(inc, off, 64);
    if (lane >= off)
      inc += o;
  }
  __syncthreads();  // s_w may still be read by a previous call
  if (lane == 63)
    s_w[w] = inc;
  __syncthreads();
  uint32_t base = 0, all = 0;
  const int nw = blockDim.x >> 6;
  for (int k = 0; k < nw; ++k)
  {
    const uint32_t x = s_w[k];
    base += k < w ? x : 0u;
    all += x;
  }
  total = all;
  return base + inc - v;
}

// sum of counts[0 .. blockIdx.x) (computed by the first wavefront, returned to every thread)
__device__ inline uint32_t blocks_ahead(const uint32_t* __restrict__ counts)
{
  __shared__ uint32_t s_ahead;
  if (threadIdx.x < 64)
  {
    uint32_t a = 0;
    for (unsigned b = threadIdx.x; b < blockIdx.x; b += 64)
      a += counts[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
      a += __shfl_xor(a, off, 64);
    if (threadIdx.x == 0)
      s_ahead = a;
  }
  __syncthreads();
  const uint32_t r = s_ahead;
  __syncthreads();
  return r;
}

// -- VoxelGrid: sorted (leaf key, point index) pairs -> one centroid per leaf, in ascending leaf order
__device__ inline bool vg_is_head(const uint32_t* __restrict__ key, long long i, long long n_finite)
{
  return i < n_finite && (i == 0 || key[i] != key[i - 1]);
}

__global__ __launch_bounds__(CP_THREADS) void vg_head_count_kernel(const uint32_t* __restrict__ key, long long n_finite,
                                                                   uint32_t* __restrict__ block_heads)
{
  const long long base = static_cast<long long>(blockIdx.x) * CP_BLOCK + threadIdx.x * 4;
  uint32_t c = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    c += vg_is_head(key, base + k, n_finite) ? 1u : 0u;
  uint32_t total;
  block_scan_excl(c, total);
  if (threadIdx.x == 0)
    block_heads[blockIdx.x] = total;
}

// pcl::CentroidPoint<PointXYZIL> of one leaf: xyz summed as floats in the order the points arrive (the stable sort keeps
// the input order inside a leaf; PCL's own std::sort leaves that order unspecified), divided by the count; label = the most
// frequent one, the smallest on a tie (AccumulatorLabel walks a std::map with a strict `>`).
// The work-group's CP_BLOCK sorted entries sit in LDS (keys + gathered points: every global load of the leaf sums was
// issued up front, in parallel); a leaf that runs past the work-group's last entry is finished from global memory.
struct LeafEntries
{
  const uint32_t* s_key;  // LDS, CP_BLOCK entries from sorted position `base`
  const float4* s_pt;
  long long base, n_finite;
  const float4* pts;  // global
  const uint32_t* key;
  const uint32_t* val;
  __device__ inline uint32_t key_at(long long i) const
  {
    return i < base + CP_BLOCK ? s_key[i - base] : key[i];
  }
  __device__ inline float4 pt_at(long long i) const
  {
    return i < base + CP_BLOCK ? s_pt[i - base] : pts[val[i]];
  }
};

__device__ inline float4 vg_leaf_centroid(const LeafEntries& le, long long s)
{
  const uint32_t k0 = le.key_at(s);
  long long e = s + 1;
  while (e < le.n_finite && le.key_at(e) == k0)
    ++e;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  const uint32_t first_label = __float_as_uint(le.pt_at(s).w);
  bool uniform = true;
  for (long long k = s; k < e; ++k)
  {
    const float4 p = le.pt_at(k);
    sx += p.x;
    sy += p.y;
    sz += p.z;
    uniform = uniform && __float_as_uint(p.w) == first_label;
  }
  uint32_t best_label = first_label;
  if (!uniform)
  {
    uint32_t best_count = 0;
    for (long long k = s; k < e; ++k)
    {
      const uint32_t lab = __float_as_uint(le.pt_at(k).w);
      uint32_t c = 0;
      for (long long j = s; j < e; ++j)
        c += __float_as_uint(le.pt_at(j).w) == lab ? 1u : 0u;
      if (c > best_count || (c == best_count && lab < best_label))
      {
        best_count = c;
        best_label = lab;
      }
    }
  }
  const float cnt = static_cast<float>(e - s);
  return make_float4(sx / cnt, sy / cnt, sz / cnt, __uint_as_float(best_label));
}

// the thread that sits on the first entry of a leaf writes the leaf's centroid; the last work-group leaves the number of
// leaves in *n_leaves
__global__ __launch_bounds__(CP_THREADS) void vg_centroid_compact_kernel(const float4* __restrict__ pts,
                                                                         const uint32_t* __restrict__ key,
                                                                         const uint32_t* __restrict__ val, long long n_finite,
                                                                         const uint32_t* __restrict__ block_heads,
                                                                         float4* __restrict__ out, uint32_t* __restrict__ n_leaves)
{
  __shared__ uint32_t s_key[CP_BLOCK];
  __shared__ float4 s_pt[CP_BLOCK];
  const long long block_base = static_cast<long long>(blockIdx.x) * CP_BLOCK;
  // coalesced: thread t stages entries t, t + 256, t + 512, t + 768 of the work-group's range
#pragma unroll
  for (int k = 0; k < 4; ++k)
  {
    const int j = threadIdx.x + k * CP_THREADS;
    const long long i = block_base + j;
    if (i < n_finite)
    {
      s_key[j] = key[i];
      s_pt[j] = pts[val[i]];
    }
  }
  __syncthreads();
  const long long base = block_base + threadIdx.x * 4;
  bool head[4];
  uint32_t c = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k)
  {
    const long long i = base + k;
    head[k] = i < n_finite && (i == 0 || (i > block_base ? s_key[i - block_base - 1] : key[i - 1]) != s_key[i - block_base]);
    c += head[k] ? 1u : 0u;
  }
  const uint32_t ahead = blocks_ahead(block_heads);
  uint32_t total;
  uint32_t leaf = ahead + block_scan_excl(c, total);
  const LeafEntries le{ s_key, s_pt, block_base, n_finite, pts, key, val };
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (head[k])
      out[leaf++] = vg_leaf_centroid(le, base + k);
  if (blockIdx.x + 1 == gridDim.x && threadIdx.x == 0)
    *n_leaves = ahead + total;
}

// -- the clip filters of both models (likelihood.cpp:84-93, beam.cpp:103-112), one pass over the cloud for the two of them.
// The reference's lambda returns true (= erase) for x^2 + y^2 > far^2, < near^2, z < z_min or z_max < z; a NaN makes
// every comparison false, so such a point is kept there too.
struct ClipParams
{
  float near_sq, far_sq, z_min, z_max;
  int enabled;
};

__device__ inline bool clip_keeps(const float4 p, const ClipParams& c)
{
  const float r2 = p.x * p.x + p.y * p.y;
  const bool erase = r2 > c.far_sq || r2 < c.near_sq || p.z < c.z_min || c.z_max < p.z;
  return !erase;
}

// n = *n_dev when n_dev is given (the VoxelGrid's leaf count, which the host has not seen yet), n_host otherwise
__global__ __launch_bounds__(CP_THREADS) void clip2_count_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ n_dev,
                                                                 long long n_host, ClipParams c0, ClipParams c1,
                                                                 uint32_t* __restrict__ counts /* [2][gridDim.x] */)
{
  const long long n = n_dev ? static_cast<long long>(*n_dev) : n_host;
  const long long base = static_cast<long long>(blockIdx.x) * CP_BLOCK + threadIdx.x * 4;
  uint32_t k0 = 0, k1 = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (base + k < n)
    {
      const float4 p = pts[base + k];
      k0 += (c0.enabled && clip_keeps(p, c0)) ? 1u : 0u;
      k1 += (c1.enabled && clip_keeps(p, c1)) ? 1u : 0u;
    }
  uint32_t t0, t1;
  block_scan_excl(k0, t0);
  block_scan_excl(k1, t1);
  if (threadIdx.x == 0)
  {
    counts[blockIdx.x] = t0;
    counts[gridDim.x + blockIdx.x] = t1;
  }
}

__global__ __launch_bounds__(CP_THREADS) void clip2_compact_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ n_dev,
                                                                   long long n_host, ClipParams c0, ClipParams c1,
                                                                   const uint32_t* __restrict__ counts, float4* __restrict__ out0,
                                                                   float4* __restrict__ out1, uint32_t* __restrict__ totals2)
{
  const long long n = n_dev ? static_cast<long long>(*n_dev) : n_host;
  const long long base = static_cast<long long>(blockIdx.x) * CP_BLOCK + threadIdx.x * 4;
  float4 p[4];
  bool keep0[4], keep1[4];
  uint32_t k0 = 0, k1 = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k)
  {
    keep0[k] = keep1[k] = false;
    if (base + k < n)
    {
      p[k] = pts[base + k];
      keep0[k] = c0.enabled && clip_keeps(p[k], c0);
      keep1[k] = c1.enabled && clip_keeps(p[k], c1);
    }
    k0 += keep0[k] ? 1u : 0u;
    k1 += keep1[k] ? 1u : 0u;
  }
  const uint32_t ahead0 = blocks_ahead(counts), ahead1 = blocks_ahead(counts + gridDim.x);
  uint32_t t0, t1;
  uint32_t d0 = ahead0 + block_scan_excl(k0, t0);
  uint32_t d1 = ahead1 + block_scan_excl(k1, t1);
#pragma unroll
  for (int k = 0; k < 4; ++k)
  {
    if (keep0[k])
      out0[d0++] = p[k];
    if (keep1[k])
      out1[d1++] = p[k];
  }
  if (blockIdx.x + 1 == gridDim.x && threadIdx.x == 0)
  {
    totals2[0] = ahead0 + t0;
    totals2[1] = ahead1 + t1;
  }
}

// out[pos[i]] = pts[i] where pos = exclusive scan of the keep flags (a kept point is one whose scan value steps)
__global__ void compact_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ pos, long long n,
                               float4* __restrict__ out)
{
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  if (pos[i + 1] != pos[i])
    out[pos[i]] = pts[i];
}

// -- the sampler's `output->push_back(pc->points[i])` for both models in one launch, with the min / max of the likelihood
// sample (the Morton key's origin): out_a[k] = src_a[idx_a[k]] (k < n_a), out_b[k] = src_b[idx_b[k]] (k < n_b). An index
// outside its cloud raises error flag 1 (and reads point 0).
__global__ __launch_bounds__(256) void gather2_minmax_kernel(const float4* __restrict__ src_a, long long n_src_a,
                                                             const uint32_t* __restrict__ idx_a, long long n_a,
                                                             float4* __restrict__ out_a, const float4* __restrict__ src_b,
                                                             long long n_src_b, const uint32_t* __restrict__ idx_b, long long n_b,
                                                             float4* __restrict__ out_b, int* __restrict__ error, MinMaxOut mm)
{
  float mn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, mx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
  unsigned cnt = 0;
  for (long long k = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; k < n_a + n_b; k += static_cast<long long>(gridDim.x) * 256)
  {
    if (k < n_a)
    {
      uint32_t i = idx_a[k];
      if (i >= n_src_a)
      {
        *error = 1;
        i = 0;
      }
      const float4 p = src_a[i];
      out_a[k] = p;
      minmax_accumulate(p, mn, mx, cnt);
    }
    else
    {
      uint32_t i = idx_b[k - n_a];
      if (i >= n_src_b)
      {
        *error = 1;
        i = 0;
      }
      out_b[k - n_a] = src_b[i];
    }
  }
  block_minmax_finish(mn, mx, cnt, mm);
}

// ---- matched / unmatched output (src/mcl_3dl.cpp:761-805) --------------------------------------------------------------
// Nearest map point to the (already rescaled) query within `reach` cells of the cell-sorted map: min d2 below `best`
// (initialised by the caller with r2), ties to the lowest map index. Shared with radius_search_kernel's logic.
__device__ inline float cell_grid_nearest(const LikGrid& g, float qx, float qy, float qz, int reach, float best,
                                          int& best_idx)
{
  const float fx = floorf((qx - g.ox) * g.inv_cell), fy = floorf((qy - g.oy) * g.inv_cell),
              fz = floorf((qz - g.oz) * g.inv_cell);
  if (!(fx >= -static_cast<float>(reach) && fy >= -static_cast<float>(reach) && fz >= -static_cast<float>(reach) &&
        fx <= static_cast<float>(g.nx - 1 + reach) && fy <= static_cast<float>(g.ny - 1 + reach) &&
        fz <= static_cast<float>(g.nz - 1 + reach)))
    return best;
  const int cx = static_cast<int>(fx), cy = static_cast<int>(fy), cz = static_cast<int>(fz);
  const int x0 = max(cx - reach, 0), x1 = min(cx + reach, g.nx - 1);
  const int y0 = max(cy - reach, 0), y1 = min(cy + reach, g.ny - 1);
  const int z0 = max(cz - reach, 0), z1 = min(cz + reach, g.nz - 1);
  if (x0 > x1)
    return best;
  for (int z = z0; z <= z1; ++z)
    for (int y = y0; y <= y1; ++y)
    {
      const size_t row = (static_cast<size_t>(z) * g.ny + y) * g.nx;
      const uint32_t s = g.cell_start[row + x0], e = g.cell_start[row + x1 + 1];
      for (uint32_t k = s; k < e; ++k)
      {
        const float4 p = g.pts[k];
        const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
        float d2 = dx * dx;
        d2 = d2 + dy * dy;
        d2 = d2 + dz * dz;
        const int idx = static_cast<int>(__float_as_uint(p.w));
        if (d2 < best || (d2 == best && best_idx >= 0 && idx < best_idx))
        {
          best = d2;
          best_idx = idx;
        }
      }
    }
  return best;
}

// One thread per point of the down-sampled scan: transform by the expectation pose (State6DOF::transform), search with
// unmatch_output_dist, classify: cls = 2 unmatched (no neighbour), 1 matched (sqdist < match_output_dist^2, compared in
// double like the reference), 0 neither. The transformed point is written to out_xyz4.
__global__ void match_split_kernel(const float4* __restrict__ pts, long long n, Vec3f pos, Quat rot_normalised, LikGrid g,
                                   LikParams prm, float r2_unmatch, int reach, double match_dist_sq,
                                   float4* __restrict__ out_xyz4, uint32_t* __restrict__ flag_match,
                                   uint32_t* __restrict__ flag_unmatch)
{
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i > n)
    return;
  if (i == n)
  {
    flag_match[i] = 0;
    flag_unmatch[i] = 0;
    return;
  }
  const float4 v = pts[i];
  const Vec3f t = vadd(qrot(rot_normalised, Vec3f{ v.x, v.y, v.z }), pos);
  float qx = t.x, qy = t.y, qz = t.z;
  if (prm.has_weight)
  {
    qx = t.x * prm.wx;
    qy = t.y * prm.wy;
    qz = t.z * prm.wz;
  }
  // The nearest map point within unmatch_output_dist: first the 27 cells around the query's own — a point outside that block
  // is at least one cell edge away, so a neighbour found nearer than that (with a margin for the float cell assignment) is the
  // global nearest and the (2 reach + 1)^3 cells of the whole radius (343 for 0.5 m over 0.2 m cells) are not visited; most
  // points of a scan lie on the map. The minimum — all the classification reads — is the same either way.
  int idx = -1;
  float d2 = r2_unmatch;
  bool done = false;
  if (reach > 1)
  {
    const float cell = 1.0f / g.inv_cell;
    const float sure = cell * 0.9999f;
    d2 = cell_grid_nearest(g, qx, qy, qz, 1, r2_unmatch, idx);
    done = idx >= 0 && d2 <= sure * sure;
    if (!done)
    {
      idx = -1;
      d2 = r2_unmatch;
    }
  }
  if (!done)
    d2 = cell_grid_nearest(g, qx, qy, qz, reach, r2_unmatch, idx);
  out_xyz4[i] = make_float4(t.x, t.y, t.z, v.w);
  flag_unmatch[i] = idx < 0 ? 1u : 0u;
  flag_match[i] = (idx >= 0 && static_cast<double>(d2) < match_dist_sq) ? 1u : 0u;
}

// Both order-preserving compactions of match_split in one launch, written as xyz triples where the caller reads them
// (page-locked host memory): pos_a / pos_b = exclusive scans of the two flag arrays (n + 1 entries), out_* may be null
// (counts only), writes beyond cap_* are dropped (the host reports the overflow from the counts).
__global__ void match_emit_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ pos_a,
                                  const uint32_t* __restrict__ pos_b, long long n, unsigned long long cap_a,
                                  unsigned long long cap_b, float* __restrict__ out_a, float* __restrict__ out_b,
                                  uint32_t* __restrict__ counts2)
{
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i == 0)
  {
    counts2[0] = pos_a[n];
    counts2[1] = pos_b[n];
  }
  if (i >= n)
    return;
  const uint32_t a = pos_a[i], b = pos_b[i];
  const bool in_a = pos_a[i + 1] != a && out_a && a < cap_a, in_b = pos_b[i + 1] != b && out_b && b < cap_b;
  if (!in_a && !in_b)
    return;
  const float4 p = pts[i];
  if (in_a)
  {
    out_a[3ull * a + 0] = p.x;
    out_a[3ull * a + 1] = p.y;
    out_a[3ull * a + 2] = p.z;
  }
  if (in_b)
  {
    out_b[3ull * b + 0] = p.x;
    out_b[3ull * b + 1] = p.y;
    out_b[3ull * b + 2] = p.z;
  }
}
}  // namespace mcl3dl
