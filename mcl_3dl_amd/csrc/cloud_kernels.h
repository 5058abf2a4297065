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

#include "device_math.h"
#include "map_structs.h"
#pragma clang fp contract(off)

namespace mcl3dl
{
__device__ inline bool finite3(const float4 p)
{
  return isfinite(p.x) && isfinite(p.y) && isfinite(p.z);
}

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

// ---- min / max over the finite points (pcl::getMinMax3D) -------------------------------------------------------------
// block partials: [6] = min xyz, max xyz; [6] as float = number of finite points in the block (exact below 2^24)
__global__ __launch_bounds__(256) void cloud_minmax_kernel(const float4* __restrict__ pts, long long n,
                                                           float* __restrict__ block_out, unsigned* __restrict__ block_cnt)
{
  __shared__ float s[6][256];
  __shared__ unsigned s_n[256];
  float mn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, mx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
  unsigned cnt = 0;
  for (long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<long long>(gridDim.x) * 256)
  {
    const float4 p = pts[i];
    if (!finite3(p))
      continue;
    ++cnt;
    mn[0] = fminf(mn[0], p.x);
    mn[1] = fminf(mn[1], p.y);
    mn[2] = fminf(mn[2], p.z);
    mx[0] = fmaxf(mx[0], p.x);
    mx[1] = fmaxf(mx[1], p.y);
    mx[2] = fmaxf(mx[2], p.z);
  }
  for (int a = 0; a < 3; ++a)
  {
    s[a][threadIdx.x] = mn[a];
    s[3 + a][threadIdx.x] = mx[a];
  }
  s_n[threadIdx.x] = cnt;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1)
  {
    if (threadIdx.x < off)
    {
      for (int a = 0; a < 3; ++a)
      {
        s[a][threadIdx.x] = fminf(s[a][threadIdx.x], s[a][threadIdx.x + off]);
        s[3 + a][threadIdx.x] = fmaxf(s[3 + a][threadIdx.x], s[3 + a][threadIdx.x + off]);
      }
      s_n[threadIdx.x] += s_n[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x < 6)
    block_out[6 * blockIdx.x + threadIdx.x] = s[threadIdx.x][0];
  if (threadIdx.x == 0)
    block_cnt[blockIdx.x] = s_n[0];
}

// one wavefront: the lanes stride the block partials, then butterfly reductions (min / max / integer sum: any order, same
// result). Launched with 64 threads.
__global__ void cloud_minmax_final(const float* __restrict__ block_out, const unsigned* __restrict__ block_cnt, int nb,
                                   float* __restrict__ out6, unsigned long long* __restrict__ out_cnt)
{
  if (blockIdx.x != 0 || threadIdx.x >= 64)
    return;
  float mn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, mx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
  unsigned long long cnt = 0;
  for (int b = threadIdx.x; b < nb; b += 64)
  {
    for (int a = 0; a < 3; ++a)
    {
      mn[a] = fminf(mn[a], block_out[6 * b + a]);
      mx[a] = fmaxf(mx[a], block_out[6 * b + 3 + a]);
    }
    cnt += block_cnt[b];
  }
  for (int off = 32; off > 0; off >>= 1)
  {
    for (int a = 0; a < 3; ++a)
    {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64));
    }
    cnt += __shfl_xor(cnt, off, 64);
  }
  if (threadIdx.x == 0)
  {
    for (int a = 0; a < 3; ++a)
    {
      out6[a] = mn[a];
      out6[3 + a] = mx[a];
    }
    *out_cnt = cnt;
  }
}

// ---- pcl::VoxelGrid ---------------------------------------------------------------------------------------------------
struct VoxelGridParams
{
  float inv_leaf[3];  // Eigen::Array4f::Ones() / leaf_size
  int min_b[3];       // floor(min_p * inv_leaf)
  int mul[3];         // divb_mul_: 1, div_b[0], div_b[0] * div_b[1]
};

// leaf index of every point — the expression of voxel_grid.hpp: ijk = int(floor(p * inv_leaf) - float(min_b)),
// idx = ijk . divb_mul. Non-finite points get the key 0xffffffff (they sort behind every leaf and are dropped).
__global__ void vg_key_kernel(const float4* __restrict__ pts, long long n, VoxelGridParams vp, uint32_t* __restrict__ key,
                              uint32_t* __restrict__ val)
{
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const float4 p = pts[i];
  uint32_t k = 0xffffffffu;
  if (finite3(p))
  {
    const int i0 = static_cast<int>(floorf(p.x * vp.inv_leaf[0]) - static_cast<float>(vp.min_b[0]));
    const int i1 = static_cast<int>(floorf(p.y * vp.inv_leaf[1]) - static_cast<float>(vp.min_b[1]));
    const int i2 = static_cast<int>(floorf(p.z * vp.inv_leaf[2]) - static_cast<float>(vp.min_b[2]));
    k = static_cast<uint32_t>(i0 * vp.mul[0] + i1 * vp.mul[1] + i2 * vp.mul[2]);
  }
  key[i] = k;
  val[i] = static_cast<uint32_t>(i);
}

// head[i] = 1 where a new leaf starts in the sorted key array (n = number of finite points); head[n] = 0 (scan slot)
__global__ void vg_heads_kernel(const uint32_t* __restrict__ key, long long n, uint32_t* __restrict__ head)
{
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i > n)
    return;
  head[i] = (i < n && (i == 0 || key[i] != key[i - 1])) ? 1u : 0u;
}

// start[leaf] = first sorted position of the leaf; leaf = exclusive scan of head at a head position
__global__ void vg_starts_kernel(const uint32_t* __restrict__ key, const uint32_t* __restrict__ leaf_of, long long n,
                                 uint32_t n_leaves, uint32_t* __restrict__ start)
{
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i > n)
    return;
  if (i == n)
  {
    start[n_leaves] = static_cast<uint32_t>(n);
    return;
  }
  if (i == 0 || key[i] != key[i - 1])
    start[leaf_of[i]] = static_cast<uint32_t>(i);
}

// One thread per leaf: pcl::CentroidPoint<PointXYZIL> — xyz summed as floats in the order the points arrive (the stable
// sort keeps the input order inside a leaf; PCL's own std::sort leaves that order unspecified), divided by the count;
// label = the most frequent one, the smallest on a tie (AccumulatorLabel walks a std::map with a strict `>`).
__global__ void vg_centroid_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ val,
                                   const uint32_t* __restrict__ start, uint32_t n_leaves, float4* __restrict__ out)
{
  const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n_leaves)
    return;
  const uint32_t s = start[l], e = start[l + 1];
  float sx = 0.f, sy = 0.f, sz = 0.f;
  const uint32_t first_label = __float_as_uint(pts[val[s]].w);
  bool uniform = true;
  for (uint32_t k = s; k < e; ++k)
  {
    const float4 p = pts[val[k]];
    sx += p.x;
    sy += p.y;
    sz += p.z;
    uniform = uniform && __float_as_uint(p.w) == first_label;
  }
  uint32_t best_label = first_label;
  if (!uniform)
  {
    uint32_t best_count = 0;
    for (uint32_t k = s; k < e; ++k)
    {
      const uint32_t lab = __float_as_uint(pts[val[k]].w);
      uint32_t c = 0;
      for (uint32_t j = s; j < e; ++j)
        c += __float_as_uint(pts[val[j]].w) == lab ? 1u : 0u;
      if (c > best_count || (c == best_count && lab < best_label))
      {
        best_count = c;
        best_label = lab;
      }
    }
  }
  const float cnt = static_cast<float>(e - s);
  out[l] = make_float4(sx / cnt, sy / cnt, sz / cnt, __uint_as_float(best_label));
}

// ---- clip filter + order-preserving compaction -------------------------------------------------------------------------
// flag[i] = 1 if the point is KEPT. The reference's lambda returns true (= erase) for x^2 + y^2 > far^2, < near^2,
// z < z_min or z_max < z; a NaN makes every comparison false, so such a point is kept there too.
__global__ void clip_flag_kernel(const float4* __restrict__ pts, long long n, float near_sq, float far_sq, float z_min,
                                 float z_max, uint32_t* __restrict__ flag)
{
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i > n)
    return;
  if (i == n)
  {
    flag[i] = 0;  // scan slot
    return;
  }
  const float4 p = pts[i];
  const float r2 = p.x * p.x + p.y * p.y;
  const bool erase = r2 > far_sq || r2 < near_sq || p.z < z_min || z_max < p.z;
  flag[i] = erase ? 0u : 1u;
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

// out[k] = src[idx[k]]; an index outside [0, n_src) raises the error flag (and reads point 0)
__global__ void gather_kernel(const float4* __restrict__ src, long long n_src, const uint32_t* __restrict__ idx,
                              long long n, float4* __restrict__ out, int* __restrict__ error)
{
  const long long k = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k >= n)
    return;
  uint32_t i = idx[k];
  if (i >= n_src)
  {
    *error = 1;
    i = 0;
  }
  out[k] = src[i];
}

// ---- scan ordering (the device form of api_core.inl:order_scan) --------------------------------------------------------
__device__ inline uint32_t spread10(uint32_t v)
{
  // 10 bits -> every third bit (the low 30 bits of the host's 64-bit morton3 spread)
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

// 30-bit Morton key of a likelihood scan point: 0.25 m cells from the cloud's minimum corner, clamped to 10 bits per axis
__global__ void order_morton_key_kernel(const float4* __restrict__ pts, long long n, const float* __restrict__ min3,
                                        uint32_t* __restrict__ key, uint32_t* __restrict__ val)
{
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const float4 p = pts[i];
  const float c[3] = { p.x, p.y, p.z };
  uint32_t q[3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
  {
    const float f = (c[a] - min3[a]) * 4.0f;
    q[a] = (f >= 0.f) ? (f < 1023.f ? static_cast<uint32_t>(f) : 1023u) : 0u;
  }
  key[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
  val[i] = static_cast<uint32_t>(i);
}

// key = squared range of a beam point from its scan origin, as float bits (non-negative floats order like unsigned ints)
__global__ void order_range_key_kernel(const float4* __restrict__ pts, long long n, const float4* __restrict__ origins,
                                       uint32_t n_o, uint32_t* __restrict__ key, uint32_t* __restrict__ val,
                                       int* __restrict__ error)
{
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const float4 p = pts[i];
  uint32_t og = __float_as_uint(p.w);
  if (og >= n_o)
  {
    *error = 2;
    og = 0;
  }
  const float4 o = origins[og];
  const float dx = p.x - o.x, dy = p.y - o.y, dz = p.z - o.z;
  key[i] = __float_as_uint(dx * dx + dy * dy + dz * dz);
  val[i] = static_cast<uint32_t>(i);
}

// out[k] = src[val[k]] (w kept: the beam scan carries the origin id there; the likelihood scan's is overwritten with 0),
// perm[k] = val[k]
__global__ void order_apply_kernel(const float4* __restrict__ src, const uint32_t* __restrict__ val, long long n,
                                   int zero_w, float4* __restrict__ out, uint32_t* __restrict__ perm)
{
  const long long k = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k >= n)
    return;
  float4 p = src[val[k]];
  if (zero_w)
    p.w = 0.f;
  out[k] = p;
  if (perm)
    perm[k] = val[k];
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
  int idx = -1;
  const float d2 = cell_grid_nearest(g, qx, qy, qz, reach, r2_unmatch, idx);
  out_xyz4[i] = make_float4(t.x, t.y, t.z, v.w);
  flag_unmatch[i] = idx < 0 ? 1u : 0u;
  flag_match[i] = (idx >= 0 && static_cast<double>(d2) < match_dist_sq) ? 1u : 0u;
}
}  // namespace mcl3dl
