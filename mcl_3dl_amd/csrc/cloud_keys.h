// cloud_keys.h — the sort keys of the cloud path as device functions (used by the fused key-and-sort kernels of
// sort_kernels.h): the Morton key and the range key of api_core.inl:order_scan, and pcl::VoxelGrid's leaf index.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#pragma clang fp contract(off)

namespace mcl3dl
{
__device__ inline bool finite3(const float4 p)
{
  return isfinite(p.x) && isfinite(p.y) && isfinite(p.z);
}

// ---- pcl::VoxelGrid ---------------------------------------------------------------------------------------------------
struct VoxelGridParams
{
  float inv_leaf[3];       // Eigen::Array4f::Ones() / leaf_size
  int min_b[3];            // floor(min_p * inv_leaf)
  int mul[3];              // divb_mul_: 1, div_b[0], div_b[0] * div_b[1]
  uint32_t nonfinite_key;  // key of a point with a non-finite coordinate: above every leaf index (sorted last, dropped)
};

// leaf index of a point — the expression of voxel_grid.hpp: ijk = int(floor(p * inv_leaf) - float(min_b)),
// idx = ijk . divb_mul
__device__ inline uint32_t voxel_leaf_key(const float4 p, const VoxelGridParams& vp)
{
  if (!finite3(p))
    return vp.nonfinite_key;
  const int i0 = static_cast<int>(floorf(p.x * vp.inv_leaf[0]) - static_cast<float>(vp.min_b[0]));
  const int i1 = static_cast<int>(floorf(p.y * vp.inv_leaf[1]) - static_cast<float>(vp.min_b[1]));
  const int i2 = static_cast<int>(floorf(p.z * vp.inv_leaf[2]) - static_cast<float>(vp.min_b[2]));
  return static_cast<uint32_t>(i0 * vp.mul[0] + i1 * vp.mul[1] + i2 * vp.mul[2]);
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

// Morton key of a likelihood scan point: 0.25 m cells from the cloud's minimum corner, 10 bits per axis, of which the
// MCL3DL_MORTON_BITS most significant bits the cloud's extent can set are kept (mm6 = the cloud's {min x, y, z, max x, y, z},
// device memory). The order only decides which evaluations share a work-group. 16 bits = two radix passes (four launches
// behind the key-making one) instead of the three a 22-bit key costs or the four of the full 30-bit key: cells of 1 m for a
// cloud up to 32 m across (2 m up to 64 m, ...). Measured at C2 (round-4 sessions s2 / s4, drivers in the git history: profiles/r04b_time8d_C2_m16.json,
// r04d_time8d_C2_m16.json against the 22-bit build in the same session): likelihood kernel +0.5 % (0.2305 -> 0.2326 ms
// device-resident step), host-buffer update -2.5 % (0.2918 -> 0.2843 ms: two launches fewer in front of the likelihood
// kernel, where every dependent launch costs 3-4 us); with 8 bits (one pass) the kernel loses 13 %. Round 3 measured 16 / 22 /
// 30 bits at C2 / C5 / the map of centroids within 1 % of each other (round-3 session s22, git history). api_core.inl:order_scan makes the
// same key on the host.
#ifndef MCL3DL_MORTON_BITS
#define MCL3DL_MORTON_BITS 16   // (A/B builds: 22 / 8 — csrc/Makefile:variants-morton)
#endif
__device__ inline uint32_t morton_key_drop(const float* __restrict__ mm6)
{
  uint32_t cells = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a)
  {
    const float e = (mm6[3 + a] - mm6[a]) * 4.0f;
    const uint32_t c = (e >= 0.f) ? (e < 1023.f ? static_cast<uint32_t>(e) : 1023u) : 0u;
    cells = c > cells ? c : cells;
  }
  const uint32_t bits = cells ? 32u - static_cast<uint32_t>(__clz(static_cast<int>(cells))) : 0u;
  return 3u * bits > MCL3DL_MORTON_BITS ? 3u * bits - MCL3DL_MORTON_BITS : 0u;
}

__device__ inline uint32_t morton_scan_key(const float4 p, const float* __restrict__ mm6)
{
  const float c[3] = { p.x, p.y, p.z };
  uint32_t q[3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
  {
    const float f = (c[a] - mm6[a]) * 4.0f;
    q[a] = (f >= 0.f) ? (f < 1023.f ? static_cast<uint32_t>(f) : 1023u) : 0u;
  }
  return (spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2)) >> morton_key_drop(mm6);
}

// squared range of a beam point from its scan origin, as float bits (non-negative floats order like unsigned ints)
__device__ inline uint32_t range_scan_key(const float4 p, const float4* __restrict__ origins, uint32_t n_o, int* __restrict__ error)
{
  uint32_t og = __float_as_uint(p.w);
  if (og >= n_o)
  {
    *error = 2;
    og = 0;
  }
  const float4 o = origins[og];
  const float dx = p.x - o.x, dy = p.y - o.y, dz = p.z - o.z;
  return __float_as_uint(dx * dx + dy * dy + dz * dz);
}
}  // namespace mcl3dl
