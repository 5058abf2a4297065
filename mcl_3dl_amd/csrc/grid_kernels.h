// grid_kernels.h — device side of the two linear-time map structures: the cell-sorted exact-NN grid (LikGrid: radius
// search, matched / unmatched, lik_index = 0, the K-bar statistics) and the DDA occupancy + voxel index (DdaGrid). Both
// are counting sorts of the map by a cell id: key per point -> stable radix sort of (key, map index) -> gather; the run
// delimiters come from a histogram (one atomic per point) and an exclusive scan. Same keys, same order inside a cell
// (ascending map index) as a sequential counting sort on the host: the structures are bit-identical to what
// host_map_compilers.h:build_*_grid_host lays out (tests/test_gpu_map_path.py compares results under both builders).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "map_structs.h"
#pragma clang fp contract(off)

namespace mcl3dl
{
// PointRepresentation::vectorize (one float product per coordinate when a dist_weight is set); w = map index
__global__ void grid_rescale_kernel(const float4* __restrict__ map, long long n, float wx, float wy, float wz, int has_weight,
                                    float4* __restrict__ out)
{
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const float4 p = map[i];
  out[i] = make_float4(has_weight ? p.x * wx : p.x, has_weight ? p.y * wy : p.y, has_weight ? p.z * wz : p.z,
                       __uint_as_float(static_cast<uint32_t>(i)));
}

struct CellGeom
{
  float ox, oy, oz, inv;
  int nx, ny, nz;
};

// cell of every rescaled point — floorf((s - o) * inv), the expression nearest_d2 uses for its queries — clamped into the grid
__global__ void lik_cell_key_kernel(const float4* __restrict__ sp, long long n, CellGeom g, uint32_t* __restrict__ key,
                                    uint32_t* __restrict__ val, uint32_t* __restrict__ count)
{
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const float4 p = sp[i];
  int cx = static_cast<int>(floorf((p.x - g.ox) * g.inv));
  int cy = static_cast<int>(floorf((p.y - g.oy) * g.inv));
  int cz = static_cast<int>(floorf((p.z - g.oz) * g.inv));
  cx = min(max(cx, 0), g.nx - 1);
  cy = min(max(cy, 0), g.ny - 1);
  cz = min(max(cz, 0), g.nz - 1);
  const uint32_t k = static_cast<uint32_t>((static_cast<size_t>(cz) * g.ny + cy) * g.nx + cx);
  key[i] = k;
  val[i] = static_cast<uint32_t>(i);
  atomicAdd(&count[k], 1u);
}

// ---- the cell grid after a map update: merge of the BASE map's grid with the update's points (round 5) -------------------
// pc_map2 = pc_map + pc_update (src/mcl_3dl.cpp:150): the update's points have the highest map indices, so inside a cell they
// follow the base points — the merged, cell-sorted array is the base array with every cell's run shifted by the number of
// update points in the cells before it, and the update's points (sorted by cell, stable) appended to their cells' runs.
// ukey = the update's cell keys, ascending (n_u of them). No histogram, no sort of the map: one binary search per cell / point.
__device__ inline uint32_t lower_bound_u32(const uint32_t* __restrict__ a, uint32_t n, uint32_t v)
{
  uint32_t lo = 0, hi = n;
  while (lo < hi)
  {
    const uint32_t mid = (lo + hi) >> 1;
    if (a[mid] < v)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

__device__ inline uint32_t lik_cell_of(const float4 p, const CellGeom& g)
{
  int cx = static_cast<int>(floorf((p.x - g.ox) * g.inv));
  int cy = static_cast<int>(floorf((p.y - g.oy) * g.inv));
  int cz = static_cast<int>(floorf((p.z - g.oz) * g.inv));
  cx = min(max(cx, 0), g.nx - 1);
  cy = min(max(cy, 0), g.ny - 1);
  cz = min(max(cz, 0), g.nz - 1);
  return static_cast<uint32_t>((static_cast<size_t>(cz) * g.ny + cy) * g.nx + cx);
}

__global__ void lik_update_key_kernel(const float4* __restrict__ sp_upd, int n_u, CellGeom g, uint32_t* __restrict__ key,
                                      uint32_t* __restrict__ val)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_u)
    return;
  key[i] = lik_cell_of(sp_upd[i], g);
  val[i] = static_cast<uint32_t>(i);
}

// (measured and not kept: one search per run of eight consecutive cells / points followed by a linear walk of the keys —
// 0.138 ms instead of 0.112 at C2: the strided runs cost more in uncoalesced accesses than the searches they save)
__global__ void lik_merge_cells_kernel(const uint32_t* __restrict__ base_start, long long n_cell_plus_1,
                                       const uint32_t* __restrict__ ukey, uint32_t n_u, uint32_t* __restrict__ out_start)
{
  const long long c = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (c < n_cell_plus_1)
    out_start[c] = base_start[c] + lower_bound_u32(ukey, n_u, static_cast<uint32_t>(c));
}

__global__ void lik_merge_points_kernel(const float4* __restrict__ base_pts, long long n_base,
                                        const uint32_t* __restrict__ base_start, CellGeom g,
                                        const float4* __restrict__ sp_upd, const uint32_t* __restrict__ ukey,
                                        const uint32_t* __restrict__ uval, uint32_t n_u, float4* __restrict__ out_pts)
{
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t < n_base)
  {
    const float4 p = base_pts[t];
    out_pts[t + lower_bound_u32(ukey, n_u, lik_cell_of(p, g))] = p;
  }
  else if (t < n_base + n_u)
  {
    const uint32_t j = static_cast<uint32_t>(t - n_base);
    out_pts[static_cast<size_t>(base_start[ukey[j] + 1u]) + j] = sp_upd[uval[j]];
  }
}

__global__ void grid_gather_kernel(const float4* __restrict__ src, const uint32_t* __restrict__ val, long long n,
                                   float4* __restrict__ dst)
{
  const long long k = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k < n)
    dst[k] = src[val[k]];
}

struct DdaGeom
{
  float mnx, mny, mnz;
  double grid;
  int nx, ny;
  unsigned long long total;
  int bnx, bny;
};

// RaycastUsingDDA::updatePointCloud / toIndex / getArrayIndex (raycast_using_dda.h:162-190,205-210,225-228): voxel =
// trunc((p - min) / grid) with a float difference and a double division, x-fastest array index in int arithmetic;
// setExists -> the voxel's bit in its 4x4x4 brick word
__global__ void dda_voxel_key_kernel(const float4* __restrict__ map, long long n, DdaGeom g, uint32_t* __restrict__ key,
                                     uint32_t* __restrict__ val, uint32_t* __restrict__ count,
                                     unsigned long long* __restrict__ bricks, int* __restrict__ err)
{
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const float4 p = map[i];
  const int c0 = static_cast<int>(static_cast<double>(p.x - g.mnx) / g.grid);
  const int c1 = static_cast<int>(static_cast<double>(p.y - g.mny) / g.grid);
  const int c2 = static_cast<int>(static_cast<double>(p.z - g.mnz) / g.grid);
  const size_t v = static_cast<size_t>(c0 + c1 * g.nx + c2 * (g.nx * g.ny));
  val[i] = static_cast<uint32_t>(i);
  if (v >= g.total)
  {
    key[i] = 0xffffffffu;  // "map point falls outside its own DDA grid"
    atomicMax(err, 1);
    return;
  }
  key[i] = static_cast<uint32_t>(v);
  atomicAdd(&count[v], 1u);
  const size_t brick = (static_cast<size_t>(c2 >> 2) * g.bny + (c1 >> 2)) * g.bnx + (c0 >> 2);
  atomicOr(&bricks[brick], 1ull << (((c2 & 3) << 4) | ((c1 & 3) << 2) | (c0 & 3)));
}

// points in voxel order (insertion = map order inside a voxel: the sort is stable) + their original map index
__global__ void dda_gather_kernel(const float4* __restrict__ map, const uint32_t* __restrict__ val, long long n,
                                  float4* __restrict__ pts, uint32_t* __restrict__ index)
{
  const long long k = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k >= n)
    return;
  const uint32_t i = val[k];
  pts[k] = map[i];
  index[k] = i;
}

// ---- the DDA overlay of a map update (DdaGrid::ov_*) ---------------------------------------------------------------------
// voxel keys of the update's points (all inside the grid: the host checked them against the grid's bounds)
__global__ void dda_overlay_key_kernel(const float4* __restrict__ upd, int n, DdaGeom g, uint32_t* __restrict__ key,
                                       uint32_t* __restrict__ val)
{
  const int i = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n)
    return;
  const float4 p = upd[i];
  const int c0 = static_cast<int>(static_cast<double>(p.x - g.mnx) / g.grid);
  const int c1 = static_cast<int>(static_cast<double>(p.y - g.mny) / g.grid);
  const int c2 = static_cast<int>(static_cast<double>(p.z - g.mnz) / g.grid);
  key[i] = static_cast<uint32_t>(c0 + c1 * g.nx + c2 * (g.nx * g.ny));
  val[i] = static_cast<uint32_t>(i);
}

__device__ inline void dda_brick_bit(const DdaGeom& g, uint32_t v, size_t& brick, unsigned long long& bit)
{
  const int plane = g.nx * g.ny;
  const int c2 = static_cast<int>(v) / plane, rem = static_cast<int>(v) - c2 * plane;
  const int c1 = rem / g.nx, c0 = rem - c1 * g.nx;
  brick = (static_cast<size_t>(c2 >> 2) * g.bny + (c1 >> 2)) * g.bnx + (c0 >> 2);
  bit = 1ull << (((c2 & 3) << 4) | ((c1 & 3) << 2) | (c0 & 3));
}

// the previous overlay goes: a voxel that holds no base point is empty again
__global__ void dda_overlay_clear_kernel(const uint32_t* __restrict__ old_key, int n, DdaGeom g,
                                         const uint32_t* __restrict__ vox_start, unsigned long long* __restrict__ bricks)
{
  const int k = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
  if (k >= n)
    return;
  const uint32_t v = old_key[k];
  if (vox_start[v + 1] != vox_start[v])
    return;
  size_t brick;
  unsigned long long bit;
  dda_brick_bit(g, v, brick, bit);
  atomicAnd(&bricks[brick], ~bit);
}

// the sorted overlay arrays + the occupancy bits of the new update
__global__ void dda_overlay_set_kernel(const float4* __restrict__ upd, const uint32_t* __restrict__ skey,
                                       const uint32_t* __restrict__ sval, int n, DdaGeom g, uint32_t* __restrict__ ov_key,
                                       float4* __restrict__ ov_pts, uint32_t* __restrict__ ov_idx,
                                       unsigned long long* __restrict__ bricks)
{
  const int k = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
  if (k >= n)
    return;
  const uint32_t v = skey[k], i = sval[k];
  ov_key[k] = v;
  ov_pts[k] = upd[i];
  ov_idx[k] = i;
  size_t brick;
  unsigned long long bit;
  dda_brick_bit(g, v, brick, bit);
  atomicOr(&bricks[brick], bit);
}

}  // namespace mcl3dl
