// kernels.h — hand-written gfx950 (CDNA4, wave64) kernels of the LiDAR measurement update.
//
//   likelihood_kernel   one work-group per particle; lanes stride the (spatially ordered) scan, transform each
//                       point by the particle pose, gather the exact nearest map point from the cell-sorted map,
//                       accumulate the score in fp64 per lane, __shfl reduce per wave, LDS across waves.
//   beam_kernel         one lane per (particle, beam point) ray: Amanatides-Woo walk through the occupancy bitmap,
//                       per-voxel point tests, penalty counted with an integer atomic per particle.
//   beam_finalize       penalty count -> product of penalties (bit-exact table) clamped at beam_likelihood_min.
//   pf_*                weight update, deterministic fp64 reductions, normalisation + entropy.
//
// These are gather / traversal kernels (bound by L2/HBM reads and the texture-addresser, not by MFMA):
// there is no dense contraction anywhere on this path, so no matrix-core code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.h"
#include "map_compiler.h"

#pragma clang fp contract(off)

namespace mcl3dl
{
// ---------------------------------------------------------------------------------------------------------
// Device-resident map structures
// ---------------------------------------------------------------------------------------------------------
// Exact nearest-neighbour grid over the dist_weight-rescaled map (replaces ChunkedKdtree + pcl::KdTreeFLANN).
// Cells are cubes of edge `cell` >= match_dist_min * 1.01; points are sorted by cell, x fastest, so the 3x3x3
// neighbourhood of a query is 9 contiguous runs (one per (y,z) row), each delimited by two cell_start entries.
// Two padding cells on every side make every neighbour index of an in-range query valid.
struct LikGrid
{
  const uint32_t* cell_start;  // [nx*ny*nz + 1]
  const float4* pts;           // [n_m] rescaled x,y,z ; w = original map index (bits)
  float ox, oy, oz;            // origin (rescaled coordinates)
  float inv_cell;
  int nx, ny, nz;
};

struct LikParams
{
  float wx, wy, wz;  // dist_weight (1,1,1 when unset)
  int has_weight;
  float match_dist_min;
  float r2;  // (float)((double)r*(double)r), pcl::KdTreeFLANN::radiusSearch
  float match_dist_flat;
  float match_weight;
};

// DDA occupancy (replaces RaycastUsingDDA::point_exists_ / points_, raycast_using_dda.h:280-281).
struct DdaGrid
{
  const unsigned long long* bricks;  // occupancy: one 64-bit word per 4x4x4 voxel brick (bit = z<<4 | y<<2 | x), bricks x fastest
  int bnx, bny, bnz;                 // brick-grid extent = ceil(n / 4)
  const uint32_t* vox_start;  // [total + 1] CSR into pts (voxel order, insertion order inside a voxel)
  const float4* pts;          // x,y,z (unscaled map coordinates), w = label bits
  const uint32_t* pt_index;   // original map index of pts[k]
  float min_x, min_y, min_z;
  float max_x, max_y, max_z;
  int nx, ny, nz;
  double grid;             // dda_grid_size_
  double ray_angle_half;   // ray_angle_half_
  double min_dist_thr_sq;  // min_dist_thr_sq_
  float hit_tolerance_f;   // (float)hit_tolerance_  (Vec3::operator*(float))
};

struct BeamParams
{
  float sin_total_ref;
  float hit_range_sq;
  uint32_t filter_label_max;
  int short_only;
  float beam_likelihood_min;
};

// ---------------------------------------------------------------------------------------------------------
// Likelihood-field model: LidarMeasurementModelLikelihood::measure, src/lidar_measurement_model_likelihood.cpp:105-139
// ---------------------------------------------------------------------------------------------------------
// Nearest rescaled map point to q among the 27 cells around it; returns min d2 (FLT_MAX if none).
template <bool STATS>
__device__ inline float nearest_d2(const LikGrid& g, float qx, float qy, float qz, unsigned& n_tested)
{
  // cell of the query; (q - o) * inv is the same float expression the host used to bin the map points
  const float fx = floorf((qx - g.ox) * g.inv_cell);
  const float fy = floorf((qy - g.oy) * g.inv_cell);
  const float fz = floorf((qz - g.oz) * g.inv_cell);
  float best = 3.0e38f;
  // written so that NaN coordinates fall through to "not found"
  if (!(fx >= 1.0f && fy >= 1.0f && fz >= 1.0f && fx <= static_cast<float>(g.nx - 2) &&
        fy <= static_cast<float>(g.ny - 2) && fz <= static_cast<float>(g.nz - 2)))
    return best;
  const int cx = static_cast<int>(fx), cy = static_cast<int>(fy), cz = static_cast<int>(fz);
  uint32_t rs[9], re[9];
#pragma unroll
  for (int r = 0; r < 9; ++r)
  {
    const int dz = r / 3 - 1, dy = r % 3 - 1;
    const size_t row = (static_cast<size_t>(cz + dz) * g.ny + (cy + dy)) * g.nx + cx;
    rs[r] = g.cell_start[row - 1];
    re[r] = g.cell_start[row + 2];
  }
#pragma unroll
  for (int r = 0; r < 9; ++r)
  {
    for (uint32_t k = rs[r]; k < re[r]; ++k)
    {
      const float4 p = g.pts[k];
      // flann::L2_Simple<float>: ((0 + dx*dx) + dy*dy) + dz*dz
      const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
      float d2 = dx * dx;
      d2 = d2 + dy * dy;
      d2 = d2 + dz * dz;
      best = d2 < best ? d2 : best;
      if (STATS)
        ++n_tested;
    }
  }
  return best;
}

// Same query against the candidate-voxel index (map_compiler.h): the voxel of q holds every map point that can be the
// nearest neighbour within r of a query inside it, so min d2 over that run == min d2 over the whole map.
template <bool STATS>
__device__ inline float nearest_d2_cand(const CandGrid& g, float qx, float qy, float qz, unsigned& n_tested)
{
  const float fx = floorf((qx - g.ox) * g.inv_e);
  const float fy = floorf((qy - g.oy) * g.inv_e);
  const float fz = floorf((qz - g.oz) * g.inv_e);
  float best = 3.0e38f;
  if (!(fx >= 0.0f && fy >= 0.0f && fz >= 0.0f && fx <= static_cast<float>(g.nvx - 1) &&
        fy <= static_cast<float>(g.nvy - 1) && fz <= static_cast<float>(g.nvz - 1)))
    return best;
  const int vx = static_cast<int>(fx), vy = static_cast<int>(fy), vz = static_cast<int>(fz);
  const int b = g.brick_table[(static_cast<size_t>(vz >> 3) * g.nby + (vy >> 3)) * g.nbx + (vx >> 3)];
  if (b < 0)
    return best;
  const size_t v = static_cast<size_t>(b) * 512 + (((vz & 7) << 6) | ((vy & 7) << 3) | (vx & 7));
  const uint32_t s = g.vox_start[v], e = g.vox_start[v + 1];
  for (uint32_t k = s; k < e; ++k)
  {
    const float4 p = g.cand[k];
    const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
    float d2 = dx * dx;
    d2 = d2 + dy * dy;
    d2 = d2 + dz * dz;
    best = d2 < best ? d2 : best;
    if (STATS)
      ++n_tested;
  }
  return best;
}

// flann::L2_Simple<float>: ((0 + dx*dx) + dy*dy) + dz*dz, float, no contraction
__device__ inline float d2_simple(float qx, float qy, float qz, float px, float py, float pz)
{
  const float dx = qx - px, dy = qy - py, dz = qz - pz;
  float d2 = dx * dx;
  d2 = d2 + dy * dy;
  d2 = d2 + dz * dz;
  return d2;
}

// ChunkedKdtree::radiusSearch(p, radius, id, sqdist, 1) as a stand-alone query (include/mcl_3dl/chunked_kdtree.h:217-237):
// nearest map point with d2 < (float)(radius*radius) in the rescaled metric, ANY radius (the node also searches with
// unmatch_output_dist, src/mcl_3dl.cpp:780, and global_localization_grid, :1058-1070). Walks the cell-sorted map over
// ceil(radius / cell) cells each way: one contiguous run per (y,z) row. Ties in d2 resolve to the lowest map index.
__global__ void radius_search_kernel(const float* __restrict__ query_xyz, int n, LikGrid g, LikParams prm, float radius,
                                     float r2, int reach, int* __restrict__ out_index, float* __restrict__ out_sqdist)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  float qx = query_xyz[3 * i], qy = query_xyz[3 * i + 1], qz = query_xyz[3 * i + 2];
  if (prm.has_weight)
  {
    qx = qx * prm.wx;
    qy = qy * prm.wy;
    qz = qz * prm.wz;
  }
  (void)radius;
  float best = r2;
  int best_idx = -1;
  const float fx = floorf((qx - g.ox) * g.inv_cell), fy = floorf((qy - g.oy) * g.inv_cell),
              fz = floorf((qz - g.oz) * g.inv_cell);
  // NaN / far-away queries: comparisons fail -> no neighbour
  if (fx >= -static_cast<float>(reach) && fy >= -static_cast<float>(reach) && fz >= -static_cast<float>(reach) &&
      fx <= static_cast<float>(g.nx - 1 + reach) && fy <= static_cast<float>(g.ny - 1 + reach) &&
      fz <= static_cast<float>(g.nz - 1 + reach))
  {
    const int cx = static_cast<int>(fx), cy = static_cast<int>(fy), cz = static_cast<int>(fz);
    const int x0 = max(cx - reach, 0), x1 = min(cx + reach, g.nx - 1);
    const int y0 = max(cy - reach, 0), y1 = min(cy + reach, g.ny - 1);
    const int z0 = max(cz - reach, 0), z1 = min(cz + reach, g.nz - 1);
    if (x0 <= x1)
      for (int z = z0; z <= z1; ++z)
        for (int y = y0; y <= y1; ++y)
        {
          const size_t row = (static_cast<size_t>(z) * g.ny + y) * g.nx;
          const uint32_t s = g.cell_start[row + x0], e = g.cell_start[row + x1 + 1];
          for (uint32_t k = s; k < e; ++k)
          {
            const float4 p = g.pts[k];
            const float d2 = d2_simple(qx, qy, qz, p.x, p.y, p.z);
            const int idx = static_cast<int>(__float_as_uint(p.w));
            if (d2 < best || (d2 == best && best_idx >= 0 && idx < best_idx))
            {
              best = d2;
              best_idx = idx;
            }
          }
        }
  }
  out_index[i] = best_idx;
  if (out_sqdist)
    out_sqdist[i] = best_idx >= 0 ? best : -1.0f;
}

// MODE 2: fat voxel records — brick table, then ONE 64-byte line holding the voxel's candidates (overflow runs for
// voxels with more than 5).

template <bool STATS>
__device__ inline float nearest_d2_rec(const RecGrid& g, float qx, float qy, float qz, unsigned& n_tested)
{
  const float fx = floorf((qx - g.ox) * g.inv_e);
  const float fy = floorf((qy - g.oy) * g.inv_e);
  const float fz = floorf((qz - g.oz) * g.inv_e);
  float best = 3.0e38f;
  if (!(fx >= 0.0f && fy >= 0.0f && fz >= 0.0f && fx <= static_cast<float>(g.nvx - 1) &&
        fy <= static_cast<float>(g.nvy - 1) && fz <= static_cast<float>(g.nvz - 1)))
    return best;
  const int vx = static_cast<int>(fx), vy = static_cast<int>(fy), vz = static_cast<int>(fz);
  const int b = g.brick_table[(static_cast<size_t>(vz >> 3) * g.nby + (vy >> 3)) * g.nbx + (vx >> 3)];
  if (b < 0)
    return best;
  const size_t v = static_cast<size_t>(b) * 512 + (((vz & 7) << 6) | ((vy & 7) << 3) | (vx & 7));
  const float4* r = g.rec + 4 * v;
  const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
  const uint32_t count = __float_as_uint(r0.x);
  if (count == 0)
    return best;
  if (STATS)
    n_tested += count;
  if (count <= 5)
  {
    float d;
    d = d2_simple(qx, qy, qz, r0.y, r0.z, r0.w);
    best = d;
    d = d2_simple(qx, qy, qz, r1.x, r1.y, r1.z);
    best = (count > 1 && d < best) ? d : best;
    d = d2_simple(qx, qy, qz, r1.w, r2.x, r2.y);
    best = (count > 2 && d < best) ? d : best;
    d = d2_simple(qx, qy, qz, r2.z, r2.w, r3.x);
    best = (count > 3 && d < best) ? d : best;
    d = d2_simple(qx, qy, qz, r3.y, r3.z, r3.w);
    best = (count > 4 && d < best) ? d : best;
    return best;
  }
  float d;
  d = d2_simple(qx, qy, qz, r0.z, r0.w, r1.x);
  best = d;
  d = d2_simple(qx, qy, qz, r1.y, r1.z, r1.w);
  best = d < best ? d : best;
  d = d2_simple(qx, qy, qz, r2.x, r2.y, r2.z);
  best = d < best ? d : best;
  d = d2_simple(qx, qy, qz, r2.w, r3.x, r3.y);
  best = d < best ? d : best;
  const float* o = reinterpret_cast<const float*>(g.ovf) + 16 * static_cast<size_t>(__float_as_uint(r0.y));
  for (uint32_t j = 0; j < count - 4; ++j)
  {
    const float* s = o + 16 * (j / 5) + 3 * (j % 5);
    d = d2_simple(qx, qy, qz, s[0], s[1], s[2]);
    best = d < best ? d : best;
  }
  return best;
}

// MODE 0: 27-cell scan of the cell-sorted map (canonical structure of SURVEY.md §8d; also the STATS/K-bar counter)
// MODE 1: candidate-voxel index
template <int BLOCK, int MODE, bool STATS>
__global__ __launch_bounds__(BLOCK) void likelihood_kernel(const float* __restrict__ pose7,
                                                           const float4* __restrict__ scan, int n_s, LikGrid g,
                                                           CandGrid cg, RecGrid rg, LikParams prm,
                                                           float* __restrict__ out_lik,
                                                           float* __restrict__ out_ratio,
                                                           double* __restrict__ out_tested)
{
  const int p = blockIdx.x;
  const float* ps = pose7 + 7 * static_cast<size_t>(p);
  const Vec3f pos = { ps[0], ps[1], ps[2] };
  const Quat rot = qnormalized(Quat{ ps[3], ps[4], ps[5], ps[6] });  // state_6dof.h:217

  double acc = 0.0;   // sum of float terms, each exactly representable: fp64 sum is exact to ~1e-16
  unsigned num = 0;   // matched points
  unsigned tested = 0;
  for (int i = threadIdx.x; i < n_s; i += BLOCK)
  {
    const float4 v = scan[i];
    // State6DOF::transform, state_6dof.h:219-223
    const Vec3f t = vadd(qrot(rot, Vec3f{ v.x, v.y, v.z }), pos);
    // PointRepresentation::vectorize: rescale by dist_weight (one rounding per coordinate)
    float qx = t.x, qy = t.y, qz = t.z;
    if (prm.has_weight)
    {
      qx = t.x * prm.wx;
      qy = t.y * prm.wy;
      qz = t.z * prm.wz;
    }
    const float d2 = MODE == 0 ? nearest_d2<STATS>(g, qx, qy, qz, tested) :
                     MODE == 1 ? nearest_d2_cand<STATS>(cg, qx, qy, qz, tested) :
                                 nearest_d2_rec<STATS>(rg, qx, qy, qz, tested);
    if (d2 < prm.r2)  // radiusSearch found a neighbour (strict <)
    {
      const float s = sqrtf(d2);
      const float dist = prm.match_dist_min - (s > prm.match_dist_flat ? s : prm.match_dist_flat);  // :128
      if (!(dist < 0.0f))                                                                           // :129
      {
        acc += static_cast<double>(dist * prm.match_weight);  // :132 (float product, then accumulated)
        ++num;
      }
    }
  }
  // wavefront __shfl reduction, then across the work-group's waves through LDS
  __shared__ double s_acc[BLOCK / 64];
  __shared__ unsigned s_num[BLOCK / 64];
  __shared__ unsigned s_tested[BLOCK / 64];
  acc = wave_sum(acc);
  num = wave_sum(num);
  if (STATS)
    tested = wave_sum(tested);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
  {
    s_acc[wave] = acc;
    s_num[wave] = num;
    if (STATS)
      s_tested[wave] = tested;
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    double a = 0.0;
    unsigned n = 0, tt = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w)
    {
      a += s_acc[w];
      n += s_num[w];
      if (STATS)
        tt += s_tested[w];
    }
    if (out_lik)
      out_lik[p] = static_cast<float>(a);
    if (out_ratio)
      out_ratio[p] = static_cast<float>(n) / static_cast<float>(n_s);  // :136
    if (STATS && out_tested)
      out_tested[p] = static_cast<double>(tt);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Small-scan variant (global localisation: hundreds of thousands of particles x 8..32 points each,
// src/lidar_measurement_model_likelihood.cpp:63-77): a wavefront is shared by 64 / W particles, W = the scan size rounded
// up to a power of two; lane = (particle, point). Poses differ between the lanes of a wave, so each lane normalises its
// own quaternion; the W terms of a particle are reduced with width-W shuffles (fp64, fixed order).
// ---------------------------------------------------------------------------------------------------------
template <int W, int MODE>
__global__ __launch_bounds__(256) void likelihood_small_kernel(const float* __restrict__ pose7, int n_p,
                                                               const float4* __restrict__ scan, int n_s, LikGrid g,
                                                               CandGrid cg, RecGrid rg, LikParams prm,
                                                               float* __restrict__ out_lik, float* __restrict__ out_ratio)
{
  const long long gt = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  const long long p = gt / W;
  const int i = static_cast<int>(gt % W);
  double acc = 0.0;
  unsigned num = 0;
  if (p < n_p && i < n_s)
  {
    const float* ps = pose7 + 7 * p;
    const Vec3f pos = { ps[0], ps[1], ps[2] };
    const Quat rot = qnormalized(Quat{ ps[3], ps[4], ps[5], ps[6] });
    const float4 v = scan[i];
    const Vec3f t = vadd(qrot(rot, Vec3f{ v.x, v.y, v.z }), pos);
    float qx = t.x, qy = t.y, qz = t.z;
    if (prm.has_weight)
    {
      qx = t.x * prm.wx;
      qy = t.y * prm.wy;
      qz = t.z * prm.wz;
    }
    unsigned dummy = 0;
    const float d2 = MODE == 0 ? nearest_d2<false>(g, qx, qy, qz, dummy) :
                     MODE == 1 ? nearest_d2_cand<false>(cg, qx, qy, qz, dummy) :
                                 nearest_d2_rec<false>(rg, qx, qy, qz, dummy);
    if (d2 < prm.r2)
    {
      const float s = sqrtf(d2);
      const float dist = prm.match_dist_min - (s > prm.match_dist_flat ? s : prm.match_dist_flat);
      if (!(dist < 0.0f))
      {
        acc = static_cast<double>(dist * prm.match_weight);
        num = 1;
      }
    }
  }
#pragma unroll
  for (int off = W / 2; off > 0; off >>= 1)
  {
    acc += __shfl_down(acc, off, W);
    num += __shfl_down(num, off, W);
  }
  if (i == 0 && p < n_p)
  {
    if (out_lik)
      out_lik[p] = static_cast<float>(acc);
    if (out_ratio)
      out_ratio[p] = static_cast<float>(num) / static_cast<float>(n_s);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Tile-major variant for large scans: one work-group = one 256-point scan tile x G particles.
//
//  * the scan point of each lane stays in registers for all G particles; the G (normalised) poses are staged through
//    LDS once per work-group and read back as broadcasts;
//  * blockIdx -> (tile, particle group) is XCD-aware: work-groups are dispatched round-robin over the 8 XCDs
//    (block b runs on XCD b % 8), so XCD x is given the tiles t == x (mod 8) and walks them one after the other over all
//    particle groups. A tile is a spatially compact patch (Morton order), so the voxel records it touches under every
//    particle pose (~1 MB) stay resident in that XCD's 4 MB L2 instead of every work-group sweeping the whole scan;
//  * per-(particle, lane) float terms go to LDS and are summed in fp64 in a fixed order (deterministic), one partial per
//    (tile, particle); lik_finalize_kernel adds the tiles in order.
// Same per-point arithmetic as likelihood_kernel — identical terms — only the (fp64) summation order differs.
template <int G, int MODE>
__global__ __launch_bounds__(256) void likelihood_tiled_kernel(const float* __restrict__ pose7, int n_p,
                                                               const float4* __restrict__ scan, int n_s, int n_tiles,
                                                               int n_groups, LikGrid g, CandGrid cg, RecGrid rg,
                                                               LikParams prm, double* __restrict__ partial_sum,
                                                               unsigned* __restrict__ partial_cnt,
                                                               const uint32_t* __restrict__ scan_perm,
                                                               float* __restrict__ strict_terms)
{
  // strict_terms != nullptr ("strict_order" option): besides the fp64 partials, every float term is stored at
  // [original scan index][particle] so that lik_strict_sum_kernel can add them in the reference's own order.
  __shared__ float s_pose[G][8];        // px,py,pz, qx,qy,qz,qw (normalised), valid
  __shared__ float s_term[G][256];
  __shared__ unsigned s_cnt[G][4];
  const int xcd = blockIdx.x & 7;
  const int seq = blockIdx.x >> 3;
  const int tile = (seq / n_groups) * 8 + xcd;
  const int group = seq % n_groups;
  if (tile >= n_tiles)
    return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t < G)
  {
    const int p = group * G + t;
    float v = 0.f;
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
      v = 1.f;
    }
    s_pose[t][7] = v;
  }
  const int i = tile * 256 + t;
  const bool have_point = i < n_s;
  const float4 v = have_point ? scan[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const int n_valid = min(G, n_p - group * G);
  for (int k = 0; k < n_valid; ++k)
  {
    const Vec3f pos = { s_pose[k][0], s_pose[k][1], s_pose[k][2] };
    const Quat rot = { s_pose[k][3], s_pose[k][4], s_pose[k][5], s_pose[k][6] };
    float term = 0.f;
    bool matched = false;
    if (have_point)
    {
      const Vec3f tp = vadd(qrot(rot, Vec3f{ v.x, v.y, v.z }), pos);
      float qx = tp.x, qy = tp.y, qz = tp.z;
      if (prm.has_weight)
      {
        qx = tp.x * prm.wx;
        qy = tp.y * prm.wy;
        qz = tp.z * prm.wz;
      }
      unsigned dummy = 0;
      const float d2 = MODE == 0 ? nearest_d2<false>(g, qx, qy, qz, dummy) :
                       MODE == 1 ? nearest_d2_cand<false>(cg, qx, qy, qz, dummy) :
                                   nearest_d2_rec<false>(rg, qx, qy, qz, dummy);
      if (d2 < prm.r2)
      {
        const float s = sqrtf(d2);
        const float dist = prm.match_dist_min - (s > prm.match_dist_flat ? s : prm.match_dist_flat);
        if (!(dist < 0.0f))
        {
          term = dist * prm.match_weight;
          matched = true;
        }
      }
    }
    s_term[k][t] = term;
    if (strict_terms && have_point)
      strict_terms[static_cast<size_t>(scan_perm[i]) * n_p + (group * G + k)] = term;
    const unsigned long long m = __ballot(matched);
    if (lane == 0)
      s_cnt[k][wave] = static_cast<unsigned>(__popcll(m));
  }
  __syncthreads();
  // fixed-order fp64 reduction: 256 / G lanes per particle, each sums a contiguous segment (bank-rotated reads)
  constexpr int LPP = 256 / G;        // lanes per particle
  constexpr int SEG = 256 / LPP;      // = G terms per lane
  const int pk = t / LPP, seg = t % LPP;
  double acc = 0.0;
  if (pk < n_valid)
  {
#pragma unroll 8
    for (int j = 0; j < SEG; ++j)
    {
      const int jj = (j + t) % SEG;
      acc += static_cast<double>(s_term[pk][seg * SEG + jj]);
    }
  }
#pragma unroll
  for (int off = LPP / 2; off > 0; off >>= 1)
    acc += __shfl_down(acc, off, LPP);
  if (seg == 0 && pk < n_valid)
  {
    const size_t o = static_cast<size_t>(tile) * n_p + (group * G + pk);
    partial_sum[o] = acc;
    partial_cnt[o] = s_cnt[pk][0] + s_cnt[pk][1] + s_cnt[pk][2] + s_cnt[pk][3];
  }
}

__global__ void lik_finalize_kernel(const double* __restrict__ partial_sum, const unsigned* __restrict__ partial_cnt,
                                    int n_tiles, int n_p, int n_s, float* __restrict__ out_lik,
                                    float* __restrict__ out_ratio)
{
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_p)
    return;
  double a = 0.0;
  unsigned n = 0;
  for (int tl = 0; tl < n_tiles; ++tl)
  {
    a += partial_sum[static_cast<size_t>(tl) * n_p + p];
    n += partial_cnt[static_cast<size_t>(tl) * n_p + p];
  }
  if (out_lik)
    out_lik[p] = static_cast<float>(a);
  if (out_ratio)
    out_ratio[p] = static_cast<float>(n) / static_cast<float>(n_s);
}

// "strict_order": score_like += dist * match_weight in the reference's own order (likelihood.cpp:120-134): one lane per
// particle walks the scan in ORIGINAL order, float adds, sequentially. Unmatched points hold 0 (x + 0.0f == x), so
// the result is the reference's float, bit for bit. Loads run DEPTH ahead of the dependent add chain.
__global__ __launch_bounds__(64) void lik_strict_sum_kernel(const float* __restrict__ terms, int n_s, int n_p,
                                                            float* __restrict__ out_lik)
{
  const int p = blockIdx.x * 64 + threadIdx.x;
  if (p >= n_p)
    return;
  constexpr int DEPTH = 32;
  float score = 0.0f;
  int i = 0;
  for (; i + DEPTH <= n_s; i += DEPTH)
  {
    float v[DEPTH];
#pragma unroll
    for (int j = 0; j < DEPTH; ++j)
      v[j] = terms[static_cast<size_t>(i + j) * n_p + p];
#pragma unroll
    for (int j = 0; j < DEPTH; ++j)
      score += v[j];
  }
  for (; i < n_s; ++i)
    score += terms[static_cast<size_t>(i) * n_p + p];
  out_lik[p] = score;
}

// "strict_order": pf::measure's `sum += p.probability_` (pf.h:255-260) as a float, sequentially, by one lane; the result
// replaces the fp64 tree sum in packed[0] so that pf_apply_kernel divides by exactly the reference's float.
__global__ void pf_strict_sum_kernel(const float* __restrict__ w_new, int n, double* __restrict__ packed)
{
  if (blockIdx.x != 0 || threadIdx.x != 0)
    return;
  float sum = 0.0f;
  int i = 0;
  for (; i + 16 <= n; i += 16)
  {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j)
      v[j] = w_new[i + j];
#pragma unroll
    for (int j = 0; j < 16; ++j)
      sum += v[j];
  }
  for (; i < n; ++i)
    sum += w_new[i];
  packed[0] = static_cast<double>(sum);
}

// n_s == 0: (likelihood 1, quality 0), src/lidar_measurement_model_likelihood.cpp:111-114
__global__ void fill_kernel(float* a, float va, float* b, float vb, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
  {
    if (a)
      a[i] = va;
    if (b)
      b[i] = vb;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Beam model: RaycastUsingDDA (include/mcl_3dl/raycasts/raycast_using_dda.h) +
//             LidarMeasurementModelBeam::getBeamStatus / measure (src/lidar_measurement_model_beam.cpp:124-192)
// ---------------------------------------------------------------------------------------------------------
struct RayStats
{
  unsigned long long steps, occupied, tested;
};

// Casts one ray; returns BeamStatus (0 SHORT, 1 HIT, 2 LONG, 3 TOTAL_REFLECTION). *hit = original map index of the
// collided point (-1 if the ray was exhausted).
//
// TRACE = true is the introspection variant behind mcl3dl_hip_dda_trace: the very same walk, but every visited voxel
// centre (fromIndex, raycast_using_dda.h:219-223) is recorded and the walk stops at the first collision regardless of
// label, exactly like the reference's waypoint test harness (test/src/test_raycast_dda.cpp:157-183).
struct RayTrace
{
  float* xyz;     // [max * 3]
  int max;
  int n;          // voxels visited (may exceed max; only the first max are stored)
  int collided;   // 1 if the walk ended on a collision
};

template <bool STATS, bool TRACE = false>
__device__ inline int cast_ray(const DdaGrid& g, const BeamParams& bp, Vec3f b, Vec3f e_org, int* hit,
                               unsigned& st_steps, unsigned& st_occ, unsigned& st_tested, RayTrace* tr = nullptr)
{
  *hit = -1;
  // isPointWithinMap, raycast_using_dda.h:260-270  -> max_movement_ = 0 -> getNextCastResult false -> LONG
  if ((b.x < g.min_x) || (g.max_x < b.x) || (b.y < g.min_y) || (g.max_y < b.y) || (b.z < g.min_z) || (g.max_z < b.z))
    return 2;
  // setRay, :76-103
  const Vec3f diff = vsub(e_org, b);
  const float nrm = sqrtf(vdot(diff, diff));
  const Vec3f dir = { diff.x / nrm, diff.y / nrm, diff.z / nrm };
  const Vec3f e = vadd(e_org, vscale(dir, g.hit_tolerance_f));
  // toIndex, :205-210: float difference, double division, truncation toward zero
  const int bx = static_cast<int>(static_cast<double>(b.x - g.min_x) / g.grid);
  const int by = static_cast<int>(static_cast<double>(b.y - g.min_y) / g.grid);
  const int bz = static_cast<int>(static_cast<double>(b.z - g.min_z) / g.grid);
  const int ex = static_cast<int>(static_cast<double>(e.x - g.min_x) / g.grid);
  const int ey = static_cast<int>(static_cast<double>(e.y - g.min_y) / g.grid);
  const int ez = static_cast<int>(static_cast<double>(e.z - g.min_z) / g.grid);
  const int dix = ex - bx, diy = ey - by, diz = ez - bz;
  const int max_movement = abs(dix) + abs(diy) + abs(diz);
  const int sx = dix < 0 ? -1 : 1, sy = diy < 0 ? -1 : 1, sz = diz < 0 ? -1 : 1;
  const float inf = __builtin_inff();
  float iex = inf, iey = inf, iez = inf, tdx = inf, tdy = inf, tdz = inf;
  if (dix != 0)
  {
    const double nearest = (dir.x < 0) ? bx * g.grid + g.min_x : (bx + 1) * g.grid + g.min_x;
    iex = static_cast<float>(fabs((nearest - b.x) / dir.x));
    tdx = static_cast<float>(fabs(g.grid / dir.x));
  }
  if (diy != 0)
  {
    const double nearest = (dir.y < 0) ? by * g.grid + g.min_y : (by + 1) * g.grid + g.min_y;
    iey = static_cast<float>(fabs((nearest - b.y) / dir.y));
    tdy = static_cast<float>(fabs(g.grid / dir.y));
  }
  if (diz != 0)
  {
    const double nearest = (dir.z < 0) ? bz * g.grid + g.min_z : (bz + 1) * g.grid + g.min_z;
    iez = static_cast<float>(fabs((nearest - b.z) / dir.z));
    tdz = static_cast<float>(fabs(g.grid / dir.z));
  }
  float tmx = iex, tmy = iey, tmz = iez;
  int cx = bx, cy = by, cz = bz;
  int pos = 0;
  const int plane = g.nx * g.ny;
  // The occupancy word of the brick the ray is currently in stays in registers: stepping inside a brick is pure ALU,
  // a (dependent, high-latency) load happens only when the ray enters a new 4x4x4 brick.
  int cur_brick = -1;
  unsigned long long word = 0ull;
  // Two nested loops instead of one ("while-while" traversal): the inner loop only WALKS — to the next occupied voxel
  // or to the end of the ray — and the point tests of that voxel run after it. On a wavefront the inner loop ends when
  // every ray has found its voxel (or run out), so the long, double-precision test body executes once per round for all
  // 64 rays together instead of once per step for whichever ray happens to sit on an occupied voxel (a ray visits
  // ~1.0 occupied voxel on its way: measured). The per-ray sequence of operations is unchanged.
  for (;;)
  {
    bool found = false;
    for (;;)
    {
      // getNextCastResult, :106-159
      ++pos;
      if (pos >= max_movement)
        break;
      // axis choice of :114-147 (strict <, ties fall to the later axis), written branch-free so the 64 rays of a
      // wavefront do not serialise on three divergent bodies; only the chosen axis changes (incrementIndex, :192-203).
      const bool x_first = tmx < tmy;
      const bool ax = x_first && (tmx < tmz);
      const bool ay = !x_first && (tmy < tmz);
      const bool az = !(ax || ay);
      cx += ax ? sx : 0;
      cy += ay ? sy : 0;
      cz += az ? sz : 0;
      const float nx_t = iex + tdx * static_cast<float>(abs(cx - bx));
      const float ny_t = iey + tdy * static_cast<float>(abs(cy - by));
      const float nz_t = iez + tdz * static_cast<float>(abs(cz - bz));
      tmx = ax ? nx_t : tmx;
      tmy = ay ? ny_t : tmy;
      tmz = az ? nz_t : tmz;
      // only the moved index can have left the grid (the others were checked when they moved; begin is inside the map)
      const bool inside = static_cast<unsigned>(cx) < static_cast<unsigned>(g.nx) &&
                          static_cast<unsigned>(cy) < static_cast<unsigned>(g.ny) &&
                          static_cast<unsigned>(cz) < static_cast<unsigned>(g.nz);
      if (!inside)
        break;
      if (STATS)
        ++st_steps;
      if (TRACE)
      {
        if (tr->n < tr->max)
        {
          tr->xyz[3 * tr->n + 0] = static_cast<float>((cx + 0.5) * g.grid + g.min_x);
          tr->xyz[3 * tr->n + 1] = static_cast<float>((cy + 0.5) * g.grid + g.min_y);
          tr->xyz[3 * tr->n + 2] = static_cast<float>((cz + 0.5) * g.grid + g.min_z);
        }
        ++tr->n;
      }
      // hasIntersection, :237-258: occupancy bit first
      const int brick = ((cz >> 2) * g.bny + (cy >> 2)) * g.bnx + (cx >> 2);  // < 2^31 / 64 (total voxels < 2^31)
      if (brick != cur_brick)
      {
        cur_brick = brick;
        word = g.bricks[brick];
      }
      if ((word >> (((cz & 3) << 4) | ((cy & 3) << 2) | (cx & 3))) & 1ull)
      {
        found = true;
        break;
      }
    }
    if (!found)
      break;  // ray exhausted (or left the grid): LONG
    if (STATS)
      ++st_occ;
    const int v = cx + cy * g.nx + cz * plane;  // getArrayIndex, :225-228 (int arithmetic there too)
    const uint32_t k0 = g.vox_start[v], k1 = g.vox_start[v + 1];
    int collided = -1;
    float4 cp = { 0, 0, 0, 0 };
    for (uint32_t k = k0; k < k1; ++k)
    {
      const float4 t = g.pts[k];
      if (STATS)
        ++st_tested;
      const Vec3f rel = { t.x - b.x, t.y - b.y, t.z - b.z };
      const double foot = static_cast<double>(fabsf(vdot(rel, dir)));
      const double a = g.ray_angle_half * foot;
      const double a2 = a * a;
      const double thr = a2 < g.min_dist_thr_sq ? g.min_dist_thr_sq : a2;
      const double dist_sq = static_cast<double>(vdot(rel, rel)) - foot * foot;
      if (dist_sq < thr)
      {
        collided = static_cast<int>(k);
        cp = t;
        break;
      }
    }
    if (collided < 0)
      continue;
    if (TRACE)
    {
      tr->collided = 1;
      *hit = static_cast<int>(g.pt_index[collided]);
      return 0;
    }
    // getBeamStatus, beam.cpp:164-187
    if (__float_as_uint(cp.w) > bp.filter_label_max)
      continue;
    *hit = static_cast<int>(g.pt_index[collided]);
    if (1.0f > bp.sin_total_ref)  // DDA always reports sin_angle_ = 1.0 (raycast_using_dda.h:152)
    {
      const double ddx = static_cast<double>(e_org.x - cp.x), ddy = static_cast<double>(e_org.y - cp.y),
                   ddz = static_cast<double>(e_org.z - cp.z);
      const float distance_from_point_sq = static_cast<float>(ddx * ddx + ddy * ddy + ddz * ddz);
      return distance_from_point_sq < bp.hit_range_sq ? 1 : 0;
    }
    return 3;
  }
  return 2;
}

// One lane per (particle, beam point).  scan_beam.w = origin index (PointXYZIL::label of the scan point).
template <bool STATS>
__global__ __launch_bounds__(256) void beam_kernel(const float* __restrict__ pose7, const float4* __restrict__ scan,
                                                   int n_b, const float4* __restrict__ origins, long long n_rays,
                                                   DdaGrid g, BeamParams bp, unsigned* __restrict__ penalty_count,
                                                   RayStats* __restrict__ stats)
{
  const long long ray = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  unsigned st_steps = 0, st_occ = 0, st_tested = 0;
  if (ray < n_rays)
  {
    const long long p = ray / n_b;
    const int i = static_cast<int>(ray - p * n_b);
    const float* ps = pose7 + 7 * p;
    const Vec3f pos = { ps[0], ps[1], ps[2] };
    const Quat raw = { ps[3], ps[4], ps[5], ps[6] };
    const Quat rot = qnormalized(raw);
    const float4 v = scan[i];
    const Vec3f end = vadd(qrot(rot, Vec3f{ v.x, v.y, v.z }), pos);  // beam.cpp:139 (transform)
    const float4 og = origins[__float_as_uint(v.w)];
    const Vec3f begin = vadd(pos, qrot(raw, Vec3f{ og.x, og.y, og.z }));  // beam.cpp:145: s.pos_ + s.rot_ * origin
    int hit;
    const int status = cast_ray<STATS>(g, bp, begin, end, &hit, st_steps, st_occ, st_tested);
    if ((status == 0) || (!bp.short_only && (status == 2)))  // beam.cpp:146
      atomicAdd(&penalty_count[p], 1u);
  }
  if (STATS)
  {
    atomicAdd(&stats->steps, static_cast<unsigned long long>(st_steps));
    atomicAdd(&stats->occupied, static_cast<unsigned long long>(st_occ));
    atomicAdd(&stats->tested, static_cast<unsigned long long>(st_tested));
  }
}

// score_beam = beam_likelihood_^k by k float multiplications (table built on the host the same way), then the
// clamp of beam.cpp:151-152.
__global__ void beam_finalize_kernel(const unsigned* __restrict__ penalty_count, const float* __restrict__ pow_table,
                                     float beam_likelihood_min, float* __restrict__ out_beam, int n_p)
{
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n_p)
  {
    float s = pow_table[penalty_count[p]];
    if (s < beam_likelihood_min)
      s = beam_likelihood_min;
    out_beam[p] = s;
  }
}

// LidarMeasurementModelBeam::getBeamStatus for explicit rays (debug-marker path, src/mcl_3dl.cpp:471-478).
__global__ void beam_status_kernel(const float* __restrict__ begin_xyz, const float* __restrict__ end_xyz, int n,
                                   DdaGrid g, BeamParams bp, int* __restrict__ status, int* __restrict__ hit_index)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  unsigned a = 0, b = 0, c = 0;
  int hit;
  const int s = cast_ray<false>(g, bp, Vec3f{ begin_xyz[3 * i], begin_xyz[3 * i + 1], begin_xyz[3 * i + 2] },
                                Vec3f{ end_xyz[3 * i], end_xyz[3 * i + 1], end_xyz[3 * i + 2] }, &hit, a, b, c);
  status[i] = s;
  if (hit_index)
    hit_index[i] = (s == 2) ? -1 : hit;
}

// One ray, one lane: the waypoint introspection used by the known-answer tests.
__global__ void dda_trace_kernel(Vec3f begin, Vec3f end, DdaGrid g, BeamParams bp, float* __restrict__ out_xyz,
                                 int max_out, int* __restrict__ out3 /* n, collided, hit index */)
{
  if (blockIdx.x != 0 || threadIdx.x != 0)
    return;
  RayTrace tr = { out_xyz, max_out, 0, 0 };
  unsigned a = 0, b = 0, c = 0;
  int hit;
  cast_ray<false, true>(g, bp, begin, end, &hit, a, b, c, &tr);
  out3[0] = tr.n;
  out3[1] = tr.collided;
  out3[2] = tr.collided ? hit : -1;
}

// ---------------------------------------------------------------------------------------------------------
// pf::ParticleFilter::measure, include/mcl_3dl/pf.h:252-279  (+ the lambda's product, src/mcl_3dl.cpp:407-424)
// ---------------------------------------------------------------------------------------------------------
constexpr int PF_BLOCK = 256;

// w_new = w * (((1 * beam) * lik) * extra); per-block partials {sum w, sum w ln w, max ratio, -min ratio}.
__global__ __launch_bounds__(PF_BLOCK) void pf_partial_kernel(const float* __restrict__ w, const float* __restrict__ lik,
                                                              const float* __restrict__ beam,
                                                              const float* __restrict__ extra,
                                                              const float* __restrict__ ratio, int n,
                                                              float* __restrict__ w_new,
                                                              double* __restrict__ block_partials)
{
  double s = 0.0, t = 0.0, rmax = 0.0, rneg = -1.0;  // match_ratio_max = 0, match_ratio_min = 1 (mcl_3dl.cpp:398-399)
  for (int i = blockIdx.x * PF_BLOCK + threadIdx.x; i < n; i += gridDim.x * PF_BLOCK)
  {
    float l = 1.0f;
    if (beam)
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

// Fixed-order reduction of the block partials (deterministic run to run). The result is written in the layout the
// update's single all-reduce(SUM) needs (mcl_3dl_amd/distributed.py): [0] sum w, [1] sum w ln w, then per rank r the pair
// [2+2r] max ratio, [3+2r] -min ratio — this rank fills its own pair and zeroes the others, so that after the SUM every
// rank holds every rank's pair. world == 1 degenerates to the plain 4 doubles.
__global__ __launch_bounds__(64) void pf_reduce_kernel(const double* __restrict__ block_partials, int n_blocks, int rank,
                                                       int world, double* __restrict__ packed)
{
  double a = 0, b = 0, c = 0.0, d = -1.0;
  for (int k = threadIdx.x; k < n_blocks; k += 64)
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
__global__ __launch_bounds__(PF_BLOCK) void pf_apply_kernel(float* __restrict__ w, const float* __restrict__ w_new,
                                                            int n, int world, const double* __restrict__ packed,
                                                            float* __restrict__ stats4)
{
  const double S = packed[0];
  const float sum_f = static_cast<float>(S);
  const bool alive = sum_f > 0.0f;
  if (alive)
  {
    for (int i = blockIdx.x * PF_BLOCK + threadIdx.x; i < n; i += gridDim.x * PF_BLOCK)
      w[i] = w_new[i] / sum_f;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && stats4)
  {
    // every rank's slot holds a value in [0,1] resp. [-1,0] (0 in both for a rank whose shard saw no ratios is impossible:
    // an empty shard reports max 0 / -min -1); the maxima over the slots are the global max ratio and -min ratio
    double rmax = packed[2], rneg = packed[3];
    for (int r = 1; r < world; ++r)
    {
      rmax = packed[2 + 2 * r] > rmax ? packed[2 + 2 * r] : rmax;
      rneg = packed[3 + 2 * r] > rneg ? packed[3 + 2 * r] : rneg;
    }
    stats4[0] = alive ? static_cast<float>(log(S) - packed[1] / S) : __builtin_nanf("");
    stats4[1] = static_cast<float>(-rneg);
    stats4[2] = static_cast<float>(rmax);
    stats4[3] = alive ? 0.0f : 1.0f;
  }
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
                                                               int* __restrict__ out_arg /*[2]*/)
{
  if (threadIdx.x < MOM_N)
  {
    double s = 0;
    for (int b = 0; b < n_blocks; ++b)  // fixed order
      s += block_mom[MOM_N * b + threadIdx.x];
    out_mom[threadIdx.x] = s;
  }
  if (threadIdx.x == 32 || threadIdx.x == 33)
  {
    const int which = threadIdx.x - 32;
    ArgMax a = block_arg[which];
    for (int b = 1; b < n_blocks; ++b)
      a = argmax_better(a, block_arg[2 * b + which]);
    out_arg[which] = a.i;
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
                                            int n_out, uint32_t* __restrict__ it_out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out)
    return;
  const float p = pscan[i];
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
