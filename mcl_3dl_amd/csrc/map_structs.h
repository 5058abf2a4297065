// map_structs.h — device-resident map structures and per-model constants shared by the kernels.
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
  int mul24_ok;                      // bnx < 2^24 and bny * bnz < 2^24: the brick index can use 24-bit multiplies
  const uint32_t* vox_start;  // [total + 1] CSR into pts (voxel order, insertion order inside a voxel)
  const float4* pts;          // x,y,z (unscaled map coordinates), w = label bits
  const uint32_t* pt_index;   // original map index of pts[k]
  // the map update (pc_map2 = pc_map + pc_update, src/mcl_3dl.cpp:150): its points are NOT in the arrays above but in a small
  // overlay sorted by voxel — a new update replaces the overlay and flips a few occupancy bits, the base arrays stay as built.
  // A voxel's points are its base run followed by its overlay run: map order, like one array over the merged cloud.
  const uint32_t* ov_key;  // [ov_n] voxel index of overlay point k, ascending (update order inside a voxel)
  const float4* ov_pts;    // [ov_n]
  const uint32_t* ov_idx;  // [ov_n] position in the update; map index = ov_base + ov_idx[k]
  int ov_n;                // 0: no overlay
  uint32_t ov_base;
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

}  // namespace mcl3dl
