// kernels.h — hand-written gfx950 (CDNA4, wave64) kernels of the LiDAR measurement update.
//
//   likelihood_kernels.h  likelihood-field model: per-particle / small-scan / tile-major (XCD-aware) kernels over the
//                         candidate-voxel index (map_compiler.h) or the cell-sorted map; strict-order sums; radius search
//   likelihood_chain_multi.h  strict_order = 3 for few particles on long scans: several tiles per work-group, a quarter of the hand-offs
//   beam_kernels.h        beam model: one lane per ray, DDA walk through 4x4x4 occupancy bricks, point tests, penalty count
//   pf_kernels.h          pf::measure (weights, deterministic fp64 reductions, normalisation, entropy) and the "next" rows
//                         (expectation / max / covariance, resampling)
//   update_kernels.h      likelihood + beam + pf::measure in ONE launch for the reference's operating range (launch-bound sizes)
//   map_compiler.h        device-side compiler of the candidate-voxel index (whole map, or the bricks a map update touches)
//   grid_kernels.h        the cell-sorted exact-NN grid and the DDA occupancy / voxel index, built on the device
//   cloud_kernels.h       scan / map preparation: PointCloud2 decode, VoxelGrid, clip + compaction, sampling gather,
//                         matched / unmatched output (each with the min / max or the count the next step needs fused in)
//   sort_kernels.h        stable radix sort of the cloud path: keys made in the first pass, points written by the last
//   cloud_keys.h          the sort keys (VoxelGrid leaf index, Morton key, range key)
//   stage_kernels.h       head and tail of a host-buffer update as one launch each: scan ordering + pose / weight take-over
//                         from page-locked host memory; lik_finalize + pf::measure with the results written back there
//
// These are gather / traversal kernels (bound by L2/HBM reads and the texture-addresser, not by MFMA):
// there is no dense contraction anywhere on this path, so no matrix-core code.
#pragma once
#include "map_structs.h"
#include "likelihood_kernels.h"
#include "likelihood_chain_multi.h"
#include "beam_kernels.h"
#include "pf_kernels.h"
#include "update_kernels.h"
#include "cloud_kernels.h"
#include "sort_kernels.h"
#include "stage_kernels.h"
#include "grid_kernels.h"
