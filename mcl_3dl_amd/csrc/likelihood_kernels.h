// likelihood_kernels.h — likelihood-field model (R3/R4/R5): nearest-neighbour queries against the three map indices, the
// per-particle / small-scan / tile-major kernels, the strict-order sums and the stand-alone radius search.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.h"
#include "float_chain.h"
#include "map_structs.h"
#pragma clang fp contract(off)

namespace mcl3dl
{
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

// MODE 2: fat voxel records — brick table, then ONE 64-byte line holding the voxel's candidates. Record layout
// (map_compiler.h:mc_write_records), four 16-byte parts:
//     part j = { candidate j: x, y, z ; w }      w: count and first overflow record, packed or plain (map_compiler.h)
// candidates 0..3 inline (unused slots hold REC_SENTINEL coordinates: their d2 never wins a minimum and never passes the
// radius test), candidates 4.. in overflow records of the same four-part layout. One part per lane of a quad is what the tiled kernel's
// cooperative fetch reads (below).

// floor to int in ONE instruction (v_cvt_flr_i32_f32; the compiler emits v_floor_f32 + v_cvt_i32_f32 for
// __float2int_rd). Same value for every input: saturating, NaN -> 0.
__device__ inline int floor_to_int(float x)
{
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// step 1 of a lookup: voxel of the query -> brick-table index and the voxel's slot inside its brick; false = outside
// the grid (a non-finite coordinate saturates to a voxel outside the grid, or for NaN lands in voxel 0 and yields NaN
// distances: no match either way). 32-bit index arithmetic (the dense brick table has < 2^31 entries, there are <= 2^22
// bricks: build_cand_grid) with 24-bit multiplies where build_cand_grid found every factor below 2^24.
__device__ inline bool rec_locate(const RecGrid& g, float qx, float qy, float qz, uint32_t& ti, uint32_t& sub)
{
  const int vx = floor_to_int((qx - g.ox) * g.inv_ex);
  const int vy = floor_to_int((qy - g.oy) * g.inv_ey);
  const int vz = floor_to_int((qz - g.oz) * g.inv_ez);
  if (g.mul24_ok)
    ti = __umul24(static_cast<uint32_t>(vz >> 3), static_cast<uint32_t>(g.nbx * g.nby)) +
         __umul24(static_cast<uint32_t>(vy >> 3), static_cast<uint32_t>(g.nbx)) + static_cast<uint32_t>(vx >> 3);
  else
    ti = (static_cast<uint32_t>(vz >> 3) * static_cast<uint32_t>(g.nby) + static_cast<uint32_t>(vy >> 3)) *
             static_cast<uint32_t>(g.nbx) + static_cast<uint32_t>(vx >> 3);
  sub = ((((static_cast<uint32_t>(vz) & 7u) << 3) | (static_cast<uint32_t>(vy) & 7u)) << 3) | (static_cast<uint32_t>(vx) & 7u);
  return static_cast<unsigned>(vx) < static_cast<unsigned>(g.nvx) && static_cast<unsigned>(vy) < static_cast<unsigned>(g.nvy) &&
         static_cast<unsigned>(vz) < static_cast<unsigned>(g.nvz);
}

// min d2 over the candidates a voxel's overflow records hold (candidates cap .. count - 1; four per 64-byte record, laid
// out like the first half of a voxel record: part j = {x, y, z, -})
// (n_records whole records: their unused slots hold the sentinel, which never wins)
__device__ inline float rec_overflow_min(const RecGrid& g, float qx, float qy, float qz, uint32_t n_records, uint32_t ext,
                                         float best)
{
  const float4* o = g.ovf + 4 * static_cast<size_t>(ext);
  for (uint32_t j = 0; j < 4u * n_records; ++j)
  {
    const float4 c = o[j];
    const float d = d2_simple(qx, qy, qz, c.x, c.y, c.z);
    best = d < best ? d : best;
  }
  return best;
}

template <bool STATS>
__device__ inline float nearest_d2_rec(const RecGrid& g, float qx, float qy, float qz, unsigned& n_tested)
{
  float best = 3.0e38f;
  uint32_t ti, sub;
  if (!rec_locate(g, qx, qy, qz, ti, sub))
    return best;
  const int b = g.brick_table[ti];
  if (b < 0)
    return best;
  const uint32_t cap = static_cast<uint32_t>(g.rec_parts);
  const float4* r = g.rec + static_cast<size_t>(cap) * ((static_cast<uint32_t>(b) << 9) | sub);
  const float4 r0 = r[0], r1 = r[1];
  const uint32_t w0 = __float_as_uint(r0.w);
  const uint32_t field = g.packed ? w0 >> g.count_shift : w0;
  const uint32_t n_records = rec_overflow_records(field, cap, g.count_is_records);
  if (!g.count_is_records && field == 0)
    return best;
  if (STATS)
    n_tested += g.count_is_records ? cap + 4u * n_records : field;
  // unused slots hold the sentinel: the minimum over all inline slots needs no look at the count
  best = fminf(d2_simple(qx, qy, qz, r0.x, r0.y, r0.z), d2_simple(qx, qy, qz, r1.x, r1.y, r1.z));
  for (uint32_t k = 2; k < cap; ++k)
  {
    const float4 c = r[k];
    best = fminf(best, d2_simple(qx, qy, qz, c.x, c.y, c.z));
  }
  // (bounded records: the overflow candidates cannot improve on a best that is within their skip bound — RecGrid::bound_step;
  // STATS counts the candidates of the canonical set either way)
  if (n_records && !(g.bound_step > 0.0f && best <= rec_bound2(g, w0)))
    best = rec_overflow_min(g, qx, qy, qz, n_records, g.packed ? (w0 & rec_ext_mask(g)) : __float_as_uint(r1.w), best);
  return best;
}

// ---- VALU-trimmed forms (profiles/r02b_valu_microbench.txt prices a wave64 v_mul/v_add_f32 at ~2.4 cycles on a SIMD and
// almost everything else — v_max, v_cndmask, v_cvt, integer multiplies, packed f32 pairs, every f64 op — at ~4.3).

// Quat::operator*(Vec3) (quat.h:139-143) = (q (x) (v,0)) (x) conj(q) with the reference's term order, minus the four
// products with the vector's zero w component. Dropping them changes nothing observable: for finite q the product is
// +-0 and x + (+-0) == x exactly (no rounding); where the dropped term could flip the SIGN of an exact zero, that sign
// never reaches a result (zeros only meet products, sums with non-zeros, and squares from here on); for a non-finite q
// both forms end in "no neighbour found". Written with scalar temporaries: one v_mul / v_add per line.
__device__ inline Vec3f qrot_trim(const Quat q, const Vec3f v)
{
  // p = q (x) (v, 0)
  const float px = (q.w * v.x + q.y * v.z) - q.z * v.y;
  const float py = (q.w * v.y + q.z * v.x) - q.x * v.z;
  const float pz = (q.w * v.z + q.x * v.y) - q.y * v.x;
  const float pw = (-(q.x * v.x) - q.y * v.y) - q.z * v.z;
  // t = p (x) conj(q), conj(q) = (-q.x, -q.y, -q.z, q.w); x * (-y) == -(x * y) bit for bit
  Vec3f t;
  t.x = ((pw * -q.x + px * q.w) + py * -q.z) - pz * -q.y;
  t.y = ((pw * -q.y + py * q.w) + pz * -q.x) - px * -q.z;
  t.z = ((pw * -q.z + pz * q.w) + px * -q.y) - py * -q.x;
  return t;
}

// ---- quad-cooperative record fetch --------------------------------------------------------------------------------
// What binds the tiled kernel is the L1's access rate (profiles/r02b_C2_pmc_summary.csv: ~1.1 cache-line accesses per
// cycle and CU, flat against -15 % VALU instructions and against twice the loads in flight per lane): a wave64 16-byte
// load whose lanes all name different records is 64 accesses, and each lane needs four of them per record. Here the four
// lanes of a quad fetch a record together — lane j reads part j of the records of lanes 0..3 of its quad, so one load
// instruction covers 16 whole records in 16 accesses instead of 64 — and compute together: lane j owns candidate j of
// each of the quad's four evaluations (the query travels by DPP), then a 4 x 4 transpose-min over the quad hands every
// lane the minimum for its own query. Same candidates, same d2 expression, and a minimum does not care about order:
// results identical to rec_min_d2.
// any lane true? (the ballot straight from the compare: no 0/1 round trip through a VGPR)
__device__ inline bool wave_any(bool p)
{
  return __builtin_amdgcn_ballot_w64(p) != 0ull;
}

template <int CTRL>
__device__ inline float quad_f(float v)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ inline uint32_t quad_u(uint32_t v)
{
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xf, 0xf, true));
}
constexpr int QUAD_BCAST0 = 0x00, QUAD_BCAST1 = 0x55, QUAD_BCAST2 = 0xAA, QUAD_BCAST3 = 0xFF;  // quad_perm:[e,e,e,e]
constexpr int QUAD_XOR1 = 0xB1, QUAD_XOR2 = 0x4E;                                              // [1,0,3,2], [2,3,0,1]

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ inline float4 buffer_load_f4(__amdgpu_buffer_rsrc_t rs, uint32_t byte_offset)
{
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, static_cast<int>(byte_offset), 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// lanes (as a 64-bit mask, straight from the compare: no 0/1 round trip through a VGPR) where a > b, unsigned
__device__ inline unsigned long long lanes_gt_u32(uint32_t a, uint32_t b)
{
  return __builtin_amdgcn_uicmp(a, b, 34 /* ICMP_UGT */);
}
constexpr unsigned long long QUAD_LANE0_MASK = 0x1111111111111111ull;

// Measured and NOT kept (C2, 0.230 ms with the code below): (1) a skewed fetch — load r of lane j = record of lane
// (j + r) & 3, which turns the transpose-min into three v_min_u32 with a quad rotation as DPP operand and no selects —
// 0.30 ms: every load instruction then touches 64 different cache lines instead of 16, and the L1's line-access rate is
// what the cooperative fetch exists to spare; (2) the wavefront's match counts kept in scalar registers and written to
// LDS once per work-group instead of once per particle (-2 VALU, +10 SALU per evaluation): 0.230 ms, no change; (3) the
// float constants (grid origin, 1 / edge, weights, r, flat, match weight) forced into vector registers because a v_mul_f32
// with an SGPR operand is priced at 4.2 instead of 2.6 cycles by profiles/r02d_valu_microbench.txt: 0.233 ms, no gain.
// One cooperative round: the quad fetches the 64-byte records `rec_index` of its four lanes from `recs` (lane j reads
// part j of each), lane j computes candidate j of evaluation e against the query of lane e, and the 4 x 4 transpose-min
// returns to every lane the minimum over the four candidates of ITS record. w[e] = the w word of the part this lane read
// of record e (part 0's w = count, part 1's w = first overflow record). Every lane of the wavefront must be active (DPP
// reads 0 from an inactive lane); a lane without a record passes any valid index and ignores the answer.
__device__ inline float quad_round(const float4* recs, uint32_t bytes32, uint32_t rec_index, float qx, float qy, float qz,
                                   int j, uint32_t (&w)[4])
{
  float4 R0, R1, R2, R3;
  if (bytes32)
  {
    // array below 4 GB: buffer loads — the (uniform) base sits in a scalar resource descriptor and the lane supplies a
    // 32-bit byte offset: one v_add_u32 with a DPP operand per load, no 64-bit address arithmetic. An offset past
    // `bytes32` (the record index of a lane without a voxel is arbitrary) reads zeros instead of faulting.
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(recs), 0, static_cast<int>(bytes32), 0x00020000);
    const uint32_t mine = rec_index << 6, part = static_cast<uint32_t>(j) << 4;
    R0 = buffer_load_f4(rs, quad_u<QUAD_BCAST0>(mine) + part);
    R1 = buffer_load_f4(rs, quad_u<QUAD_BCAST1>(mine) + part);
    R2 = buffer_load_f4(rs, quad_u<QUAD_BCAST2>(mine) + part);
    R3 = buffer_load_f4(rs, quad_u<QUAD_BCAST3>(mine) + part);
  }
  else
  {
    const float4* part = recs + j;
    R0 = part[4 * static_cast<size_t>(quad_u<QUAD_BCAST0>(rec_index))];
    R1 = part[4 * static_cast<size_t>(quad_u<QUAD_BCAST1>(rec_index))];
    R2 = part[4 * static_cast<size_t>(quad_u<QUAD_BCAST2>(rec_index))];
    R3 = part[4 * static_cast<size_t>(quad_u<QUAD_BCAST3>(rec_index))];
  }
  // candidate j of evaluation e against the query of lane e
  const float d0 = d2_simple(quad_f<QUAD_BCAST0>(qx), quad_f<QUAD_BCAST0>(qy), quad_f<QUAD_BCAST0>(qz), R0.x, R0.y, R0.z);
  const float d1 = d2_simple(quad_f<QUAD_BCAST1>(qx), quad_f<QUAD_BCAST1>(qy), quad_f<QUAD_BCAST1>(qz), R1.x, R1.y, R1.z);
  const float d2 = d2_simple(quad_f<QUAD_BCAST2>(qx), quad_f<QUAD_BCAST2>(qy), quad_f<QUAD_BCAST2>(qz), R2.x, R2.y, R2.z);
  const float d3 = d2_simple(quad_f<QUAD_BCAST3>(qx), quad_f<QUAD_BCAST3>(qy), quad_f<QUAD_BCAST3>(qz), R3.x, R3.y, R3.z);
  // 4 x 4 transpose-min: after the xor-1 step a lane holds min over {j, j^1} for evaluation (j & 1) resp. 2 + (j & 1);
  // after the xor-2 step min over the whole quad for evaluation j. A d2 is a sum of squares — never negative — so the
  // float minimum is the minimum of the bit patterns as unsigned integers (NaN, from a non-finite query, orders above
  // every number and every d2 of that query is NaN anyway): v_min_u32 needs no canonicalisation of its inputs.
  const bool odd = (j & 1) != 0, high = (j & 2) != 0;
  const uint32_t u0 = __float_as_uint(d0), u1 = __float_as_uint(d1), u2 = __float_as_uint(d2), u3 = __float_as_uint(d3);
  const uint32_t m01 = min(odd ? u1 : u0, quad_u<QUAD_XOR1>(odd ? u0 : u1));
  const uint32_t m23 = min(odd ? u3 : u2, quad_u<QUAD_XOR1>(odd ? u2 : u3));
  w[0] = __float_as_uint(R0.w);
  w[1] = __float_as_uint(R1.w);
  w[2] = __float_as_uint(R2.w);
  w[3] = __float_as_uint(R3.w);
  return __uint_as_float(min(high ? m23 : m01, quad_u<QUAD_XOR2>(high ? m01 : m23)));
}

// The same for 128-byte records (eight inline candidates): all eight 16-byte loads of the quad's four records are issued
// before any arithmetic, so that the two halves of a 128-byte line are requested together; lane j owns candidates j and
// j + 4 of each evaluation.
__device__ inline float quad_round_wide(const float4* recs, uint32_t bytes32, uint32_t voxel_index, float qx, float qy,
                                        float qz, int j, uint32_t (&w)[4])
{
  float4 R0, R1, R2, R3, S0, S1, S2, S3;
  if (bytes32)
  {
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(recs), 0, static_cast<int>(bytes32), 0x00020000);
    const uint32_t mine = voxel_index << 7, part = static_cast<uint32_t>(j) << 4;
    const uint32_t o0 = quad_u<QUAD_BCAST0>(mine) + part, o1 = quad_u<QUAD_BCAST1>(mine) + part,
                   o2 = quad_u<QUAD_BCAST2>(mine) + part, o3 = quad_u<QUAD_BCAST3>(mine) + part;
    R0 = buffer_load_f4(rs, o0);
    S0 = buffer_load_f4(rs, o0 + 64u);
    R1 = buffer_load_f4(rs, o1);
    S1 = buffer_load_f4(rs, o1 + 64u);
    R2 = buffer_load_f4(rs, o2);
    S2 = buffer_load_f4(rs, o2 + 64u);
    R3 = buffer_load_f4(rs, o3);
    S3 = buffer_load_f4(rs, o3 + 64u);
  }
  else
  {
    const float4* part = recs + j;
    const float4* p0 = part + 8 * static_cast<size_t>(quad_u<QUAD_BCAST0>(voxel_index));
    const float4* p1 = part + 8 * static_cast<size_t>(quad_u<QUAD_BCAST1>(voxel_index));
    const float4* p2 = part + 8 * static_cast<size_t>(quad_u<QUAD_BCAST2>(voxel_index));
    const float4* p3 = part + 8 * static_cast<size_t>(quad_u<QUAD_BCAST3>(voxel_index));
    R0 = p0[0];
    S0 = p0[4];
    R1 = p1[0];
    S1 = p1[4];
    R2 = p2[0];
    S2 = p2[4];
    R3 = p3[0];
    S3 = p3[4];
  }
  const float x0 = quad_f<QUAD_BCAST0>(qx), y0 = quad_f<QUAD_BCAST0>(qy), z0 = quad_f<QUAD_BCAST0>(qz);
  const float x1 = quad_f<QUAD_BCAST1>(qx), y1 = quad_f<QUAD_BCAST1>(qy), z1 = quad_f<QUAD_BCAST1>(qz);
  const float x2 = quad_f<QUAD_BCAST2>(qx), y2 = quad_f<QUAD_BCAST2>(qy), z2 = quad_f<QUAD_BCAST2>(qz);
  const float x3 = quad_f<QUAD_BCAST3>(qx), y3 = quad_f<QUAD_BCAST3>(qy), z3 = quad_f<QUAD_BCAST3>(qz);
  const uint32_t u0 = min(__float_as_uint(d2_simple(x0, y0, z0, R0.x, R0.y, R0.z)), __float_as_uint(d2_simple(x0, y0, z0, S0.x, S0.y, S0.z)));
  const uint32_t u1 = min(__float_as_uint(d2_simple(x1, y1, z1, R1.x, R1.y, R1.z)), __float_as_uint(d2_simple(x1, y1, z1, S1.x, S1.y, S1.z)));
  const uint32_t u2 = min(__float_as_uint(d2_simple(x2, y2, z2, R2.x, R2.y, R2.z)), __float_as_uint(d2_simple(x2, y2, z2, S2.x, S2.y, S2.z)));
  const uint32_t u3 = min(__float_as_uint(d2_simple(x3, y3, z3, R3.x, R3.y, R3.z)), __float_as_uint(d2_simple(x3, y3, z3, S3.x, S3.y, S3.z)));
  const bool odd = (j & 1) != 0, high = (j & 2) != 0;
  const uint32_t m01 = min(odd ? u1 : u0, quad_u<QUAD_XOR1>(odd ? u0 : u1));
  const uint32_t m23 = min(odd ? u3 : u2, quad_u<QUAD_XOR1>(odd ? u2 : u3));
  w[0] = __float_as_uint(R0.w);
  w[1] = __float_as_uint(R1.w);
  w[2] = __float_as_uint(R2.w);
  w[3] = __float_as_uint(R3.w);
  return __uint_as_float(min(high ? m23 : m01, quad_u<QUAD_XOR2>(high ? m01 : m23)));
}

// w[e] = the w word of part j of record e, as quad_round returns them: w[j] is the word of this lane's own record
// (two levels of selects on the bits of j — three v_cndmask; the chain of comparisons j == 0 ? ... : j == 1 ? ... compiled into a
// dozen exec-mask instructions per evaluation)
__device__ inline uint32_t own_word(const uint32_t (&w)[4], int j)
{
  const bool odd = (j & 1) != 0, high = (j & 2) != 0;
  const uint32_t lo = odd ? w[1] : w[0], hi = odd ? w[3] : w[2];
  return high ? hi : lo;
}

// likelihood.cpp:128-132 for one evaluation: dist = r - std::max(root, flat), and the term dist * match_weight — 0 * match_weight
// for an evaluation without a match (dist < 0: the -1 of a lane without a neighbour, or a root beyond r - flat)
__device__ inline float lik_dist(const LikParams& prm, float root)
{
  // root is a finite, non-negative float here: v_max_f32 (IEEE maxNum) returns what std::max(root, flat) returns for every flat,
  // a NaN parameter included (both give root) — ONE instruction; fmaxf() costs two canonicalising v_max on top of its own, the
  // compare-and-select form a v_mov of the uniform operand
  float m;
  asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(root), "s"(prm.match_dist_flat));
  return prm.match_dist_min - m;
}
__device__ inline float lik_term(const LikParams& prm, float dist, bool matched)
{
  return matched ? dist * prm.match_weight : 0.0f * prm.match_weight;
}

// vrec = the lane's own record index (0 for a lane without one: it reads record 0 and ignores the answer); returns
// min d2 over ALL candidates of the lane's voxel: the inline four, then — while any lane of the wavefront still has
// candidates left — one overflow record per round, fetched and reduced the same cooperative way.
__device__ inline float rec_min_d2_quad(const RecGrid& g, float qx, float qy, float qz, uint32_t vrec, bool valid, int lane)
{
  const int j = lane & 3;
  uint32_t w[4];
  // a 128-byte record: parts 0-3 carry the count and the overflow reference like a 64-byte one, parts 4-7 hold four more
  // inline candidates in the other half of the same 128-byte line
  const bool wide = g.rec_parts == 8;
  float best = wide ? quad_round_wide(g.rec, g.rec_bytes32, vrec, qx, qy, qz, j, w) :
                      quad_round(g.rec, g.rec_bytes32, vrec, qx, qy, qz, j, w);
  const uint32_t cap = wide ? 8u : 4u;
  if (g.packed)
  {
    // packed w words: part j of record j — the part this lane fetched of its OWN record — says count and overflow reference
    const uint32_t mine = own_word(w, j);
    const uint32_t thr = g.count_is_records ? g.over_thr : ((cap << g.count_shift) | ((1u << g.count_shift) - 1u));
    if (lanes_gt_u32(mine, thr) != 0ull)
    {
      const uint32_t ext = mine & rec_ext_mask(g);
      // bounded records: a lane whose best inline d2 is within the skip bound of its overflow candidates runs no round
      const bool skip = g.bound_step > 0.0f && best <= rec_bound2(g, mine);
      const uint32_t rounds = (valid && !skip) ? rec_overflow_records(mine >> g.count_shift, cap, g.count_is_records) : 0u;
      for (uint32_t r = 0; wave_any(r < rounds); ++r)
      {
        const bool more = r < rounds;
        uint32_t unused[4];
        const float m = quad_round(g.ovf, g.ovf_bytes32, more ? ext + r : 0u, qx, qy, qz, j, unused);
        best = (more && m < best) ? m : best;
      }
    }
    return best;
  }
  // overflow (more than `cap` candidates): the counts of the quad's four records sit in lane 0 (part 0's w)
  const uint32_t cmax = max(max(w[0], w[1]), max(w[2], w[3]));
  if ((lanes_gt_u32(cmax, cap) & QUAD_LANE0_MASK) != 0ull)
  {
    // count of MY record = part 0's w of record j, held by lane 0 of the quad; first overflow record = part 1's w, lane 1
    const uint32_t n0 = quad_u<QUAD_BCAST0>(w[0]), n1 = quad_u<QUAD_BCAST0>(w[1]), n2 = quad_u<QUAD_BCAST0>(w[2]),
                   n3 = quad_u<QUAD_BCAST0>(w[3]);
    const uint32_t e0 = quad_u<QUAD_BCAST1>(w[0]), e1 = quad_u<QUAD_BCAST1>(w[1]), e2 = quad_u<QUAD_BCAST1>(w[2]),
                   e3 = quad_u<QUAD_BCAST1>(w[3]);
    const uint32_t count = j == 0 ? n0 : j == 1 ? n1 : j == 2 ? n2 : n3;
    const uint32_t ext = j == 0 ? e0 : j == 1 ? e1 : j == 2 ? e2 : e3;
    const uint32_t rounds = (valid && count > cap) ? (count - cap + 3u) / 4u : 0u;
    for (uint32_t r = 0; wave_any(r < rounds); ++r)
    {
      const bool more = r < rounds;
      uint32_t unused[4];
      const float m = quad_round(g.ovf, g.ovf_bytes32, more ? ext + r : 0u, qx, qy, qz, j, unused);
      best = (more && m < best) ? m : best;
    }
  }
  return best;
}

// sqrtf for 0 <= x < r2, correctly rounded wherever its value can reach the result: v_sqrt_f32 (1 ulp) followed by the two
// fused-residual checks of the compiler's own sqrtf expansion (s - 1 ulp and s + 1 ulp against x), WITHOUT that
// expansion's input scaling for x < 2^-96 and without its inf/zero pass-through (5 + 2 instructions of 16). Below 2^-96
// the root is < 3.6e-15: whatever it is rounded to — or flushed to zero — `match_dist_min - max(root, flat)` is the same
// float as long as match_dist_min > 1.2e-7 m (launch_measure uses this form only then); x = 0 gives 0 (the int(s) -+ 1
// neighbours are NaN / the smallest denormal, both rejected by the comparisons).
__device__ inline float sqrt_in_radius(float x)
{
  float s = __builtin_amdgcn_sqrtf(x);
  const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
  const float vp = __builtin_fmaf(-s_dn, s, x), vs = __builtin_fmaf(-s_up, s, x);
  s = (vp <= 0.0f) ? s_dn : s;
  s = (vs > 0.0f) ? s_up : s;
  return s;
}

// One evaluation on the cooperative path, for kernels in which EVERY lane of the wavefront reaches this point (a lane
// without work passes have_point = false): transform, locate, cooperative fetch, distance, term. Returns the float term
// (0 when there is no match) and sets `matched`.
__device__ inline float eval_coop(const RecGrid& rg, const LikParams& prm, const Vec3f pos, const Quat rot, const float4 v,
                                  bool have_point, int lane, bool& matched)
{
  const Vec3f tp = vadd(qrot_trim(rot, Vec3f{ v.x, v.y, v.z }), pos);
  // rescale by dist_weight; without one the weights are 1.0f and x * 1.0f == x bit for bit: no select needed
  const float qx = tp.x * prm.wx, qy = tp.y * prm.wy, qz = tp.z * prm.wz;
  uint32_t ti, sub;
  const bool inside = rec_locate(rg, qx, qy, qz, ti, sub) && have_point;
  // a lane without a voxel reads the table's extra last entry, which is always -1: `valid` is then ONE compare, and its
  // ballot comes straight from that compare
  const int b = rg.brick_table[inside ? ti : rg.ti_empty];
  const bool valid = b >= 0;
  float dist = -1.0f;
  if (wave_any(valid))  // wave-uniform: a wavefront with nothing to look up skips the record loads
  {
    // with buffer loads (array below 4 GB) the record index of an invalid lane may be anything: no select
    const uint32_t rec = (static_cast<uint32_t>(b) << 9) | sub;
    const uint32_t vrec = rg.rec_bytes32 ? rec : (valid ? rec : 0u);
    const float d2 = rec_min_d2_quad(rg, qx, qy, qz, vrec, valid, lane);
    if (valid && d2 < prm.r2)
    {
      const float s = sqrt_in_radius(d2);
      dist = lik_dist(prm, s);
    }
  }
  // likelihood.cpp:129 `if (dist < 0.0) continue;` — one compare decides both the count and the term
  matched = !(dist < 0.0f);
  return lik_term(prm, dist, matched);
}

// One particle's likelihood-field score by one work-group of BLOCK threads: lanes stride the (Morton-ordered) scan.
// s_row != nullptr (the default wherever the scan fits: launch_measure): every float term goes to LDS at its ORIGINAL scan
// index (perm: device index -> index in the caller's array) and the first wavefront runs the reference's own recurrence over
// them — score_like += dist * match_weight, float, sequentially, in the order the caller's cloud holds its points
// (likelihood.cpp:120-134; float_chain.h: the whole wavefront works on it) — so the likelihood is the reference's float bit
// for bit. Unmatched points hold +0 (x + 0.0f == x). s_row: 16-byte aligned, chain_row_floats(n_s) floats.
// s_row == nullptr: fp64 per-lane accumulators, wavefront __shfl reduction, then across the wavefronts in order — the same
// terms in a fixed-order fp64 tree (within the reference's own rounding of its float, ~n_s * 6e-8 at worst).
// Thread 0 returns the sum, the match count and (STATS) the candidates tested; shared by likelihood_kernel and the one-launch
// update (update_kernels.h), which therefore produce the same bits.
template <int BLOCK, int MODE, bool STATS>
__device__ __forceinline__ void lik_particle(const Vec3f pos, const Quat rot, const float4* __restrict__ scan, int n_s,
                                             const LikGrid& g, const RecGrid& rg, const LikParams& prm,
                                             int coop, double& sum_out, unsigned& num_out, unsigned& tested_out,
                                             const uint32_t* __restrict__ perm = nullptr, float* s_row = nullptr)
{
  double acc = 0.0;   // sum of float terms, each exactly representable: fp64 sum is exact to ~1e-16
  unsigned num = 0;   // matched points
  unsigned tested = 0;
  const bool rows = !STATS && s_row != nullptr;
  if (rows)
  {
    // the zeros behind the last term (seq_sum_wave: chain_row_floats)
    if (static_cast<int>(threadIdx.x) < chain_row_floats(n_s) - n_s)
      s_row[n_s + threadIdx.x] = 0.0f;
  }
  if (MODE == 2 && !STATS && coop)
  {
    // cooperative record fetch (rec_min_d2_quad): consecutive lanes hold consecutive (Morton-ordered) scan points; every
    // lane stays in the loop until the whole work-group is done
    for (int base = 0; base < n_s; base += BLOCK)
    {
      const int i = base + static_cast<int>(threadIdx.x);
      const bool have = i < n_s;
      const float4 v = have ? scan[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      const uint32_t slot = (rows && have) ? perm[i] : 0u;
      bool matched;
      const float term = eval_coop(rg, prm, pos, rot, v, have, static_cast<int>(threadIdx.x & 63), matched);
      if (rows)
      {
        if (have)
          s_row[slot] = term;
      }
      else
        acc += static_cast<double>(term);
      num += matched ? 1u : 0u;
    }
  }
  else
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
                                 nearest_d2_rec<STATS>(rg, qx, qy, qz, tested);
    float term = 0.0f;
    if (d2 < prm.r2)  // radiusSearch found a neighbour (strict <)
    {
      const float s = sqrtf(d2);
      const float dist = prm.match_dist_min - (s > prm.match_dist_flat ? s : prm.match_dist_flat);  // :128
      if (!(dist < 0.0f))                                                                           // :129
      {
        term = dist * prm.match_weight;  // :132 (float product, then accumulated)
        if (!rows)
          acc += static_cast<double>(term);
        ++num;
      }
    }
    if (rows)
      s_row[perm[i]] = term;
  }
  // wavefront __shfl reduction, then across the work-group's waves through LDS
  __shared__ double s_acc[BLOCK / 64];
  __shared__ unsigned s_num[BLOCK / 64];
  __shared__ unsigned s_tested[BLOCK / 64];
  if (!rows)
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
  float chain = 0.0f;
  if (rows && wave == 0)
    chain = seq_sum_wave(s_row, n_s, lane);
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
    sum_out = rows ? static_cast<double>(chain) : a;  // (float -> double -> float is the identity: the callers narrow it back)
    num_out = n;
    tested_out = tt;
  }
}

// MODE 0: 27-cell scan of the cell-sorted map (canonical structure of SURVEY.md §8d; also the STATS/K-bar counter)
// MODE 2: candidate-voxel records (map_compiler.h). (MODE 1, candidate RUNS behind a CSR, lost every A/B from round 2 on and
// went in round 6.)
// (the caller-order term row of lik_particle / the one-launch update: dynamic LDS, sized by the launch)
extern __shared__ __attribute__((aligned(16))) float dyn_row[];

// (the body as a device function of the particle: likelihood_kernel is it with blockIdx.x; lik_particle_beam_kernel, update_kernels.h,
// interleaves its work-groups with the beam kernel's in one launch)
template <int BLOCK, int MODE, bool STATS>
__device__ __forceinline__ void likelihood_particle_body(const int p, const float* __restrict__ pose7,
                                                         const float4* __restrict__ scan, int n_s, LikGrid g, RecGrid rg,
                                                         LikParams prm, float* __restrict__ out_lik,
                                                         float* __restrict__ out_ratio, double* __restrict__ out_tested,
                                                         int coop, const uint32_t* __restrict__ perm)
{
  const float* ps = pose7 + 7 * static_cast<size_t>(p);
  const Vec3f pos = { ps[0], ps[1], ps[2] };
  const Quat rot = qnormalized(Quat{ ps[3], ps[4], ps[5], ps[6] });  // state_6dof.h:217
  double a = 0.0;
  unsigned n = 0, tt = 0;
  lik_particle<BLOCK, MODE, STATS>(pos, rot, scan, n_s, g, rg, prm, coop, a, n, tt, perm, perm ? dyn_row : nullptr);
  if (threadIdx.x == 0)
  {
    if (out_lik)
      out_lik[p] = static_cast<float>(a);
    if (out_ratio)
      out_ratio[p] = static_cast<float>(n) / static_cast<float>(n_s);  // :136
    if (STATS && out_tested)
      out_tested[p] = static_cast<double>(tt);
  }
}

template <int BLOCK, int MODE, bool STATS>
__global__ __launch_bounds__(BLOCK) void likelihood_kernel(const float* __restrict__ pose7,
                                                           const float4* __restrict__ scan, int n_s, LikGrid g,
                                                           RecGrid rg, LikParams prm,
                                                           float* __restrict__ out_lik,
                                                           float* __restrict__ out_ratio,
                                                           double* __restrict__ out_tested, int coop,
                                                           const uint32_t* __restrict__ perm = nullptr)
{
  likelihood_particle_body<BLOCK, MODE, STATS>(static_cast<int>(blockIdx.x), pose7, scan, n_s, g, rg, prm, out_lik, out_ratio,
                                               out_tested, coop, perm);
}

// Measured and NOT kept: a wavefront-per-particle kernel for scans of 96-1000 points with thousands of particles (no LDS,
// no barrier, shuffle-only reduction, four particles per work-group) — 4096 x 512: 23.3 us against 21.8 us for
// likelihood_kernel<256>, equal or slower at every shape tried (4096 x 96 ... 100 000 x 96): those shapes are bound by the
// latency of the pose -> table -> record chain at two rounds of resident wavefronts, not by the work-group's fixed costs.
// ---------------------------------------------------------------------------------------------------------
// Small-scan variant (global localisation: hundreds of thousands of particles x 8..32 points each,
// src/lidar_measurement_model_likelihood.cpp:63-77): a wavefront is shared by 64 / W particles, W = the scan size rounded
// up to a power of two; lane = (particle, point). Poses differ between the lanes of a wave, so each lane normalises its
// own quaternion; the W terms of a particle are reduced with width-W shuffles (fp64, fixed order).
// ---------------------------------------------------------------------------------------------------------
// perm != nullptr (the default): the W lanes of a particle park their float terms in LDS at their ORIGINAL scan indices and
// the particle's first lane adds them up as the reference does (likelihood.cpp:120-134: float, sequentially, caller's order;
// at most 32 adds) — bit-identical likelihoods. perm == nullptr: the fp64 width-W shuffle tree.
template <int W, int MODE>
__global__ __launch_bounds__(256) void likelihood_small_kernel(const float* __restrict__ pose7, int n_p,
                                                               const float4* __restrict__ scan, int n_s, LikGrid g,
                                                               RecGrid rg, LikParams prm,
                                                               float* __restrict__ out_lik, float* __restrict__ out_ratio,
                                                               int coop, const uint32_t* __restrict__ perm = nullptr)
{
  const long long gt = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  const long long p = gt / W;
  const int i = static_cast<int>(gt % W);
  float term = 0.0f;
  unsigned num = 0;
  if (MODE == 2 && coop)
  {
    // all 64 lanes go through the cooperative fetch; a lane past the last particle / point carries have = false
    const bool have = p < n_p && i < n_s;
    const float* ps = pose7 + 7 * (p < n_p ? p : 0);
    const Vec3f pos = { ps[0], ps[1], ps[2] };
    const Quat rot = qnormalized(Quat{ ps[3], ps[4], ps[5], ps[6] });
    const float4 v = scan[i < n_s ? i : 0];
    bool matched;
    term = eval_coop(rg, prm, pos, rot, v, have, static_cast<int>(threadIdx.x & 63), matched);
    num = matched ? 1u : 0u;
  }
  else if (p < n_p && i < n_s)
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
                                 nearest_d2_rec<false>(rg, qx, qy, qz, dummy);
    if (d2 < prm.r2)
    {
      const float s = sqrtf(d2);
      const float dist = prm.match_dist_min - (s > prm.match_dist_flat ? s : prm.match_dist_flat);
      if (!(dist < 0.0f))
      {
        term = dist * prm.match_weight;
        num = 1;
      }
    }
  }
  float lik;
  if (perm)
  {
    __shared__ float s_small[256];
    const int row = static_cast<int>(threadIdx.x) - i;  // first lane of this particle
    if (i < n_s)
      s_small[row + static_cast<int>(perm[i])] = term;
    __syncthreads();
    float s = 0.0f;
    if (i == 0)
      for (int j = 0; j < n_s; ++j)
        s = s + s_small[row + j];
    lik = s;
  }
  else
  {
    double acc = static_cast<double>(term);
#pragma unroll
    for (int off = W / 2; off > 0; off >>= 1)
      acc += __shfl_down(acc, off, W);
    lik = static_cast<float>(acc);
  }
#pragma unroll
  for (int off = W / 2; off > 0; off >>= 1)
    num += __shfl_down(num, off, W);
  if (i == 0 && p < n_p)
  {
    if (out_lik)
      out_lik[p] = lik;
    if (out_ratio)
      out_ratio[p] = static_cast<float>(num) / static_cast<float>(n_s);
  }
}

// ---- the reference's float sum inside the tiled kernel (CHAIN) ---------------------------------------------------------
// score_like += dist * match_weight is a float recurrence over the scan in the order the cloud holds its points
// (likelihood.cpp:124-134). When that order IS the order of the device scan array — a scan that came out of the engine's
// own preparation, or any scan under the option strict_order = 3, whose order mcl3dl_hip_scan_order reports — a tile's 256
// terms are consecutive links of every particle's chain, so the chain can run where the terms are, in LDS: the work-group of
// (tile j, particle group g) takes each particle's running sum and match count over from (tile j - 1, g), adds its 256
// terms in array order — G lanes of its first wavefront, one particle each, four terms per 16-byte LDS read — and hands
// them on; the last tile writes likelihood and match ratio. No term ever leaves the CU: no N_s x N_p array, no replay
// kernel, no partials and no finalize launch. Unmatched points hold +0 (x + 0.0f == x for the non-negative sums here), so
// the float is the reference's bit for bit.
// Hand-off: per particle two naturally aligned 8-byte words {value, tag} — {running float sum, tag}, {match count, tag} —
// each written by ONE agent-scope store and polled by ONE agent-scope load (MI355X_MICROARCH.md "handoff-1to1": data-tagged
// granules need no fence and no separate flag). tag = tag0 + tile, tag0 advancing by the tile count with every launch, so
// the words are never cleared and a stale word of an earlier launch never matches. The consumer (tile j) has a higher
// block index than its producer (tile j - 1: same row of tiles -> block - 1, previous row -> block - 8 n_groups + 7) and
// blocks are dispatched in index order, so the producer is always resident or done when the consumer polls; the poll is
// bounded all the same — a lane that gives up raises *err (page-locked: the host checks it after every synchronisation)
// instead of hanging the GPU.
// Cost model (measured, profiles/r05*): a hop is ~1.2 us of dependent adds + ~1.5-3 us of hand-off. The eight tiles of a
// row run side by side on the eight XCDs, so XCD x settles x hops behind XCD 0 (once per launch), and a row must last
// eight hops for the chain to keep up: rows of >= 512 particle groups (8192 particles) never wait.
struct LikChain
{
  unsigned long long* carry;  // [n_p][2] words {sum, tag}, {count, tag}
  uint32_t tag0;              // tag of tile 0 in this launch
  float* out_lik;             // [n_p] written by the last tile
  float* out_ratio;           // [n_p] (may be null)
  float* also_fill;           // [n_p] set to 1 on the way (may be null): the beam score of an update without beam points
  volatile unsigned* err;     // raised when a hand-off did not arrive within the poll bound
};

__device__ inline unsigned long long chain_pack(uint32_t value, uint32_t tag)
{
  return (static_cast<unsigned long long>(tag) << 32) | value;
}

// ---- deferred overflow rounds (tiled kernel, packed 64-byte records) ------------------------------------------------
// On a map of voxel-filter centroids a quarter of the voxels hold more than four candidates. A wavefront runs an overflow
// round as soon as ONE of its 64 evaluations needs it — practically always — with a quarter of its lanes doing useful work
// (or, with 128-byte records, every evaluation pays for eight candidates and two L2 requests). Here an evaluation whose
// record overflows parks its best-so-far in its own term slot and queues (overflow reference, count, lane, particle slot) in
// a per-wavefront LDS queue; whenever 64 are queued the wavefront runs ONE dense overflow round for them — each lane
// rebuilds its query from the queued lane's scan point (ds_bpermute) and the particle's pose (LDS), exactly the arithmetic
// of the first pass, so the minimum, the term and therefore every result are the same bits as without the queue.
constexpr int CHAIN_POLL_MAX = 1 << 20;  // polls of a hand-off word before the lane gives up (~1 s)
constexpr int DEFER_QCAP = 96;   // entries per wavefront: a push never finds more than 63 queued, and flushes first if it would not fit

struct DeferQueue
{
  uint32_t word[4][DEFER_QCAP];  // packed w of the record: count | (bound) | first overflow record (RecGrid::count_shift / ext_bits)
  uint16_t who[4][DEFER_QCAP];   // (particle slot k << 6) | lane
};

// The first four candidates of one evaluation on the cooperative path (all lanes of the wavefront arrive): like eval_coop,
// but a lane whose record holds more than four candidates returns `over` = true with its best-so-far in `best` and the
// record's packed word in `mine` instead of running the overflow rounds.
// over_m = the wavefront's lanes with `over` as a 64-bit mask, built from compare MASKS with scalar operations: a ballot of a
// combined boolean costs a v_cndmask + v_cmp round trip through a VGPR per use (two uses per evaluation: four of 159 instructions).
__device__ inline float eval_coop_first(const RecGrid& rg, const LikParams& prm, const Vec3f pos, const Quat rot, const float4 v,
                                        bool have_point, int lane, bool& matched, bool& over, float& best, uint32_t& mine,
                                        unsigned long long& over_m)
{
  const Vec3f tp = vadd(qrot_trim(rot, Vec3f{ v.x, v.y, v.z }), pos);
  const float qx = tp.x * prm.wx, qy = tp.y * prm.wy, qz = tp.z * prm.wz;
  uint32_t ti, sub;
  const bool inside = rec_locate(rg, qx, qy, qz, ti, sub) && have_point;
  const int b = rg.brick_table[inside ? ti : rg.ti_empty];
  const bool valid = b >= 0;
  float dist = -1.0f;
  over = false;
  over_m = 0ull;
  best = 0.0f;
  mine = 0u;
  const unsigned long long valid_m = __builtin_amdgcn_ballot_w64(valid);  // (straight from the compare)
  if (valid_m != 0ull)
  {
    const uint32_t rec = (static_cast<uint32_t>(b) << 9) | sub;
    const uint32_t vrec = rg.rec_bytes32 ? rec : (valid ? rec : 0u);
    uint32_t w[4];
    best = quad_round(rg.rec, rg.rec_bytes32, vrec, qx, qy, qz, lane & 3, w);
    mine = own_word(w, lane & 3);
    over_m = valid_m & lanes_gt_u32(mine, rg.over_thr);
    // bounded records (RecGrid::bound_step): no overflow candidate is nearer than the voxel's skip bound to ANY query inside
    // the voxel, so a best inline d2 within it is final — the evaluation is not queued
    if (rg.bound_step > 0.0f && over_m != 0ull)
    {
      // (a real branch: on the lattice map almost no wavefront holds an overflowing voxel. The empty asm keeps the compiler from
      // turning the five instructions into unconditional code + selects, which it did as soon as this function was inlined into a
      // body instead of a kernel: + 3 % on the whole kernel, profiles/r06t_refactor_check.txt)
      asm volatile("");
      over_m &= __builtin_amdgcn_fcmpf(best, rec_bound2(rg, mine), 2 /* FCMP_OGT: false for a NaN, like `>` */);
    }
    over = __builtin_amdgcn_inverse_ballot_w64(over_m);
    if (valid && !over && best < prm.r2)
    {
      const float s = sqrt_in_radius(best);
      dist = lik_dist(prm, s);
    }
  }
  matched = !(dist < 0.0f);
  return lik_term(prm, dist, matched);
}

// One dense overflow round for entries [base, base + n) of this wavefront's queue, n <= 64 (all lanes arrive; lane l takes
// entry base + l). s_term / s_cnt rows are the tiled kernel's: the parked best is replaced by the term, a match is counted.
template <int G, int LD>
__device__ inline void defer_drain(const RecGrid& rg, const LikParams& prm, const float (&s_pose)[G][8], float (&s_term)[G][LD],
                                   unsigned (&s_cnt)[G][4], const DeferQueue& q, int wave, int lane, uint32_t base, uint32_t n,
                                   const float4 v)
{
  const bool act = static_cast<uint32_t>(lane) < n;
  const uint32_t slot = act ? base + static_cast<uint32_t>(lane) : base;
  const uint32_t word = q.word[wave][slot];
  const uint32_t who = q.who[wave][slot];
  const int src = static_cast<int>(who & 63u), k = static_cast<int>(who >> 6);
  // the queued lane's scan point, from its registers
  const Vec3f pt = { __shfl(v.x, src), __shfl(v.y, src), __shfl(v.z, src) };
  const Vec3f pos = { s_pose[k][0], s_pose[k][1], s_pose[k][2] };
  const Quat rot = { s_pose[k][3], s_pose[k][4], s_pose[k][5], s_pose[k][6] };
  const Vec3f tp = vadd(qrot_trim(rot, pt), pos);
  const float qx = tp.x * prm.wx, qy = tp.y * prm.wy, qz = tp.z * prm.wz;
  const int t_src = wave * 64 + src;
  float best = s_term[k][t_src];
  const uint32_t ext = word & rec_ext_mask(rg);
  const uint32_t rounds = act ? rec_overflow_records(word >> rg.count_shift, 4u, rg.count_is_records) : 0u;
  for (uint32_t r = 0; wave_any(r < rounds); ++r)
  {
    const bool more = r < rounds;
    uint32_t unused[4];
    const float m = quad_round(rg.ovf, rg.ovf_bytes32, more ? ext + r : 0u, qx, qy, qz, lane & 3, unused);
    best = (more && m < best) ? m : best;
  }
  float dist = -1.0f;
  if (act && best < prm.r2)
  {
    const float s = sqrt_in_radius(best);
    dist = lik_dist(prm, s);
  }
  if (act)
  {
    s_term[k][t_src] = lik_term(prm, dist, !(dist < 0.0f));
    if (!(dist < 0.0f))
      atomicAdd(&s_cnt[k][wave], 1u);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Tile-major variant for large scans: one work-group = one 256-point scan tile x G particles.
//
//  * the scan point of each lane stays in registers for all G particles; the G (normalised) poses are staged through
//    LDS once per work-group and read back as broadcasts;
//  * blockIdx -> (tile, particle group) is XCD-aware: work-groups are dispatched round-robin over the 8 XCDs
//    (block b runs on XCD b % 8), so XCD x is given the tiles t == x (mod 8) and walks them one after the other over all
//    particle groups (the last n_tiles % 8 tiles are shared out by particle group). A tile is a spatially compact patch
//    (Morton order), so the voxel records it touches under every particle pose (~1 MB) stay resident in that XCD's 4 MB
//    L2 instead of every work-group sweeping the whole scan;
//  * per-(particle, lane) float terms go to LDS and are summed in fp64 in a fixed order (deterministic), one partial per
//    (tile, particle); lik_finalize_kernel adds the tiles in order.
// Same per-point arithmetic as likelihood_kernel — identical terms — only the (fp64) summation order differs.
//
// COOP (MODE 2 only): the quad-cooperative record fetch of rec_min_d2_quad plus the VALU-trimmed transform / sqrt. Same
// terms, bit for bit. MINW = wavefronts per SIMD the register allocation must leave room for (G = 32 holds 33 KB of LDS:
// 4 at most).
// DEFER (COOP only; packed 64-byte records): overflow rounds queued per wavefront and run densely (above).
// (the kernel's body as a device function of the work-group's index: likelihood_tiled_kernel below is it with blockIdx.x, the
// heterogeneous launch of both models — lik_beam_kernel, update_kernels.h — calls it with the index it assigns)
template <int G, int MODE, int MINW = 8, bool COOP = false, bool DEFER = false, bool CHAIN = false>
__device__ __forceinline__ void likelihood_tiled_body(const uint32_t block_index, const float* __restrict__ pose7, int n_p,
                                                      const float4* __restrict__ scan, int n_s, int n_tiles, int n_groups,
                                                      LikGrid g, RecGrid rg, LikParams prm,
                                                      double* __restrict__ partial_sum, unsigned* __restrict__ partial_cnt,
                                                      const uint32_t* __restrict__ scan_perm, float* __restrict__ strict_terms,
                                                      int strict_skew4, LikChain ch)
{
  // strict_terms != nullptr ("strict_order" option): besides the fp64 partials, every float term is stored at
  // [particle group][original scan index][G] so that lik_strict_sum_rows_kernel can add them in the reference's own order.
  // (CHAIN: rows padded to 260 floats — the chain's lanes read DIFFERENT rows at the same column, 16 bytes at a time: with a
  // row length of 256 they would all hit the same LDS banks)
  constexpr int TERM_LD = CHAIN ? 260 : 256;
  __shared__ float s_pose[G][8];        // px,py,pz, qx,qy,qz,qw (normalised), valid
  __shared__ __attribute__((aligned(16))) float s_term[G][TERM_LD];
  __shared__ unsigned s_cnt[G][4];
  // Work-groups are dispatched round-robin over the 8 XCDs (block b runs on XCD b % 8). The tiles of the largest
  // multiple of eight are INTERLEAVED: XCD x walks tiles x, x + 8, ... one after the other over all particle groups —
  // every XCD gets an even sample of cheap (mostly empty bricks) and expensive (along the walls) tiles; contiguous
  // ranges of tiles per XCD were measured 7 % (C2) to 13 % (C5) slower. The remaining n_tiles % 8 tiles — all of them
  // for a scan shorter than 2048 points — are cut into eight contiguous ranges of (tile, group) pairs, so that a short
  // scan still uses every XCD (32-bit arithmetic: the launch has fewer than 2^31 work-groups, plan_lik checks).
  const uint32_t xcd = block_index & 7u;
  const uint32_t seq = block_index >> 3;
  const uint32_t ng = static_cast<uint32_t>(n_groups);
  const uint32_t full_tiles = static_cast<uint32_t>(n_tiles) & ~7u;
  const uint32_t per_xcd_full = (full_tiles >> 3) * ng;
  int tile, group;
  if constexpr (CHAIN)
  {
    // rows of eight tiles, XCD x takes tile 8 row + x over all particle groups; a last partial row leaves XCDs idle (the
    // shared-out remainder below would break "the producer's block index is lower")
    const uint32_t row = seq / ng;
    tile = static_cast<int>(row * 8u + xcd);
    group = static_cast<int>(seq - row * ng);
    if (tile >= n_tiles)
      return;
  }
  else if (seq < per_xcd_full)
  {
    const uint32_t row = seq / ng;
    tile = static_cast<int>(row * 8u + xcd);
    group = static_cast<int>(seq - row * ng);
  }
  else
  {
    const uint32_t rem_items = (static_cast<uint32_t>(n_tiles) - full_tiles) * ng;
    const uint32_t per_xcd_rem = (rem_items + 7u) >> 3;
    const uint32_t s2 = seq - per_xcd_full;
    const uint32_t item = xcd * per_xcd_rem + s2;
    if (s2 >= per_xcd_rem || item >= rem_items)
      return;
    const uint32_t row = item / ng;
    tile = static_cast<int>(full_tiles + row);
    group = static_cast<int>(item - row * ng);
  }
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
  if constexpr (COOP && MODE == 2 && DEFER)
  {
    __shared__ DeferQueue s_q;
    uint32_t qn = 0;  // entries queued by this wavefront (uniform)
    for (int k = 0; k < n_valid; ++k)
    {
      const Vec3f pos = { s_pose[k][0], s_pose[k][1], s_pose[k][2] };
      const Quat rot = { s_pose[k][3], s_pose[k][4], s_pose[k][5], s_pose[k][6] };
      bool matched, over;
      float best;
      uint32_t mine;
      unsigned long long om;
      const float term = eval_coop_first(rg, prm, pos, rot, v, have_point, lane, matched, over, best, mine, om);
      s_term[k][t] = over ? best : term;
      const unsigned long long m = __builtin_amdgcn_ballot_w64(matched);
      if (lane == 0)
        s_cnt[k][wave] = static_cast<unsigned>(__popcll(m));
      if (om != 0ull)
      {
        const uint32_t n_new = static_cast<uint32_t>(__popcll(om));
        if (qn + n_new > static_cast<uint32_t>(DEFER_QCAP))
        {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          defer_drain(rg, prm, s_pose, s_term, s_cnt, s_q, wave, lane, 0u, qn, v);
          qn = 0;
        }
        if (over)
        {
          const uint32_t slot = qn + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(om >> 32),
                                                               __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(om), 0u));
          s_q.word[wave][slot] = mine;
          s_q.who[wave][slot] = static_cast<uint16_t>((static_cast<uint32_t>(k) << 6) | static_cast<uint32_t>(lane));
        }
        qn += n_new;
        if (qn >= 64u)
        {
          // the LAST 64 entries: what stays queued keeps its place
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          defer_drain(rg, prm, s_pose, s_term, s_cnt, s_q, wave, lane, qn - 64u, 64u, v);
          qn -= 64u;
        }
      }
    }
    if (qn != 0u)
    {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      defer_drain(rg, prm, s_pose, s_term, s_cnt, s_q, wave, lane, 0u, qn, v);
    }
  }
  else if constexpr (COOP && MODE == 2)
  {
    // every lane stays active through the cooperative fetch (DPP reads 0 from an inactive lane): a lane without a scan
    // point, outside the grid or without a brick simply carries valid = false
    for (int k = 0; k < n_valid; ++k)
    {
      const Vec3f pos = { s_pose[k][0], s_pose[k][1], s_pose[k][2] };
      const Quat rot = { s_pose[k][3], s_pose[k][4], s_pose[k][5], s_pose[k][6] };
      bool matched;
      const float term = eval_coop(rg, prm, pos, rot, v, have_point, lane, matched);
      s_term[k][t] = term;
      const unsigned long long m = __builtin_amdgcn_ballot_w64(matched);
      if (lane == 0)
        s_cnt[k][wave] = static_cast<unsigned>(__popcll(m));
    }
  }
  else
  for (int k = 0; k < n_valid; ++k)
  {
    const Vec3f pos = { s_pose[k][0], s_pose[k][1], s_pose[k][2] };
    const Quat rot = { s_pose[k][3], s_pose[k][4], s_pose[k][5], s_pose[k][6] };
    float term = 0.f;
    bool matched = false;
    if (have_point)
    {
      const Vec3f tp = vadd(qrot(rot, Vec3f{ v.x, v.y, v.z }), pos);
      // rescale by dist_weight; without one the weights are 1.0f and x * 1.0f == x bit for bit, so no select is needed
      const float qx = tp.x * prm.wx, qy = tp.y * prm.wy, qz = tp.z * prm.wz;
      unsigned dummy = 0;
      const float d2 = MODE == 0 ? nearest_d2<false>(g, qx, qy, qz, dummy) :
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
    const unsigned long long m = __ballot(matched);
    if (lane == 0)
      s_cnt[k][wave] = static_cast<unsigned>(__popcll(m));
  }
  __syncthreads();
  if constexpr (CHAIN)
  {
    if (wave != 0 || lane >= n_valid)
      return;
    const int k = lane;
    const size_t p = static_cast<size_t>(group) * G + k;
    unsigned long long* cw = ch.carry + 2 * p;
    float s = 0.0f;
    uint32_t cnt = 0;
    if (tile > 0)
    {
      const uint32_t want = ch.tag0 + static_cast<uint32_t>(tile) - 1u;
      unsigned long long a = 0, b = 0;
      int polls = 0;
      bool got = false;
      // (every lane polls its own two words; lanes that have theirs idle through the others' iterations)
      while (!got)
      {
        a = __hip_atomic_load(cw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b = __hip_atomic_load(cw + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        got = static_cast<uint32_t>(a >> 32) == want && static_cast<uint32_t>(b >> 32) == want;
        if (!got)
        {
          if (++polls > CHAIN_POLL_MAX)
          {
            *ch.err = 1u;
            break;
          }
          __builtin_amdgcn_s_sleep(8);
        }
      }
      s = __uint_as_float(static_cast<uint32_t>(a));
      cnt = static_cast<uint32_t>(b);
    }
    const float4* row = reinterpret_cast<const float4*>(&s_term[k][0]);
#pragma unroll 8
    for (int j = 0; j < 64; ++j)
    {
      const float4 t4 = row[j];
      s = s + t4.x;
      s = s + t4.y;
      s = s + t4.z;
      s = s + t4.w;
    }
    cnt += s_cnt[k][0] + s_cnt[k][1] + s_cnt[k][2] + s_cnt[k][3];
    if (tile == n_tiles - 1)
    {
      ch.out_lik[p] = s;
      if (ch.out_ratio)
        ch.out_ratio[p] = static_cast<float>(cnt) / static_cast<float>(n_s);
      if (ch.also_fill)
        ch.also_fill[p] = 1.0f;
    }
    else
    {
      const uint32_t tag = ch.tag0 + static_cast<uint32_t>(tile);
      __hip_atomic_store(cw, chain_pack(__float_as_uint(s), tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(cw + 1, chain_pack(cnt, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  if (strict_terms)
  {
    // rows of G floats, [group][original scan index][G]: G / 4 lanes write one row as float4s (one 16..128-byte run)
    constexpr int Q = G / 4;
    // (group regions are strict_skew4 float4s further apart than their n_s rows: see lik_strict_sum_rows_kernel)
    float4* rows = reinterpret_cast<float4*>(strict_terms) + static_cast<size_t>(group) * (static_cast<size_t>(n_s) * Q + strict_skew4);
#pragma unroll
    for (int j = 0; j < Q; ++j)
    {
      const int e = t + 256 * j, pt = e / Q, k4 = (e % Q) * 4;
      const int src = tile * 256 + pt;
      if (src < n_s)
        rows[static_cast<size_t>(scan_perm[src]) * Q + (e % Q)] =
            make_float4(s_term[k4][pt], s_term[k4 + 1][pt], s_term[k4 + 2][pt], s_term[k4 + 3][pt]);
    }
  }
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

template <int G, int MODE, int MINW = 8, bool COOP = false, bool DEFER = false, bool CHAIN = false>
__global__ __launch_bounds__(256, MINW) void likelihood_tiled_kernel(const float* __restrict__ pose7, int n_p,
                                                               const float4* __restrict__ scan, int n_s, int n_tiles,
                                                               int n_groups, LikGrid g, RecGrid rg,
                                                               LikParams prm, double* __restrict__ partial_sum,
                                                               unsigned* __restrict__ partial_cnt,
                                                               const uint32_t* __restrict__ scan_perm,
                                                               float* __restrict__ strict_terms, int strict_skew4 = 0,
                                                               LikChain ch = LikChain{})
{
  likelihood_tiled_body<G, MODE, MINW, COOP, DEFER, CHAIN>(blockIdx.x, pose7, n_p, scan, n_s, n_tiles, n_groups, g, rg, prm, partial_sum,
                                                           partial_cnt, scan_perm, strict_terms, strict_skew4, ch);
}

// Sums the per-tile partials of each particle: 32 particles per work-group, 8 lanes per particle each walk every 8th tile
// (independent loads, 1/8 of the dependent chain), then lane-slice 0 adds the 8 sub-sums in fixed order (deterministic).
// also_fill (may be null): an array of n_p floats set to 1 on the way — the beam score of an update without beam points
// (beam.cpp:130-133), which would otherwise cost a launch of its own.
__global__ __launch_bounds__(256) void lik_finalize_kernel(const double* __restrict__ partial_sum,
                                                           const unsigned* __restrict__ partial_cnt, int n_tiles, int n_p,
                                                           int n_s, float* __restrict__ out_lik,
                                                           float* __restrict__ out_ratio, float* __restrict__ also_fill)
{
  __shared__ double s_a[8][32];
  __shared__ unsigned s_n[8][32];
  const int pl = threadIdx.x & 31, slice = threadIdx.x >> 5;
  const int p = blockIdx.x * 32 + pl;
  double a = 0.0;
  unsigned n = 0;
  if (p < n_p)
  {
    for (int tl = slice; tl < n_tiles; tl += 8)
    {
      a += partial_sum[static_cast<size_t>(tl) * n_p + p];
      n += partial_cnt[static_cast<size_t>(tl) * n_p + p];
    }
  }
  s_a[slice][pl] = a;
  s_n[slice][pl] = n;
  __syncthreads();
  if (slice != 0 || p >= n_p)
    return;
#pragma unroll
  for (int k = 1; k < 8; ++k)
  {
    a += s_a[k][pl];
    n += s_n[k][pl];
  }
  if (out_lik)
    out_lik[p] = static_cast<float>(a);
  if (out_ratio)
    out_ratio[p] = static_cast<float>(n) / static_cast<float>(n_s);
  if (also_fill)
    also_fill[p] = 1.0f;
}

// "strict_order": score_like += dist * match_weight in the reference's own order (likelihood.cpp:120-134): float adds,
// sequentially, in ORIGINAL scan order. Unmatched points hold 0 (x + 0.0f == x), so the result is the reference's
// float, bit for bit. One work-group per particle group streams its [n_s][G] term rows. Wavefronts 1-3 are loaders:
// they fetch the next ROWS rows and store them TRANSPOSED ([particle][row]) into the other LDS buffer; wavefront 0 is the
// adder: lanes 0..G-1, one per particle, read four consecutive rows per ds_read_b128 and run the dependent add chain.
// The chain (n_s adds per particle) is the critical path; the loads hide behind it.
// Measured and NOT kept (round 3, C5: 512 groups x 65 536 terms, 0.68 ms): 32 KB chunks with two chunks in flight in
// registers and two work-groups per CU (0.91 ms: slower), and two adjacent groups per work-group so that 512 groups run in
// one round of 256 work-groups (0.68 ms: no change) — the kernel reads its 2.1 GB of terms at ~3.1 TB/s either way.
// What the kernel costs (C5: 512 groups x 65 536 rows of 64 B = 2.1 GB of terms, written by the tiled kernel just before):
//   * the adder's dependent chain, ~11 cycles per term: 0.34 ms for 65 536 terms whatever the number of chains a wavefront
//     carries side by side — so GPW particle groups share one adder wavefront (lane = (group, particle), GPW x G <= 64): the
//     512 groups of C5 are 256 work-groups, ONE round of one chain (round 3: one group per work-group, two rounds or two
//     work-groups per CU: 0.75 / 0.55 ms);
//   * streaming the terms: with one 64 KB batch of loads in flight per CU the stream ran at 2.9 TB/s — 5.8 us per batch,
//     i.e. bound by memory LATENCY, with either grouping (profiles/r04i_C5_strict_gpw.txt: 0.727 / 0.734 ms). Hence 15 loader
//     wavefronts per work-group with TWO chunks in flight in their registers (the chunk the adder needs next is already in
//     LDS): ~190 KB in flight per CU.
//   * where the groups' regions lie: 65 536 rows of 64 B are exactly 4 MiB, so with the regions back to back every work-group
//     (and, in the tiled kernel, every writer of one scan tile) is at the same offset modulo 4 MiB at the same time — the same
//     HBM channel. The regions are therefore `skew4` float4s (4352 B: 4 KiB + 256 B) further apart than their rows.
// CHUNK = bytes of terms per group and LDS buffer (2 x GPW x CHUNK <= 128 KB): 65536 / 32768 / 16384 for GPW 1 / 2 / 4.
constexpr int STRICT_SKEW4 = 272;

// The chunk is kept in LDS as it comes from memory ([group][row][G]): the loaders park their float4s with one ds_write_b128 each
// (round 3's form transposed it — four scattered ds_write_b32 per float4, whose bank conflicts made the parking of a chunk take as
// long as the adder needed for it; gone in round 6), and the adder lane (group, particle)
// reads ONE float per row, sixteen rows ahead of the add that consumes them: the chain then waits for the adds alone, not for
// an LDS round trip every sixteen terms (~11 -> ~6 cycles per term). Group regions sit G floats apart from a multiple of the
// bank count so that the GPW groups' rows do not collide. Same terms, same order, same float: bit-identical.
template <int G, int CHUNK, int GPW>
__global__ __launch_bounds__(1024) void lik_strict_sum_rows_kernel(const float* __restrict__ terms, int n_s, int n_p,
                                                                   int n_groups, float* __restrict__ out_lik, int skew4, int accumulate = 0)
{
  constexpr int Q = G / 4;               // float4s per row
  constexpr int ROWS = CHUNK / (4 * G);  // rows per chunk and group
  constexpr int GSTRIDE = ROWS * G + G;  // floats between the groups' regions of one buffer
  constexpr int LOADERS = 960;
  constexpr int ELEMS = GPW * ROWS * Q;  // float4s per chunk
  constexpr int PER = (ELEMS + LOADERS - 1) / LOADERS;
  constexpr int AHEAD = 16;
  static_assert(GPW * G <= 64, "one adder wavefront");
  static_assert(ROWS % AHEAD == 0, "whole batches of rows");
  __shared__ __attribute__((aligned(16))) float buf[2][GPW * GSTRIDE];
  const int group0 = blockIdx.x * GPW, t = threadIdx.x;
  const float4* base = reinterpret_cast<const float4*>(terms);
  const int n_chunks = (n_s + ROWS - 1) / ROWS;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // rows past n_s and groups past the last one arrive as zeros (x + 0.0f == x): every chunk is parked whole
  const auto fetch = [&](int c, float4 (&reg)[PER])
  {
    const int lt = t - 64;
    const int first = c * ROWS, n_rows = min(ROWS, n_s - first);
#pragma unroll
    for (int j = 0; j < PER; ++j)
    {
      const int e = lt + LOADERS * j;
      const int gl = e / (ROWS * Q), rem = e - gl * (ROWS * Q), r = rem / Q;
      const bool live = c < n_chunks && e < ELEMS && r < n_rows && group0 + gl < n_groups;
      reg[j] = live ? base[static_cast<size_t>(group0 + gl) * (static_cast<size_t>(n_s) * Q + skew4) + static_cast<size_t>(first) * Q + rem] : z4;
    }
  };
  const auto park = [&](int c, const float4 (&reg)[PER])
  {
    const int lt = t - 64;
    float* dst = buf[c & 1];
#pragma unroll
    for (int j = 0; j < PER; ++j)
    {
      const int e = lt + LOADERS * j;
      const int gl = e / (ROWS * Q), rem = e - gl * (ROWS * Q);
      if (e < ELEMS)
      {
        float* d = dst + gl * GSTRIDE + rem * 4;  // (16-byte aligned: one ds_write_b128)
        d[0] = reg[j].x;
        d[1] = reg[j].y;
        d[2] = reg[j].z;
        d[3] = reg[j].w;
      }
    }
  };
  float4 ra[PER], rb[PER];  // chunks in flight: even ones in ra, odd ones in rb
  const bool loader = t >= 64;
  if (loader && n_chunks > 0)
  {
    fetch(0, ra);
    park(0, ra);
    fetch(1, rb);
    fetch(2, ra);
  }
  __syncthreads();
  float score = 0.0f;
  if (accumulate)
  {
    // (a scan replayed in chunks of its original order: this launch continues the sums the previous chunk's launch left)
    const int p0 = (group0 + t / G) * G + (t % G);
    if (t < GPW * G && group0 + t / G < n_groups && p0 < n_p)
      score = out_lik[p0];
  }
  // one trip of the pipeline: the loaders park chunk c + 1 out of `reg` and refill it with chunk c + 3, the adder wavefront
  // runs chunk c. Called for even c with rb and for odd c with ra, so that each register array is named statically.
  const auto trip = [&](int c, float4 (&reg)[PER])
  {
    if (loader)
    {
      if (c + 1 < n_chunks)
      {
        park(c + 1, reg);
        fetch(c + 3, reg);
      }
    }
    else if (t < GPW * G)
    {
      const float* cur = buf[c & 1] + (t / G) * GSTRIDE + (t % G);
      float nx[AHEAD];
#pragma unroll
      for (int j = 0; j < AHEAD; ++j)
        nx[j] = cur[j * G];
#pragma unroll
      for (int r = 0; r < ROWS; r += AHEAD)
      {
        float now[AHEAD];
#pragma unroll
        for (int j = 0; j < AHEAD; ++j)
          now[j] = nx[j];
        if (r + AHEAD < ROWS)
        {
#pragma unroll
          for (int j = 0; j < AHEAD; ++j)
            nx[j] = cur[(r + AHEAD + j) * G];
        }
        // (the scheduler otherwise sinks these reads to just ahead of their own adds, and the chain waits for an LDS round trip
        // every sixteen terms as before)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < AHEAD; ++j)
          score += now[j];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  };
  for (int c = 0; c < n_chunks; c += 2)
  {
    trip(c, rb);
    if (c + 1 < n_chunks)
      trip(c + 1, ra);
  }
  const int p = (group0 + t / G) * G + (t % G);
  if (t < GPW * G && group0 + t / G < n_groups && p < n_p)
    out_lik[p] = score;
}

// pf::measure's `sum += p.probability_` (pf.h:255-260) as the reference runs it — a float recurrence in particle order — behind
// pf_partial / pf_reduce: the result replaces the fp64 tree sum in packed[0] so that pf_apply_kernel divides by exactly the
// reference's float. One work-group: 256 lanes stage 4096 weights at a time in LDS (coalesced), the first wavefront runs
// the recurrence over them (float_chain.h: 256 terms a pass instead of one), carrying the sum from chunk to chunk.
__global__ __launch_bounds__(256) void pf_strict_sum_kernel(const float* __restrict__ w_new, int n,
                                                            double* __restrict__ packed)
{
  __shared__ __attribute__((aligned(16))) float buf[4096 + 4];
  if (blockIdx.x != 0)
    return;
  float sum = 0.0f;
  for (int base = 0; base < n; base += 4096)
  {
    const int m = min(4096, n - base), m4 = chain_row_floats(m);
    for (int j = threadIdx.x; j < m4; j += 256)
      buf[j] = j < m ? w_new[base + j] : 0.0f;
    __syncthreads();
    if (threadIdx.x < 64)
      sum = seq_sum_wave(buf, m, static_cast<int>(threadIdx.x), sum);
    __syncthreads();
  }
  if (threadIdx.x == 0)
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

}  // namespace mcl3dl
