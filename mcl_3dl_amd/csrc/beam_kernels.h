// beam_kernels.h — beam model (R6/R7/R8): the DDA ray walk and its kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.h"
#include "map_structs.h"
#pragma clang fp contract(off)

namespace mcl3dl
{
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

// a * b + c on the low 24 bits of a and b in ONE instruction (v_mad_u32_u24; left to itself the compiler picks the 64-bit
// v_mad_u64_u32 for the first of two chained multiply-adds)
// b must be wave-uniform (a kernel argument here): it travels as the instruction's one scalar operand
__device__ inline unsigned mad24(unsigned a, unsigned b, unsigned c)
{
  unsigned r;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
  return r;
}

// any ACTIVE lane true? (the ballot straight from the predicate)
__device__ inline bool wave_any_active(bool p)
{
  return __builtin_amdgcn_ballot_w64(p) != 0ull;
}

// isPointWithinMap, raycast_using_dda.h:260-270 (a begin outside the map: max_movement_ = 0 -> getNextCastResult false -> LONG)
__device__ inline bool point_within_map(const DdaGrid& g, const Vec3f b)
{
  return !((b.x < g.min_x) || (g.max_x < b.x) || (b.y < g.min_y) || (g.max_y < b.y) || (b.z < g.min_z) || (g.max_z < b.z));
}

// Measured and NOT kept (round 3, commit ae2f9a2): the nine fp64 divisions of the ray set-up (toIndex of the end point,
// t_delta and initial_edges per axis) as reciprocal multiplies with an exactness check and the division as the rare
// fallback — bit-identical on 3 x 2^24 self-test inputs built on the undecidable cases, and no faster: C3 beam group
// 0.1263 -> 0.1270 ms, 16 384 rays per particle 4.57 -> 4.65 ms (profiles/r03m_beam_fast_div_ab.txt). The set-up is not
// where a ray's time goes; the walk is.
// toIndex, raycast_using_dda.h:205-210: float difference, double division, truncation toward zero
__device__ inline void to_index(const DdaGrid& g, const Vec3f p, int& ix, int& iy, int& iz)
{
  ix = static_cast<int>(static_cast<double>(p.x - g.min_x) / g.grid);
  iy = static_cast<int>(static_cast<double>(p.y - g.min_y) / g.grid);
  iz = static_cast<int>(static_cast<double>(p.z - g.min_z) / g.grid);
}

// The walk from a begin point that lies within the map and whose voxel (bx, by, bz) = toIndex(begin) is known.
// OVERLAY = false: compiled without the lookup of a map update's points (DdaGrid::ov_*) — for launches that know there is none
// (six VGPRs and a wavefront of occupancy less in the beam kernel).
template <bool STATS, bool TRACE = false, bool OVERLAY = true>
__device__ inline int cast_ray_from(const DdaGrid& g, const BeamParams& bp, Vec3f b, int bx, int by, int bz, Vec3f e_org,
                                    int* hit, unsigned& st_steps, unsigned& st_occ, unsigned& st_tested, RayTrace* tr = nullptr)
{
  *hit = -1;
  // setRay, :76-103
  const Vec3f diff = vsub(e_org, b);
  const float nrm = sqrtf(vdot(diff, diff));
  const Vec3f dir = { diff.x / nrm, diff.y / nrm, diff.z / nrm };
  const Vec3f e = vadd(e_org, vscale(dir, g.hit_tolerance_f));
  int ex, ey, ez;
  to_index(g, e, ex, ey, ez);
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
  uint32_t word_lo = 0u, word_hi = 0u;  // z = 0, 1 resp. z = 2, 3 of the brick
  // Two nested loops instead of one ("while-while" traversal): the inner loop only WALKS — to the next occupied voxel
  // or to the end of the ray — and the point tests of that voxel run after it. On a wavefront the inner loop ends when
  // every ray has found its voxel (or run out), so the long, double-precision test body executes once per round for all
  // 64 rays together instead of once per step for whichever ray happens to sit on an occupied voxel (a ray visits
  // ~1.0 occupied voxel on its way: measured). The per-ray sequence of operations is unchanged.
  // The walk itself is straight-line code under per-lane predicates with ONE backward branch on a ballot: a ray that has
  // found its voxel (or run out) simply stops changing its state while the others walk on. (Written with `break`s the
  // same loop compiled to ~35 scalar mask operations and 7 branches per step.)
  const int bnx = g.bnx, bny = g.bny;
  for (;;)
  {
    bool found = false;
    bool walking = true;
    do
    {
      // getNextCastResult, :106-159: ++pos; pos >= max_movement ends the ray
      const int pos1 = pos + 1;
      const bool step = walking && (pos1 < max_movement);
      pos = step ? pos1 : pos;
      // axis choice of :114-147 (strict <, ties fall to the later axis); only the chosen axis changes (incrementIndex, :192-203)
      const bool x_first = tmx < tmy;
      const bool ax = step && x_first && (tmx < tmz);
      const bool ay = step && !x_first && (tmy < tmz);
      const bool az = step && !(x_first && (tmx < tmz)) && !(!x_first && (tmy < tmz));
      cx += ax ? sx : 0;
      cy += ay ? sy : 0;
      cz += az ? sz : 0;
      // initial_edges + t_delta * abs(pos - begin) (:131, :139, :147): |(float)(c - b)| == (float)|c - b| for every int, and
      // the absolute value is a free source modifier of the multiply
      const float nx_t = iex + tdx * fabsf(static_cast<float>(cx - bx));
      const float ny_t = iey + tdy * fabsf(static_cast<float>(cy - by));
      const float nz_t = iez + tdz * fabsf(static_cast<float>(cz - bz));
      tmx = ax ? nx_t : tmx;
      tmy = ay ? ny_t : tmy;
      tmz = az ? nz_t : tmz;
      // only the moved index can have left the grid (the others were checked when they moved; begin is inside the map);
      // leaving it ends the ray
      const bool go = step && static_cast<unsigned>(cx) < static_cast<unsigned>(g.nx) &&
                      static_cast<unsigned>(cy) < static_cast<unsigned>(g.ny) &&
                      static_cast<unsigned>(cz) < static_cast<unsigned>(g.nz);
      if (STATS)
        st_steps += go ? 1u : 0u;
      if (TRACE)
      {
        if (go)
        {
          if (tr->n < tr->max)
          {
            tr->xyz[3 * tr->n + 0] = static_cast<float>((cx + 0.5) * g.grid + g.min_x);
            tr->xyz[3 * tr->n + 1] = static_cast<float>((cy + 0.5) * g.grid + g.min_y);
            tr->xyz[3 * tr->n + 2] = static_cast<float>((cz + 0.5) * g.grid + g.min_z);
          }
          ++tr->n;
        }
      }
      // hasIntersection, :237-258: occupancy bit first. Brick index < 2^31 / 64 (total voxels < 2^31); 24-bit
      // multiply-adds where build_dda_grid found every factor below 2^24 (a 32-bit integer multiply is quarter rate)
      const int brick = g.mul24_ok ? static_cast<int>(mad24(mad24(static_cast<unsigned>(cz >> 2), static_cast<unsigned>(bny),
                                                                  static_cast<unsigned>(cy >> 2)),
                                                            static_cast<unsigned>(bnx), static_cast<unsigned>(cx >> 2))) :
                                     ((cz >> 2) * bny + (cy >> 2)) * bnx + (cx >> 2);
      if (go && brick != cur_brick)
      {
        cur_brick = brick;
        const uint2 w2 = reinterpret_cast<const uint2*>(g.bricks)[brick];
        word_lo = w2.x;
        word_hi = w2.y;
      }
      // bit z * 16 + y * 4 + x of the brick word, as a 32-bit test on the half cz & 2 selects
      // a ray that did not step reads an all-ones word: "occupied" stops its walk, and `go` keeps it out of `found` — so the
      // loop condition is ONE compare whose lane mask is the ballot
      const uint32_t half = go ? ((cz & 2) ? word_hi : word_lo) : 0xffffffffu;
      walking = __builtin_amdgcn_ubfe(half, static_cast<uint32_t>((((cz & 1) << 2) | (cy & 3)) << 2 | (cx & 3)), 1u) == 0u;
      found = found || (go && !walking);
    } while (wave_any_active(walking));
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
    bool in_overlay = false;
    if (OVERLAY && collided < 0 && g.ov_n > 0)
    {
      // the update's points of this voxel, behind the base map's (they are later in pc_map2): binary search for the run
      int lo = 0, hi = g.ov_n;
      while (lo < hi)
      {
        const int mid = (lo + hi) >> 1;
        if (g.ov_key[mid] < static_cast<uint32_t>(v))
          lo = mid + 1;
        else
          hi = mid;
      }
      for (int k = lo; k < g.ov_n && g.ov_key[k] == static_cast<uint32_t>(v); ++k)
      {
        const float4 t = g.ov_pts[k];
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
          collided = k;
          cp = t;
          in_overlay = true;
          break;
        }
      }
    }
    if (collided < 0)
      continue;
    const int hit_index = (OVERLAY && in_overlay) ? static_cast<int>(g.ov_base + g.ov_idx[collided]) :
                                                    static_cast<int>(g.pt_index[collided]);
    if (TRACE)
    {
      tr->collided = 1;
      *hit = hit_index;
      return 0;
    }
    // getBeamStatus, beam.cpp:164-187
    if (__float_as_uint(cp.w) > bp.filter_label_max)
      continue;
    *hit = hit_index;
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

// Casts one ray from an arbitrary begin point (explicit rays: getBeamStatus for the debug markers, the waypoint trace).
template <bool STATS, bool TRACE = false, bool OVERLAY = true>
__device__ inline int cast_ray(const DdaGrid& g, const BeamParams& bp, Vec3f b, Vec3f e_org, int* hit,
                               unsigned& st_steps, unsigned& st_occ, unsigned& st_tested, RayTrace* tr = nullptr)
{
  *hit = -1;
  if (!point_within_map(g, b))
    return 2;
  int bx, by, bz;
  to_index(g, b, bx, by, bz);
  return cast_ray_from<STATS, TRACE, OVERLAY>(g, bp, b, bx, by, bz, e_org, hit, st_steps, st_occ, st_tested, tr);
}

// Everything about a ray that depends only on (particle, origin): all N_b rays of a particle share it, so a small
// kernel computes it once instead of every ray repeating a quaternion normalisation, a rotation and three
// double-precision divisions. Same expressions, same order: same bits.
struct BeamOrigin
{
  float4 rot;    // normalised quaternion (x, y, z, w): transforms the end points (beam.cpp:139)
  float4 begin;  // s.pos_ + s.rot_ * origin with the RAW quaternion (beam.cpp:145); w = 1 if within the map
  int4 voxel;    // toIndex(begin)
  float4 pos;    // particle position
};

// what depends only on (particle, origin): beam.cpp:139 / :145 and the begin voxel
__device__ inline BeamOrigin make_beam_origin(const float* __restrict__ pose7, long long p, const float4 og, const DdaGrid& g)
{
  const float* ps = pose7 + 7 * p;
  const Vec3f pos = { ps[0], ps[1], ps[2] };
  const Quat raw = { ps[3], ps[4], ps[5], ps[6] };
  const Quat rot = qnormalized(raw);
  const Vec3f begin = vadd(pos, qrot(raw, Vec3f{ og.x, og.y, og.z }));
  BeamOrigin r;
  r.rot = make_float4(rot.x, rot.y, rot.z, rot.w);
  const bool within = point_within_map(g, begin);
  r.begin = make_float4(begin.x, begin.y, begin.z, within ? 1.0f : 0.0f);
  int bx = 0, by = 0, bz = 0;
  if (within)
    to_index(g, begin, bx, by, bz);
  r.voxel = make_int4(bx, by, bz, 0);
  r.pos = make_float4(pos.x, pos.y, pos.z, 0.f);
  return r;
}

// Also zeroes the particle's penalty counter (the memset the beam kernel would otherwise wait for).
__global__ void beam_origin_kernel(const float* __restrict__ pose7, int n_p, const float4* __restrict__ origins, int n_o,
                                   DdaGrid g, BeamOrigin* __restrict__ out, unsigned* __restrict__ penalty_count)
{
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= static_cast<long long>(n_p) * n_o)
    return;
  const long long p = t / n_o;
  const int o = static_cast<int>(t - p * n_o);
  if (o == 0)
    penalty_count[p] = 0u;
  out[t] = make_beam_origin(pose7, p, origins[o], g);
}

// One lane per (particle, beam point).  scan_beam.w = origin index (PointXYZIL::label of the scan point).
// prepared != nullptr: the per-(particle, origin) table of beam_origin_kernel ([n_p][n_o]); nullptr: every ray computes
// its own begin point (small launches, where one more kernel launch costs more than it saves). (Round 6 measured a third form —
// every work-group prepares the entries of ITS particles in LDS, no launch, no trip through memory — level at C3 and 4.6 % slower
// at the C5 shard: one or two lanes computing while four wavefronts wait cost more than a 64-byte load: profiles/r06q_beam_tail_ab.txt.)
// (the body as a device function of the work-group's index: beam_kernel is it with blockIdx.x; lik_beam_kernel, update_kernels.h,
// interleaves it with the tiled likelihood kernel's work-groups in one launch)
// (BLOCK: rays per work-group = its thread count; 256 for beam_kernel, 64 where the launch it rides in has 64-thread work-groups)
template <bool STATS, bool OVERLAY = true, int BLOCK = 256>
__device__ __forceinline__ void beam_body(const long long block_index, const float* __restrict__ pose7,
                                          const float4* __restrict__ scan, int n_b, const float4* __restrict__ origins,
                                          long long n_rays, DdaGrid g, BeamParams bp,
                                          unsigned* __restrict__ penalty_count, RayStats* __restrict__ stats,
                                          const BeamOrigin* __restrict__ prepared, int n_o)
{
  const long long ray0 = block_index * BLOCK;
  const long long ray = ray0 + threadIdx.x;
  unsigned st_steps = 0, st_occ = 0, st_tested = 0;
  // Penalised rays are counted per particle. Thousands of rays of one particle bumping one global counter serialise in
  // L2 (same cache line), so a work-group first counts in LDS for the (at most two, when N_b >= 256) particles its 256
  // rays start with — one ballot + popcount per wavefront — and issues one global atomic per particle at the end.
  __shared__ unsigned block_count[2];
  if (threadIdx.x < 2)
    block_count[threadIdx.x] = 0u;
  __syncthreads();
  const long long p0 = ray0 / n_b;
  bool penalised = false;
  long long p = p0;
  if (ray < n_rays)
  {
    p = ray / n_b;
    const int i = static_cast<int>(ray - p * n_b);
    const float4 v = scan[i];
    int hit, status;
    if (prepared)
    {
      const BeamOrigin bo = prepared[p * n_o + __float_as_uint(v.w)];
      const Vec3f end = vadd(qrot(Quat{ bo.rot.x, bo.rot.y, bo.rot.z, bo.rot.w }, Vec3f{ v.x, v.y, v.z }),
                             Vec3f{ bo.pos.x, bo.pos.y, bo.pos.z });  // beam.cpp:139 (transform)
      status = 2;
      if (bo.begin.w != 0.0f)
        status = cast_ray_from<STATS, false, OVERLAY>(g, bp, Vec3f{ bo.begin.x, bo.begin.y, bo.begin.z }, bo.voxel.x, bo.voxel.y, bo.voxel.z,
                                      end, &hit, st_steps, st_occ, st_tested);
    }
    else
    {
      const float* ps = pose7 + 7 * p;
      const Vec3f pos = { ps[0], ps[1], ps[2] };
      const Quat raw = { ps[3], ps[4], ps[5], ps[6] };
      const Quat rot = qnormalized(raw);
      const Vec3f end = vadd(qrot(rot, Vec3f{ v.x, v.y, v.z }), pos);  // beam.cpp:139 (transform)
      const float4 og = origins[__float_as_uint(v.w)];
      const Vec3f begin = vadd(pos, qrot(raw, Vec3f{ og.x, og.y, og.z }));  // beam.cpp:145: s.pos_ + s.rot_ * origin
      status = cast_ray<STATS, false, OVERLAY>(g, bp, begin, end, &hit, st_steps, st_occ, st_tested);
    }
    penalised = (status == 0) || (!bp.short_only && (status == 2));  // beam.cpp:146
  }
  const int rel = static_cast<int>(p - p0);  // >= 0; small N_b puts many particles in one work-group
#pragma unroll
  for (int s = 0; s < 2; ++s)
  {
    const unsigned long long m = __ballot(penalised && rel == s);
    if (m != 0ull && (threadIdx.x & 63) == 0)
      atomicAdd(&block_count[s], static_cast<unsigned>(__popcll(m)));
  }
  if (penalised && rel >= 2)
    atomicAdd(&penalty_count[p], 1u);
  __syncthreads();
  if (threadIdx.x < 2 && block_count[threadIdx.x] != 0u)
    atomicAdd(&penalty_count[p0 + threadIdx.x], block_count[threadIdx.x]);
  if (STATS)
  {
    atomicAdd(&stats->steps, static_cast<unsigned long long>(st_steps));
    atomicAdd(&stats->occupied, static_cast<unsigned long long>(st_occ));
    atomicAdd(&stats->tested, static_cast<unsigned long long>(st_tested));
  }
}

template <bool STATS, bool OVERLAY = true>
__global__ __launch_bounds__(256) void beam_kernel(const float* __restrict__ pose7, const float4* __restrict__ scan,
                                                   int n_b, const float4* __restrict__ origins, long long n_rays,
                                                   DdaGrid g, BeamParams bp, unsigned* __restrict__ penalty_count,
                                                   RayStats* __restrict__ stats,
                                                   const BeamOrigin* __restrict__ prepared, int n_o)
{
  beam_body<STATS, OVERLAY>(static_cast<long long>(blockIdx.x), pose7, scan, n_b, origins, n_rays, g, bp, penalty_count, stats, prepared,
                            n_o);
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

}  // namespace mcl3dl
