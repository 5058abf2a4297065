// map_compiler.h — device-side map compiler for the likelihood-field index ("candidate voxels").
//
// What it builds (once per map stamp / search radius, entirely on the GPU):
//   a sparse two-level voxel grid over the dist_weight-rescaled map — dense table of 8x8x8-voxel bricks, bricks allocated
//   only within reach of map points — in which every voxel V (edge e = r/2) stores the EXACT candidate set
//        C(V) = { map points that can be the nearest neighbour, at distance < r, of at least one query inside V }
//   (conservatively: a superset proven below), as a contiguous run of float4 {x,y,z,index}.
//   A query then costs: 1 brick-table load + 2 run-delimiter loads + |C(V)| contiguous point loads (|C| ~ 4 on a 0.1 m
//   sampled wall) instead of 18 delimiter loads + ~25 points of the 27-cell scan — ~6x fewer cache-line accesses, which is
//   what bounds likelihood_kernel.
//
// Exactness argument (V+ = V grown by 1e-3*e on every side to absorb the float rounding of the query's voxel index):
//   (1) p is the NN of q in V+ at distance < r  =>  dmin(p,V+) <= |q-p| < r.
//   (2) for any q in V+, |q - NN(q)| <= |q - p'| <= dmax(p',V+) for every p'  =>  dmin(NN(q),V+) <= D(V) := min_p' dmax(p',V+).
//   (3) if some p' is closer than p by more than a margin m at EVERY q of V+ (the difference of squared distances is
//       linear in q, so its minimum over the box is attained at a corner and has a closed form) then p is never the NN.
//   All three tests are evaluated in fp64 with relative slack 1e-5 / absolute margin m = 1e-5*r^2, orders of magnitude above
//   the ~2e-7 relative error of the kernel's float d^2, so the float-argmin point of the query kernel is never dropped and
//   min d^2 over C(V) is bit-identical to min d^2 over the whole map (checked by every parity test).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcl3dl
{
struct CompileParams
{
  float ox, oy, oz, inv_ex, inv_ey, inv_ez;
  double ex, ey, ez;  // voxel edges (rescaled coordinates). Every step of the candidate proof is per axis (boxes V, V+), so a
                      // voxel may be a box: dist_weight stretches the map along an axis and thins its points out there in the
                      // metric — an edge stretched likewise keeps the candidates per voxel and divides the voxel count
  double grow;    // V+ = V grown by this on every side
  double r2_hi;   // (r*(1+1e-5))^2
  double margin;  // domination margin m
  int rx, ry, rz;  // voxels to visit around a point's own voxel, per axis
  int nvx, nvy, nvz, nbx, nby, nbz;
  int n_points;
};

__device__ inline int3 voxel_of(const CompileParams& c, const float4 p)
{
  // the same float expression the query kernel evaluates
  return make_int3(static_cast<int>(floorf((p.x - c.ox) * c.inv_ex)), static_cast<int>(floorf((p.y - c.oy) * c.inv_ey)),
                   static_cast<int>(floorf((p.z - c.oz) * c.inv_ez)));
}

__device__ inline long long brick_index(const CompileParams& c, int vx, int vy, int vz)
{
  return (static_cast<long long>(vz >> 3) * c.nby + (vy >> 3)) * c.nbx + (vx >> 3);
}

__device__ inline unsigned local_index(int vx, int vy, int vz)
{
  return static_cast<unsigned>(((vz & 7) << 6) | ((vy & 7) << 3) | (vx & 7));
}

// squared min / max distance from point p to the grown voxel box (fp64)
__device__ inline void box_dist2(const CompileParams& c, const float4 p, int vx, int vy, int vz, double& dmin2,
                                 double& dmax2)
{
  const double pc[3] = { static_cast<double>(p.x), static_cast<double>(p.y), static_cast<double>(p.z) };
  const double o[3] = { static_cast<double>(c.ox), static_cast<double>(c.oy), static_cast<double>(c.oz) };
  const int v[3] = { vx, vy, vz };
  const double ed[3] = { c.ex, c.ey, c.ez };
  dmin2 = 0.0;
  dmax2 = 0.0;
#pragma unroll
  for (int a = 0; a < 3; ++a)
  {
    const double lo = o[a] + v[a] * ed[a] - c.grow, hi = o[a] + (v[a] + 1) * ed[a] + c.grow;
    const double below = lo - pc[a], above = pc[a] - hi;
    const double out = below > 0 ? below : (above > 0 ? above : 0.0);
    dmin2 += out * out;
    const double f1 = fabs(pc[a] - lo), f2 = fabs(pc[a] - hi);
    const double far = f1 > f2 ? f1 : f2;
    dmax2 += far * far;
  }
}

// decode thread -> (point, voxel offset); returns false if outside the grid
__device__ inline bool visit(const CompileParams& c, const float4* __restrict__ pts, long long t, int& pi, int& vx,
                             int& vy, int& vz)
{
  const int sx = 2 * c.rx + 1, sy = 2 * c.ry + 1, sz = 2 * c.rz + 1;
  const int per = sx * sy * sz;
  pi = static_cast<int>(t / per);
  if (pi >= c.n_points)
    return false;
  const int o = static_cast<int>(t - static_cast<long long>(pi) * per);
  const int3 pv = voxel_of(c, pts[pi]);
  vx = pv.x + (o % sx) - c.rx;
  vy = pv.y + ((o / sx) % sy) - c.ry;
  vz = pv.z + (o / (sx * sy)) - c.rz;
  return vx >= 0 && vy >= 0 && vz >= 0 && vx < c.nvx && vy < c.nvy && vz < c.nvz;
}

// K1: mark every brick that holds a voxel within reach of a point
__global__ void mc_mark_bricks(CompileParams c, const float4* __restrict__ pts, int* __restrict__ brick_flag)
{
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= c.n_points)
    return;
  const int3 v = voxel_of(c, pts[pi]);
  const int x0 = max(v.x - c.rx, 0) >> 3, x1 = min(v.x + c.rx, c.nvx - 1) >> 3;
  const int y0 = max(v.y - c.ry, 0) >> 3, y1 = min(v.y + c.ry, c.nvy - 1) >> 3;
  const int z0 = max(v.z - c.rz, 0) >> 3, z1 = min(v.z + c.rz, c.nvz - 1) >> 3;
  for (int z = z0; z <= z1; ++z)
    for (int y = y0; y <= y1; ++y)
      for (int x = x0; x <= x1; ++x)
        brick_flag[(static_cast<long long>(z) * c.nby + y) * c.nbx + x] = 1;
}

// brick_flag (0/1) + its exclusive scan -> brick id or -1
__global__ void mc_brick_ids(const int* __restrict__ flag, const uint32_t* __restrict__ scanned, int* __restrict__ table,
                             long long n)
{
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n)
    table[i] = flag[i] ? static_cast<int>(scanned[i]) : -1;
}

__global__ void mc_fill_u32(uint32_t* a, uint32_t v, long long n)
{
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n)
    a[i] = v;
}

// K3: D^2(V) = min over points within reach of dmax^2(p, V+), kept as float bits rounded UP (positive floats order like uints)
__global__ void mc_scatter_dmax(CompileParams c, const float4* __restrict__ pts, const int* __restrict__ table,
                                uint32_t* __restrict__ d2bits, long long n_threads)
{
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n_threads)
    return;
  int pi, vx, vy, vz;
  if (!visit(c, pts, t, pi, vx, vy, vz))
    return;
  double dmin2, dmax2;
  box_dist2(c, pts[pi], vx, vy, vz, dmin2, dmax2);
  if (dmin2 > c.r2_hi)
    return;
  const int b = table[brick_index(c, vx, vy, vz)];
  if (b < 0)
    return;
  float up = static_cast<float>(dmax2);
  if (static_cast<double>(up) < dmax2)
    up = __uint_as_float(__float_as_uint(up) + 1u);
  atomicMin(&d2bits[static_cast<size_t>(b) * 512 + local_index(vx, vy, vz)], __float_as_uint(up));
}

__device__ inline bool prelim_test(const CompileParams& c, const float4 p, int vx, int vy, int vz, uint32_t dbits)
{
  double dmin2, dmax2;
  box_dist2(c, p, vx, vy, vz, dmin2, dmax2);
  const double d2 = static_cast<double>(__uint_as_float(dbits)) * (1.0 + 2e-5) + 1e-12;
  return dmin2 <= c.r2_hi && dmin2 <= d2;
}

// K4 / K5: preliminary candidates = points with dmin <= min(r, D) (slack included); count, then fill
template <bool FILL>
__global__ void mc_prelim(CompileParams c, const float4* __restrict__ pts, const int* __restrict__ table,
                          const uint32_t* __restrict__ d2bits, uint32_t* __restrict__ count,
                          const uint32_t* __restrict__ pstart, uint32_t* __restrict__ prelim, long long n_threads)
{
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n_threads)
    return;
  int pi, vx, vy, vz;
  if (!visit(c, pts, t, pi, vx, vy, vz))
    return;
  const int b = table[brick_index(c, vx, vy, vz)];
  if (b < 0)
    return;
  const size_t v = static_cast<size_t>(b) * 512 + local_index(vx, vy, vz);
  if (!prelim_test(c, pts[pi], vx, vy, vz, d2bits[v]))
    return;
  const uint32_t slot = atomicAdd(&count[v], 1u);
  if (FILL)
    prelim[pstart[v] + slot] = static_cast<uint32_t>(pi);
}

// K6 (mc_prune_boxed below): per voxel, drop every candidate that another candidate beats by more than the margin
// everywhere in V+; the survivors are compacted to the front of the voxel's preliminary run in ascending point order.
// brick id -> brick coordinate (inverse of the table), so a voxel index decodes to its integer coordinates
__global__ void mc_brick_coords(const int* __restrict__ table, int nbx, int nby, long long n_table,
                                int* __restrict__ brick_xyz)
{
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_table)
    return;
  const int b = table[i];
  if (b < 0)
    return;
  const long long plane = static_cast<long long>(nbx) * nby;
  brick_xyz[3 * b + 0] = static_cast<int>(i % nbx);
  brick_xyz[3 * b + 1] = static_cast<int>((i / nbx) % nby);
  brick_xyz[3 * b + 2] = static_cast<int>(i / plane);
}

// one voxel, one thread
__device__ inline void prune_voxel_serial(const CompileParams& c, const float4* __restrict__ pts,
                                          const int* __restrict__ brick_xyz, const uint32_t* __restrict__ pstart,
                                          uint32_t* __restrict__ prelim, uint32_t* __restrict__ kept_count, long long v)
{
  const uint32_t s = pstart[v], e = pstart[v + 1];
  if (s == e)
  {
    kept_count[v] = 0;
    return;
  }
  const int b = static_cast<int>(v >> 9);
  const int l = static_cast<int>(v & 511);
  const int vc[3] = { brick_xyz[3 * b + 0] * 8 + (l & 7), brick_xyz[3 * b + 1] * 8 + ((l >> 3) & 7),
                      brick_xyz[3 * b + 2] * 8 + (l >> 6) };
  const double o[3] = { static_cast<double>(c.ox), static_cast<double>(c.oy), static_cast<double>(c.oz) };
  double ctr[3];
  const double ed[3] = { c.ex, c.ey, c.ez };
  const double hx = 0.5 * c.ex + c.grow, hy = 0.5 * c.ey + c.grow, hz = 0.5 * c.ez + c.grow;
  for (int a = 0; a < 3; ++a)
    ctr[a] = o[a] + (vc[a] + 0.5) * ed[a];
  // pass 1: mark dominated candidates (bit 31 of the stored id). Every pair is compared while the list is short (the
  // normal case: ~10 entries); a long list (huge match_dist_min relative to the map spacing) is compared against its
  // PRUNE_K entries nearest to the voxel centre only — the likely dominators — which keeps the work linear in the list
  // length. Pruning less is always safe: the result is a superset of the exact candidate set either way.
  constexpr int PRUNE_K = 32;
  const uint32_t k_all = e - s;
  uint32_t near_idx[PRUNE_K];
  double near_pp[PRUNE_K];
  uint32_t n_near = 0;
  if (k_all > PRUNE_K)
  {
    for (uint32_t i = s; i < e; ++i)
    {
      const float4 p = pts[prelim[i] & 0x7fffffffu];
      const double px = p.x - ctr[0], py = p.y - ctr[1], pz = p.z - ctr[2];
      const double pp = px * px + py * py + pz * pz;
      if (n_near < PRUNE_K || pp < near_pp[n_near - 1])
      {
        uint32_t pos = n_near < PRUNE_K ? n_near++ : PRUNE_K - 1;
        while (pos > 0 && near_pp[pos - 1] > pp)
        {
          near_pp[pos] = near_pp[pos - 1];
          near_idx[pos] = near_idx[pos - 1];
          --pos;
        }
        near_pp[pos] = pp;
        near_idx[pos] = i;
      }
    }
  }
  const uint32_t n_rivals = k_all > PRUNE_K ? n_near : k_all;
  for (uint32_t i = s; i < e; ++i)
  {
    const float4 p = pts[prelim[i] & 0x7fffffffu];
    const double px = p.x - ctr[0], py = p.y - ctr[1], pz = p.z - ctr[2];
    const double pp = px * px + py * py + pz * pz;
    bool dominated = false;
    for (uint32_t jj = 0; jj < n_rivals && !dominated; ++jj)
    {
      const uint32_t j = k_all > PRUNE_K ? near_idx[jj] : s + jj;
      if (j == i)
        continue;
      const float4 q = pts[prelim[j] & 0x7fffffffu];
      const double qx = q.x - ctr[0], qy = q.y - ctr[1], qz = q.z - ctr[2];
      // g(x) = |x-p|^2 - |x-q|^2 = 2 x.(q-p) + |p|^2 - |q|^2 ; its minimum over the box [-half, half]^3
      const double cx = 2.0 * (qx - px), cy = 2.0 * (qy - py), cz = 2.0 * (qz - pz);
      const double gmin = pp - (qx * qx + qy * qy + qz * qz) - (hx * fabs(cx) + hy * fabs(cy) + hz * fabs(cz));
      dominated = gmin > c.margin;
    }
    if (dominated)
      prelim[i] |= 0x80000000u;
  }
  // (Rounds 4-5 carried a pass 1b for crowded voxels — every sub-box of a subdivision of V+ with its own dominator, a sharper
  // proof: it moved the tiled kernel on maps of centroids by <= 1.2 %, profiles/r04ac_quarter_bounds_ab.txt, and went in round 6.)
  // pass 2: compact survivors to the front, ordered by their distance to the voxel box (ties: ascending point id): the
  // candidates a record holds INLINE are then the ones nearest to the voxel — the likely winners — and the nearest of the
  // overflow candidates is the first of them, which is what the record's skip bound is taken from (mc_write_records). The
  // order changes no result (a query takes the minimum over all candidates). Selection sort; runs are short — a run of
  // more than 32 survivors keeps the ascending-id order (its bound is then the minimum over all overflow candidates anyway).
  uint32_t n = 0;
  for (uint32_t i = s; i < e; ++i)
  {
    const uint32_t id = prelim[i];
    if (!(id & 0x80000000u))
    {
      prelim[i] = prelim[s + n];
      prelim[s + n] = id;
      ++n;
    }
  }
  constexpr uint32_t SORT_MAX = 32;
  double key[SORT_MAX];
  if (n <= SORT_MAX)
    for (uint32_t i = 0; i < n; ++i)
    {
      double dmin2, dmax2;
      box_dist2(c, pts[prelim[s + i] & 0x7fffffffu], vc[0], vc[1], vc[2], dmin2, dmax2);
      key[i] = dmin2;
    }
  for (uint32_t i = 0; i + 1 < n; ++i)
  {
    uint32_t m = i;
    for (uint32_t j = i + 1; j < n; ++j)
    {
      const uint32_t a = prelim[s + j] & 0x7fffffffu, bb = prelim[s + m] & 0x7fffffffu;
      const bool before = n <= SORT_MAX ? (key[j] < key[m] || (key[j] == key[m] && a < bb)) : a < bb;
      if (before)
        m = j;
    }
    const uint32_t tmp = prelim[s + i];
    prelim[s + i] = prelim[s + m];
    prelim[s + m] = tmp;
    if (n <= SORT_MAX)
    {
      const double tk = key[i];
      key[i] = key[m];
      key[m] = tk;
    }
  }
  kept_count[v] = n;
}

// longer_than = 0: every voxel; > 0: only the voxels whose run is longer (the rest is mc_prune_coop's)
__global__ void mc_prune_boxed(CompileParams c, const float4* __restrict__ pts, const int* __restrict__ brick_xyz,
                               const uint32_t* __restrict__ pstart, uint32_t* __restrict__ prelim,
                               uint32_t* __restrict__ kept_count, long long n_vox, uint32_t longer_than)
{
  const long long v = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (v < n_vox && (longer_than == 0u || pstart[v + 1] - pstart[v] > longer_than))
    prune_voxel_serial(c, pts, brick_xyz, pstart, prelim, kept_count, v);
}

// The same pruning with L lanes per voxel (runs of up to 32 preliminary candidates — the normal case; longer runs are left to a
// mc_prune_boxed launch behind this one, whose scratch arrays this kernel then does not carry): the candidates sit in LDS, every lane tests its own candidates against all rivals, the
// survivors are ranked by (distance to the voxel box, point id) instead of selection-sorted. Same fp64 expressions pair by
// pair, same decisions, the same unique order — so the same records; what changes is the critical path: one thread per voxel
// walks ~k^2 fp64 tests with its arrays in scratch memory (0.75 - 1 ms for the 172 k voxels of a map update, more than the rest
// of the update together; what the pass costs now: profiles/r04o_map_update_C2_kernel_stats.csv), sixteen lanes share them.
constexpr int PRUNE_COOP_MAX = 32;
constexpr int PRUNE_LONG_MAX = 256;  // runs of 33 .. 256: one wavefront per voxel (mc_prune_long); longer: one thread

template <int L>
__global__ __launch_bounds__(256) void mc_prune_coop(CompileParams c, const float4* __restrict__ pts,
                                                     const int* __restrict__ brick_xyz, const uint32_t* __restrict__ pstart,
                                                     uint32_t* __restrict__ prelim, uint32_t* __restrict__ kept_count,
                                                     long long n_vox, uint32_t* __restrict__ long_list,
                                                     uint32_t* __restrict__ long_count)
{
  constexpr int G = 256 / L;  // voxels per work-group
  __shared__ double s_px[G][PRUNE_COOP_MAX], s_py[G][PRUNE_COOP_MAX], s_pz[G][PRUNE_COOP_MAX], s_pp[G][PRUNE_COOP_MAX];
  __shared__ double s_key[G][PRUNE_COOP_MAX];
  __shared__ uint32_t s_id[G][PRUNE_COOP_MAX];
  const int grp = static_cast<int>(threadIdx.x) / L, g = static_cast<int>(threadIdx.x) % L;
  const long long v = static_cast<long long>(blockIdx.x) * G + grp;
  const bool valid = v < n_vox;
  uint32_t s = 0, e = 0;
  if (valid)
  {
    s = pstart[v];
    e = pstart[v + 1];
  }
  const uint32_t k = e - s;
  const bool coop = valid && k > 0 && k <= static_cast<uint32_t>(PRUNE_COOP_MAX);
  if (valid && g == 0 && k == 0)
    kept_count[v] = 0;
  if (valid && g == 0 && k > static_cast<uint32_t>(PRUNE_COOP_MAX) && k <= static_cast<uint32_t>(PRUNE_LONG_MAX))
    long_list[atomicAdd(long_count, 1u)] = static_cast<uint32_t>(v);  // mc_prune_long's work (any order)
  int vc[3] = { 0, 0, 0 };
  double ctr[3] = { 0, 0, 0 };
  const double ed[3] = { c.ex, c.ey, c.ez };
  const double hx = 0.5 * c.ex + c.grow, hy = 0.5 * c.ey + c.grow, hz = 0.5 * c.ez + c.grow;
  if (coop)
  {
    const int b = static_cast<int>(v >> 9);
    const int l = static_cast<int>(v & 511);
    vc[0] = brick_xyz[3 * b + 0] * 8 + (l & 7);
    vc[1] = brick_xyz[3 * b + 1] * 8 + ((l >> 3) & 7);
    vc[2] = brick_xyz[3 * b + 2] * 8 + (l >> 6);
    const double o[3] = { static_cast<double>(c.ox), static_cast<double>(c.oy), static_cast<double>(c.oz) };
    for (int a = 0; a < 3; ++a)
      ctr[a] = o[a] + (vc[a] + 0.5) * ed[a];
    for (uint32_t i = g; i < k; i += L)
    {
      const uint32_t id = prelim[s + i] & 0x7fffffffu;
      const float4 p = pts[id];
      const double px = p.x - ctr[0], py = p.y - ctr[1], pz = p.z - ctr[2];
      s_id[grp][i] = id;
      s_px[grp][i] = px;
      s_py[grp][i] = py;
      s_pz[grp][i] = pz;
      s_pp[grp][i] = px * px + py * py + pz * pz;
      double dmin2, dmax2;
      box_dist2(c, p, vc[0], vc[1], vc[2], dmin2, dmax2);
      s_key[grp][i] = dmin2;
    }
  }
  __syncthreads();
  const auto group_or = [](uint32_t m)
  {
#pragma unroll
    for (int d = 1; d < L; d <<= 1)
      m |= __shfl_xor(m, d, L);
    return m;
  };
  // pass 1: dominated by ONE rival everywhere in V+ (any entry of the run is a rival)
  uint32_t dom = 0u;
  if (coop)
    for (uint32_t i = g; i < k; i += L)
    {
      const double px = s_px[grp][i], py = s_py[grp][i], pz = s_pz[grp][i], pp = s_pp[grp][i];
      bool dominated = false;
      for (uint32_t j = 0; j < k && !dominated; ++j)
      {
        if (j == i)
          continue;
        const double qx = s_px[grp][j], qy = s_py[grp][j], qz = s_pz[grp][j];
        const double cx = 2.0 * (qx - px), cy = 2.0 * (qy - py), cz = 2.0 * (qz - pz);
        const double gmin = pp - (qx * qx + qy * qy + qz * qz) - (hx * fabs(cx) + hy * fabs(cy) + hz * fabs(cz));
        dominated = gmin > c.margin;
      }
      dom |= dominated ? 1u << i : 0u;
    }
  dom = group_or(dom);
  // pass 2: the survivors in the order of (distance to the voxel box, point id); the dropped entries behind them, flagged
  if (coop)
  {
    const uint32_t n = k - static_cast<uint32_t>(__popc(dom));
    for (uint32_t i = g; i < k; i += L)
    {
      const uint32_t id = s_id[grp][i];
      uint32_t rank = 0;
      if (dom & (1u << i))
        rank = n + static_cast<uint32_t>(__popc(dom & ((1u << i) - 1u)));
      else
      {
        const double key = s_key[grp][i];
        for (uint32_t j = 0; j < k; ++j)
        {
          if (dom & (1u << j))
            continue;
          const double kj = s_key[grp][j];
          rank += (kj < key || (kj == key && s_id[grp][j] < id)) ? 1u : 0u;
        }
      }
      prelim[s + rank] = (dom & (1u << i)) ? (id | 0x80000000u) : id;
    }
    if (g == 0)
      kept_count[v] = n;
  }
}

// Runs of PRUNE_COOP_MAX + 1 .. PRUNE_LONG_MAX preliminary candidates (voxels between two close surfaces, raw maps): one
// wavefront per voxel, taken from the list mc_prune_coop wrote. prune_voxel_serial's long-run rule, stated for 64 lanes: the
// rivals are the PRUNE_K entries nearest to the voxel centre — the 32 smallest by (squared distance, position in the run),
// which is what the serial insertion keeps —, every entry is tested against them, no sub-box refinement, survivors ordered by
// (distance to the voxel box, id) when at most 32 survive and by id otherwise.
__global__ __launch_bounds__(256) void mc_prune_long(CompileParams c, const float4* __restrict__ pts,
                                                     const int* __restrict__ brick_xyz, const uint32_t* __restrict__ pstart,
                                                     uint32_t* __restrict__ prelim, uint32_t* __restrict__ kept_count,
                                                     const uint32_t* __restrict__ long_list,
                                                     const uint32_t* __restrict__ long_count)
{
  constexpr int L = 64, G = 4, K = 32, M = PRUNE_LONG_MAX;
  __shared__ double s_px[G][M], s_py[G][M], s_pz[G][M], s_pp[G][M], s_key[G][M];
  __shared__ uint32_t s_id[G][M], s_near[G][K], s_alive[G][M];
  const int grp = static_cast<int>(threadIdx.x) / L, g = static_cast<int>(threadIdx.x) % L;
  const uint32_t n_work = *long_count;
  // (every wavefront of a work-group makes the same number of trips: the barriers below are work-group barriers)
  for (uint32_t w0 = blockIdx.x * G; w0 < n_work; w0 += gridDim.x * G)
  {
    const uint32_t w = w0 + static_cast<uint32_t>(grp);
    const bool valid = w < n_work;
    const long long v = valid ? static_cast<long long>(long_list[w]) : 0;
    const uint32_t s = valid ? pstart[v] : 0u, e = valid ? pstart[v + 1] : 0u;
    const uint32_t k = e - s;
    int vc[3] = { 0, 0, 0 };
    double ctr[3] = { 0, 0, 0 };
    const double ed[3] = { c.ex, c.ey, c.ez };
    const double hx = 0.5 * c.ex + c.grow, hy = 0.5 * c.ey + c.grow, hz = 0.5 * c.ez + c.grow;
    if (valid)
    {
      const int b = static_cast<int>(v >> 9);
      const int l = static_cast<int>(v & 511);
      vc[0] = brick_xyz[3 * b + 0] * 8 + (l & 7);
      vc[1] = brick_xyz[3 * b + 1] * 8 + ((l >> 3) & 7);
      vc[2] = brick_xyz[3 * b + 2] * 8 + (l >> 6);
      const double o[3] = { static_cast<double>(c.ox), static_cast<double>(c.oy), static_cast<double>(c.oz) };
      for (int a = 0; a < 3; ++a)
        ctr[a] = o[a] + (vc[a] + 0.5) * ed[a];
      for (uint32_t i = g; i < k; i += L)
      {
        const uint32_t id = prelim[s + i] & 0x7fffffffu;
        const float4 p = pts[id];
        const double px = p.x - ctr[0], py = p.y - ctr[1], pz = p.z - ctr[2];
        s_id[grp][i] = id;
        s_px[grp][i] = px;
        s_py[grp][i] = py;
        s_pz[grp][i] = pz;
        s_pp[grp][i] = px * px + py * py + pz * pz;
        double dmin2, dmax2;
        box_dist2(c, p, vc[0], vc[1], vc[2], dmin2, dmax2);
        s_key[grp][i] = dmin2;
      }
    }
    __syncthreads();
    // the K rivals: rank by (pp, position)
    if (valid)
      for (uint32_t i = g; i < k; i += L)
      {
        const double pp = s_pp[grp][i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < k; ++j)
        {
          const double pj = s_pp[grp][j];
          rank += (pj < pp || (pj == pp && j < i)) ? 1u : 0u;
        }
        if (rank < static_cast<uint32_t>(K))
          s_near[grp][rank] = i;
      }
    __syncthreads();
    if (valid)
      for (uint32_t i = g; i < k; i += L)
      {
        const double px = s_px[grp][i], py = s_py[grp][i], pz = s_pz[grp][i], pp = s_pp[grp][i];
        bool dominated = false;
        for (int jj = 0; jj < K && !dominated; ++jj)
        {
          const uint32_t j = s_near[grp][jj];
          if (j == i)
            continue;
          const double qx = s_px[grp][j], qy = s_py[grp][j], qz = s_pz[grp][j];
          const double cx = 2.0 * (qx - px), cy = 2.0 * (qy - py), cz = 2.0 * (qz - pz);
          const double gmin = pp - (qx * qx + qy * qy + qz * qz) - (hx * fabs(cx) + hy * fabs(cy) + hz * fabs(cz));
          dominated = gmin > c.margin;
        }
        s_alive[grp][i] = dominated ? 0u : 1u;
      }
    __syncthreads();
    if (valid)
    {
      uint32_t n = 0;
      for (uint32_t j = 0; j < k; ++j)
        n += s_alive[grp][j];
      const bool by_key = n <= 32u;
      for (uint32_t i = g; i < k; i += L)
      {
        const uint32_t id = s_id[grp][i];
        uint32_t rank = 0;
        if (s_alive[grp][i])
        {
          const double key = s_key[grp][i];
          for (uint32_t j = 0; j < k; ++j)
          {
            if (!s_alive[grp][j])
              continue;
            const double kj = s_key[grp][j];
            const uint32_t idj = s_id[grp][j];
            rank += (by_key ? (kj < key || (kj == key && idj < id)) : idj < id) ? 1u : 0u;
          }
        }
        else
        {
          rank = n;
          for (uint32_t j = 0; j < i; ++j)
            rank += s_alive[grp][j] ? 0u : 1u;
        }
        prelim[s + rank] = s_alive[grp][i] ? id : (id | 0x80000000u);
      }
      if (g == 0)
        kept_count[v] = n;
    }
    __syncthreads();  // the next trip overwrites the staged run
  }
}


// ---- "fat" voxel records (query MODE 2) -------------------------------------------------------------------------
// One 64-byte record per voxel of every allocated brick, so a query is ONE cache line after the brick table. Four 16-byte
// parts, one per lane of a quad in the tiled kernel's cooperative fetch:
//   part j   : { candidate j: x, y, z ; w }
//   overflow : the same four-part layout       contiguous, candidates 4, 5, ... four per record (unused slots: sentinel)
// Unused candidate slots hold REC_SENTINEL coordinates. The w words, two forms (RecGrid::packed):
//   packed   : w of parts 0..3 all = (candidate count << 26) | first overflow record — the lane of a quad that fetched part j
//              of ITS OWN record (the cooperative fetch reads part j of the records of lanes 0..3) has count and overflow
//              reference in the word it loaded, no cross-lane traffic. Needs every count <= 63 and fewer than 2^26
//              overflow records; the compiler falls back to the plain form otherwise.
//   plain    : w of part 0 = candidate count, w of part 1 = first overflow record (count > cap)
//   bounded  : like packed, with a skip bound between count and reference — w = (overflow records << 28) | (bound << 22) |
//              first overflow record; the count field holds the NUMBER OF OVERFLOW RECORDS ceil((count - 4) / 4), all a query
//              needs (unused slots hold the sentinel), so 4 bits cover 64 candidates. bound = floor(63 x dmin / r) (rounded down with margin), dmin = the distance from the voxel box V+ to
//              the NEAREST of the voxel's overflow candidates: no query inside the voxel is closer than dmin to any of them, so
//              a lane whose best inline d2 is <= (bound x r / 63)^2 skips its overflow records — the minimum cannot change
//              (exactness: RecGrid::bound_step). Needs every count <= 63 and fewer than 2^22 overflow records; 64-byte records.
constexpr float REC_SENTINEL = 1.0e18f;
constexpr uint32_t REC_EXT_BITS = 26u, REC_EXT_MASK = (1u << REC_EXT_BITS) - 1u, REC_COUNT_MAX = 63u;
constexpr uint32_t REC2_EXT_BITS = 22u, REC2_COUNT_SHIFT = 28u, REC2_BOUND_MAX = 63u;

struct RecGrid
{
  const int32_t* brick_table;
  const float4* rec;  // [n_bricks*512][rec_parts]
  const float4* ovf;  // [n_overflow][4]
  float ox, oy, oz, inv_ex, inv_ey, inv_ez;
  int nvx, nvy, nvz, nbx, nby, nbz;
  int mul24_ok;  // nbx * nby and every brick coordinate < 2^24: the table index can use 24-bit multiplies
  int off32_ok;  // the record array is smaller than 4 GB: byte offsets fit 32 bits
  uint32_t rec_bytes32, ovf_bytes32;  // size of rec / ovf in bytes when below 4 GB (buffer loads), else 0
  int rec_parts;                      // 16-byte parts (= inline candidates) per voxel record: 4 (64 bytes) or 8 (128 bytes)
  uint32_t ti_empty;                  // index of the table's extra last entry, always -1 (lanes without a voxel read it)
  int packed;                         // w words in a packed form (above): count field = w >> count_shift, reference = w & ext mask
  uint32_t count_shift, ext_bits;     // 26 / 26 (packed), 28 / 22 (bounded)
  int count_is_records;               // bounded form: the count field is the number of overflow records, not of candidates
  uint32_t over_thr;                  // a packed word above this has overflow records: (4 << 26) | low bits, resp. 2^28 - 1
  // bounded form only (0 = no bound): a lane's overflow records cannot improve on a best inline d2 <= (bound * bound_step)^2.
  // bound_step = float(r / 63); the compiler rounds the stored level down with a relative margin of 2e-4, orders of magnitude
  // above the float rounding of this product and of the kernel's d2 (2e-7), so (bound * bound_step)^2 stays below the d2 the
  // kernel would compute for every overflow candidate at every query inside the voxel: skipping is exact.
  float bound_step;
  // (A dense form — every brick of the grid allocated in table order, brick id == table index, no table load in the kernels —
  // was measured in round 6 and not kept: one load instruction less per evaluation, but 4.9 GB instead of 0.51 GB at C2, +8 % L2
  // requests, +39 % L2 misses, 0.2362 against 0.2314 ms; twice as slow at 32 768 particles: profiles/r06g_tiled_levers_ab.txt.)
};

__host__ __device__ inline uint32_t rec_ext_mask(const RecGrid& g)
{
  return (1u << g.ext_bits) - 1u;
}
// overflow records a voxel with this count field references (cap = inline candidates per record)
__host__ __device__ inline uint32_t rec_overflow_records(uint32_t field, uint32_t cap, int count_is_records)
{
  return count_is_records ? field : (field > cap ? (field - cap + 3u) / 4u : 0u);
}
// squared skip bound of a packed word (0 when the index carries no bounds: never skip)
__device__ inline float rec_bound2(const RecGrid& g, uint32_t w)
{
  const float bd = static_cast<float>((w >> REC2_EXT_BITS) & REC2_BOUND_MAX) * g.bound_step;
  return bd * bd;
}

// Grid-stride: every wavefront keeps its three tallies in scalar registers and a work-group issues three global atomics in
// total (one atomic per wavefront and counter used to serialise 125 000 wavefronts on one cache line: 1.3 ms of a 15 ms build).
__global__ __launch_bounds__(256) void mc_count_overflow(const uint32_t* __restrict__ kept_count, uint32_t* __restrict__ n_ovf,
                                                         long long n_vox, unsigned long long* __restrict__ hist3, uint32_t cap)
{
  __shared__ unsigned long long s_h[5];
  if (threadIdx.x < 5)
    s_h[threadIdx.x] = 0ull;
  __syncthreads();
  unsigned long long h_any = 0, h_4 = 0, h_8 = 0, h_max = 0, h_15 = 0;  // wave-uniform
  // hist3[0] = voxels with at least one candidate, [1] = voxels with more than four, [2] = with more than eight (the index
  // picks its voxel edge and its record size from their ratios: host_map_compilers.h), [3] = with more than REC_COUNT_MAX
  // (any of those: no packed w words), [4] = unused
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long base = static_cast<long long>(blockIdx.x) * blockDim.x; base < n_vox; base += stride)
  {
    const long long v = base + threadIdx.x;
    const uint32_t c = v < n_vox ? kept_count[v] : 0u;
    if (v < n_vox)
      n_ovf[v] = c > cap ? (c - cap + 3) / 4 : 0u;
    h_any += static_cast<unsigned long long>(__popcll(__ballot(c > 0u)));
    h_4 += static_cast<unsigned long long>(__popcll(__ballot(c > 4u)));
    h_8 += static_cast<unsigned long long>(__popcll(__ballot(c > 8u)));
    h_max += static_cast<unsigned long long>(__popcll(__ballot(c > REC_COUNT_MAX)));
  }
  if (hist3 && (threadIdx.x & 63) == 0)
  {
    atomicAdd(&s_h[0], h_any);
    atomicAdd(&s_h[1], h_4);
    atomicAdd(&s_h[2], h_8);
    atomicAdd(&s_h[3], h_max);
    atomicAdd(&s_h[4], h_15);
  }
  __syncthreads();
  if (hist3 && threadIdx.x < 5 && s_h[threadIdx.x])
    atomicAdd(&hist3[threadIdx.x], s_h[threadIdx.x]);
}

// cap = inline candidates per voxel record (4: 64-byte records, 8: 128-byte records); part j = {candidate j: x, y, z; w},
// w words in the packed or the plain form (RecGrid); candidates cap.. go to four-candidate overflow records
// fmt: 0 = plain w words, 1 = packed, 2 = bounded (needs cp, brick_xyz and r for the bound)
__global__ void mc_write_records(const float4* __restrict__ pts, const uint32_t* __restrict__ pstart,
                                 const uint32_t* __restrict__ prelim, const uint32_t* __restrict__ kept_count,
                                 const uint32_t* __restrict__ ovf_start, float* __restrict__ rec,
                                 float* __restrict__ ovf, long long n_vox, uint32_t cap, int fmt, CompileParams cp,
                                 const int* __restrict__ brick_xyz, double r)
{
  const long long v = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (v >= n_vox)
    return;
  const uint32_t c = kept_count[v], src = pstart[v];
  uint32_t packed_w = (c << REC_EXT_BITS) | (c > cap ? ovf_start[v] : 0u);
  if (fmt == 2)
  {
    uint32_t level = 0;
    if (c > cap)
    {
      // distance from V+ to the nearest overflow candidate (the first one when the run is sorted by it; the minimum over
      // all of them costs nothing more and holds for unsorted long runs too), as a level of r / 63, rounded DOWN
      const int b = static_cast<int>(v >> 9), l = static_cast<int>(v & 511);
      const int vx = brick_xyz[3 * b + 0] * 8 + (l & 7), vy = brick_xyz[3 * b + 1] * 8 + ((l >> 3) & 7),
                vz = brick_xyz[3 * b + 2] * 8 + (l >> 6);
      double nearest2 = 1.0e300;
      for (uint32_t k = cap; k < c; ++k)
      {
        double dmin2, dmax2;
        box_dist2(cp, pts[prelim[src + k] & 0x7fffffffu], vx, vy, vz, dmin2, dmax2);
        nearest2 = dmin2 < nearest2 ? dmin2 : nearest2;
      }
      const double lv = floor(static_cast<double>(REC2_BOUND_MAX) * sqrt(nearest2) / r * (1.0 - 2.0e-4));
      level = lv <= 0.0 ? 0u : (lv >= static_cast<double>(REC2_BOUND_MAX) ? REC2_BOUND_MAX : static_cast<uint32_t>(lv));
    }
    const uint32_t n_rec = c > cap ? (c - cap + 3u) / 4u : 0u;
    packed_w = (n_rec << REC2_COUNT_SHIFT) | (level << REC2_EXT_BITS) | (c > cap ? ovf_start[v] : 0u);
  }
  // unused candidate slots hold REC_SENTINEL: a point so far away that its d2 (~3e36, finite) never wins a minimum and
  // never passes the radius test, so a query may take the minimum over all inline slots without looking at the count
  float4* dst = reinterpret_cast<float4*>(rec) + static_cast<size_t>(cap) * v;
  const uint32_t inline_n = c < cap ? c : cap;
  for (uint32_t k = 0; k < cap; ++k)
  {
    float4 o = make_float4(REC_SENTINEL, REC_SENTINEL, REC_SENTINEL, 0.f);
    if (k < inline_n)
    {
      const float4 p = pts[prelim[src + k] & 0x7fffffffu];
      o.x = p.x;
      o.y = p.y;
      o.z = p.z;
    }
    if (fmt != 0)
    {
      if (k < 4)
        o.w = __uint_as_float(packed_w);
    }
    else
    {
      if (k == 0)
        o.w = __uint_as_float(c);
      if (k == 1)
        o.w = __uint_as_float(c > cap ? ovf_start[v] : 0u);
    }
    dst[k] = o;
  }
  if (c > cap)
  {
    float* o = ovf + 16 * static_cast<size_t>(ovf_start[v]);
    for (uint32_t k = cap; k < c; ++k)
    {
      const float4 p = pts[prelim[src + k] & 0x7fffffffu];
      const uint32_t j = k - cap;
      float* slot = o + 4 * j;  // record j / 4, part j % 4
      slot[0] = p.x;
      slot[1] = p.y;
      slot[2] = p.z;
    }
  }
}

// ---- map updates: re-compile only the bricks a changed set of points can reach (host_map_compilers.h:update_cand_grid) ----
// new_flag[i] = 1 for a dirty brick that has no id yet
__global__ void mc_new_brick_flags(const int* __restrict__ dirty, const int* __restrict__ table, uint32_t* __restrict__ new_flag,
                                   long long n_table)
{
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i > n_table)
    return;
  new_flag[i] = (i < n_table && dirty[i] && table[i] < 0) ? 1u : 0u;
}

// assigns the new ids (appended behind the existing bricks) and builds the sub-problem's view: sub_table = dense ids of
// the dirty bricks only, sub_main[sub id] = id in the main record array, sub_bxyz = brick coordinates
__global__ void mc_dirty_tables(const int* __restrict__ dirty, const uint32_t* __restrict__ new_rank,
                                const uint32_t* __restrict__ dirty_rank, uint32_t n_bricks_old, int nbx, int nby,
                                long long n_table, int* __restrict__ table, int* __restrict__ sub_table,
                                int* __restrict__ sub_main, int* __restrict__ sub_bxyz)
{
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_table)
    return;
  if (!dirty[i])
  {
    sub_table[i] = -1;
    return;
  }
  int id = table[i];
  if (id < 0)
  {
    id = static_cast<int>(n_bricks_old + new_rank[i]);
    table[i] = id;
  }
  const int sub = static_cast<int>(dirty_rank[i]);
  sub_table[i] = sub;
  sub_main[sub] = id;
  const long long plane = static_cast<long long>(nbx) * nby;
  sub_bxyz[3 * sub + 0] = static_cast<int>(i % nbx);
  sub_bxyz[3 * sub + 1] = static_cast<int>((i / nbx) % nby);
  sub_bxyz[3 * sub + 2] = static_cast<int>(i / plane);
}

// flag[i] = 1 if point i can reach a voxel of a dirty brick (its reach box touches one); flag[n] = 0 (scan slot)
__global__ void mc_relevant_points(CompileParams c, const float4* __restrict__ pts, const int* __restrict__ dirty,
                                   uint32_t* __restrict__ flag)
{
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi > c.n_points)
    return;
  if (pi == c.n_points)
  {
    flag[pi] = 0;
    return;
  }
  const int3 v = voxel_of(c, pts[pi]);
  const int x0 = max(v.x - c.rx, 0) >> 3, x1 = min(v.x + c.rx, c.nvx - 1) >> 3;
  const int y0 = max(v.y - c.ry, 0) >> 3, y1 = min(v.y + c.ry, c.nvy - 1) >> 3;
  const int z0 = max(v.z - c.rz, 0) >> 3, z1 = min(v.z + c.rz, c.nvz - 1) >> 3;
  uint32_t f = 0;
  for (int z = z0; z <= z1; ++z)
    for (int y = y0; y <= y1; ++y)
      for (int x = x0; x <= x1; ++x)
        f |= dirty[(static_cast<long long>(z) * c.nby + y) * c.nbx + x] ? 1u : 0u;
  flag[pi] = f;
}

// out[pos[i]] = pts[i] for flagged points (pos = exclusive scan of the flags): order kept, so candidate lists come out in
// the same ascending map order a whole-map compile produces
__global__ void mc_compact_points(const float4* __restrict__ pts, const uint32_t* __restrict__ pos, int n,
                                  float4* __restrict__ out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  if (pos[i + 1] != pos[i])
    out[pos[i]] = pts[i];
}

// copies the freshly compiled records of the dirty bricks into the main array; overflow references are rebased onto the
// main overflow array, where the new overflow records were appended at ovf_base. The overflow records the replaced
// voxels of EXISTING bricks referenced become orphans: counted into *orphaned.
__global__ void mc_install_records(const float4* __restrict__ sub_rec, const int* __restrict__ sub_main, uint32_t ovf_base,
                                   uint32_t n_bricks_old, long long n_sub_vox, float4* __restrict__ rec,
                                   unsigned long long* __restrict__ orphaned, uint32_t cap, int packed, uint32_t count_shift,
                                   int count_is_records)
{
  const long long v = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (v >= n_sub_vox)
    return;
  const int sub = static_cast<int>(v >> 9);
  const uint32_t brick = static_cast<uint32_t>(sub_main[sub]);
  const size_t dst = ((static_cast<size_t>(brick) << 9) | static_cast<size_t>(v & 511)) * cap;
  const size_t src = static_cast<size_t>(v) * cap;
  if (brick < n_bricks_old)
  {
    const uint32_t w0 = __float_as_uint(rec[dst].w);
    const uint32_t old_records = rec_overflow_records(packed ? w0 >> count_shift : w0, cap, count_is_records);
    if (old_records)
      atomicAdd(orphaned, static_cast<unsigned long long>(old_records));
  }
  float4 r0 = sub_rec[src];
  float4 r1 = sub_rec[src + 1];
  if (packed)
  {
    // the caller made sure ovf_base + (the sub-compile's overflow records) stays below 2^ext_bits: the sum cannot carry into
    // the bound / the count
    const uint32_t w = __float_as_uint(r0.w);
    const uint32_t moved = rec_overflow_records(w >> count_shift, cap, count_is_records) ? w + ovf_base : w;
    r0.w = __uint_as_float(moved);
    r1.w = __uint_as_float(moved);
    rec[dst + 0] = r0;
    rec[dst + 1] = r1;
    for (uint32_t k = 2; k < cap; ++k)
    {
      float4 rk = sub_rec[src + k];
      if (k < 4)
        rk.w = __uint_as_float(moved);
      rec[dst + k] = rk;
    }
    return;
  }
  if (__float_as_uint(r0.w) > cap)
    r1.w = __uint_as_float(__float_as_uint(r1.w) + ovf_base);
  rec[dst + 0] = r0;
  rec[dst + 1] = r1;
  for (uint32_t k = 2; k < cap; ++k)
    rec[dst + k] = sub_rec[src + k];
}

// ---- reclaiming orphaned overflow records (host_map_compilers.h:compact_overflow) ------------------------------------------
// A map update appends the overflow records of the bricks it re-compiles and orphans the ones those bricks referenced before;
// mapcloud_update replaces the previous update every time (src/mcl_3dl.cpp:141-153), so the orphans pile up at the rate of
// the live update's records. Counting pass: overflow records each voxel references, from its own record's w word.
__global__ void mc_ovf_counts(const float4* __restrict__ rec, long long n_vox, uint32_t cap, int packed, uint32_t count_shift,
                              int count_is_records, uint32_t* __restrict__ n_ovf /*[n_vox + 1]*/)
{
  const long long v = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (v > n_vox)
    return;
  uint32_t n = 0;
  if (v < n_vox)
  {
    const uint32_t w0 = __float_as_uint(rec[static_cast<size_t>(v) * cap].w);
    n = rec_overflow_records(packed ? w0 >> count_shift : w0, cap, count_is_records);
  }
  n_ovf[v] = n;
}

// moves every voxel's overflow records to new_start[v] .. of a fresh array and rewrites the reference in its record
__global__ void mc_ovf_move(float4* __restrict__ rec, long long n_vox, uint32_t cap, int packed, uint32_t count_shift,
                            uint32_t ext_bits, int count_is_records, const uint32_t* __restrict__ new_start,
                            const float4* __restrict__ old_ovf, float4* __restrict__ new_ovf)
{
  const long long v = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (v >= n_vox)
    return;
  float4* r = rec + static_cast<size_t>(v) * cap;
  const uint32_t w0 = __float_as_uint(r[0].w);
  const uint32_t n = rec_overflow_records(packed ? w0 >> count_shift : w0, cap, count_is_records);
  if (n == 0)
    return;
  const uint32_t ext_mask = (1u << ext_bits) - 1u;
  const uint32_t old_ext = packed ? (w0 & ext_mask) : __float_as_uint(r[1].w);
  const uint32_t dst = new_start[v];
  for (uint32_t j = 0; j < 4u * n; ++j)
    new_ovf[4 * static_cast<size_t>(dst) + j] = old_ovf[4 * static_cast<size_t>(old_ext) + j];
  if (packed)
  {
    const float w = __uint_as_float((w0 & ~ext_mask) | dst);  // count (and bound) kept
    for (uint32_t k = 0; k < 4u; ++k)
      r[k].w = w;
  }
  else
    r[1].w = __uint_as_float(dst);
}

// ---- exclusive scan of uint32 (3 levels of 1024-element tiles cover 2^30 elements) --------------------------------
constexpr int SCAN_TILE = 1024;  // 256 threads x 4 elements

__global__ __launch_bounds__(256) void scan_tiles(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                  uint32_t* __restrict__ tile_sums, long long n)
{
  __shared__ uint32_t sh[256];
  const long long base = static_cast<long long>(blockIdx.x) * SCAN_TILE + threadIdx.x * 4;
  uint32_t v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    v[k] = (base + k < n) ? in[base + k] : 0u;
  const uint32_t mine = v[0] + v[1] + v[2] + v[3];
  sh[threadIdx.x] = mine;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1)  // Hillis-Steele inclusive scan of the 256 thread sums
  {
    const uint32_t add = threadIdx.x >= off ? sh[threadIdx.x - off] : 0u;
    __syncthreads();
    sh[threadIdx.x] += add;
    __syncthreads();
  }
  uint32_t run = sh[threadIdx.x] - mine;
#pragma unroll
  for (int k = 0; k < 4; ++k)
  {
    if (base + k < n)
      out[base + k] = run;
    run += v[k];
  }
  if (threadIdx.x == 255 && tile_sums)
    tile_sums[blockIdx.x] = sh[255];
}

__global__ void scan_add_offsets(uint32_t* __restrict__ out, const uint32_t* __restrict__ tile_offsets, long long n)
{
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n)
    out[i] += tile_offsets[i / SCAN_TILE];
}

__global__ void sum_u32_to_u64(const uint32_t* __restrict__ in, long long n, unsigned long long* __restrict__ total)
{
  // one atomic per work-group (one per wavefront made 4096 of them on one address: 50 us for any n — twice per map update)
  __shared__ unsigned long long part[4];
  unsigned long long s = 0;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x)
    s += in[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0)
    part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    const unsigned long long b = part[0] + part[1] + part[2] + part[3];
    if (b)
      atomicAdd(total, b);
  }
}

// work-groups for sum_u32_to_u64 over n values: ~4096 values each, at most 1024
inline unsigned sum_blocks(long long n)
{
  const long long b = (n + 4095) / 4096;
  return static_cast<unsigned>(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}
}  // namespace mcl3dl
