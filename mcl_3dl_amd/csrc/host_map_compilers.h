// host_map_compilers.h — part of the single translation unit mcl3dl_hip.hip: host drivers of the three map structures
// (cell-sorted exact-NN grid, DDA occupancy bricks + voxel index, candidate-voxel records; device side in
// map_compiler.h) and the device prefix scans they share.
#pragma once

namespace
{
int build_lik_grid(mcl3dl_hip_ctx* ctx);  // host_grid_builders.h
int build_dda_grid(mcl3dl_hip_ctx* ctx);
int ensure_map_dev(mcl3dl_hip_ctx* ctx);  // host_grid_builders.h: the map as a device cloud {x, y, z, label}
int cloud_minmax(mcl3dl_hip_ctx* ctx, const float4* pts, long long n, float* host6, unsigned long long* host_cnt);  // host_cloud.h

// ---- map compiler: exact-NN grid -----------------------------------------------------------------------
// Replaces ChunkedKdtree::setInputCloud + pcl::KdTreeFLANN::setInputCloud.  The reference's chunking is a memory
// device (20 m chunks with duplicated margins, chunked_kdtree.h:124-216) whose query result equals the global
// nearest neighbour within the radius whenever radius <= max_search_radius; the grid gives that result directly.
// Host form (option grid_build_host = 1): the sequential counting sort the device builder (host_grid_builders.h) is
// checked against.
int build_lik_grid_host(mcl3dl_hip_ctx* ctx)
{
  const size_t n = ctx->map_xyz.size() / 3;
  const float cell = ctx->match_dist_min * 1.01f;
  if (!(cell > 0.f) || !std::isfinite(cell))
    return ctx->fail(-3, "match_dist_min must be positive and finite");
  const float inv = 1.0f / cell;
  std::vector<float> s(3 * n);
  float mn[3] = { 0, 0, 0 }, mx[3] = { 0, 0, 0 };
  for (size_t i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a)
    {
      // PointRepresentation::vectorize: one float product per coordinate
      const float v = ctx->has_weight ? ctx->map_xyz[3 * i + a] * ctx->weight[a] : ctx->map_xyz[3 * i + a];
      if (!std::isfinite(v))
        return ctx->fail(-3, "map point %zu is not finite", i);
      s[3 * i + a] = v;
      if (i == 0 || v < mn[a])
        mn[a] = v;
      if (i == 0 || v > mx[a])
        mx[a] = v;
    }
  float o[3];
  int dim[3];
  double total = 1;
  for (int a = 0; a < 3; ++a)
  {
    o[a] = mn[a] - 2.0f * cell;
    dim[a] = static_cast<int>(floorf((mx[a] - o[a]) * inv)) + 3;
    total *= dim[a];
  }
  if (total > 3.0e9)
    return ctx->fail(-4, "likelihood grid would need %.3g cells (map extent too large for the dense index)", total);
  const size_t ncell = static_cast<size_t>(dim[0]) * dim[1] * dim[2];
  std::vector<uint32_t> cell_of(n);
  std::vector<uint32_t> start(ncell + 1, 0);
  for (size_t i = 0; i < n; ++i)
  {
    int c[3];
    for (int a = 0; a < 3; ++a)
    {
      c[a] = static_cast<int>(floorf((s[3 * i + a] - o[a]) * inv));  // same expression as the kernel's
      c[a] = std::min(std::max(c[a], 0), dim[a] - 1);
    }
    cell_of[i] = static_cast<uint32_t>((static_cast<size_t>(c[2]) * dim[1] + c[1]) * dim[0] + c[0]);
    ++start[cell_of[i] + 1];
  }
  for (size_t c = 0; c < ncell; ++c)
    start[c + 1] += start[c];
  std::vector<uint32_t> fill(start.begin(), start.end() - 1);
  std::vector<float4> pts(n);
  for (size_t i = 0; i < n; ++i)
  {
    const uint32_t dst = fill[cell_of[i]]++;
    pts[dst] = make_float4(s[3 * i], s[3 * i + 1], s[3 * i + 2], bits_to_float(static_cast<uint32_t>(i)));
  }
  TRY(ensure(ctx, ctx->lik_pts, sizeof(float4) * n));
  TRY(ensure(ctx, ctx->lik_cells, sizeof(uint32_t) * (ncell + 1)));
  TRY(h2d(ctx, ctx->lik_pts.p, pts.data(), sizeof(float4) * n));
  TRY(h2d(ctx, ctx->lik_cells.p, start.data(), sizeof(uint32_t) * (ncell + 1)));
  TRY(sync_stream(ctx));
  ctx->lg.cell_start = ctx->lik_cells.as<uint32_t>();
  ctx->lg.pts = ctx->lik_pts.as<float4>();
  ctx->lg.ox = o[0];
  ctx->lg.oy = o[1];
  ctx->lg.oz = o[2];
  ctx->lg.inv_cell = inv;
  ctx->lg.nx = dim[0];
  ctx->lg.ny = dim[1];
  ctx->lg.nz = dim[2];
  ctx->footprint[0] = sizeof(float4) * n;
  ctx->footprint[1] = sizeof(uint32_t) * (ncell + 1);
  ctx->lik_dirty = false;
  return 0;
}

// ---- map compiler: DDA occupancy -------------------------------------------------------------------------
// RaycastUsingDDA::updatePointCloud / setExists, include/mcl_3dl/raycasts/raycast_using_dda.h:162-190,230-235:
// AABB by getMinMax3D, map_size = (size_t)((max-min)/grid)+1, voxel = trunc((p-min)/grid) (float difference,
// double division), x-fastest array index; per voxel the points stay in insertion (map) order.
// the ray-test constants of a DdaGrid (shared by both builders)
void dda_ray_constants(const mcl3dl_hip_ctx* ctx, DdaGrid& g)
{
  g.grid = static_cast<double>(ctx->dda_grid_size);
  g.ray_angle_half = static_cast<double>(ctx->ray_angle_half);
  // RaycastUsingDDA ctor, raycast_using_dda.h:59: map_grid_size_y appears twice (reference quirk, kept)
  const double gx = ctx->map_grid[0], gy = ctx->map_grid[1];
  g.min_dist_thr_sq = gx * gx + gy * gy + gy * gy;
  g.hit_tolerance_f = static_cast<float>(static_cast<double>(ctx->hit_range));
}

// Host form (option grid_build_host = 1); device form in host_grid_builders.h.
int build_dda_grid_host(mcl3dl_hip_ctx* ctx)
{
  const size_t n = ctx->map_xyz.size() / 3;
  const double grid = static_cast<double>(ctx->dda_grid_size);
  if (!(grid > 0))
    return ctx->fail(-3, "dda_grid_size must be positive");
  float mn[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, mx[3] = { -3.4e38f, -3.4e38f, -3.4e38f };
  for (size_t i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a)
    {
      const float v = ctx->map_xyz[3 * i + a];
      if (v < mn[a])
        mn[a] = v;
      if (v > mx[a])
        mx[a] = v;
    }
  int dim[3];
  double total_d = 1;
  for (int a = 0; a < 3; ++a)
  {
    dim[a] = static_cast<int>(static_cast<size_t>((mx[a] - mn[a]) / grid) + 1);
    total_d *= dim[a];
  }
  if (total_d >= 2147483647.0)  // the reference keeps point_total in an int (raycast_using_dda.h:176)
    return ctx->fail(-4, "DDA grid would need %.3g voxels (>= 2^31)", total_d);
  const size_t total = static_cast<size_t>(total_d);
  std::vector<uint32_t> vox(n);
  std::vector<uint32_t> start(total + 1, 0);
  const int bdim[3] = { (dim[0] + 3) / 4, (dim[1] + 3) / 4, (dim[2] + 3) / 4 };
  std::vector<unsigned long long> bits(static_cast<size_t>(bdim[0]) * bdim[1] * bdim[2], 0ull);
  for (size_t i = 0; i < n; ++i)
  {
    int c[3];
    for (int a = 0; a < 3; ++a)
      c[a] = static_cast<int>(static_cast<double>(ctx->map_xyz[3 * i + a] - mn[a]) / grid);
    const size_t v = static_cast<size_t>(c[0] + c[1] * dim[0] + c[2] * (dim[0] * dim[1]));
    if (v >= total)
      return ctx->fail(-3, "map point %zu falls outside its own DDA grid", i);
    vox[i] = static_cast<uint32_t>(v);
    ++start[v + 1];
    const size_t brick = (static_cast<size_t>(c[2] >> 2) * bdim[1] + (c[1] >> 2)) * bdim[0] + (c[0] >> 2);
    bits[brick] |= 1ull << (((c[2] & 3) << 4) | ((c[1] & 3) << 2) | (c[0] & 3));
  }
  for (size_t v = 0; v < total; ++v)
    start[v + 1] += start[v];
  std::vector<uint32_t> fill(start.begin(), start.end() - 1);
  std::vector<float4> pts(n);
  std::vector<uint32_t> index(n);
  for (size_t i = 0; i < n; ++i)  // ascending i: insertion order preserved inside a voxel
  {
    const uint32_t dst = fill[vox[i]]++;
    pts[dst] = make_float4(ctx->map_xyz[3 * i], ctx->map_xyz[3 * i + 1], ctx->map_xyz[3 * i + 2],
                           bits_to_float(ctx->map_label[i]));
    index[dst] = static_cast<uint32_t>(i);
  }
  TRY(ensure(ctx, ctx->dda_bits, sizeof(unsigned long long) * bits.size()));
  TRY(ensure(ctx, ctx->dda_start, sizeof(uint32_t) * (total + 1)));
  TRY(ensure(ctx, ctx->dda_pts, sizeof(float4) * n));
  TRY(ensure(ctx, ctx->dda_index, sizeof(uint32_t) * n));
  TRY(h2d(ctx, ctx->dda_bits.p, bits.data(), sizeof(unsigned long long) * bits.size()));
  TRY(h2d(ctx, ctx->dda_start.p, start.data(), sizeof(uint32_t) * (total + 1)));
  TRY(h2d(ctx, ctx->dda_pts.p, pts.data(), sizeof(float4) * n));
  TRY(h2d(ctx, ctx->dda_index.p, index.data(), sizeof(uint32_t) * n));
  TRY(sync_stream(ctx));
  DdaGrid& g = ctx->dg;
  g.bricks = ctx->dda_bits.as<unsigned long long>();
  g.bnx = bdim[0];
  g.bny = bdim[1];
  g.bnz = bdim[2];
  g.mul24_ok = (bdim[0] < (1 << 24) && static_cast<long long>(bdim[1]) * bdim[2] < (1ll << 24)) ? 1 : 0;
  g.vox_start = ctx->dda_start.as<uint32_t>();
  g.pts = ctx->dda_pts.as<float4>();
  g.pt_index = ctx->dda_index.as<uint32_t>();
  g.min_x = mn[0];
  g.min_y = mn[1];
  g.min_z = mn[2];
  g.max_x = mx[0];
  g.max_y = mx[1];
  g.max_z = mx[2];
  g.nx = dim[0];
  g.ny = dim[1];
  g.nz = dim[2];
  g.ov_n = 0;  // the host builder puts every point into the one array: a map update then rebuilds
  ctx->dda_overlay_ok = false;
  dda_ray_constants(ctx, g);
  ctx->footprint[2] = sizeof(unsigned long long) * bits.size();
  ctx->footprint[3] = sizeof(uint32_t) * (total + 1);
  ctx->footprint[4] = sizeof(float4) * n + sizeof(uint32_t) * n;
  ctx->dda_dirty = false;
  return 0;
}

// two events that are destroyed on every return path of the builders below
struct EventPairRaii
{
  hipEvent_t a = nullptr, b = nullptr;
  ~EventPairRaii()
  {
    if (a)
      (void)hipEventDestroy(a);
    if (b)
      (void)hipEventDestroy(b);
  }
};

// Scratch memory of the map compilers. Blocks come from — and go back to — a pool in the context (scratch_alloc /
// scratch_release) instead of hipMalloc / hipFree: a map update used two dozen of each (every hipFree a device
// synchronisation), a millisecond and more of a 4 ms mapcloud_update (src/mcl_3dl.cpp:141-153). All work is enqueued on the
// context's ONE stream, so a block handed out again is written only behind the kernels that still read it. After a whole-map
// build the pool is trimmed (scratch_trim): the big temporaries of a 10 M-point compile do not stay resident.
struct TempBuf
{
  void* p = nullptr;
  mcl3dl_hip_ctx* owner = nullptr;
  ~TempBuf()
  {
    if (!p)
      return;
    if (owner)
      scratch_release(owner, p);
    else
      (void)hipFree(p);
  }
};

int scratch_alloc(mcl3dl_hip_ctx* ctx, TempBuf& b, size_t bytes)
{
  if (bytes == 0)
    bytes = 16;
  // best fit among the free blocks that are not wastefully large
  int best = -1;
  for (size_t k = 0; k < ctx->scratch.size(); ++k)
  {
    const mcl3dl_hip_ctx::ScratchBlk& blk = ctx->scratch[k];
    if (!blk.used && blk.cap >= bytes && blk.cap <= 4 * bytes + (1u << 16) &&
        (best < 0 || blk.cap < ctx->scratch[static_cast<size_t>(best)].cap))
      best = static_cast<int>(k);
  }
  if (best >= 0)
  {
    ctx->scratch[static_cast<size_t>(best)].used = true;
    b.p = ctx->scratch[static_cast<size_t>(best)].p;
    b.owner = ctx;
    return 0;
  }
  void* p = nullptr;
  const size_t cap = bytes + bytes / 8;
  if (hipMalloc(&p, cap) != hipSuccess)
  {
    // out of memory with idle blocks parked: give them back and try once more
    (void)hipGetLastError();
    scratch_trim(ctx, 0);
    HIP_TRY(hipMalloc(&p, cap));
  }
  ctx->scratch.push_back({ p, cap, true });
  b.p = p;
  b.owner = ctx;
  return 0;
}

// ---- map compiler: candidate-voxel index (device side in map_compiler.h) ------------------------------------------
int device_exclusive_scan(mcl3dl_hip_ctx* ctx, uint32_t* data, long long n)  // in place
{
  if (n <= 0)
    return 0;
  const long long tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  // the per-tile sums live in pooled scratch memory: no allocation, no synchronisation (whoever gets the block next writes it
  // behind these kernels, on the same stream)
  TempBuf sums_buf;
  uint32_t* sums = nullptr;
  if (tiles > 1)
  {
    TRY(scratch_alloc(ctx, sums_buf, sizeof(uint32_t) * tiles));
    sums = static_cast<uint32_t*>(sums_buf.p);
  }
  hipLaunchKernelGGL(scan_tiles, dim3(static_cast<unsigned>(tiles)), dim3(256), 0, ctx->stream, data, data, sums, n);
  if (tiles > 1)
  {
    TRY(device_exclusive_scan(ctx, sums, tiles));
    hipLaunchKernelGGL(scan_add_offsets, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, ctx->stream, data, sums, n);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// The same scan without allocation or synchronisation: `ws` holds the per-tile sums of every level
// (>= n / 1023 + 4 entries).
int device_exclusive_scan_ws(mcl3dl_hip_ctx* ctx, uint32_t* data, long long n, uint32_t* ws)
{
  if (n <= 0)
    return 0;
  const long long tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  uint32_t* sums = tiles > 1 ? ws : nullptr;
  hipLaunchKernelGGL(scan_tiles, dim3(static_cast<unsigned>(tiles)), dim3(256), 0, ctx->stream, data, data, sums, n);
  if (tiles > 1)
  {
    TRY(device_exclusive_scan_ws(ctx, sums, tiles, ws + tiles));
    hipLaunchKernelGGL(scan_add_offsets, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, ctx->stream, data,
                       sums, n);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// Geometry of the candidate-voxel grid for a point set with the given bounds (rescaled coordinates).
// How much longer than the base edge a voxel is along each axis: 1 without a dist_weight; with one (option "cand_aniso",
// default on) the axis weight relative to the smallest weight, capped — src/mcl_3dl.cpp:1270 stretches the map along an axis
// (the shipped default: z x 5, src/parameters.cpp:108-110), which thins the map's points out along it in the metric the query
// ball lives in, so a voxel stretched the same way holds the candidates an unstretched map's cube would, and there are that
// many fewer voxels (walls at z x 5: a fifth; the +-r shell around a floor: 2-3 layers instead of 4-5).
void cand_axis_stretch(const mcl3dl_hip_ctx* ctx, double stretch[3])
{
  stretch[0] = stretch[1] = stretch[2] = 1.0;
  if (!ctx->cand_aniso_active || !ctx->has_weight)
    return;
  double wmin = 0.0;
  for (int a = 0; a < 3; ++a)
  {
    const double w = std::fabs(static_cast<double>(ctx->weight[a]));
    if (w > 0.0 && (wmin == 0.0 || w < wmin))
      wmin = w;
  }
  if (!(wmin > 0.0))
    return;
  for (int a = 0; a < 3; ++a)
  {
    const double w = std::fabs(static_cast<double>(ctx->weight[a]));
    stretch[a] = std::min(std::max(w / wmin, 1.0), ctx->cand_aniso_max);
  }
}

int cand_geometry(mcl3dl_hip_ctx* ctx, double voxel_ratio, const double stretch[3], const float mn[3], const float mx[3],
                  CompileParams* out, long long* n_table_out)
{
  const double r = static_cast<double>(ctx->match_dist_min);
  float e_f[3];
  for (int a = 0; a < 3; ++a)
  {
    e_f[a] = static_cast<float>(r * voxel_ratio * stretch[a]);
    if (!(e_f[a] > 0.f) || !std::isfinite(e_f[a]))
      return ctx->fail(-3, "bad candidate voxel edge");
  }
  CompileParams cp{};
  cp.ex = static_cast<double>(e_f[0]);
  cp.ey = static_cast<double>(e_f[1]);
  cp.ez = static_cast<double>(e_f[2]);
  cp.inv_ex = 1.0f / e_f[0];
  cp.inv_ey = 1.0f / e_f[1];
  cp.inv_ez = 1.0f / e_f[2];
  const double ed[3] = { cp.ex, cp.ey, cp.ez };
  cp.grow = 1e-3 * std::min(cp.ex, std::min(cp.ey, cp.ez));
  const double r_hi = r * (1.0 + 1e-5);
  cp.r2_hi = r_hi * r_hi;
  cp.margin = 1e-5 * r * r;
  int reach[3];
  for (int a = 0; a < 3; ++a)
    reach[a] = static_cast<int>(std::floor((r_hi + cp.grow) / ed[a])) + 1;
  cp.rx = reach[0];
  cp.ry = reach[1];
  cp.rz = reach[2];
  float o[3];
  int nv[3], nb[3];
  double n_table_d = 1;
  for (int a = 0; a < 3; ++a)
  {
    // Phase: maps that come out of a voxel filter sit on a lattice; with the origin ON that lattice every voxel face
    // coincides with a Voronoi face of the map and each voxel keeps 3 candidates per axis instead of the 2 a generic
    // position needs. Half a voxel of phase puts lattice maps in the generic position; arbitrary maps do not care.
    o[a] = mn[a] - static_cast<float>((reach[a] + 1 + ctx->cand_phase) * ed[a]);
    nv[a] = static_cast<int>(std::floor((static_cast<double>(mx[a]) - o[a]) / ed[a])) + reach[a] + 2;
    nb[a] = (nv[a] + 7) / 8;
    n_table_d *= nb[a];
  }
  if (n_table_d > 2.0e9)
    return ctx->fail(-4, "candidate index would need %.3g bricks in its dense table", n_table_d);
  cp.ox = o[0];
  cp.oy = o[1];
  cp.oz = o[2];
  cp.nvx = nv[0];
  cp.nvy = nv[1];
  cp.nvz = nv[2];
  cp.nbx = nb[0];
  cp.nby = nb[1];
  cp.nbz = nb[2];
  *out = cp;
  *n_table_out = static_cast<long long>(n_table_d);
  return 0;
}

// Rescaled map points (PointRepresentation::vectorize) in map order, w = original index; bounds on request.
int rescaled_points(mcl3dl_hip_ctx* ctx, size_t first, size_t count, std::vector<float4>& sp, float mn[3], float mx[3])
{
  sp.resize(count);
  for (size_t k = 0; k < count; ++k)
  {
    const size_t i = first + k;
    float v[3];
    for (int a = 0; a < 3; ++a)
    {
      v[a] = ctx->has_weight ? ctx->map_xyz[3 * i + a] * ctx->weight[a] : ctx->map_xyz[3 * i + a];
      if (!std::isfinite(v[a]))
        return ctx->fail(-3, "map point %zu is not finite", i);
      if (mn && (k == 0 || v[a] < mn[a]))
        mn[a] = v[a];
      if (mx && (k == 0 || v[a] > mx[a]))
        mx[a] = v[a];
    }
    sp[k] = make_float4(v[0], v[1], v[2], bits_to_float(static_cast<uint32_t>(i)));
  }
  return 0;
}

// size of a device array for a buffer-load descriptor: its bytes when they fit 32 bits, else 0 (64-bit addressing)
uint32_t bytes32(unsigned long long bytes)
{
  return bytes < (1ull << 32) ? static_cast<uint32_t>(bytes) : 0u;
}

// The compiler proper: for the bricks `table` names (ids 0 .. n_bricks - 1, every other entry -1; bxyz = their brick
// coordinates) and the points `pts` (cp.n_points of them, device), the candidate set of every voxel: D^2 scatter ->
// preliminary lists -> domination prune -> records of `cap` 16-byte parts into rec_out[n_bricks * 512 * cap * 4 floats] (device) with their
// overflow records in a fresh allocation (ovf_out, *n_ovf of them). Used for the whole map (build_cand_grid) and for the
// bricks a map update touches (update_cand_grid).
// the fields of RecGrid that say how the records' w words are read (map_compiler.h: plain / packed / bounded)
void set_word_format(RecGrid& g, int fmt, float r)
{
  g.packed = fmt != 0 ? 1 : 0;
  g.count_shift = fmt == 2 ? REC2_COUNT_SHIFT : REC_EXT_BITS;
  g.ext_bits = fmt == 2 ? REC2_EXT_BITS : REC_EXT_BITS;
  g.count_is_records = fmt == 2 ? 1 : 0;
  g.over_thr = fmt == 2 ? (1u << REC2_COUNT_SHIFT) - 1u : ((4u << REC_EXT_BITS) | REC_EXT_MASK);
  g.bound_step = fmt == 2 ? r / static_cast<float>(REC2_BOUND_MAX) : 0.0f;
}

int word_format(const RecGrid& g)
{
  return !g.packed ? 0 : (g.bound_step > 0.0f ? 2 : 1);
}

struct CompileOutput
{
  unsigned long long total = 0, kept = 0;
  // voxels with candidates, with more than four, with more than eight, with more than REC_COUNT_MAX, with more than REC2_COUNT_MAX
  unsigned long long hist3[5] = { 0, 0, 0, 0, 0 };
  uint32_t n_ovf = 0;
  int packed = 0;  // form of the w words written: 0 plain, 1 packed, 2 bounded (map_compiler.h)
  // CSR form (lik_index 1): kept counts per voxel stay in d_count, runs in d_pstart / d_prelim
  TempBuf d_count, d_pstart, d_prelim, d_ovf_data;
};

// want_packed: -1 = the richest form of the w words the counts and ovf_base + the overflow records allow — bounded, packed,
// plain (whole-map build; options cand_packed / cand_bound restrict the choice), 0 = plain, 1 = packed or nothing, 2 = bounded or
// nothing (a map update into an index of that form): returns 1 without writing records when that is impossible.
int compile_bricks(mcl3dl_hip_ctx* ctx, const CompileParams& cp, const float4* pts, const int* table, const int* bxyz,
                   uint32_t n_bricks, bool records, float* rec_out, CompileOutput* out, uint32_t cap = 4,
                   int want_packed = -1, uint32_t ovf_base = 0)
{
  const size_t n = static_cast<size_t>(cp.n_points);
  const long long n_vox = static_cast<long long>(n_bricks) * 512;
  const long long n_threads = static_cast<long long>(n) * (2 * cp.rx + 1) * (2 * cp.ry + 1) * (2 * cp.rz + 1);
  const unsigned blocks_t = static_cast<unsigned>((n_threads + 255) / 256);
  if ((n_threads + 255) / 256 > 0x7fffffffLL)
    return ctx->fail(-4, "candidate index: too many (point, voxel) pairs");
  const unsigned blocks_v = static_cast<unsigned>((n_vox + 1 + 255) / 256);
  TempBuf d_d2, d_total;
  TempBuf &d_count = out->d_count, &d_pstart = out->d_pstart, &d_prelim = out->d_prelim;
  TRY(scratch_alloc(ctx, d_d2, sizeof(uint32_t) * n_vox));
  TRY(scratch_alloc(ctx, d_count, sizeof(uint32_t) * (n_vox + 1)));
  TRY(scratch_alloc(ctx, d_pstart, sizeof(uint32_t) * (n_vox + 1)));
  TRY(scratch_alloc(ctx, d_total, sizeof(unsigned long long)));
  hipLaunchKernelGGL(mc_fill_u32, dim3(blocks_v), dim3(256), 0, ctx->stream, static_cast<uint32_t*>(d_d2.p), 0x7f800000u,
                     n_vox);
  if (n)
    hipLaunchKernelGGL(mc_scatter_dmax, dim3(blocks_t), dim3(256), 0, ctx->stream, cp, pts, table,
                       static_cast<uint32_t*>(d_d2.p), n_threads);
  HIP_TRY(hipMemsetAsync(d_count.p, 0, sizeof(uint32_t) * (n_vox + 1), ctx->stream));
  if (n)
    hipLaunchKernelGGL((mc_prelim<false>), dim3(blocks_t), dim3(256), 0, ctx->stream, cp, pts, table,
                       static_cast<const uint32_t*>(d_d2.p), static_cast<uint32_t*>(d_count.p),
                       static_cast<const uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), n_threads);
  // total preliminary candidates must fit the 32-bit run delimiters
  unsigned long long total = 0;
  HIP_TRY(hipMemsetAsync(d_total.p, 0, sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(sum_u32_to_u64, dim3(sum_blocks(n_vox)), dim3(256), 0, ctx->stream, static_cast<const uint32_t*>(d_count.p),
                     n_vox, static_cast<unsigned long long*>(d_total.p));
  TRY(d2h(ctx, &total, d_total.p, sizeof(total)));
  TRY(sync_stream(ctx));
  if (total >= 0xfffffff0ULL)
    return ctx->fail(-4, "candidate index: %llu preliminary candidates exceed 32-bit offsets", total);
  HIP_TRY(hipMemcpyAsync(d_pstart.p, d_count.p, sizeof(uint32_t) * (n_vox + 1), hipMemcpyDeviceToDevice, ctx->stream));
  TRY(device_exclusive_scan(ctx, static_cast<uint32_t*>(d_pstart.p), n_vox + 1));
  TRY(scratch_alloc(ctx, d_prelim, sizeof(uint32_t) * (total ? total : 1)));
  HIP_TRY(hipMemsetAsync(d_count.p, 0, sizeof(uint32_t) * (n_vox + 1), ctx->stream));
  if (n)
    hipLaunchKernelGGL((mc_prelim<true>), dim3(blocks_t), dim3(256), 0, ctx->stream, cp, pts, table,
                       static_cast<const uint32_t*>(d_d2.p), static_cast<uint32_t*>(d_count.p),
                       static_cast<const uint32_t*>(d_pstart.p), static_cast<uint32_t*>(d_prelim.p), n_threads);
  // prune; d_count becomes the kept count per voxel
  if (ctx->cand_prune_coop)
  {
    // runs of up to 32: sixteen lanes per voxel; 33 .. 256: one wavefront per voxel, from a list the first kernel writes; the
    // rest: one thread per voxel
    constexpr int LANES = 16;
    TempBuf d_long;
    TRY(scratch_alloc(ctx, d_long, sizeof(uint32_t) * (static_cast<size_t>(n_vox) + 1)));
    uint32_t* long_count = static_cast<uint32_t*>(d_long.p);
    HIP_TRY(hipMemsetAsync(long_count, 0, sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL((mc_prune_coop<LANES>), dim3(static_cast<unsigned>((n_vox + 256 / LANES - 1) / (256 / LANES))), dim3(256),
                       0, ctx->stream, cp, pts, bxyz, static_cast<const uint32_t*>(d_pstart.p),
                       static_cast<uint32_t*>(d_prelim.p), static_cast<uint32_t*>(d_count.p), n_vox, long_count + 1, long_count);
    hipLaunchKernelGGL(mc_prune_long, dim3(static_cast<unsigned>(std::min<long long>((n_vox + 3) / 4, 2048))), dim3(256), 0,
                       ctx->stream, cp, pts, bxyz, static_cast<const uint32_t*>(d_pstart.p),
                       static_cast<uint32_t*>(d_prelim.p), static_cast<uint32_t*>(d_count.p), long_count + 1, long_count);
    hipLaunchKernelGGL(mc_prune_boxed, dim3(blocks_v), dim3(256), 0, ctx->stream, cp, pts, bxyz,
                       static_cast<const uint32_t*>(d_pstart.p), static_cast<uint32_t*>(d_prelim.p),
                       static_cast<uint32_t*>(d_count.p), n_vox, static_cast<uint32_t>(PRUNE_LONG_MAX));
  }
  else
    hipLaunchKernelGGL(mc_prune_boxed, dim3(blocks_v), dim3(256), 0, ctx->stream, cp, pts, bxyz,
                       static_cast<const uint32_t*>(d_pstart.p), static_cast<uint32_t*>(d_prelim.p),
                       static_cast<uint32_t*>(d_count.p), n_vox, 0u);
  unsigned long long kept = 0;
  HIP_TRY(hipMemsetAsync(d_total.p, 0, sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(sum_u32_to_u64, dim3(sum_blocks(n_vox)), dim3(256), 0, ctx->stream, static_cast<const uint32_t*>(d_count.p),
                     n_vox, static_cast<unsigned long long*>(d_total.p));
  TRY(d2h(ctx, &kept, d_total.p, sizeof(kept)));
  TRY(sync_stream(ctx));
  out->total = total;
  out->kept = kept;
  if (!records)
    return 0;
  // fat records: overflow slots per voxel -> exclusive scan -> write
  TempBuf d_ovf;
  TRY(scratch_alloc(ctx, d_ovf, sizeof(uint32_t) * (n_vox + 1)));
  HIP_TRY(hipMemsetAsync(d_ovf.p, 0, sizeof(uint32_t) * (n_vox + 1), ctx->stream));
  TempBuf d_hist;
  TRY(scratch_alloc(ctx, d_hist, 5 * sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(d_hist.p, 0, 5 * sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(mc_count_overflow, dim3(std::min(blocks_v, 4096u)), dim3(256), 0, ctx->stream,
                     static_cast<const uint32_t*>(d_count.p), static_cast<uint32_t*>(d_ovf.p), n_vox,
                     static_cast<unsigned long long*>(d_hist.p), cap);
  TRY(device_exclusive_scan(ctx, static_cast<uint32_t*>(d_ovf.p), n_vox + 1));
  uint32_t n_ovf = 0;
  TRY(d2h(ctx, &n_ovf, static_cast<uint32_t*>(d_ovf.p) + n_vox, sizeof(uint32_t)));
  TRY(d2h(ctx, out->hist3, d_hist.p, 5 * sizeof(unsigned long long)));
  TRY(sync_stream(ctx));
  const bool can_pack = out->hist3[3] == 0 && static_cast<unsigned long long>(ovf_base) + n_ovf <= REC_EXT_MASK;
  const bool can_bound = cap == 4 && out->hist3[3] == 0 &&
                         static_cast<unsigned long long>(ovf_base) + n_ovf < (1ull << REC2_EXT_BITS);
  if ((want_packed == 1 && !can_pack) || (want_packed == 2 && !can_bound))
    return 1;
  if (want_packed >= 0)
    out->packed = want_packed;
  else
    out->packed = (can_bound && ctx->cand_packed && ctx->cand_bound) ? 2 : (can_pack && ctx->cand_packed) ? 1 : 0;
  TRY(scratch_alloc(ctx, out->d_ovf_data, 64ull * (n_ovf ? n_ovf : 1)));
  {
    // unused candidate slots of an overflow record hold the sentinel, like those of a voxel record
    const long long words = 16ll * (n_ovf ? n_ovf : 1);
    uint32_t sentinel_bits;
    const float sentinel = REC_SENTINEL;
    memcpy(&sentinel_bits, &sentinel, sizeof(sentinel_bits));
    hipLaunchKernelGGL(mc_fill_u32, dim3(static_cast<unsigned>((words + 255) / 256)), dim3(256), 0, ctx->stream,
                       static_cast<uint32_t*>(out->d_ovf_data.p), sentinel_bits, words);
  }
  hipLaunchKernelGGL(mc_write_records, dim3(blocks_v), dim3(256), 0, ctx->stream, pts,
                     static_cast<const uint32_t*>(d_pstart.p), static_cast<const uint32_t*>(d_prelim.p),
                     static_cast<const uint32_t*>(d_count.p), static_cast<const uint32_t*>(d_ovf.p), rec_out,
                     static_cast<float*>(out->d_ovf_data.p), n_vox, cap, out->packed, cp, bxyz,
                     static_cast<double>(ctx->match_dist_min));
  HIP_TRY(hipGetLastError());
  out->n_ovf = n_ovf;
  return 0;
}

// returns 0, an error (< 0), or RC_OVER_BUDGET: the records would exceed option "index_budget_bytes" (ctx->cand_need_bytes says
// by how much; nothing has been allocated or replaced: the previous index, if any, is still in place)
constexpr int RC_OVER_BUDGET = 1;

int build_cand_grid_at(mcl3dl_hip_ctx* ctx, double voxel_ratio, uint32_t cap = 4)
{
  const unsigned long long rec_bytes = 16ull * cap;  // per voxel
  const size_t n = ctx->map_xyz.size() / 3;
  EventPairRaii evs;
  HIP_TRY(hipEventCreate(&evs.a));
  HIP_TRY(hipEventCreate(&evs.b));
  const hipEvent_t ev0 = evs.a, ev1 = evs.b;
  HIP_TRY(hipEventRecord(ev0, ctx->stream));
  // the rescaled points (PointRepresentation::vectorize; w = map index) and their bounds, on the device from the device copy
  // of the map; they stay there: a map update (update_cand_grid) re-compiles single bricks from them
  TRY(ensure_map_dev(ctx));
  TRY(ensure(ctx, ctx->cand_all_pts, sizeof(float4) * n));
  hipLaunchKernelGGL(grid_rescale_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, ctx->stream,
                     ctx->map_dev.as<float4>(), static_cast<long long>(n), ctx->weight[0], ctx->weight[1], ctx->weight[2],
                     ctx->has_weight ? 1 : 0, ctx->cand_all_pts.as<float4>());
  float mm[6];
  unsigned long long n_finite = 0;
  TRY(cloud_minmax(ctx, ctx->cand_all_pts.as<float4>(), static_cast<long long>(n), mm, &n_finite));
  if (n_finite != n)
    return ctx->fail(-3, "%llu map point(s) are not finite", static_cast<unsigned long long>(n) - n_finite);
  const float* mn = mm;
  const float* mx = mm + 3;
  CompileParams cp{};
  long long n_table = 0;
  double stretch[3];
  cand_axis_stretch(ctx, stretch);
  TRY(cand_geometry(ctx, voxel_ratio, stretch, mn, mx, &cp, &n_table));
  cp.n_points = static_cast<int>(n);

  TempBuf d_flag, d_scan, d_bxyz;
  TRY(scratch_alloc(ctx, d_flag, sizeof(int) * n_table));
  TRY(scratch_alloc(ctx, d_scan, sizeof(uint32_t) * (n_table + 1)));
  HIP_TRY(hipMemsetAsync(d_flag.p, 0, sizeof(int) * n_table, ctx->stream));
  const float4* pts = ctx->cand_all_pts.as<float4>();
  hipLaunchKernelGGL(mc_mark_bricks, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, ctx->stream, cp, pts,
                     static_cast<int*>(d_flag.p));
  HIP_TRY(hipMemsetAsync(d_scan.p, 0, sizeof(uint32_t) * (n_table + 1), ctx->stream));
  HIP_TRY(hipMemcpyAsync(d_scan.p, d_flag.p, sizeof(int) * n_table, hipMemcpyDeviceToDevice, ctx->stream));
  TRY(device_exclusive_scan(ctx, static_cast<uint32_t*>(d_scan.p), n_table + 1));
  uint32_t n_bricks = 0;
  TRY(d2h(ctx, &n_bricks, static_cast<uint32_t*>(d_scan.p) + n_table, sizeof(uint32_t)));
  TRY(sync_stream(ctx));
  if (n_bricks == 0 || n_bricks > (1u << 22))
    return ctx->fail(-4, "candidate index: %u bricks", n_bricks);
  ctx->cand_need_bytes = static_cast<double>(rec_bytes) * 512.0 * n_bricks;
  if (ctx->lik_index == 2 && ctx->index_budget_bytes > 0.0 && ctx->cand_need_bytes > ctx->index_budget_bytes)
    return RC_OVER_BUDGET;
  // one entry more than the grid has bricks: entry n_table is always -1, the entry lanes without a voxel read (eval_coop)
  TRY(ensure(ctx, ctx->cand_table, sizeof(int) * (n_table + 1)));
  int* table = ctx->cand_table.as<int>();
  hipLaunchKernelGGL(mc_brick_ids, dim3(static_cast<unsigned>((n_table + 255) / 256)), dim3(256), 0, ctx->stream,
                     static_cast<const int*>(d_flag.p), static_cast<const uint32_t*>(d_scan.p), table, n_table);
  HIP_TRY(hipMemsetAsync(table + n_table, 0xff, sizeof(int), ctx->stream));
  TRY(scratch_alloc(ctx, d_bxyz, sizeof(int) * 3 * n_bricks));
  hipLaunchKernelGGL(mc_brick_coords, dim3(static_cast<unsigned>((n_table + 255) / 256)), dim3(256), 0, ctx->stream,
                     table, cp.nbx, cp.nby, n_table, static_cast<int*>(d_bxyz.p));
  const long long n_vox = static_cast<long long>(n_bricks) * 512;
  const bool records = ctx->lik_index == 2;
  if (records)
    TRY(ensure(ctx, ctx->cand_rec, rec_bytes * static_cast<size_t>(n_vox)));
  CompileOutput co;
  TRY(compile_bricks(ctx, cp, pts, table, static_cast<const int*>(d_bxyz.p), n_bricks, records,
                     records ? ctx->cand_rec.as<float>() : nullptr, &co, cap));
  const unsigned long long total = co.total, kept = co.kept;
  ctx->cand_cp = cp;
  ctx->cand_n_table = n_table;
  ctx->cand_n_bricks = n_bricks;
  ctx->cand_n_points = n;
  if (records)
  {
    const uint32_t n_ovf = co.n_ovf;
    TRY(ensure(ctx, ctx->cand_ovf, 64ull * (n_ovf ? n_ovf : 1)));
    HIP_TRY(hipMemcpyAsync(ctx->cand_ovf.p, co.d_ovf_data.p, 64ull * (n_ovf ? n_ovf : 1), hipMemcpyDeviceToDevice,
                           ctx->stream));
    ctx->cand_n_ovf = n_ovf;
    ctx->cand_ovf_leaked = 0;
    HIP_TRY(hipEventRecord(ev1, ctx->stream));
    TRY(sync_stream(ctx));
    float ms2 = 0.f;
    HIP_TRY(hipEventSynchronize(ev1));
    HIP_TRY(hipEventElapsedTime(&ms2, ev0, ev1));
    RecGrid& g = ctx->rg;
    g.brick_table = table;
    g.rec = ctx->cand_rec.as<float4>();
    g.ovf = ctx->cand_ovf.as<float4>();
    g.ox = cp.ox;
    g.oy = cp.oy;
    g.oz = cp.oz;
    g.inv_ex = cp.inv_ex;
    g.inv_ey = cp.inv_ey;
    g.inv_ez = cp.inv_ez;
    g.nvx = cp.nvx;
    g.nvy = cp.nvy;
    g.nvz = cp.nvz;
    g.nbx = cp.nbx;
    g.nby = cp.nby;
    g.nbz = cp.nbz;
    g.mul24_ok = (static_cast<long long>(cp.nbx) * cp.nby < (1ll << 24) && cp.nbz < (1 << 24)) ? 1 : 0;
    g.off32_ok = (rec_bytes * static_cast<unsigned long long>(n_vox) < (1ull << 32)) ? 1 : 0;
    g.rec_bytes32 = bytes32(rec_bytes * static_cast<unsigned long long>(n_vox));
    g.rec_parts = static_cast<int>(cap);
    set_word_format(g, co.packed, ctx->match_dist_min);
    ctx->cand_parts = cap;
    g.ovf_bytes32 = bytes32(64ull * (n_ovf ? n_ovf : 1));
    g.ti_empty = static_cast<uint32_t>(n_table);
    ctx->footprint[5] = sizeof(int) * n_table;
    ctx->footprint[6] = rec_bytes * static_cast<size_t>(n_vox);
    ctx->footprint[7] = 64ull * n_ovf;
    ctx->cand_stats[0] = n_bricks;
    ctx->cand_stats[1] = static_cast<double>(total);
    ctx->cand_stats[2] = static_cast<double>(kept);
    ctx->cand_stats[3] = ms2;
    ctx->cand_stats[4] = static_cast<double>(co.hist3[0]);
    ctx->cand_stats[5] = static_cast<double>(co.hist3[1]);
    ctx->cand_over8 = static_cast<double>(co.hist3[2]);
    ctx->cand_stats[6] = n_ovf;
    ctx->cand_stats[7] = cp.ex / static_cast<double>(ctx->match_dist_min);
    ctx->cand_edge_ratio[0] = cp.ex / static_cast<double>(ctx->match_dist_min);
    ctx->cand_edge_ratio[1] = cp.ey / static_cast<double>(ctx->match_dist_min);
    ctx->cand_edge_ratio[2] = cp.ez / static_cast<double>(ctx->match_dist_min);
    ctx->cand_dirty = false;
    return 0;
  }
  return ctx->fail(-3, "the candidate index is built for lik_index = 2 only");
}

// The voxel edge: option "cand_voxel_ratio" x match_dist_min, or — ratio 0, the default — chosen from the map itself: r / 2
// unless more than a quarter of the voxels with candidates hold more than the four a record has room for (maps of
// voxel-filter centroids rather than lattice points: DESIGN.md section 6), then 0.36 r: the second fetch round per
// evaluation costs more than the larger table (measured: jittered C2 0.49 -> 0.37 ms, lattice C2 +2 %).
// One attempt under the footprint budget (option "index_budget_bytes", 0 = none): when the records of the wanted voxel edge
// would exceed it, the edge is coarsened (records ~ 1 / edge^2 on a map of surfaces; the mark-and-count passes that price an
// edge cost a millisecond and allocate nothing) until they fit — larger voxels hold more candidates, the tiled kernel's queued
// overflow rounds take them, results stay the same bits — and when even the coarsest edge the record format can describe
// (1.5 r: a voxel then reaches the 63 candidates the packed word counts) does not fit, the build fails with what it would need.
int build_cand_grid_budgeted(mcl3dl_hip_ctx* ctx, double ratio, uint32_t cap, double* ratio_used = nullptr)
{
  // the budget: option index_budget_bytes; -1 (default) = a quarter of the device's memory, 0 = none
  if (ctx->index_budget_opt < 0.0)
  {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess)
    {
      (void)hipGetLastError();
      total_b = 0;
    }
    ctx->index_budget_bytes = 0.25 * static_cast<double>(total_b);
  }
  else
    ctx->index_budget_bytes = ctx->index_budget_opt;
  // cubes first (fastest: profiles/r05e_aniso_ab.txt); boxes that follow the dist_weight when the cubes do not fit the budget
  // (option cand_aniso = 2, the default; 1 = always boxes, 0 = never); then coarser voxels
  const bool can_stretch = ctx->has_weight && ctx->cand_aniso != 0;
  ctx->cand_aniso_active = ctx->cand_aniso == 1 && can_stretch;
  const double asked = ratio;
  const bool asked_aniso = ctx->cand_aniso_active;
  // (every step either switches to boxes — once — or coarsens the edge by at least 8 %, so the loop reaches 1.5 r within 20
  // steps from any edge the options allow; it ends only at 1.5 r: the message below then tells the truth)
  for (int attempt = 0; attempt < 64; ++attempt)
  {
    const int rc = build_cand_grid_at(ctx, ratio, cap);
    if (rc != RC_OVER_BUDGET)
    {
      if (rc == 0 && ratio_used)
        *ratio_used = ratio;
      // whatever differs from what was asked for is said (diagnostics: mcl3dl_hip_index_note), not only visible in
      // cand_edge_ratio_* 
      if (rc == 0 && (ratio != asked || ctx->cand_aniso_active != asked_aniso))
      {
        char note[256];
        snprintf(note, sizeof(note), "index budget %.3g bytes: %s voxels of %.3g x match_dist_min instead of cubes of %.3g x",
                 ctx->index_budget_bytes, ctx->cand_aniso_active ? "dist_weight-stretched" : "cubic", ratio, asked);
        ctx->index_note = note;
      }
      return rc;
    }
    if (can_stretch && !ctx->cand_aniso_active)
    {
      ctx->cand_aniso_active = true;
      continue;
    }
    if (ratio >= 1.5)
      break;
    ratio = std::min(1.5, ratio * std::max(1.08, std::sqrt(ctx->cand_need_bytes / ctx->index_budget_bytes) * 1.08));
  }
  return ctx->fail(-4, "the candidate index needs %.3g bytes of records at a voxel edge of %.3g x match_dist_min (the coarsest "
                       "the record format describes is 1.5 x), the budget (option index_budget_bytes) is %.3g",
                   ctx->cand_need_bytes, ratio, ctx->index_budget_bytes);
}

int build_cand_grid(mcl3dl_hip_ctx* ctx)
{
  const uint32_t forced = ctx->cand_record_parts == 8 ? 8u : ctx->cand_record_parts == 4 ? 4u : 0u;
  if (ctx->cand_voxel_ratio > 0.0)
    return build_cand_grid_budgeted(ctx, ctx->cand_voxel_ratio, forced ? forced : 4u);
  double base = 0.5;
  TRY(build_cand_grid_budgeted(ctx, 0.5, forced ? forced : 4u, &base));
  const double crowded = forced == 8 ? ctx->cand_over8 : ctx->cand_stats[5];  // voxels whose candidates do not fit the record
  if (base == 0.5 && ctx->lik_index == 2 && ctx->cand_stats[4] > 0 && crowded / ctx->cand_stats[4] > 0.25)
  {
    // a crowded map (voxel-filter centroids rather than a lattice): smaller voxels, and — unless the record size is forced
    // or the tiled kernel queues its overflow rounds (below) — 128-byte records with eight inline candidates when they stay
    // below 16 GB (estimated from the first build: a voxel of 0.36 r has (0.5 / 0.36)^3 times as many of them). Measured on
    // the jittered C2 map with immediate overflow rounds: 0.42 ms (0.5 r, 64 B) -> 0.364 (0.36 r, 64 B) -> 0.338 (0.36 r,
    // 128 B); on a lattice the wide record costs 20 %, so it is never the default there.
    const double first_ms = ctx->cand_stats[3];
    const bool defer = ctx->lik_defer != 0 && ctx->cand_packed != 0 && ctx->rg.packed != 0;
    const double fine = 0.36;  // with the queue: 0.30 r measured 1.6 % faster at +-0.045 m, 2.6 % slower at +-0.02 m, twice the memory
    const double est_bricks = ctx->cand_stats[0] * std::pow(0.5 / fine, 3.0);
    const double est_bytes = 128.0 * 512.0 * est_bricks;
    // with the tiled kernel's overflow rounds deferred (lik_defer, packed records) the 64-byte record wins again:
    // 0.326 ms against 0.340 with 128-byte records at 0.36 r on that map, 0.316 at 0.30 r (profiles/r03u_defer_ab.txt)
    const uint32_t cap = forced ? forced : (defer ? 4u : est_bytes < 16.0e9 ? 8u : 4u);
    // the finer index is an optimisation of a valid one: only attempt it where it fits (brick limit of the dense table,
    // memory, the footprint budget), and if it fails all the same, put the r / 2 index back — a map the coarser default
    // handles must keep working
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    const double need = (cap == 8 ? 128.0 : 64.0) * 512.0 * est_bricks * 1.5;
    const bool in_budget = ctx->index_budget_bytes <= 0.0 || need / 1.5 * 1.1 < ctx->index_budget_bytes;
    if (est_bricks < 0.9 * static_cast<double>(1u << 22) && need < 0.8 * static_cast<double>(free_b) && in_budget)
    {
      const int rc = build_cand_grid_at(ctx, fine, cap);
      if (rc == 0)
        ctx->cand_stats[3] += first_ms;
      else if (rc == RC_OVER_BUDGET)
        ctx->index_note = "finer candidate index not built (over index_budget_bytes): voxel edge r / 2 kept";
      else
      {
        const std::string why = ctx->err;
        ctx->cand_dirty = true;  // its buffers may have been re-allocated under the first index
        TRY(build_cand_grid_budgeted(ctx, 0.5, forced ? forced : 4u));
        ctx->cand_stats[3] += first_ms;
        ctx->index_note = "finer candidate index not built (" + why + "): voxel edge r / 2 kept";
      }
    }
  }
  return 0;
}

// grow a context buffer KEEPING its content (ensure() drops it)
int ensure_keep(mcl3dl_hip_ctx* ctx, DevBuf& b, size_t keep_bytes, size_t bytes)
{
  if (b.cap >= bytes)
    return 0;
  void* old = b.p;
  const size_t cap = bytes + bytes / 2;
  void* fresh = nullptr;
  HIP_TRY(hipMalloc(&fresh, cap));
  if (old && keep_bytes)
    HIP_TRY(hipMemcpyAsync(fresh, old, keep_bytes, hipMemcpyDeviceToDevice, ctx->stream));
  TRY(sync_stream(ctx));
  if (old)
    HIP_TRY(hipFree(old));
  b.p = fresh;
  b.cap = cap;
  ++ctx->generation;
  return 0;
}

// Reclaims the overflow records that map updates orphaned: every voxel's live records move to the front of a fresh array, in
// voxel order (one pass over the records' w words, a scan, one copy pass: ~0.3 ms for the 8 M voxels of C2's map against the
// 12 ms of the whole-map rebuild that used to reclaim them).
int compact_overflow(mcl3dl_hip_ctx* ctx)
{
  const long long n_vox = static_cast<long long>(ctx->cand_n_bricks) * 512;
  const uint32_t cap = ctx->cand_parts;
  TempBuf d_cnt;
  TRY(scratch_alloc(ctx, d_cnt, sizeof(uint32_t) * static_cast<size_t>(n_vox + 1)));
  uint32_t* cnt = static_cast<uint32_t*>(d_cnt.p);
  hipLaunchKernelGGL(mc_ovf_counts, dim3(static_cast<unsigned>((n_vox + 1 + 255) / 256)), dim3(256), 0, ctx->stream,
                     ctx->cand_rec.as<float4>(), n_vox, cap, ctx->rg.packed, ctx->rg.count_shift, ctx->rg.count_is_records, cnt);
  TRY(device_exclusive_scan(ctx, cnt, n_vox + 1));
  uint32_t live = 0;
  TRY(d2h(ctx, &live, cnt + n_vox, sizeof(uint32_t)));
  TRY(sync_stream(ctx));
  void* fresh = nullptr;
  const size_t cap_bytes = 64ull * (static_cast<size_t>(live) + live / 2 + 1024);
  HIP_TRY(hipMalloc(&fresh, cap_bytes));
  hipLaunchKernelGGL(mc_ovf_move, dim3(static_cast<unsigned>((n_vox + 255) / 256)), dim3(256), 0, ctx->stream,
                     ctx->cand_rec.as<float4>(), n_vox, cap, ctx->rg.packed, ctx->rg.count_shift, ctx->rg.ext_bits, ctx->rg.count_is_records, cnt,
                     ctx->cand_ovf.as<float4>(),
                     static_cast<float4*>(fresh));
  HIP_TRY(hipGetLastError());
  TRY(sync_stream(ctx));
  if (ctx->cand_ovf.p)
    HIP_TRY(hipFree(ctx->cand_ovf.p));
  ctx->cand_ovf.p = fresh;
  ctx->cand_ovf.cap = cap_bytes;
  ctx->cand_n_ovf = live;
  ctx->cand_ovf_leaked = 0;
  ctx->rg.ovf = ctx->cand_ovf.as<float4>();
  ctx->rg.ovf_bytes32 = bytes32(64ull * (live ? live : 1));
  ctx->footprint[7] = 64ull * live;
  ctx->cand_stats[6] = live;
  ++ctx->cand_ovf_compactions;
  ++ctx->generation;
  return 0;
}

// The map changed from (base + old update) to (base + new update) — mapcloud_update, src/mcl_3dl.cpp:141-153,1355-1362:
// pc_map2 = pc_map + pc_update. Only voxels within reach of a removed or an added point can change their candidate set,
// so only the bricks holding such voxels are compiled again, from the points that can reach them, and installed over
// their old records (new bricks are appended). A brick's new records equal what a whole-map compile would write for it:
// same candidate sets, same ascending map order. Falls back to a full rebuild (cand_dirty) when nothing is built yet,
// when a point falls outside the grid the index was laid out for, or when the index is not in record form.
// ctx->map_xyz already holds base + new update; old_update = the rescaled points that were removed (host).
int update_cand_grid(mcl3dl_hip_ctx* ctx, size_t n_base, const std::vector<float4>& old_update, double* stats5)
{
  const size_t n_total = ctx->map_xyz.size() / 3;
  if (stats5)
    for (int i = 0; i < 6; ++i)
      stats5[i] = 0;
  if (ctx->cand_dirty || ctx->lik_index != 2 || ctx->cand_n_points != n_base + old_update.size())
  {
    if (stats5)
      stats5[5] = ctx->cand_dirty ? 1 : ctx->lik_index != 2 ? 2 : 3;
    ctx->cand_dirty = true;
    return 0;
  }
  CompileParams cp = ctx->cand_cp;
  const long long n_table = ctx->cand_n_table;
  std::vector<float4> fresh;
  TRY(rescaled_points(ctx, n_base, n_total - n_base, fresh, nullptr, nullptr));
  // every added point, with its reach, must lie inside the grid the index was laid out for
  for (const float4& p : fresh)
  {
    const int v[3] = { static_cast<int>(floorf((p.x - cp.ox) * cp.inv_ex)), static_cast<int>(floorf((p.y - cp.oy) * cp.inv_ey)),
                       static_cast<int>(floorf((p.z - cp.oz) * cp.inv_ez)) };
    const int nv[3] = { cp.nvx, cp.nvy, cp.nvz };
    const int reach[3] = { cp.rx, cp.ry, cp.rz };
    for (int a = 0; a < 3; ++a)
      if (v[a] - reach[a] - 1 < 0 || v[a] + reach[a] + 1 >= nv[a])
      {
        if (stats5)
          stats5[5] = 4;
        ctx->cand_dirty = true;
        return 0;
      }
  }
  EventPairRaii evs;
  HIP_TRY(hipEventCreate(&evs.a));
  HIP_TRY(hipEventCreate(&evs.b));
  const hipEvent_t ev0 = evs.a, ev1 = evs.b;
  HIP_TRY(hipEventRecord(ev0, ctx->stream));
  // 1. dirty bricks: within reach of a removed or an added point
  TempBuf d_dirty, d_old, d_newflag, d_dirtyrank, d_sub_table, d_sub_main, d_sub_bxyz, d_relflag, d_rel, d_subrec;
  TRY(scratch_alloc(ctx, d_dirty, sizeof(int) * n_table));
  HIP_TRY(hipMemsetAsync(d_dirty.p, 0, sizeof(int) * n_table, ctx->stream));
  if (!old_update.empty())
  {
    TRY(scratch_alloc(ctx, d_old, sizeof(float4) * old_update.size()));
    TRY(h2d(ctx, d_old.p, old_update.data(), sizeof(float4) * old_update.size()));
    CompileParams c2 = cp;
    c2.n_points = static_cast<int>(old_update.size());
    hipLaunchKernelGGL(mc_mark_bricks, dim3(static_cast<unsigned>((old_update.size() + 255) / 256)), dim3(256), 0,
                       ctx->stream, c2, static_cast<const float4*>(d_old.p), static_cast<int*>(d_dirty.p));
  }
  TRY(ensure_keep(ctx, ctx->cand_all_pts, sizeof(float4) * n_base, sizeof(float4) * n_total));
  if (!fresh.empty())
  {
    float4* dst = ctx->cand_all_pts.as<float4>() + n_base;
    TRY(h2d(ctx, dst, fresh.data(), sizeof(float4) * fresh.size()));
    CompileParams c2 = cp;
    c2.n_points = static_cast<int>(fresh.size());
    hipLaunchKernelGGL(mc_mark_bricks, dim3(static_cast<unsigned>((fresh.size() + 255) / 256)), dim3(256), 0, ctx->stream,
                       c2, static_cast<const float4*>(dst), static_cast<int*>(d_dirty.p));
  }
  // 2. ids: new bricks are appended; dense sub ids for the dirty ones
  const unsigned blocks_tab = static_cast<unsigned>((n_table + 1 + 255) / 256);
  TRY(scratch_alloc(ctx, d_newflag, sizeof(uint32_t) * (n_table + 1)));
  TRY(scratch_alloc(ctx, d_dirtyrank, sizeof(uint32_t) * (n_table + 1)));
  hipLaunchKernelGGL(mc_new_brick_flags, dim3(blocks_tab), dim3(256), 0, ctx->stream, static_cast<const int*>(d_dirty.p),
                     ctx->cand_table.as<int>(), static_cast<uint32_t*>(d_newflag.p), n_table);
  HIP_TRY(hipMemsetAsync(d_dirtyrank.p, 0, sizeof(uint32_t) * (n_table + 1), ctx->stream));
  HIP_TRY(hipMemcpyAsync(d_dirtyrank.p, d_dirty.p, sizeof(int) * n_table, hipMemcpyDeviceToDevice, ctx->stream));
  TRY(device_exclusive_scan(ctx, static_cast<uint32_t*>(d_newflag.p), n_table + 1));
  TRY(device_exclusive_scan(ctx, static_cast<uint32_t*>(d_dirtyrank.p), n_table + 1));
  uint32_t n_new = 0, n_dirty = 0;
  TRY(d2h(ctx, &n_new, static_cast<uint32_t*>(d_newflag.p) + n_table, sizeof(uint32_t)));
  TRY(d2h(ctx, &n_dirty, static_cast<uint32_t*>(d_dirtyrank.p) + n_table, sizeof(uint32_t)));
  TRY(sync_stream(ctx));
  const uint32_t n_bricks_old = ctx->cand_n_bricks;
  if (n_dirty == 0)
  {
    if (stats5)
      stats5[5] = 6;
    ctx->cand_n_points = n_total;
    return 0;
  }
  if (static_cast<unsigned long long>(n_bricks_old) + n_new > (1u << 22))
  {
    if (stats5)
      stats5[5] = 5;
    ctx->cand_dirty = true;  // too many bricks: start over
    return 0;
  }
  // orphaned overflow records (the previous updates' appended ones, mostly) are reclaimed once they outnumber half of the
  // array: a compaction pass, not a rebuild
  if (ctx->cand_ovf_leaked > std::max<uint32_t>(4096u, ctx->cand_n_ovf / 2))
    TRY(compact_overflow(ctx));
  const uint32_t n_bricks = n_bricks_old + n_new;
  const uint32_t cap = ctx->cand_parts;
  const unsigned long long rec_bytes = 16ull * cap;
  TRY(ensure_keep(ctx, ctx->cand_rec, rec_bytes * 512 * n_bricks_old, rec_bytes * 512 * n_bricks));
  TRY(scratch_alloc(ctx, d_sub_table, sizeof(int) * n_table));
  TRY(scratch_alloc(ctx, d_sub_main, sizeof(int) * n_dirty));
  TRY(scratch_alloc(ctx, d_sub_bxyz, sizeof(int) * 3 * n_dirty));
  hipLaunchKernelGGL(mc_dirty_tables, dim3(blocks_tab), dim3(256), 0, ctx->stream, static_cast<const int*>(d_dirty.p),
                     static_cast<const uint32_t*>(d_newflag.p), static_cast<const uint32_t*>(d_dirtyrank.p), n_bricks_old,
                     cp.nbx, cp.nby, n_table, ctx->cand_table.as<int>(), static_cast<int*>(d_sub_table.p),
                     static_cast<int*>(d_sub_main.p), static_cast<int*>(d_sub_bxyz.p));
  // 3. the points that can reach a dirty brick, in map order
  cp.n_points = static_cast<int>(n_total);
  TRY(scratch_alloc(ctx, d_relflag, sizeof(uint32_t) * (n_total + 1)));
  hipLaunchKernelGGL(mc_relevant_points, dim3(static_cast<unsigned>((n_total + 1 + 255) / 256)), dim3(256), 0, ctx->stream,
                     cp, ctx->cand_all_pts.as<float4>(), static_cast<const int*>(d_dirty.p),
                     static_cast<uint32_t*>(d_relflag.p));
  TRY(device_exclusive_scan(ctx, static_cast<uint32_t*>(d_relflag.p), static_cast<long long>(n_total) + 1));
  uint32_t n_rel = 0;
  TRY(d2h(ctx, &n_rel, static_cast<uint32_t*>(d_relflag.p) + n_total, sizeof(uint32_t)));
  TRY(sync_stream(ctx));
  TRY(scratch_alloc(ctx, d_rel, sizeof(float4) * (n_rel ? n_rel : 1)));
  hipLaunchKernelGGL(mc_compact_points, dim3(static_cast<unsigned>((n_total + 255) / 256)), dim3(256), 0, ctx->stream,
                     ctx->cand_all_pts.as<float4>(), static_cast<const uint32_t*>(d_relflag.p), static_cast<int>(n_total),
                     static_cast<float4*>(d_rel.p));
  // 4. compile the dirty bricks and install them
  const long long n_sub_vox = static_cast<long long>(n_dirty) * 512;
  TRY(scratch_alloc(ctx, d_subrec, rec_bytes * static_cast<size_t>(n_sub_vox)));
  cp.n_points = static_cast<int>(n_rel);
  CompileOutput co;
  {
    const int rc = compile_bricks(ctx, cp, static_cast<const float4*>(d_rel.p), static_cast<const int*>(d_sub_table.p),
                                  static_cast<const int*>(d_sub_bxyz.p), n_dirty, true, static_cast<float*>(d_subrec.p), &co,
                                  cap, word_format(ctx->rg), ctx->cand_n_ovf);
    if (rc == 1)
    {
      // the update does not fit the index's w words (a voxel with more candidates than the count field holds, or more
      // overflow records than the reference field addresses): the next query rebuilds the whole index, which then picks a
      // form that fits
      if (stats5)
        stats5[5] = 7;
      ctx->cand_dirty = true;
      return 0;
    }
    if (rc != 0)
      return rc;
  }
  const uint32_t ovf_base = ctx->cand_n_ovf;
  if (co.n_ovf)
  {
    TRY(ensure_keep(ctx, ctx->cand_ovf, 64ull * ovf_base, 64ull * (static_cast<size_t>(ovf_base) + co.n_ovf)));
    HIP_TRY(hipMemcpyAsync(ctx->cand_ovf.as<char>() + 64ull * ovf_base, co.d_ovf_data.p, 64ull * co.n_ovf,
                           hipMemcpyDeviceToDevice, ctx->stream));
  }
  TempBuf d_orphan;
  TRY(scratch_alloc(ctx, d_orphan, sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(d_orphan.p, 0, sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(mc_install_records, dim3(static_cast<unsigned>((n_sub_vox + 255) / 256)), dim3(256), 0, ctx->stream,
                     static_cast<const float4*>(d_subrec.p), static_cast<const int*>(d_sub_main.p), ovf_base, n_bricks_old,
                     n_sub_vox, ctx->cand_rec.as<float4>(), static_cast<unsigned long long*>(d_orphan.p), cap, ctx->rg.packed,
                     ctx->rg.count_shift, ctx->rg.count_is_records);
  HIP_TRY(hipGetLastError());
  unsigned long long orphaned = 0;
  TRY(d2h(ctx, &orphaned, d_orphan.p, sizeof(orphaned)));
  HIP_TRY(hipEventRecord(ev1, ctx->stream));
  TRY(sync_stream(ctx));
  float ms = 0.f;
  HIP_TRY(hipEventSynchronize(ev1));
  HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
  ctx->cand_n_bricks = n_bricks;
  ctx->cand_n_ovf = ovf_base + co.n_ovf;
  ctx->cand_ovf_leaked += static_cast<uint32_t>(orphaned);  // reclaimed by compact_overflow (or the next full rebuild)
  ctx->cand_n_points = n_total;
  ctx->rg.brick_table = ctx->cand_table.as<int>();
  ctx->rg.rec = ctx->cand_rec.as<float4>();
  ctx->rg.ovf = ctx->cand_ovf.as<float4>();
  ctx->rg.off32_ok = (rec_bytes * 512 * n_bricks < (1ull << 32)) ? 1 : 0;
  ctx->rg.rec_bytes32 = bytes32(rec_bytes * 512 * n_bricks);
  ctx->rg.ovf_bytes32 = bytes32(64ull * (ctx->cand_n_ovf ? ctx->cand_n_ovf : 1));
  ctx->footprint[6] = rec_bytes * 512 * n_bricks;
  ctx->footprint[7] = 64ull * ctx->cand_n_ovf;
  ctx->cand_stats[0] = n_bricks;
  ctx->cand_stats[6] = ctx->cand_n_ovf;
  ++ctx->generation;
  if (stats5)
  {
    stats5[0] = n_dirty;
    stats5[1] = n_new;
    stats5[2] = n_rel;
    stats5[3] = ms;
    stats5[4] = co.n_ovf;
  }
  return 0;
}

}  // namespace
