// mcl3dl_hip.hip — host side of the C ABI declared in include/mcl3dl_hip.h: context, map compiler
// (cell-sorted exact-NN grid, DDA occupancy), scan ordering, kernel launches, hipEvent timing.
// Device code lives in kernels.h.  gfx950 only; there is no CPU fallback anywhere in this file.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "kernels.h"
#include "mcl3dl_hip.h"

#pragma clang fp contract(off)

using namespace mcl3dl;

namespace
{
struct DevBuf
{
  void* p = nullptr;
  size_t cap = 0;
  template <typename T>
  T* as() const
  {
    return static_cast<T*>(p);
  }
};

struct EventPair
{
  hipEvent_t start, stop;
  int kernel;
};
}  // namespace

struct mcl3dl_hip_ctx
{
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  // The two LiDAR models are independent until pf::measure: the beam kernels run on a second stream, forked from and
  // joined back into `stream` with events, so their (VALU-heavy, memory-light) waves fill the slots the likelihood
  // kernel leaves idle while it waits on L2.
  hipStream_t aux_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int overlap_models = 1;
  std::string err;

  // host copy of the map (kept to rebuild the device structures when parameters change)
  std::vector<float> map_xyz;
  std::vector<uint32_t> map_label;
  uint64_t stamp = 0;
  bool has_map = false;
  bool has_weight = false;
  float weight[3] = { 1.f, 1.f, 1.f };

  // LidarMeasurementModelLikelihoodParameters defaults, include/mcl_3dl/parameters.h:74-76
  float match_dist_min = 0.2f, match_dist_flat = 0.05f, match_weight = 5.0f;
  // LidarMeasurementModelBeamParameters defaults, include/mcl_3dl/parameters.h:96-112
  float map_grid[3] = { 0.1f, 0.1f, 0.1f };
  float dda_grid_size = 0.2f;
  float ray_angle_half = static_cast<float>(0.25 * M_PI / 180.0);
  float hit_range = 0.3f;
  float beam_likelihood_min = 0.2f;
  uint32_t beam_num_points = 3;
  float ang_total_ref = static_cast<float>(M_PI / 6.0);
  uint32_t filter_label_max = 0xFFFFFFFFu;
  int short_only = 1;
  // derived, src/lidar_measurement_model_beam.cpp:65-67
  float hit_range_sq = 0, beam_likelihood = 0, sin_total_ref = 0;

  bool lik_dirty = true, dda_dirty = true, cand_dirty = true;
  DevBuf lik_pts, lik_cells;
  LikGrid lg{};
  // candidate-voxel index (map_compiler.h): lik_index 1 = use it for measure(), 0 = 27-cell scan of the cell grid
  int lik_index = 2;
  int lik_small = 1;       // 1 = several particles share a wavefront when the scan has <= 32 points
  int lik_tiled = 1;       // 1 = tile-major XCD-aware kernel for large scans, 0 = one work-group per particle always
  int lik_group = 16;      // particles per work-group of the tiled kernel (16 or 32)
  DevBuf lik_partial_sum, lik_partial_cnt;
  int strict_order = 0;    // 1 = add the likelihood terms / the weights in the reference's float order (single GPU)
  DevBuf scan_perm, strict_terms;
  double cand_voxel_ratio = 0.5;  // voxel edge / match_dist_min
  double cand_phase = 0.5;        // grid origin shifted by this fraction of a voxel (see build_cand_grid)
  DevBuf cand_table, cand_start, cand_pts, cand_rec, cand_ovf;
  CandGrid cg{};
  RecGrid rg{};
  double cand_stats[4] = { 0, 0, 0, 0 };  // bricks, voxels with candidates, candidates, build ms
  DevBuf dda_bits, dda_start, dda_pts, dda_index;
  DdaGrid dg{};
  uint64_t footprint[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };

  // scans of the current update
  DevBuf scan_lik, scan_beam, origins, pow_table;
  size_t n_s = 0, n_b = 0, n_o = 0;
  bool has_scan = false;
  bool pow_table_dirty = true;

  // work buffers
  DevBuf pose, lik, ratio, beam, weightb, wnew, extra, penalty, block_partials, partial4, stats4, ray_stats,
      tested, ray_begin, ray_end, ray_status, ray_hit, mom_blocks, mom_arg, mom_out, mom_idx, subset;

  // resampling plan
  std::vector<float> rs_keys;        // accumulated probabilities, in particles_dup_ order after std::sort
  std::vector<uint32_t> rs_order;    // which particle sits at each position of particles_dup_
  std::vector<uint32_t> rs_source, rs_slot;
  size_t rs_n = 0, rs_n_out = 0, rs_n_dup = 0;
  float rs_pstep = 0.f;
  bool rs_planned = false;
  DevBuf rs_d_keys, rs_d_pscan, rs_d_it, rs_d_source, rs_d_slot, rs_d_noise, rs_d_in, rs_d_out, rs_d_order, rs_d_flag,
      rs_d_ws, rs_d_dup8;
  bool rs_sorted = false;  // std::sort had ties to order: rs_order is not the identity

  // mcl3dl_hip_update_device: the launch sequence of one device-resident update, captured into a hipGraph the second
  // time the same arguments arrive and replayed afterwards (small updates are launch-bound: 8-10 launches of a few
  // microseconds each). `generation` counts everything that can change what gets enqueued — parameters, options, map,
  // stream, scan sizes, any device buffer that had to be reallocated.
  uint64_t generation = 0;
  int use_graph = 0;
  struct UpdateKey
  {
    const void* p[8];
    size_t n_p;
    uint64_t generation;
    bool operator==(const UpdateKey& o) const
    {
      return memcmp(p, o.p, sizeof(p)) == 0 && n_p == o.n_p && generation == o.generation;
    }
  };
  UpdateKey graph_key{}, seen_key{}, failed_key{};
  bool have_seen = false, have_failed = false;
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  uint64_t graph_replays = 0, graph_captures = 0;
  std::string graph_note;  // why the last capture attempt fell back to plain launches (diagnostics)

  // Pinned staging for the host-buffer entry points: small copies go through page-locked memory so that
  // hipMemcpyAsync really is asynchronous (a pageable copy costs a driver-side staging round trip each); results are
  // handed to the caller's arrays when the stream is synchronised (sync_stream).
  struct StageChunk
  {
    char* p;
    size_t cap;
  };
  struct StagedResult
  {
    void* user;
    const void* staged;
    size_t bytes;
  };
  std::vector<StageChunk> stage;
  size_t stage_cur = 0, stage_off = 0;
  std::vector<StagedResult> stage_out;
  // host-side scan staging (kept in the context so that it outlives the asynchronous copies)
  std::vector<float4> h_scan_lik, h_scan_beam, h_origins;
  std::vector<uint32_t> h_scan_perm;

  // timing
  bool timing = false;
  unsigned timing_mask = 0xffffffffu;  // bit k = time kernel group k (MCL3DL_KERNEL_*); each timed group costs two event records
  std::vector<EventPair> pending;
  std::vector<hipEvent_t> free_events;
  double kernel_ms[MCL3DL_KERNEL_COUNT] = { 0, 0, 0 };
  uint64_t kernel_launches[MCL3DL_KERNEL_COUNT] = { 0, 0, 0 };

  int fail(int code, const char* fmt, ...)
  {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    err = buf;
    stage_out.clear();  // results of a failed call are not delivered (their destinations may be gone)
    return code;
  }
};

namespace
{
#define HIP_TRY(expr)                                                                            \
  do                                                                                             \
  {                                                                                              \
    const hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                        \
      return ctx->fail(-2, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

#define TRY(expr)        \
  do                     \
  {                      \
    const int r_ = (expr); \
    if (r_ != 0)         \
      return r_;         \
  } while (0)

inline float bits_to_float(uint32_t u)
{
  float f;
  memcpy(&f, &u, sizeof(f));
  return f;
}

int ensure(mcl3dl_hip_ctx* ctx, DevBuf& b, size_t bytes)
{
  if (bytes == 0)
    bytes = 16;
  if (b.cap >= bytes)
    return 0;
  if (b.p)
  {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  const size_t cap = bytes + bytes / 4;
  HIP_TRY(hipMalloc(&b.p, cap));
  b.cap = cap;
  ++ctx->generation;  // a captured update graph holds the old address
  return 0;
}

constexpr size_t STAGE_MAX_COPY = 4u << 20;  // larger copies go straight from / to the caller's (pageable) memory

// bump allocation in page-locked chunks; everything is released for reuse by sync_stream. nullptr = allocation failed
// (the caller then falls back to a direct copy).
void* stage_alloc(mcl3dl_hip_ctx* ctx, size_t bytes)
{
  bytes = (bytes + 255) & ~static_cast<size_t>(255);
  while (ctx->stage_cur < ctx->stage.size())
  {
    mcl3dl_hip_ctx::StageChunk& ch = ctx->stage[ctx->stage_cur];
    if (ctx->stage_off + bytes <= ch.cap)
    {
      void* p = ch.p + ctx->stage_off;
      ctx->stage_off += bytes;
      return p;
    }
    ++ctx->stage_cur;
    ctx->stage_off = 0;
  }
  const size_t last = ctx->stage.empty() ? (512u << 10) : ctx->stage.back().cap;
  const size_t cap = std::max(bytes, 2 * last);
  void* p = nullptr;
  if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess)
  {
    (void)hipGetLastError();
    return nullptr;
  }
  ctx->stage.push_back({ static_cast<char*>(p), cap });
  ctx->stage_cur = ctx->stage.size() - 1;
  ctx->stage_off = bytes;
  return p;
}

int h2d(mcl3dl_hip_ctx* ctx, void* dst, const void* src, size_t bytes)
{
  if (bytes == 0)
    return 0;
  if (bytes <= STAGE_MAX_COPY)
  {
    if (void* p = stage_alloc(ctx, bytes))
    {
      memcpy(p, src, bytes);
      HIP_TRY(hipMemcpyAsync(dst, p, bytes, hipMemcpyHostToDevice, ctx->stream));
      return 0;
    }
  }
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  return 0;
}

// The data is in `dst` only after sync_stream().
int d2h(mcl3dl_hip_ctx* ctx, void* dst, const void* src, size_t bytes)
{
  if (bytes == 0)
    return 0;
  if (bytes <= STAGE_MAX_COPY)
  {
    if (void* p = stage_alloc(ctx, bytes))
    {
      HIP_TRY(hipMemcpyAsync(p, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
      ctx->stage_out.push_back({ dst, p, bytes });
      return 0;
    }
  }
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  return 0;
}

// hipStreamSynchronize + hand the staged results to the caller's arrays + recycle the staging memory.
int sync_stream(mcl3dl_hip_ctx* ctx)
{
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (const mcl3dl_hip_ctx::StagedResult& r : ctx->stage_out)
    memcpy(r.user, r.staged, r.bytes);
  ctx->stage_out.clear();
  ctx->stage_cur = 0;
  ctx->stage_off = 0;
  return 0;
}

// ---- timing -----------------------------------------------------------------------------------------
int timing_begin(mcl3dl_hip_ctx* ctx, int kernel, EventPair* ep, hipStream_t on = nullptr)
{
  if (!on)
    on = ctx->stream;
  ep->start = nullptr;
  if (!ctx->timing || !(ctx->timing_mask & (1u << kernel)))
    return 0;
  hipEvent_t ev[2];
  for (int i = 0; i < 2; ++i)
  {
    if (!ctx->free_events.empty())
    {
      ev[i] = ctx->free_events.back();
      ctx->free_events.pop_back();
    }
    else
    {
      HIP_TRY(hipEventCreate(&ev[i]));
    }
  }
  ep->start = ev[0];
  ep->stop = ev[1];
  ep->kernel = kernel;
  HIP_TRY(hipEventRecord(ep->start, on));
  return 0;
}

int timing_end(mcl3dl_hip_ctx* ctx, const EventPair& ep, hipStream_t on = nullptr)
{
  if (!ctx->timing || !ep.start)
    return 0;
  HIP_TRY(hipEventRecord(ep.stop, on ? on : ctx->stream));
  ctx->pending.push_back(ep);
  return 0;
}

int timing_collect(mcl3dl_hip_ctx* ctx)
{
  if (ctx->pending.empty())
    return 0;
  TRY(sync_stream(ctx));
  HIP_TRY(hipStreamSynchronize(ctx->aux_stream));
  for (const EventPair& ep : ctx->pending)
  {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ep.start, ep.stop));
    ctx->kernel_ms[ep.kernel] += ms;
    ctx->kernel_launches[ep.kernel] += 1;
    ctx->free_events.push_back(ep.start);
    ctx->free_events.push_back(ep.stop);
  }
  ctx->pending.clear();
  return 0;
}

// ---- map compiler: exact-NN grid -----------------------------------------------------------------------
// Replaces ChunkedKdtree::setInputCloud + pcl::KdTreeFLANN::setInputCloud.  The reference's chunking is a memory
// device (20 m chunks with duplicated margins, chunked_kdtree.h:124-216) whose query result equals the global
// nearest neighbour within the radius whenever radius <= max_search_radius; the grid gives that result directly.
int build_lik_grid(mcl3dl_hip_ctx* ctx)
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
int build_dda_grid(mcl3dl_hip_ctx* ctx)
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
  g.grid = grid;
  g.ray_angle_half = static_cast<double>(ctx->ray_angle_half);
  // RaycastUsingDDA ctor, raycast_using_dda.h:59: map_grid_size_y appears twice (reference quirk, kept)
  const double gx = ctx->map_grid[0], gy = ctx->map_grid[1];
  g.min_dist_thr_sq = gx * gx + gy * gy + gy * gy;
  g.hit_tolerance_f = static_cast<float>(static_cast<double>(ctx->hit_range));
  ctx->footprint[2] = sizeof(unsigned long long) * bits.size();
  ctx->footprint[3] = sizeof(uint32_t) * (total + 1);
  ctx->footprint[4] = sizeof(float4) * n + sizeof(uint32_t) * n;
  ctx->dda_dirty = false;
  return 0;
}

// ---- map compiler: candidate-voxel index (device side in map_compiler.h) ------------------------------------------
int device_exclusive_scan(mcl3dl_hip_ctx* ctx, uint32_t* data, long long n)  // in place
{
  if (n <= 0)
    return 0;
  const long long tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  uint32_t* sums = nullptr;
  if (tiles > 1)
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&sums), sizeof(uint32_t) * tiles));
  hipLaunchKernelGGL(scan_tiles, dim3(static_cast<unsigned>(tiles)), dim3(256), 0, ctx->stream, data, data, sums, n);
  if (tiles > 1)
  {
    const int rc = device_exclusive_scan(ctx, sums, tiles);
    if (rc == 0)
      hipLaunchKernelGGL(scan_add_offsets, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, ctx->stream,
                         data, sums, n);
    TRY(sync_stream(ctx));
    HIP_TRY(hipFree(sums));
    if (rc != 0)
      return rc;
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

struct TempBuf
{
  void* p = nullptr;
  ~TempBuf()
  {
    if (p)
      (void)hipFree(p);
  }
};

int build_cand_grid(mcl3dl_hip_ctx* ctx)
{
  const size_t n = ctx->map_xyz.size() / 3;
  const double r = static_cast<double>(ctx->match_dist_min);
  const float e_f = static_cast<float>(r * ctx->cand_voxel_ratio);
  if (!(e_f > 0.f) || !std::isfinite(e_f))
    return ctx->fail(-3, "bad candidate voxel edge");
  hipEvent_t ev0, ev1;
  HIP_TRY(hipEventCreate(&ev0));
  HIP_TRY(hipEventCreate(&ev1));
  HIP_TRY(hipEventRecord(ev0, ctx->stream));
  // rescaled points in map order (PointRepresentation::vectorize), w = original index
  std::vector<float4> sp(n);
  float mn[3] = { 0, 0, 0 }, mx[3] = { 0, 0, 0 };
  for (size_t i = 0; i < n; ++i)
  {
    float v[3];
    for (int a = 0; a < 3; ++a)
    {
      v[a] = ctx->has_weight ? ctx->map_xyz[3 * i + a] * ctx->weight[a] : ctx->map_xyz[3 * i + a];
      if (!std::isfinite(v[a]))
        return ctx->fail(-3, "map point %zu is not finite", i);
      if (i == 0 || v[a] < mn[a])
        mn[a] = v[a];
      if (i == 0 || v[a] > mx[a])
        mx[a] = v[a];
    }
    sp[i] = make_float4(v[0], v[1], v[2], bits_to_float(static_cast<uint32_t>(i)));
  }
  CompileParams cp{};
  cp.e = static_cast<double>(e_f);
  cp.inv_e = 1.0f / e_f;
  cp.grow = 1e-3 * cp.e;
  const double r_hi = r * (1.0 + 1e-5);
  cp.r2_hi = r_hi * r_hi;
  cp.margin = 1e-5 * r * r;
  cp.reach = static_cast<int>(std::floor((r_hi + cp.grow) / cp.e)) + 1;
  cp.n_points = static_cast<int>(n);
  float o[3];
  int nv[3], nb[3];
  double n_table_d = 1;
  for (int a = 0; a < 3; ++a)
  {
    // Phase: maps that come out of a voxel filter sit on a lattice; with the origin ON that lattice every voxel face
    // coincides with a Voronoi face of the map and each voxel keeps 3 candidates per axis instead of the 2 a generic
    // position needs. Half a voxel of phase puts lattice maps in the generic position; arbitrary maps do not care.
    o[a] = mn[a] - static_cast<float>((cp.reach + 1 + ctx->cand_phase) * cp.e);
    nv[a] = static_cast<int>(std::floor((static_cast<double>(mx[a]) - o[a]) / cp.e)) + cp.reach + 2;
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
  const long long n_table = static_cast<long long>(n_table_d);

  TempBuf d_pts, d_flag, d_scan, d_d2, d_count, d_pstart, d_prelim, d_bxyz, d_total;
  HIP_TRY(hipMalloc(&d_pts.p, sizeof(float4) * n));
  TRY(h2d(ctx, d_pts.p, sp.data(), sizeof(float4) * n));
  HIP_TRY(hipMalloc(&d_flag.p, sizeof(int) * n_table));
  HIP_TRY(hipMalloc(&d_scan.p, sizeof(uint32_t) * (n_table + 1)));
  HIP_TRY(hipMemsetAsync(d_flag.p, 0, sizeof(int) * n_table, ctx->stream));
  const float4* pts = static_cast<const float4*>(d_pts.p);
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
  TRY(ensure(ctx, ctx->cand_table, sizeof(int) * n_table));
  int* table = ctx->cand_table.as<int>();
  hipLaunchKernelGGL(mc_brick_ids, dim3(static_cast<unsigned>((n_table + 255) / 256)), dim3(256), 0, ctx->stream,
                     static_cast<const int*>(d_flag.p), static_cast<const uint32_t*>(d_scan.p), table, n_table);
  HIP_TRY(hipMalloc(&d_bxyz.p, sizeof(int) * 3 * n_bricks));
  hipLaunchKernelGGL(mc_brick_coords, dim3(static_cast<unsigned>((n_table + 255) / 256)), dim3(256), 0, ctx->stream,
                     table, cp.nbx, cp.nby, n_table, static_cast<int*>(d_bxyz.p));

  const long long n_vox = static_cast<long long>(n_bricks) * 512;
  const int side = 2 * cp.reach + 1;
  const long long n_threads = static_cast<long long>(n) * side * side * side;
  const unsigned blocks_t = static_cast<unsigned>((n_threads + 255) / 256);
  if ((n_threads + 255) / 256 > 0x7fffffffLL)
    return ctx->fail(-4, "candidate index: too many (point, voxel) pairs");
  const unsigned blocks_v = static_cast<unsigned>((n_vox + 1 + 255) / 256);
  HIP_TRY(hipMalloc(&d_d2.p, sizeof(uint32_t) * n_vox));
  HIP_TRY(hipMalloc(&d_count.p, sizeof(uint32_t) * (n_vox + 1)));
  HIP_TRY(hipMalloc(&d_pstart.p, sizeof(uint32_t) * (n_vox + 1)));
  HIP_TRY(hipMalloc(&d_total.p, sizeof(unsigned long long)));
  hipLaunchKernelGGL(mc_fill_u32, dim3(blocks_v), dim3(256), 0, ctx->stream, static_cast<uint32_t*>(d_d2.p), 0x7f800000u,
                     n_vox);
  hipLaunchKernelGGL(mc_scatter_dmax, dim3(blocks_t), dim3(256), 0, ctx->stream, cp, pts, table,
                     static_cast<uint32_t*>(d_d2.p), n_threads);
  HIP_TRY(hipMemsetAsync(d_count.p, 0, sizeof(uint32_t) * (n_vox + 1), ctx->stream));
  hipLaunchKernelGGL((mc_prelim<false>), dim3(blocks_t), dim3(256), 0, ctx->stream, cp, pts, table,
                     static_cast<const uint32_t*>(d_d2.p), static_cast<uint32_t*>(d_count.p),
                     static_cast<const uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), n_threads);
  // total preliminary candidates must fit the 32-bit run delimiters
  unsigned long long total = 0;
  HIP_TRY(hipMemsetAsync(d_total.p, 0, sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(sum_u32_to_u64, dim3(1024), dim3(256), 0, ctx->stream, static_cast<const uint32_t*>(d_count.p),
                     n_vox, static_cast<unsigned long long*>(d_total.p));
  TRY(d2h(ctx, &total, d_total.p, sizeof(total)));
  TRY(sync_stream(ctx));
  if (total >= 0xfffffff0ULL)
    return ctx->fail(-4, "candidate index: %llu preliminary candidates exceed 32-bit offsets", total);
  HIP_TRY(hipMemcpyAsync(d_pstart.p, d_count.p, sizeof(uint32_t) * (n_vox + 1), hipMemcpyDeviceToDevice, ctx->stream));
  TRY(device_exclusive_scan(ctx, static_cast<uint32_t*>(d_pstart.p), n_vox + 1));
  HIP_TRY(hipMalloc(&d_prelim.p, sizeof(uint32_t) * (total ? total : 1)));
  HIP_TRY(hipMemsetAsync(d_count.p, 0, sizeof(uint32_t) * (n_vox + 1), ctx->stream));
  hipLaunchKernelGGL((mc_prelim<true>), dim3(blocks_t), dim3(256), 0, ctx->stream, cp, pts, table,
                     static_cast<const uint32_t*>(d_d2.p), static_cast<uint32_t*>(d_count.p),
                     static_cast<const uint32_t*>(d_pstart.p), static_cast<uint32_t*>(d_prelim.p), n_threads);
  // prune; d_count becomes the kept count per voxel
  hipLaunchKernelGGL(mc_prune_boxed, dim3(blocks_v), dim3(256), 0, ctx->stream, cp, pts, static_cast<const int*>(d_bxyz.p),
                     static_cast<const uint32_t*>(d_pstart.p), static_cast<uint32_t*>(d_prelim.p),
                     static_cast<uint32_t*>(d_count.p), n_vox);
  unsigned long long kept = 0;
  HIP_TRY(hipMemsetAsync(d_total.p, 0, sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(sum_u32_to_u64, dim3(1024), dim3(256), 0, ctx->stream, static_cast<const uint32_t*>(d_count.p),
                     n_vox, static_cast<unsigned long long*>(d_total.p));
  TRY(d2h(ctx, &kept, d_total.p, sizeof(kept)));
  TRY(sync_stream(ctx));
  if (ctx->lik_index == 2)
  {
    // fat records: overflow slots per voxel -> exclusive scan -> write
    TempBuf d_ovf;
    HIP_TRY(hipMalloc(&d_ovf.p, sizeof(uint32_t) * (n_vox + 1)));
    HIP_TRY(hipMemsetAsync(d_ovf.p, 0, sizeof(uint32_t) * (n_vox + 1), ctx->stream));
    hipLaunchKernelGGL(mc_count_overflow, dim3(blocks_v), dim3(256), 0, ctx->stream,
                       static_cast<const uint32_t*>(d_count.p), static_cast<uint32_t*>(d_ovf.p), n_vox);
    TRY(device_exclusive_scan(ctx, static_cast<uint32_t*>(d_ovf.p), n_vox + 1));
    uint32_t n_ovf = 0;
    TRY(d2h(ctx, &n_ovf, static_cast<uint32_t*>(d_ovf.p) + n_vox, sizeof(uint32_t)));
    TRY(sync_stream(ctx));
    TRY(ensure(ctx, ctx->cand_rec, 64ull * static_cast<size_t>(n_vox)));
    TRY(ensure(ctx, ctx->cand_ovf, 64ull * (n_ovf ? n_ovf : 1)));
    HIP_TRY(hipMemsetAsync(ctx->cand_ovf.p, 0, 64ull * (n_ovf ? n_ovf : 1), ctx->stream));
    hipLaunchKernelGGL(mc_write_records, dim3(blocks_v), dim3(256), 0, ctx->stream, pts,
                       static_cast<const uint32_t*>(d_pstart.p), static_cast<const uint32_t*>(d_prelim.p),
                       static_cast<const uint32_t*>(d_count.p), static_cast<const uint32_t*>(d_ovf.p),
                       ctx->cand_rec.as<float>(), ctx->cand_ovf.as<float>(), n_vox);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev1, ctx->stream));
    TRY(sync_stream(ctx));
    float ms2 = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms2, ev0, ev1));
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    RecGrid& g = ctx->rg;
    g.brick_table = table;
    g.rec = ctx->cand_rec.as<float4>();
    g.ovf = ctx->cand_ovf.as<float4>();
    g.ox = cp.ox;
    g.oy = cp.oy;
    g.oz = cp.oz;
    g.inv_e = cp.inv_e;
    g.nvx = cp.nvx;
    g.nvy = cp.nvy;
    g.nvz = cp.nvz;
    g.nbx = cp.nbx;
    g.nby = cp.nby;
    g.nbz = cp.nbz;
    g.mul24_ok = (static_cast<long long>(cp.nbx) * cp.nby < (1ll << 24) && cp.nbz < (1 << 24)) ? 1 : 0;
    ctx->footprint[5] = sizeof(int) * n_table;
    ctx->footprint[6] = 64ull * static_cast<size_t>(n_vox);
    ctx->footprint[7] = 64ull * n_ovf;
    ctx->cand_stats[0] = n_bricks;
    ctx->cand_stats[1] = static_cast<double>(total);
    ctx->cand_stats[2] = static_cast<double>(kept);
    ctx->cand_stats[3] = ms2;
    ctx->cand_dirty = false;
    return 0;
  }
  TRY(ensure(ctx, ctx->cand_start, sizeof(uint32_t) * (n_vox + 1)));
  TRY(ensure(ctx, ctx->cand_pts, sizeof(float4) * (kept ? kept : 1)));
  HIP_TRY(hipMemsetAsync(static_cast<uint32_t*>(d_count.p) + n_vox, 0, sizeof(uint32_t), ctx->stream));
  HIP_TRY(hipMemcpyAsync(ctx->cand_start.p, d_count.p, sizeof(uint32_t) * (n_vox + 1), hipMemcpyDeviceToDevice,
                         ctx->stream));
  TRY(device_exclusive_scan(ctx, ctx->cand_start.as<uint32_t>(), n_vox + 1));
  hipLaunchKernelGGL(mc_write_final, dim3(blocks_v), dim3(256), 0, ctx->stream, pts,
                     static_cast<const uint32_t*>(d_pstart.p), static_cast<const uint32_t*>(d_prelim.p),
                     ctx->cand_start.as<uint32_t>(), ctx->cand_pts.as<float4>(), n_vox);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(ev1, ctx->stream));
  TRY(sync_stream(ctx));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
  (void)hipEventDestroy(ev0);
  (void)hipEventDestroy(ev1);
  CandGrid& g = ctx->cg;
  g.brick_table = table;
  g.vox_start = ctx->cand_start.as<uint32_t>();
  g.cand = ctx->cand_pts.as<float4>();
  g.ox = cp.ox;
  g.oy = cp.oy;
  g.oz = cp.oz;
  g.inv_e = cp.inv_e;
  g.nvx = cp.nvx;
  g.nvy = cp.nvy;
  g.nvz = cp.nvz;
  g.nbx = cp.nbx;
  g.nby = cp.nby;
  g.nbz = cp.nbz;
  ctx->footprint[5] = sizeof(int) * n_table;
  ctx->footprint[6] = sizeof(uint32_t) * (n_vox + 1);
  ctx->footprint[7] = sizeof(float4) * kept;
  ctx->cand_stats[0] = n_bricks;
  ctx->cand_stats[1] = static_cast<double>(total);
  ctx->cand_stats[2] = static_cast<double>(kept);
  ctx->cand_stats[3] = ms;
  ctx->cand_dirty = false;
  return 0;
}

int ensure_structures(mcl3dl_hip_ctx* ctx, bool need_lik, bool need_dda, bool need_cells = false)
{
  if (!ctx->has_map)
    return ctx->fail(-5, "no map: call mcl3dl_hip_set_map first");
  if (need_lik && (ctx->lik_index == 0 || need_cells) && ctx->lik_dirty)
    TRY(build_lik_grid(ctx));
  if (need_lik && ctx->lik_index >= 1 && !need_cells && ctx->cand_dirty)
    TRY(build_cand_grid(ctx));
  if (need_dda && ctx->dda_dirty)
    TRY(build_dda_grid(ctx));
  return 0;
}

LikParams lik_params(const mcl3dl_hip_ctx* ctx)
{
  LikParams p;
  p.wx = ctx->weight[0];
  p.wy = ctx->weight[1];
  p.wz = ctx->weight[2];
  p.has_weight = ctx->has_weight ? 1 : 0;
  p.match_dist_min = ctx->match_dist_min;
  // pcl::KdTreeFLANN::radiusSearch: (float)(radius * radius) with radius widened to double
  p.r2 = static_cast<float>(static_cast<double>(ctx->match_dist_min) * static_cast<double>(ctx->match_dist_min));
  p.match_dist_flat = ctx->match_dist_flat;
  p.match_weight = ctx->match_weight;
  return p;
}

BeamParams beam_params(const mcl3dl_hip_ctx* ctx)
{
  BeamParams p;
  p.sin_total_ref = ctx->sin_total_ref;
  p.hit_range_sq = ctx->hit_range_sq;
  p.filter_label_max = ctx->filter_label_max;
  p.short_only = ctx->short_only;
  p.beam_likelihood_min = ctx->beam_likelihood_min;
  return p;
}

// LidarMeasurementModelBeam::refreshParameters, src/lidar_measurement_model_beam.cpp:65-67 (host libm, like the reference)
void beam_refresh(mcl3dl_hip_ctx* ctx)
{
  ctx->hit_range_sq = static_cast<float>(std::pow(static_cast<double>(ctx->hit_range), 2));
  ctx->beam_likelihood = static_cast<float>(
      std::pow(static_cast<double>(ctx->beam_likelihood_min), 1.0 / static_cast<float>(ctx->beam_num_points)));
  ctx->sin_total_ref = sinf(ctx->ang_total_ref);
  ctx->pow_table_dirty = true;
}

// 3-D Morton key of a scan point (robot frame), 0.25 m cells: neighbouring lanes of a wavefront then gather from
// neighbouring map cells.
uint64_t morton3(uint32_t x, uint32_t y, uint32_t z)
{
  auto spread = [](uint64_t v)
  {
    v &= 0x1fffff;
    v = (v | v << 32) & 0x1f00000000ffffULL;
    v = (v | v << 16) & 0x1f0000ff0000ffULL;
    v = (v | v << 8) & 0x100f00f00f00f00fULL;
    v = (v | v << 4) & 0x10c30c30c30c30c3ULL;
    v = (v | v << 2) & 0x1249249249249249ULL;
    return v;
  };
  return spread(x) | (spread(y) << 1) | (spread(z) << 2);
}

int launch_measure(mcl3dl_hip_ctx* ctx, const float* d_pose, size_t n_p, float* d_lik, float* d_ratio, float* d_beam,
                   bool stats, double* stats6)
{
  if (!ctx->has_scan)
    return ctx->fail(-5, "no scan uploaded: call mcl3dl_hip_upload_scan first");
  if (n_p == 0)
    return 0;
  if (n_p > 0x7fffffffu)
    return ctx->fail(-3, "too many particles");
  const bool want_lik = (d_lik || d_ratio || stats);
  const bool want_beam = (d_beam || stats);
  TRY(ensure_structures(ctx, want_lik && ctx->n_s > 0, want_beam && ctx->n_b > 0, stats));
  const int np = static_cast<int>(n_p);
  bool beam_forked = false;
  // ---- beam model (enqueued first: on its own stream when both models run, see mcl3dl_hip_ctx::aux_stream)
  if (want_beam)
  {
    if (ctx->n_b == 0)
    {
      if (!stats)
        hipLaunchKernelGGL(fill_kernel, dim3((np + 255) / 256), dim3(256), 0, ctx->stream, d_beam, 1.0f,
                           static_cast<float*>(nullptr), 0.0f, np);
    }
    else
    {
      if (ctx->pow_table_dirty)
      {
        // score_beam *= beam_likelihood_ repeated k times (beam.cpp:148), float
        std::vector<float> table(ctx->n_b + 1);
        table[0] = 1.0f;
        for (size_t k = 1; k <= ctx->n_b; ++k)
          table[k] = table[k - 1] * ctx->beam_likelihood;
        TRY(ensure(ctx, ctx->pow_table, sizeof(float) * table.size()));
        TRY(h2d(ctx, ctx->pow_table.p, table.data(), sizeof(float) * table.size()));
        TRY(sync_stream(ctx));
        ctx->pow_table_dirty = false;
      }
      const BeamParams bp = beam_params(ctx);
      const long long n_rays = static_cast<long long>(n_p) * static_cast<long long>(ctx->n_b);
      const long long blocks = (n_rays + 255) / 256;
      if (blocks > 0x7fffffffLL)
        return ctx->fail(-3, "too many rays for one launch");
      TRY(ensure(ctx, ctx->penalty, sizeof(unsigned) * n_p));
      const bool overlap = ctx->overlap_models && !stats && want_lik && ctx->n_s > 0;
      hipStream_t bs = overlap ? ctx->aux_stream : ctx->stream;
      if (overlap)
      {
        HIP_TRY(hipEventRecord(ctx->ev_fork, ctx->stream));
        HIP_TRY(hipStreamWaitEvent(bs, ctx->ev_fork, 0));
      }
      EventPair ep{};
      if (!stats)
        TRY(timing_begin(ctx, MCL3DL_KERNEL_BEAM, &ep, bs));
      HIP_TRY(hipMemsetAsync(ctx->penalty.p, 0, sizeof(unsigned) * n_p, bs));
      if (stats)
      {
        TRY(ensure(ctx, ctx->ray_stats, sizeof(RayStats)));
        HIP_TRY(hipMemsetAsync(ctx->ray_stats.p, 0, sizeof(RayStats), bs));
        hipLaunchKernelGGL((beam_kernel<true>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, bs, d_pose,
                           ctx->scan_beam.as<float4>(), static_cast<int>(ctx->n_b), ctx->origins.as<float4>(), n_rays,
                           ctx->dg, bp, ctx->penalty.as<unsigned>(), ctx->ray_stats.as<RayStats>());
      }
      else
      {
        hipLaunchKernelGGL((beam_kernel<false>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, bs, d_pose,
                           ctx->scan_beam.as<float4>(), static_cast<int>(ctx->n_b), ctx->origins.as<float4>(), n_rays,
                           ctx->dg, bp, ctx->penalty.as<unsigned>(), static_cast<RayStats*>(nullptr));
        hipLaunchKernelGGL(beam_finalize_kernel, dim3((np + 255) / 256), dim3(256), 0, bs,
                           ctx->penalty.as<unsigned>(), ctx->pow_table.as<float>(), ctx->beam_likelihood_min, d_beam,
                           np);
        TRY(timing_end(ctx, ep, bs));
      }
      if (overlap)
      {
        HIP_TRY(hipEventRecord(ctx->ev_join, bs));
        beam_forked = true;
      }
    }
    HIP_TRY(hipGetLastError());
  }
  // ---- likelihood-field model
  if (want_lik)
  {
    if (ctx->n_s == 0)
    {
      if (!stats)
        hipLaunchKernelGGL(fill_kernel, dim3((np + 255) / 256), dim3(256), 0, ctx->stream, d_lik, 1.0f, d_ratio, 0.0f,
                           np);
    }
    else
    {
      const LikParams lp = lik_params(ctx);
      const int ns = static_cast<int>(ctx->n_s);
      EventPair ep{};
      if (stats)
      {
        TRY(ensure(ctx, ctx->tested, sizeof(double) * n_p));
        hipLaunchKernelGGL((likelihood_kernel<256, 0, true>), dim3(np), dim3(256), 0, ctx->stream, d_pose,
                           ctx->scan_lik.as<float4>(), ns, ctx->lg, ctx->cg, ctx->rg, lp, nullptr, nullptr,
                           ctx->tested.as<double>());
      }
      else
      {
        TRY(timing_begin(ctx, MCL3DL_KERNEL_LIKELIHOOD, &ep));
        const float4* scan = ctx->scan_lik.as<float4>();
        const bool tiled = (ctx->lik_tiled && ns >= 1024 && np >= 64) || ctx->strict_order;
        float* strict_terms = nullptr;
        if (ctx->strict_order)
        {
          const size_t G = static_cast<size_t>(ctx->lik_group);  // rows of G floats per particle group
          TRY(ensure(ctx, ctx->strict_terms, sizeof(float) * static_cast<size_t>(ns) * ((n_p + G - 1) / G) * G));
          strict_terms = ctx->strict_terms.as<float>();
        }
        const bool small = !tiled && ns <= 32 && np >= 256 && ctx->lik_small;
        if (small)
        {
          int W = 1;
          while (W < ns)
            W <<= 1;
          const long long blocks = (static_cast<long long>(np) * W + 255) / 256;
          if (blocks > 0x7fffffffLL)
            return ctx->fail(-3, "too many work-groups for the small-scan likelihood kernel");
#define LAUNCH_SMALL(WW, MODE)                                                                                         \
  hipLaunchKernelGGL((likelihood_small_kernel<WW, MODE>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0,           \
                     ctx->stream, d_pose, np, scan, ns, ctx->lg, ctx->cg, ctx->rg, lp, d_lik, d_ratio)
#define LAUNCH_SMALL_W(MODE)       \
  switch (W)                       \
  {                                \
    case 1: LAUNCH_SMALL(1, MODE); break;   \
    case 2: LAUNCH_SMALL(2, MODE); break;   \
    case 4: LAUNCH_SMALL(4, MODE); break;   \
    case 8: LAUNCH_SMALL(8, MODE); break;   \
    case 16: LAUNCH_SMALL(16, MODE); break; \
    default: LAUNCH_SMALL(32, MODE); break; \
  }
          if (ctx->lik_index == 2)
          {
            LAUNCH_SMALL_W(2)
          }
          else if (ctx->lik_index == 1)
          {
            LAUNCH_SMALL_W(1)
          }
          else
          {
            LAUNCH_SMALL_W(0)
          }
#undef LAUNCH_SMALL_W
#undef LAUNCH_SMALL
        }
        else if (tiled)
        {
          const int G = ctx->lik_group;
          const int n_tiles = (ns + 255) / 256, n_groups = (np + G - 1) / G;
          const long long blocks = static_cast<long long>((n_tiles + 7) / 8) * 8 * n_groups;
          if (blocks > 0x7fffffffLL)
            return ctx->fail(-3, "too many work-groups for the tiled likelihood kernel");
          TRY(ensure(ctx, ctx->lik_partial_sum, sizeof(double) * static_cast<size_t>(n_tiles) * n_p));
          TRY(ensure(ctx, ctx->lik_partial_cnt, sizeof(unsigned) * static_cast<size_t>(n_tiles) * n_p));
#define LAUNCH_TILED(GG, MODE)                                                                                         \
  hipLaunchKernelGGL((likelihood_tiled_kernel<GG, MODE>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0,           \
                     ctx->stream, d_pose, np, scan, ns, n_tiles, n_groups, ctx->lg, ctx->cg, ctx->rg, lp,              \
                     ctx->lik_partial_sum.as<double>(), ctx->lik_partial_cnt.as<unsigned>(),                          \
                     ctx->scan_perm.as<uint32_t>(), strict_terms)
          if (G == 8)
          {
            if (ctx->lik_index == 2)
              LAUNCH_TILED(8, 2);
            else if (ctx->lik_index == 1)
              LAUNCH_TILED(8, 1);
            else
              LAUNCH_TILED(8, 0);
          }
          else if (G == 32)
          {
            if (ctx->lik_index == 2)
              LAUNCH_TILED(32, 2);
            else if (ctx->lik_index == 1)
              LAUNCH_TILED(32, 1);
            else
              LAUNCH_TILED(32, 0);
          }
          else
          {
            if (ctx->lik_index == 2)
              LAUNCH_TILED(16, 2);
            else if (ctx->lik_index == 1)
              LAUNCH_TILED(16, 1);
            else
              LAUNCH_TILED(16, 0);
          }
#undef LAUNCH_TILED
          hipLaunchKernelGGL(lik_finalize_kernel, dim3((np + 31) / 32), dim3(256), 0, ctx->stream,
                             ctx->lik_partial_sum.as<double>(), ctx->lik_partial_cnt.as<unsigned>(), n_tiles, np, ns,
                             d_lik, d_ratio);
          if (strict_terms && d_lik)
          {
            if (G == 8)
              hipLaunchKernelGGL(lik_strict_sum_kernel<8>, dim3(n_groups), dim3(256), 0, ctx->stream, strict_terms, ns, np,
                                 d_lik);
            else if (G == 32)
              hipLaunchKernelGGL(lik_strict_sum_kernel<32>, dim3(n_groups), dim3(256), 0, ctx->stream, strict_terms, ns,
                                 np, d_lik);
            else
              hipLaunchKernelGGL(lik_strict_sum_kernel<16>, dim3(n_groups), dim3(256), 0, ctx->stream, strict_terms, ns,
                                 np, d_lik);
          }
        }
        else
        {
#define LAUNCH_LIK(BLOCK, MODE)                                                                                   \
  hipLaunchKernelGGL((likelihood_kernel<BLOCK, MODE, false>), dim3(np), dim3(BLOCK), 0, ctx->stream, d_pose, scan, ns, \
                     ctx->lg, ctx->cg, ctx->rg, lp, d_lik, d_ratio, nullptr)
        if (ctx->lik_index == 2)
        {
          if (ns <= 128)
            LAUNCH_LIK(64, 2);
          else
            LAUNCH_LIK(256, 2);
        }
        else if (ctx->lik_index == 1)
        {
          if (ns <= 128)
            LAUNCH_LIK(64, 1);
          else
            LAUNCH_LIK(256, 1);
        }
        else
        {
          if (ns <= 128)
            LAUNCH_LIK(64, 0);
          else
            LAUNCH_LIK(256, 0);
        }
#undef LAUNCH_LIK
        }
        TRY(timing_end(ctx, ep));
      }
    }
    HIP_TRY(hipGetLastError());
  }
  if (beam_forked)
    HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));  // later work on `stream` sees the beam scores
  if (stats)
  {
    std::vector<double> tested(ctx->n_s ? n_p : 0);
    RayStats rs{ 0, 0, 0 };
    if (ctx->n_s)
      TRY(d2h(ctx, tested.data(), ctx->tested.p, sizeof(double) * n_p));
    if (ctx->n_b)
      TRY(d2h(ctx, &rs, ctx->ray_stats.p, sizeof(RayStats)));
    TRY(sync_stream(ctx));
    stats6[0] = std::accumulate(tested.begin(), tested.end(), 0.0);
    stats6[1] = static_cast<double>(n_p) * static_cast<double>(ctx->n_s);
    stats6[2] = static_cast<double>(rs.steps);
    stats6[3] = static_cast<double>(rs.occupied);
    stats6[4] = static_cast<double>(rs.tested);
    stats6[5] = static_cast<double>(n_p) * static_cast<double>(ctx->n_b);
  }
  return 0;
}

int pf_blocks(size_t n)
{
  const size_t b = (n + PF_BLOCK - 1) / PF_BLOCK;
  return static_cast<int>(std::min<size_t>(std::max<size_t>(b, 1), 1024));
}
}  // namespace

// =================================================================================================================
// C ABI
// =================================================================================================================
extern "C"
{
int mcl3dl_hip_abi_version(void)
{
  return MCL3DL_HIP_ABI_VERSION;
}

int mcl3dl_hip_create(mcl3dl_hip_ctx** out, int device_id)
{
  if (!out)
    return -1;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return -6;  // no GPU: there is deliberately no CPU fallback
  if (device_id < 0 || device_id >= count)
    return -6;
  mcl3dl_hip_ctx* ctx = new mcl3dl_hip_ctx;
  ctx->device = device_id;
  if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess)
  {
    delete ctx;
    return -2;
  }
  ctx->stream = ctx->own_stream;
  if (hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess)
  {
    delete ctx;
    return -2;
  }
  beam_refresh(ctx);
  *out = ctx;
  return 0;
}

void mcl3dl_hip_destroy(mcl3dl_hip_ctx* ctx)
{
  if (!ctx)
    return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->aux_stream)
    (void)hipStreamSynchronize(ctx->aux_stream);
  DevBuf* bufs[] = { &ctx->cand_table, &ctx->cand_start, &ctx->cand_pts, &ctx->cand_rec, &ctx->cand_ovf, &ctx->lik_partial_sum, &ctx->lik_partial_cnt, &ctx->scan_perm, &ctx->strict_terms, &ctx->mom_blocks, &ctx->mom_arg, &ctx->mom_out, &ctx->mom_idx,
                     &ctx->subset, &ctx->rs_d_keys, &ctx->rs_d_pscan, &ctx->rs_d_it, &ctx->rs_d_source, &ctx->rs_d_slot,
                     &ctx->rs_d_noise, &ctx->rs_d_in, &ctx->rs_d_out, &ctx->lik_pts, &ctx->lik_cells, &ctx->dda_bits, &ctx->dda_start, &ctx->dda_pts, &ctx->dda_index,
                     &ctx->scan_lik, &ctx->scan_beam, &ctx->origins, &ctx->pow_table, &ctx->pose, &ctx->lik,
                     &ctx->ratio, &ctx->beam, &ctx->weightb, &ctx->wnew, &ctx->extra, &ctx->penalty,
                     &ctx->block_partials, &ctx->partial4, &ctx->stats4, &ctx->ray_stats, &ctx->tested,
                     &ctx->ray_begin, &ctx->ray_end, &ctx->ray_status, &ctx->ray_hit };
  for (DevBuf* b : bufs)
    if (b->p)
      (void)hipFree(b->p);
  for (const EventPair& ep : ctx->pending)
  {
    (void)hipEventDestroy(ep.start);
    (void)hipEventDestroy(ep.stop);
  }
  for (hipEvent_t e : ctx->free_events)
    (void)hipEventDestroy(e);
  if (ctx->graph_exec)
    (void)hipGraphExecDestroy(ctx->graph_exec);
  if (ctx->graph)
    (void)hipGraphDestroy(ctx->graph);
  for (const mcl3dl_hip_ctx::StageChunk& ch : ctx->stage)
    (void)hipHostFree(ch.p);
  if (ctx->ev_fork)
    (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join)
    (void)hipEventDestroy(ctx->ev_join);
  if (ctx->aux_stream)
    (void)hipStreamDestroy(ctx->aux_stream);
  if (ctx->own_stream)
    (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

const char* mcl3dl_hip_last_error(const mcl3dl_hip_ctx* ctx)
{
  return ctx ? ctx->err.c_str() : "null context";
}

int mcl3dl_hip_set_stream(mcl3dl_hip_ctx* ctx, void* hip_stream)
{
  if (!ctx)
    return -1;
  ++ctx->generation;
  TRY(sync_stream(ctx));
  ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
  return 0;
}

void* mcl3dl_hip_get_stream(mcl3dl_hip_ctx* ctx)
{
  return ctx ? static_cast<void*>(ctx->stream) : nullptr;
}

int mcl3dl_hip_synchronize(mcl3dl_hip_ctx* ctx)
{
  if (!ctx)
    return -1;
  TRY(sync_stream(ctx));
  return 0;
}

int mcl3dl_hip_set_map(mcl3dl_hip_ctx* ctx, const float* xyz, const uint32_t* label, size_t n_m, uint64_t stamp,
                       const float* dist_weight)
{
  if (!ctx)
    return -1;
  ++ctx->generation;
  if (!xyz || n_m == 0)
    return ctx->fail(-3, "empty map");
  if (n_m > 0xfffffff0u)
    return ctx->fail(-3, "map too large (index must fit 32 bits)");
  HIP_TRY(hipSetDevice(ctx->device));
  ctx->map_xyz.assign(xyz, xyz + 3 * n_m);
  if (label)
    ctx->map_label.assign(label, label + n_m);
  else
    ctx->map_label.assign(n_m, 0u);
  ctx->stamp = stamp;
  ctx->has_weight = dist_weight != nullptr;
  for (int a = 0; a < 3; ++a)
    ctx->weight[a] = dist_weight ? dist_weight[a] : 1.0f;
  ctx->has_map = true;
  ctx->lik_dirty = true;
  ctx->cand_dirty = true;
  ctx->dda_dirty = true;
  return 0;
}

int mcl3dl_hip_set_likelihood_params(mcl3dl_hip_ctx* ctx, float match_dist_min, float match_dist_flat,
                                     float match_weight)
{
  if (!ctx)
    return -1;
  ++ctx->generation;
  if (!(match_dist_min > 0.f))
    return ctx->fail(-3, "match_dist_min must be > 0");
  if (match_dist_min != ctx->match_dist_min)
    ctx->lik_dirty = ctx->cand_dirty = true;  // cell / voxel edges follow the search radius
  ctx->match_dist_min = match_dist_min;
  ctx->match_dist_flat = match_dist_flat;
  ctx->match_weight = match_weight;
  return 0;
}

int mcl3dl_hip_set_beam_params(mcl3dl_hip_ctx* ctx, float map_grid_x, float map_grid_y, float map_grid_z,
                               float dda_grid_size, float ray_angle_half, float hit_range, float beam_likelihood_min,
                               uint32_t num_points, float ang_total_ref, uint32_t filter_label_max,
                               int add_penalty_short_only_mode)
{
  if (!ctx)
    return -1;
  ++ctx->generation;
  if (!(dda_grid_size > 0.f))
    return ctx->fail(-3, "dda_grid_size must be > 0");
  ctx->map_grid[0] = map_grid_x;
  ctx->map_grid[1] = map_grid_y;
  ctx->map_grid[2] = map_grid_z;
  ctx->dda_grid_size = dda_grid_size;
  ctx->ray_angle_half = ray_angle_half;
  ctx->hit_range = hit_range;
  ctx->beam_likelihood_min = beam_likelihood_min;
  ctx->beam_num_points = num_points;
  ctx->ang_total_ref = ang_total_ref;
  ctx->filter_label_max = filter_label_max;
  ctx->short_only = add_penalty_short_only_mode ? 1 : 0;
  ctx->dda_dirty = true;  // refreshParameters re-creates the raycaster (beam.cpp:69-79)
  beam_refresh(ctx);
  return 0;
}

// sync_at_end = false: the caller synchronises the stream itself before it returns (the staging vectors live in the
// context, so nothing here dies earlier).
static int upload_scan_impl(mcl3dl_hip_ctx* ctx, const float* scan_lik_xyz, size_t n_s, const float* scan_beam_xyz,
                            const uint32_t* scan_beam_origin, size_t n_b, const float* origins, size_t n_o,
                            bool sync_at_end)
{
  if (!ctx)
    return -1;
  if ((n_s && !scan_lik_xyz) || (n_b && (!scan_beam_xyz || !origins || n_o == 0)))
    return ctx->fail(-3, "null scan array");
  if (n_s > 0x7fffffffu || n_b > 0x7fffffffu)
    return ctx->fail(-3, "scan too large");
  HIP_TRY(hipSetDevice(ctx->device));
  // likelihood scan: spatial (Morton) order. The score is a sum, so the order only changes which lanes work together.
  std::vector<float4>& lik = ctx->h_scan_lik;
  lik.resize(n_s);
  if (n_s)
  {
    float mn[3] = { scan_lik_xyz[0], scan_lik_xyz[1], scan_lik_xyz[2] };
    for (size_t i = 0; i < n_s; ++i)
      for (int a = 0; a < 3; ++a)
        mn[a] = std::min(mn[a], scan_lik_xyz[3 * i + a]);
    // 30-bit Morton key (10 bits per axis, 0.25 m cells) + 3-pass LSD radix sort: ~0.1 ms for 16 k points on one core
    std::vector<uint32_t>& idx = ctx->h_scan_perm;
    std::vector<uint32_t> key(n_s), key2(n_s), idx2(n_s);
    idx.resize(n_s);
    for (size_t i = 0; i < n_s; ++i)
    {
      uint32_t c[3];
      for (int a = 0; a < 3; ++a)
      {
        const float f = (scan_lik_xyz[3 * i + a] - mn[a]) * 4.0f;
        c[a] = (f >= 0.f) ? (f < 1023.f ? static_cast<uint32_t>(f) : 1023u) : 0u;
      }
      key[i] = static_cast<uint32_t>(morton3(c[0], c[1], c[2]));
      idx[i] = static_cast<uint32_t>(i);
    }
    for (int pass = 0; pass < 3; ++pass)
    {
      uint32_t hist[1025] = { 0 };
      const int shift = 10 * pass;
      for (size_t i = 0; i < n_s; ++i)
        ++hist[((key[i] >> shift) & 1023u) + 1];
      for (int b = 0; b < 1024; ++b)
        hist[b + 1] += hist[b];
      for (size_t i = 0; i < n_s; ++i)
      {
        const uint32_t dst = hist[(key[i] >> shift) & 1023u]++;
        key2[dst] = key[i];
        idx2[dst] = idx[i];
      }
      key.swap(key2);
      idx.swap(idx2);
    }
    for (size_t k = 0; k < n_s; ++k)
    {
      const uint32_t i = idx[k];
      lik[k] = make_float4(scan_lik_xyz[3 * i], scan_lik_xyz[3 * i + 1], scan_lik_xyz[3 * i + 2], 0.f);
    }
    TRY(ensure(ctx, ctx->scan_perm, sizeof(uint32_t) * n_s));
    TRY(h2d(ctx, ctx->scan_perm.p, idx.data(), sizeof(uint32_t) * n_s));
  }
  // beam scan: ordered by range from its scan origin. A ray walks ~range/dda_grid voxels and (its end point being a
  // measured surface) ends near its last voxel, so the 64 rays of a wavefront finish together instead of idling behind the
  // longest one. The beam score is a count of penalised rays, so the order is free.
  std::vector<float4>& beam = ctx->h_scan_beam;
  beam.resize(n_b);
  if (n_b)
  {
    std::vector<std::pair<float, uint32_t>> keys(n_b);
    for (size_t i = 0; i < n_b; ++i)
    {
      const uint32_t og = scan_beam_origin ? scan_beam_origin[i] : 0u;
      if (og >= n_o)
        return ctx->fail(-3, "beam point %zu names origin %u but only %zu origins were given", i, og, n_o);
      const float dx = scan_beam_xyz[3 * i] - origins[3 * og], dy = scan_beam_xyz[3 * i + 1] - origins[3 * og + 1],
                  dz = scan_beam_xyz[3 * i + 2] - origins[3 * og + 2];
      keys[i] = { dx * dx + dy * dy + dz * dz, static_cast<uint32_t>(i) };
    }
    std::sort(keys.begin(), keys.end());
    for (size_t k = 0; k < n_b; ++k)
    {
      const uint32_t i = keys[k].second;
      const uint32_t og = scan_beam_origin ? scan_beam_origin[i] : 0u;
      beam[k] = make_float4(scan_beam_xyz[3 * i], scan_beam_xyz[3 * i + 1], scan_beam_xyz[3 * i + 2], bits_to_float(og));
    }
  }
  std::vector<float4>& org = ctx->h_origins;
  org.resize(n_o);
  for (size_t i = 0; i < n_o; ++i)
    org[i] = make_float4(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2], 0.f);
  TRY(ensure(ctx, ctx->scan_lik, sizeof(float4) * n_s));
  TRY(ensure(ctx, ctx->scan_beam, sizeof(float4) * n_b));
  TRY(ensure(ctx, ctx->origins, sizeof(float4) * n_o));
  TRY(h2d(ctx, ctx->scan_lik.p, lik.data(), sizeof(float4) * n_s));
  TRY(h2d(ctx, ctx->scan_beam.p, beam.data(), sizeof(float4) * n_b));
  TRY(h2d(ctx, ctx->origins.p, org.data(), sizeof(float4) * n_o));
  if (sync_at_end)
    TRY(sync_stream(ctx));
  if (n_b != ctx->n_b)
    ctx->pow_table_dirty = true;
  if (n_s != ctx->n_s || n_b != ctx->n_b || n_o != ctx->n_o || !ctx->has_scan)
    ++ctx->generation;
  ctx->n_s = n_s;
  ctx->n_b = n_b;
  ctx->n_o = n_o;
  ctx->has_scan = true;
  return 0;
}

int mcl3dl_hip_upload_scan(mcl3dl_hip_ctx* ctx, const float* scan_lik_xyz, size_t n_s, const float* scan_beam_xyz,
                           const uint32_t* scan_beam_origin, size_t n_b, const float* origins, size_t n_o)
{
  return upload_scan_impl(ctx, scan_lik_xyz, n_s, scan_beam_xyz, scan_beam_origin, n_b, origins, n_o, true);
}

int mcl3dl_hip_measure_device(mcl3dl_hip_ctx* ctx, const float* d_pose, size_t n_p, float* d_lik, float* d_match_ratio,
                              float* d_beam)
{
  if (!ctx)
    return -1;
  HIP_TRY(hipSetDevice(ctx->device));
  return launch_measure(ctx, d_pose, n_p, d_lik, d_match_ratio, d_beam, false, nullptr);
}

int mcl3dl_hip_workload_stats(mcl3dl_hip_ctx* ctx, const float* d_pose, size_t n_p, double* stats6)
{
  if (!ctx || !stats6)
    return -1;
  HIP_TRY(hipSetDevice(ctx->device));
  return launch_measure(ctx, d_pose, n_p, nullptr, nullptr, nullptr, true, stats6);
}

int mcl3dl_hip_pf_partial_device(mcl3dl_hip_ctx* ctx, const float* d_weight, const float* d_lik, const float* d_beam,
                                 const float* d_extra, const float* d_match_ratio, size_t n_p, int rank, int world,
                                 double* d_packed)
{
  if (!ctx)
    return -1;
  if (n_p == 0 || n_p > 0x7fffffffu)
    return ctx->fail(-3, "bad particle count");
  if (world < 1 || rank < 0 || rank >= world || world > 4096)
    return ctx->fail(-3, "bad rank/world (%d/%d)", rank, world);
  HIP_TRY(hipSetDevice(ctx->device));
  const int nb = pf_blocks(n_p);
  TRY(ensure(ctx, ctx->wnew, sizeof(float) * n_p));
  TRY(ensure(ctx, ctx->block_partials, sizeof(double) * 4 * nb));
  EventPair ep{};
  TRY(timing_begin(ctx, MCL3DL_KERNEL_PF, &ep));
  hipLaunchKernelGGL(pf_partial_kernel, dim3(nb), dim3(PF_BLOCK), 0, ctx->stream, d_weight, d_lik, d_beam, d_extra,
                     d_match_ratio, static_cast<int>(n_p), ctx->wnew.as<float>(), ctx->block_partials.as<double>());
  hipLaunchKernelGGL(pf_reduce_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->block_partials.as<double>(), nb, rank,
                     world, d_packed);
  if (ctx->strict_order && world == 1)
    hipLaunchKernelGGL(pf_strict_sum_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->wnew.as<float>(),
                       static_cast<int>(n_p), d_packed);
  TRY(timing_end(ctx, ep));
  HIP_TRY(hipGetLastError());
  return 0;
}

int mcl3dl_hip_pf_apply_device(mcl3dl_hip_ctx* ctx, float* d_weight_inout, size_t n_p, int world,
                               const double* d_packed, float* d_stats4)
{
  if (!ctx)
    return -1;
  if (n_p == 0 || n_p > 0x7fffffffu)
    return ctx->fail(-3, "bad particle count");
  if (world < 1 || world > 4096)
    return ctx->fail(-3, "bad world size %d", world);
  HIP_TRY(hipSetDevice(ctx->device));
  EventPair ep{};
  TRY(timing_begin(ctx, MCL3DL_KERNEL_PF, &ep));
  hipLaunchKernelGGL(pf_apply_kernel, dim3(pf_blocks(n_p)), dim3(PF_BLOCK), 0, ctx->stream, d_weight_inout,
                     ctx->wnew.as<float>(), static_cast<int>(n_p), world, d_packed, d_stats4);
  TRY(timing_end(ctx, ep));
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- host entry points -------------------------------------------------------------------------------------
namespace
{
int enqueue_update(mcl3dl_hip_ctx* ctx, const float* d_pose, size_t n_p, float* d_weight, const float* d_extra,
                   float* d_lik, float* d_ratio, float* d_beam, float* d_stats4)
{
  TRY(launch_measure(ctx, d_pose, n_p, d_lik, d_ratio, d_beam, false, nullptr));
  TRY(mcl3dl_hip_pf_partial_device(ctx, d_weight, d_lik, d_beam, d_extra, d_ratio, n_p, 0, 1, ctx->partial4.as<double>()));
  TRY(mcl3dl_hip_pf_apply_device(ctx, d_weight, n_p, 1, ctx->partial4.as<double>(), d_stats4));
  return 0;
}

void drop_graph(mcl3dl_hip_ctx* ctx)
{
  if (ctx->graph_exec)
    (void)hipGraphExecDestroy(ctx->graph_exec);
  if (ctx->graph)
    (void)hipGraphDestroy(ctx->graph);
  ctx->graph_exec = nullptr;
  ctx->graph = nullptr;
}
}  // namespace

int mcl3dl_hip_update_device(mcl3dl_hip_ctx* ctx, const float* d_pose, size_t n_p, float* d_weight_inout,
                             const float* d_extra, float* d_lik, float* d_match_ratio, float* d_beam, float* d_stats4)
{
  if (!ctx)
    return -1;
  if (n_p == 0 || n_p > 0x7fffffffu)
    return ctx->fail(-3, "bad particle count");
  if (!d_pose || !d_weight_inout || !d_stats4)
    return ctx->fail(-3, "null pose / weight / stats array");
  if (!ctx->has_scan)
    return ctx->fail(-5, "no scan uploaded: call mcl3dl_hip_upload_scan first");
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t fb = sizeof(float) * n_p;
  if (!d_lik)
  {
    TRY(ensure(ctx, ctx->lik, fb));
    d_lik = ctx->lik.as<float>();
  }
  if (!d_match_ratio)
  {
    TRY(ensure(ctx, ctx->ratio, fb));
    d_match_ratio = ctx->ratio.as<float>();
  }
  if (!d_beam)
  {
    TRY(ensure(ctx, ctx->beam, fb));
    d_beam = ctx->beam.as<float>();
  }
  TRY(ensure(ctx, ctx->partial4, sizeof(double) * 4));
  mcl3dl_hip_ctx::UpdateKey key{};
  key.p[0] = d_pose;
  key.p[1] = d_weight_inout;
  key.p[2] = d_extra;
  key.p[3] = d_lik;
  key.p[4] = d_match_ratio;
  key.p[5] = d_beam;
  key.p[6] = d_stats4;
  key.p[7] = ctx->stream;
  key.n_p = n_p;
  key.generation = ctx->generation;
  const bool graphs = ctx->use_graph && !ctx->timing;
  if (graphs && ctx->graph_exec && ctx->graph_key == key)
  {
    HIP_TRY(hipGraphLaunch(ctx->graph_exec, ctx->stream));
    ++ctx->graph_replays;
    return 0;
  }
  // First sighting of these arguments: run eagerly (this is also what builds the map structures and sizes every work
  // buffer). Second sighting: nothing is left to build or allocate, so the same calls can be captured.
  // (the structure checks mirror ensure_structures / launch_measure: whatever this update needs must already exist)
  const bool need_lik = ctx->n_s > 0, need_dda = ctx->n_b > 0;
  const bool built = ctx->has_map && !(need_lik && ctx->lik_index == 0 && ctx->lik_dirty) &&
                     !(need_lik && ctx->lik_index >= 1 && ctx->cand_dirty) &&
                     !(need_dda && (ctx->dda_dirty || ctx->pow_table_dirty));
  const bool capture = graphs && built && ctx->have_seen && ctx->seen_key == key &&
                       !(ctx->have_failed && ctx->failed_key == key);
  if (!capture)
  {
    TRY(enqueue_update(ctx, d_pose, n_p, d_weight_inout, d_extra, d_lik, d_match_ratio, d_beam, d_stats4));
    // enqueue_update may itself have moved the generation on (first-use allocations): remember the state it left
    key.generation = ctx->generation;
    ctx->seen_key = key;
    ctx->have_seen = true;
    return 0;
  }
  drop_graph(ctx);
  HIP_TRY(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
  const int rc = enqueue_update(ctx, d_pose, n_p, d_weight_inout, d_extra, d_lik, d_match_ratio, d_beam, d_stats4);
  hipGraph_t g = nullptr;
  const hipError_t e_end = hipStreamEndCapture(ctx->stream, &g);
  bool ok = rc == 0 && e_end == hipSuccess && g != nullptr && ctx->generation == key.generation;
  if (ok)
  {
    ctx->graph = g;
    ok = hipGraphInstantiate(&ctx->graph_exec, g, nullptr, nullptr, 0) == hipSuccess;
  }
  else if (g)
    (void)hipGraphDestroy(g);
  if (!ok)
  {
    // not capturable in this state (e.g. a buffer had to grow): forget it and run the plain sequence
    const hipError_t e_last = hipGetLastError();
    char why[256];
    snprintf(why, sizeof(why), "update graph not captured: rc=%d end=%s last=%s graph=%p generation %llu -> %llu", rc,
             hipGetErrorString(e_end), hipGetErrorString(e_last), static_cast<void*>(g),
             static_cast<unsigned long long>(key.generation), static_cast<unsigned long long>(ctx->generation));
    ctx->graph_note = why;
    drop_graph(ctx);
    ctx->failed_key = key;
    ctx->have_failed = true;
    return enqueue_update(ctx, d_pose, n_p, d_weight_inout, d_extra, d_lik, d_match_ratio, d_beam, d_stats4);
  }
  ctx->graph_key = key;
  ++ctx->graph_captures;
  HIP_TRY(hipGraphLaunch(ctx->graph_exec, ctx->stream));
  ++ctx->graph_replays;
  return 0;
}

const char* mcl3dl_hip_graph_note(const mcl3dl_hip_ctx* ctx)
{
  return ctx ? ctx->graph_note.c_str() : "";
}

int mcl3dl_hip_graph_stats(mcl3dl_hip_ctx* ctx, uint64_t* captures, uint64_t* replays)
{
  if (!ctx)
    return -1;
  if (captures)
    *captures = ctx->graph_captures;
  if (replays)
    *replays = ctx->graph_replays;
  return 0;
}

int mcl3dl_hip_measure_batch(mcl3dl_hip_ctx* ctx, const float* pose, size_t n_p, const float* scan_lik_xyz, size_t n_s,
                             const float* scan_beam_xyz, const uint32_t* scan_beam_origin, size_t n_b,
                             const float* origins, size_t n_o, float* out_lik, float* out_match_ratio, float* out_beam)
{
  if (!ctx)
    return -1;
  if (n_p == 0)
    return 0;
  if (!pose)
    return ctx->fail(-3, "null pose array");
  HIP_TRY(hipSetDevice(ctx->device));
  TRY(upload_scan_impl(ctx, scan_lik_xyz, n_s, scan_beam_xyz, scan_beam_origin, n_b, origins, n_o, false));
  TRY(ensure(ctx, ctx->pose, sizeof(float) * 7 * n_p));
  TRY(ensure(ctx, ctx->lik, sizeof(float) * n_p));
  TRY(ensure(ctx, ctx->ratio, sizeof(float) * n_p));
  TRY(ensure(ctx, ctx->beam, sizeof(float) * n_p));
  TRY(h2d(ctx, ctx->pose.p, pose, sizeof(float) * 7 * n_p));
  const bool lik_wanted = out_lik || out_match_ratio;
  TRY(launch_measure(ctx, ctx->pose.as<float>(), n_p, lik_wanted ? ctx->lik.as<float>() : nullptr,
                     lik_wanted ? ctx->ratio.as<float>() : nullptr, out_beam ? ctx->beam.as<float>() : nullptr, false,
                     nullptr));
  if (out_lik)
    TRY(d2h(ctx, out_lik, ctx->lik.p, sizeof(float) * n_p));
  if (out_match_ratio)
    TRY(d2h(ctx, out_match_ratio, ctx->ratio.p, sizeof(float) * n_p));
  if (out_beam)
    TRY(d2h(ctx, out_beam, ctx->beam.p, sizeof(float) * n_p));
  TRY(sync_stream(ctx));
  return 0;
}

int mcl3dl_hip_pf_measure(mcl3dl_hip_ctx* ctx, float* weight_inout, const float* lik, const float* beam,
                          const float* extra, const float* match_ratio, size_t n_p, float* entropy,
                          float* match_ratio_min, float* match_ratio_max, int* restored)
{
  if (!ctx)
    return -1;
  if (n_p == 0)
    return ctx->fail(-3, "no particles");
  if (!weight_inout || !lik)
    return ctx->fail(-3, "null weight / likelihood array");
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t fb = sizeof(float) * n_p;
  TRY(ensure(ctx, ctx->weightb, fb));
  TRY(ensure(ctx, ctx->lik, fb));
  TRY(ensure(ctx, ctx->beam, fb));
  TRY(ensure(ctx, ctx->extra, fb));
  TRY(ensure(ctx, ctx->ratio, fb));
  TRY(ensure(ctx, ctx->partial4, sizeof(double) * 4));
  TRY(ensure(ctx, ctx->stats4, sizeof(float) * 4));
  TRY(h2d(ctx, ctx->weightb.p, weight_inout, fb));
  TRY(h2d(ctx, ctx->lik.p, lik, fb));
  if (beam)
    TRY(h2d(ctx, ctx->beam.p, beam, fb));
  if (extra)
    TRY(h2d(ctx, ctx->extra.p, extra, fb));
  if (match_ratio)
    TRY(h2d(ctx, ctx->ratio.p, match_ratio, fb));
  TRY(mcl3dl_hip_pf_partial_device(ctx, ctx->weightb.as<float>(), ctx->lik.as<float>(),
                                   beam ? ctx->beam.as<float>() : nullptr, extra ? ctx->extra.as<float>() : nullptr,
                                   match_ratio ? ctx->ratio.as<float>() : nullptr, n_p, 0, 1, ctx->partial4.as<double>()));
  TRY(mcl3dl_hip_pf_apply_device(ctx, ctx->weightb.as<float>(), n_p, 1, ctx->partial4.as<double>(),
                                 ctx->stats4.as<float>()));
  float st[4];
  TRY(d2h(ctx, weight_inout, ctx->weightb.p, fb));
  TRY(d2h(ctx, st, ctx->stats4.p, sizeof(st)));
  TRY(sync_stream(ctx));
  if (entropy)
    *entropy = st[0];
  if (match_ratio_min)
    *match_ratio_min = st[1];
  if (match_ratio_max)
    *match_ratio_max = st[2];
  if (restored)
    *restored = st[3] != 0.0f;
  return 0;
}

int mcl3dl_hip_measure_update(mcl3dl_hip_ctx* ctx, const float* pose, const float* extra, float* weight_inout,
                              size_t n_p, const float* scan_lik_xyz, size_t n_s, const float* scan_beam_xyz,
                              const uint32_t* scan_beam_origin, size_t n_b, const float* origins, size_t n_o,
                              float* out_lik, float* out_match_ratio, float* out_beam, float* entropy,
                              float* match_ratio_min, float* match_ratio_max, int* restored)
{
  if (!ctx)
    return -1;
  if (n_p == 0)
    return ctx->fail(-3, "no particles");
  if (!pose || !weight_inout)
    return ctx->fail(-3, "null pose / weight array");
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t fb = sizeof(float) * n_p;
  TRY(upload_scan_impl(ctx, scan_lik_xyz, n_s, scan_beam_xyz, scan_beam_origin, n_b, origins, n_o, false));
  TRY(ensure(ctx, ctx->pose, sizeof(float) * 7 * n_p));
  TRY(ensure(ctx, ctx->weightb, fb));
  TRY(ensure(ctx, ctx->lik, fb));
  TRY(ensure(ctx, ctx->ratio, fb));
  TRY(ensure(ctx, ctx->beam, fb));
  TRY(ensure(ctx, ctx->extra, fb));
  TRY(ensure(ctx, ctx->partial4, sizeof(double) * 4));
  TRY(ensure(ctx, ctx->stats4, sizeof(float) * 4));
  TRY(h2d(ctx, ctx->pose.p, pose, sizeof(float) * 7 * n_p));
  TRY(h2d(ctx, ctx->weightb.p, weight_inout, fb));
  if (extra)
    TRY(h2d(ctx, ctx->extra.p, extra, fb));
  TRY(launch_measure(ctx, ctx->pose.as<float>(), n_p, ctx->lik.as<float>(), ctx->ratio.as<float>(),
                     ctx->beam.as<float>(), false, nullptr));
  TRY(mcl3dl_hip_pf_partial_device(ctx, ctx->weightb.as<float>(), ctx->lik.as<float>(), ctx->beam.as<float>(),
                                   extra ? ctx->extra.as<float>() : nullptr, ctx->ratio.as<float>(), n_p, 0, 1,
                                   ctx->partial4.as<double>()));
  TRY(mcl3dl_hip_pf_apply_device(ctx, ctx->weightb.as<float>(), n_p, 1, ctx->partial4.as<double>(),
                                 ctx->stats4.as<float>()));
  float st[4];
  TRY(d2h(ctx, weight_inout, ctx->weightb.p, fb));
  TRY(d2h(ctx, st, ctx->stats4.p, sizeof(st)));
  if (out_lik)
    TRY(d2h(ctx, out_lik, ctx->lik.p, fb));
  if (out_match_ratio)
    TRY(d2h(ctx, out_match_ratio, ctx->ratio.p, fb));
  if (out_beam)
    TRY(d2h(ctx, out_beam, ctx->beam.p, fb));
  TRY(sync_stream(ctx));
  if (entropy)
    *entropy = st[0];
  if (match_ratio_min)
    *match_ratio_min = st[1];
  if (match_ratio_max)
    *match_ratio_max = st[2];
  if (restored)
    *restored = st[3] != 0.0f;
  return 0;
}

int mcl3dl_hip_beam_status(mcl3dl_hip_ctx* ctx, const float* begin_xyz, const float* end_xyz, size_t n, int32_t* status,
                           int32_t* hit_index)
{
  if (!ctx)
    return -1;
  if (n == 0)
    return 0;
  if (!begin_xyz || !end_xyz || !status)
    return ctx->fail(-3, "null ray array");
  if (n > 0x7fffffffu)
    return ctx->fail(-3, "too many rays");
  HIP_TRY(hipSetDevice(ctx->device));
  TRY(ensure_structures(ctx, false, true));
  TRY(ensure(ctx, ctx->ray_begin, sizeof(float) * 3 * n));
  TRY(ensure(ctx, ctx->ray_end, sizeof(float) * 3 * n));
  TRY(ensure(ctx, ctx->ray_status, sizeof(int) * n));
  TRY(ensure(ctx, ctx->ray_hit, sizeof(int) * n));
  TRY(h2d(ctx, ctx->ray_begin.p, begin_xyz, sizeof(float) * 3 * n));
  TRY(h2d(ctx, ctx->ray_end.p, end_xyz, sizeof(float) * 3 * n));
  const int ni = static_cast<int>(n);
  hipLaunchKernelGGL(beam_status_kernel, dim3((ni + 63) / 64), dim3(64), 0, ctx->stream, ctx->ray_begin.as<float>(),
                     ctx->ray_end.as<float>(), ni, ctx->dg, beam_params(ctx), ctx->ray_status.as<int>(),
                     ctx->ray_hit.as<int>());
  HIP_TRY(hipGetLastError());
  TRY(d2h(ctx, status, ctx->ray_status.p, sizeof(int) * n));
  if (hit_index)
    TRY(d2h(ctx, hit_index, ctx->ray_hit.p, sizeof(int) * n));
  TRY(sync_stream(ctx));
  return 0;
}

int mcl3dl_hip_dda_trace(mcl3dl_hip_ctx* ctx, const float* begin3, const float* end3, float* out_xyz, int max_out,
                         int* n_visited, int* collided, int* hit_index)
{
  if (!ctx)
    return -1;
  if (!begin3 || !end3 || max_out < 0 || (max_out > 0 && !out_xyz))
    return ctx->fail(-3, "bad trace arguments");
  HIP_TRY(hipSetDevice(ctx->device));
  TRY(ensure_structures(ctx, false, true));
  TRY(ensure(ctx, ctx->ray_begin, sizeof(float) * 3 * static_cast<size_t>(max_out)));
  TRY(ensure(ctx, ctx->ray_status, sizeof(int) * 3));
  hipLaunchKernelGGL(dda_trace_kernel, dim3(1), dim3(64), 0, ctx->stream, Vec3f{ begin3[0], begin3[1], begin3[2] },
                     Vec3f{ end3[0], end3[1], end3[2] }, ctx->dg, beam_params(ctx), ctx->ray_begin.as<float>(), max_out,
                     ctx->ray_status.as<int>());
  HIP_TRY(hipGetLastError());
  int out3[3] = { 0, 0, -1 };
  TRY(d2h(ctx, out3, ctx->ray_status.p, sizeof(out3)));
  TRY(sync_stream(ctx));
  const int n_copy = std::min(out3[0], max_out);
  if (n_copy > 0)
  {
    TRY(d2h(ctx, out_xyz, ctx->ray_begin.p, sizeof(float) * 3 * static_cast<size_t>(n_copy)));
    TRY(sync_stream(ctx));
  }
  if (n_visited)
    *n_visited = out3[0];
  if (collided)
    *collided = out3[1];
  if (hit_index)
    *hit_index = out3[2];
  return 0;
}

// ---- R4 as a stand-alone query: ChunkedKdtree::radiusSearch -------------------------------------------------------------
int mcl3dl_hip_radius_search(mcl3dl_hip_ctx* ctx, const float* query_xyz, size_t n, float radius, int32_t* out_index,
                             float* out_sqdist)
{
  if (!ctx)
    return -1;
  if (n == 0)
    return 0;
  if (!query_xyz || !out_index || n > 0x7fffffffu || !(radius > 0.f))
    return ctx->fail(-3, "bad arguments to radius_search");
  HIP_TRY(hipSetDevice(ctx->device));
  TRY(ensure_structures(ctx, true, false, true));  // the cell-sorted map
  const float cell = 1.0f / ctx->lg.inv_cell;
  const int reach = static_cast<int>(std::ceil(radius / cell)) + 1;
  if (reach > 64)
    return ctx->fail(-3, "radius %.3g is more than 64 cells of the map index", radius);
  TRY(ensure(ctx, ctx->ray_begin, sizeof(float) * 3 * n));
  TRY(ensure(ctx, ctx->ray_hit, sizeof(int) * n));
  TRY(ensure(ctx, ctx->ray_end, sizeof(float) * n));
  TRY(h2d(ctx, ctx->ray_begin.p, query_xyz, sizeof(float) * 3 * n));
  const LikParams lp = lik_params(ctx);
  const float r2 = static_cast<float>(static_cast<double>(radius) * static_cast<double>(radius));
  const int ni = static_cast<int>(n);
  hipLaunchKernelGGL(radius_search_kernel, dim3((ni + 63) / 64), dim3(64), 0, ctx->stream, ctx->ray_begin.as<float>(), ni,
                     ctx->lg, lp, radius, r2, reach, ctx->ray_hit.as<int>(), ctx->ray_end.as<float>());
  HIP_TRY(hipGetLastError());
  TRY(d2h(ctx, out_index, ctx->ray_hit.p, sizeof(int) * n));
  if (out_sqdist)
    TRY(d2h(ctx, out_sqdist, ctx->ray_end.p, sizeof(float) * n));
  TRY(sync_stream(ctx));
  return 0;
}

// ---- "next" row: expectation / max / covariance ----------------------------------------------------------------------
// Quat(const Vec3& forward, const Vec3& up_raw), include/mcl_3dl/quat.h:61-80 (host, float with double square roots)
static Quat quat_from_front_up(Vec3f forward, Vec3f up_raw)
{
  auto normalized = [](Vec3f a)
  {
    const float n = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    return Vec3f{ a.x / n, a.y / n, a.z / n };
  };
  auto cross = [](Vec3f a, Vec3f q)
  { return Vec3f{ a.y * q.z - a.z * q.y, a.z * q.x - a.x * q.z, a.x * q.y - a.y * q.x }; };
  const Vec3f xv = normalized(forward);
  const Vec3f yv = normalized(cross(up_raw, xv));
  const Vec3f zv = normalized(cross(xv, yv));
  Quat q;
  q.w = static_cast<float>(std::sqrt(std::max(0.0, 1.0 + xv.x + yv.y + zv.z)) / 2.0);
  q.x = static_cast<float>(std::sqrt(std::max(0.0, 1.0 + xv.x - yv.y - zv.z)) / 2.0);
  q.y = static_cast<float>(std::sqrt(std::max(0.0, 1.0 - xv.x + yv.y - zv.z)) / 2.0);
  q.z = static_cast<float>(std::sqrt(std::max(0.0, 1.0 - xv.x - yv.y + zv.z)) / 2.0);
  if (zv.y - yv.z > 0)
    q.x = -q.x;
  if (xv.z - zv.x > 0)
    q.y = -q.y;
  if (yv.x - xv.y > 0)
    q.z = -q.z;
  return q;
}

int mcl3dl_hip_expectation_device(mcl3dl_hip_ctx* ctx, const float* d_pose, const float* d_weight, const float* d_bias,
                                  size_t n, float* out_mean7, float* out_total, int32_t* out_max_index,
                                  int32_t* out_max_biased_index)
{
  if (!ctx)
    return -1;
  if (n == 0 || n > 0x7fffffffu || !d_pose || !d_weight)
    return ctx->fail(-3, "bad arguments to expectation");
  HIP_TRY(hipSetDevice(ctx->device));
  const int nb = pf_blocks(n);
  TRY(ensure(ctx, ctx->mom_blocks, sizeof(double) * MOM_N * nb));
  TRY(ensure(ctx, ctx->mom_arg, sizeof(ArgMax) * 2 * nb));
  TRY(ensure(ctx, ctx->mom_out, sizeof(double) * COV_N));
  TRY(ensure(ctx, ctx->mom_idx, sizeof(int) * 2));
  hipLaunchKernelGGL(pf_moments_kernel, dim3(nb), dim3(PF_BLOCK), 0, ctx->stream, d_pose, d_weight, d_bias,
                     static_cast<int>(n), ctx->mom_blocks.as<double>(), ctx->mom_arg.as<ArgMax>());
  hipLaunchKernelGGL(pf_moments_reduce_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->mom_blocks.as<double>(),
                     ctx->mom_arg.as<ArgMax>(), nb, ctx->mom_out.as<double>(), ctx->mom_idx.as<int>(),
                     static_cast<double*>(nullptr));
  HIP_TRY(hipGetLastError());
  double m[MOM_N];
  int arg[2];
  TRY(d2h(ctx, m, ctx->mom_out.p, sizeof(m)));
  TRY(d2h(ctx, arg, ctx->mom_idx.p, sizeof(arg)));
  TRY(sync_stream(ctx));
  // ParticleWeightedMeanQuat::getMean, state_6dof.h:345-350
  const float p_sum = static_cast<float>(m[0]);
  const Quat q = quat_from_front_up(Vec3f{ static_cast<float>(m[4]), static_cast<float>(m[5]), static_cast<float>(m[6]) },
                                    Vec3f{ static_cast<float>(m[7]), static_cast<float>(m[8]), static_cast<float>(m[9]) });
  if (out_mean7)
  {
    out_mean7[0] = static_cast<float>(m[1]) / p_sum;
    out_mean7[1] = static_cast<float>(m[2]) / p_sum;
    out_mean7[2] = static_cast<float>(m[3]) / p_sum;
    out_mean7[3] = q.x;
    out_mean7[4] = q.y;
    out_mean7[5] = q.z;
    out_mean7[6] = q.w;
  }
  if (out_total)
    *out_total = p_sum;
  if (out_max_index)
    *out_max_index = arg[0];
  if (out_max_biased_index)
    *out_max_biased_index = arg[1];
  return 0;
}

int mcl3dl_hip_covariance_device(mcl3dl_hip_ctx* ctx, const float* d_pose, const float* d_weight, size_t n_particles,
                                 const uint32_t* d_subset, size_t n_subset, const float* mean7, float* out_cov36)
{
  if (!ctx)
    return -1;
  const size_t n = d_subset ? n_subset : n_particles;
  if (n == 0 || n > 0x7fffffffu || !d_pose || !d_weight || !mean7 || !out_cov36)
    return ctx->fail(-3, "bad arguments to covariance");
  HIP_TRY(hipSetDevice(ctx->device));
  const int nb = pf_blocks(n);
  TRY(ensure(ctx, ctx->mom_blocks, sizeof(double) * COV_N * nb));
  TRY(ensure(ctx, ctx->mom_out, sizeof(double) * COV_N));
  const Vec3f exp_rpy = quat_get_rpy(Quat{ mean7[3], mean7[4], mean7[5], mean7[6] });  // host libm, like the reference
  hipLaunchKernelGGL(pf_covariance_kernel, dim3(nb), dim3(PF_BLOCK), 0, ctx->stream, d_pose, d_weight, d_subset,
                     static_cast<int>(n), mean7[0], mean7[1], mean7[2], exp_rpy, ctx->mom_blocks.as<double>());
  hipLaunchKernelGGL(pf_covariance_reduce_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->mom_blocks.as<double>(), nb,
                     ctx->mom_out.as<double>());
  HIP_TRY(hipGetLastError());
  double s[COV_N];
  TRY(d2h(ctx, s, ctx->mom_out.p, sizeof(s)));
  TRY(sync_stream(ctx));
  const float p_sum = static_cast<float>(s[21]);
  int idx = 0;
  for (int j = 0; j < 6; ++j)
    for (int k = j; k < 6; ++k)
    {
      const float v = static_cast<float>(s[idx++]) / p_sum;  // pf.h:351-357
      out_cov36[6 * j + k] = v;
      out_cov36[6 * k + j] = v;
    }
  return 0;
}

// ---- the same reductions over particle shards (one record per GPU, combined after an all-gather / all-reduce) -------
int mcl3dl_hip_moments_partial_device(mcl3dl_hip_ctx* ctx, const float* d_pose, const float* d_weight,
                                      const float* d_bias, size_t n, double* d_out16)
{
  if (!ctx)
    return -1;
  if (n == 0 || n > 0x7fffffffu || !d_pose || !d_weight || !d_out16)
    return ctx->fail(-3, "bad arguments to moments_partial");
  HIP_TRY(hipSetDevice(ctx->device));
  const int nb = pf_blocks(n);
  TRY(ensure(ctx, ctx->mom_blocks, sizeof(double) * MOM_N * nb));
  TRY(ensure(ctx, ctx->mom_arg, sizeof(ArgMax) * 2 * nb));
  TRY(ensure(ctx, ctx->mom_out, sizeof(double) * COV_N));
  TRY(ensure(ctx, ctx->mom_idx, sizeof(int) * 2));
  hipLaunchKernelGGL(pf_moments_kernel, dim3(nb), dim3(PF_BLOCK), 0, ctx->stream, d_pose, d_weight, d_bias,
                     static_cast<int>(n), ctx->mom_blocks.as<double>(), ctx->mom_arg.as<ArgMax>());
  hipLaunchKernelGGL(pf_moments_reduce_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->mom_blocks.as<double>(),
                     ctx->mom_arg.as<ArgMax>(), nb, ctx->mom_out.as<double>(), ctx->mom_idx.as<int>(), d_out16);
  HIP_TRY(hipGetLastError());
  return 0;
}

int mcl3dl_hip_moments_finish(const double* parts16, int world, const uint64_t* index_offset, float* out_mean7,
                              float* out_total, int64_t* out_max_index, int64_t* out_max_biased_index)
{
  if (!parts16 || world < 1)
    return -3;
  double m[MOM_N] = { 0 };
  for (int r = 0; r < world; ++r)  // rank order: deterministic
    for (int k = 0; k < MOM_N; ++k)
      m[k] += parts16[16 * r + k];
  // pf.h:361-390: the first particle holding the maximum wins (strict <), shards are in particle order
  float best[2] = { -1.0f, -1.0f };
  int64_t arg[2] = { 0, 0 };
  for (int r = 0; r < world; ++r)
    for (int w = 0; w < 2; ++w)
    {
      const float v = static_cast<float>(parts16[16 * r + MOM_N + 2 * w]);
      if (v > best[w])
      {
        best[w] = v;
        arg[w] = static_cast<int64_t>(parts16[16 * r + MOM_N + 2 * w + 1]) +
                 static_cast<int64_t>(index_offset ? index_offset[r] : 0);
      }
    }
  const float p_sum = static_cast<float>(m[0]);
  const Quat q = quat_from_front_up(Vec3f{ static_cast<float>(m[4]), static_cast<float>(m[5]), static_cast<float>(m[6]) },
                                    Vec3f{ static_cast<float>(m[7]), static_cast<float>(m[8]), static_cast<float>(m[9]) });
  if (out_mean7)
  {
    out_mean7[0] = static_cast<float>(m[1]) / p_sum;
    out_mean7[1] = static_cast<float>(m[2]) / p_sum;
    out_mean7[2] = static_cast<float>(m[3]) / p_sum;
    out_mean7[3] = q.x;
    out_mean7[4] = q.y;
    out_mean7[5] = q.z;
    out_mean7[6] = q.w;
  }
  if (out_total)
    *out_total = p_sum;
  if (out_max_index)
    *out_max_index = arg[0];
  if (out_max_biased_index)
    *out_max_biased_index = arg[1];
  return 0;
}

int mcl3dl_hip_covariance_partial_device(mcl3dl_hip_ctx* ctx, const float* d_pose, const float* d_weight,
                                         size_t n_particles, const uint32_t* d_subset, size_t n_subset,
                                         const float* mean7, double* d_out22)
{
  if (!ctx)
    return -1;
  const size_t n = d_subset ? n_subset : n_particles;
  if (n > 0x7fffffffu || !d_pose || !d_weight || !mean7 || !d_out22)
    return ctx->fail(-3, "bad arguments to covariance_partial");
  HIP_TRY(hipSetDevice(ctx->device));
  if (n == 0)
  {
    HIP_TRY(hipMemsetAsync(d_out22, 0, sizeof(double) * COV_N, ctx->stream));
    return 0;
  }
  const int nb = pf_blocks(n);
  TRY(ensure(ctx, ctx->mom_blocks, sizeof(double) * COV_N * nb));
  const Vec3f exp_rpy = quat_get_rpy(Quat{ mean7[3], mean7[4], mean7[5], mean7[6] });
  hipLaunchKernelGGL(pf_covariance_kernel, dim3(nb), dim3(PF_BLOCK), 0, ctx->stream, d_pose, d_weight, d_subset,
                     static_cast<int>(n), mean7[0], mean7[1], mean7[2], exp_rpy, ctx->mom_blocks.as<double>());
  hipLaunchKernelGGL(pf_covariance_reduce_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->mom_blocks.as<double>(), nb,
                     d_out22);
  HIP_TRY(hipGetLastError());
  return 0;
}

int mcl3dl_hip_covariance_finish(const double* sums22, float* out_cov36)
{
  if (!sums22 || !out_cov36)
    return -3;
  const float p_sum = static_cast<float>(sums22[21]);
  int idx = 0;
  for (int j = 0; j < 6; ++j)
    for (int k = j; k < 6; ++k)
    {
      const float v = static_cast<float>(sums22[idx++]) / p_sum;  // pf.h:351-357
      out_cov36[6 * j + k] = v;
      out_cov36[6 * k + j] = v;
    }
  return 0;
}

int mcl3dl_hip_expectation(mcl3dl_hip_ctx* ctx, const float* pose, const float* weight, const float* bias, size_t n,
                           float* out_mean7, float* out_total, int32_t* out_max_index, int32_t* out_max_biased_index)
{
  if (!ctx)
    return -1;
  if (n == 0 || !pose || !weight)
    return ctx->fail(-3, "bad arguments to expectation");
  HIP_TRY(hipSetDevice(ctx->device));
  TRY(ensure(ctx, ctx->pose, sizeof(float) * 7 * n));
  TRY(ensure(ctx, ctx->weightb, sizeof(float) * n));
  TRY(ensure(ctx, ctx->extra, sizeof(float) * n));
  TRY(h2d(ctx, ctx->pose.p, pose, sizeof(float) * 7 * n));
  TRY(h2d(ctx, ctx->weightb.p, weight, sizeof(float) * n));
  if (bias)
    TRY(h2d(ctx, ctx->extra.p, bias, sizeof(float) * n));
  return mcl3dl_hip_expectation_device(ctx, ctx->pose.as<float>(), ctx->weightb.as<float>(),
                                       bias ? ctx->extra.as<float>() : nullptr, n, out_mean7, out_total, out_max_index,
                                       out_max_biased_index);
}

int mcl3dl_hip_covariance(mcl3dl_hip_ctx* ctx, const float* pose, const float* weight, size_t n, const uint32_t* subset,
                          size_t n_subset, const float* mean7, float* out_cov36)
{
  if (!ctx)
    return -1;
  if (n == 0 || !pose || !weight)
    return ctx->fail(-3, "bad arguments to covariance");
  if (subset)
    for (size_t i = 0; i < n_subset; ++i)
      if (subset[i] >= n)
        return ctx->fail(-3, "subset index %u out of range", subset[i]);
  HIP_TRY(hipSetDevice(ctx->device));
  TRY(ensure(ctx, ctx->pose, sizeof(float) * 7 * n));
  TRY(ensure(ctx, ctx->weightb, sizeof(float) * n));
  TRY(h2d(ctx, ctx->pose.p, pose, sizeof(float) * 7 * n));
  TRY(h2d(ctx, ctx->weightb.p, weight, sizeof(float) * n));
  if (subset)
  {
    TRY(ensure(ctx, ctx->subset, sizeof(uint32_t) * n_subset));
    TRY(h2d(ctx, ctx->subset.p, subset, sizeof(uint32_t) * n_subset));
  }
  return mcl3dl_hip_covariance_device(ctx, ctx->pose.as<float>(), ctx->weightb.as<float>(), n,
                                      subset ? ctx->subset.as<uint32_t>() : nullptr, n_subset, mean7, out_cov36);
}

// ---- "next" row: resample / resizeParticle ---------------------------------------------------------------------------
int mcl3dl_hip_resample_begin(mcl3dl_hip_ctx* ctx, const float* weight, size_t n, size_t n_out, float* out_pstep)
{
  if (!ctx)
    return -1;
  if (!weight || n == 0 || n_out == 0 || n > 0x7fffffffu || n_out > 0x7fffffffu)
    return ctx->fail(-3, "bad arguments to resample_begin");
  // accum += p.probability_ ; p.accum_probability_ = accum   (pf.h:193-197 / 401-405): a float recurrence in particle
  // order, so it runs on the host (one add per particle)
  ctx->rs_keys.resize(n);
  float accum = 0;
  bool ties = false;
  for (size_t i = 0; i < n; ++i)
  {
    const float prev = accum;
    accum += weight[i];
    ties = ties || (i > 0 && !(prev < accum));
    ctx->rs_keys[i] = accum;
  }
  // std::sort(particles_dup_) (pf.h:200 / 408). Ascending and tie-free input is left as it is by any sort; with ties
  // (weight-0 particles) libstdc++'s introsort decides who leads each tie group, so the very same std::sort runs here
  // (the comparison looks at the accumulated probability only, like Particle::operator<, pf.h:104-107).
  ctx->rs_sorted = ties;
  if (ties)
  {
    std::vector<std::pair<float, uint32_t>> dup(n);
    for (size_t i = 0; i < n; ++i)
      dup[i] = { ctx->rs_keys[i], static_cast<uint32_t>(i) };
    std::sort(dup.begin(), dup.end(),
              [](const std::pair<float, uint32_t>& a, const std::pair<float, uint32_t>& b) { return a.first < b.first; });
    ctx->rs_order.resize(n);
    for (size_t i = 0; i < n; ++i)
    {
      ctx->rs_keys[i] = dup[i].first;
      ctx->rs_order[i] = dup[i].second;
    }
  }
  ctx->rs_n = n;
  ctx->rs_n_out = n_out;
  ctx->rs_pstep = accum / n_out;  // pf.h:202 / 410 (float / size_t)
  ctx->rs_planned = false;
  if (out_pstep)
    *out_pstep = ctx->rs_pstep;
  HIP_TRY(hipSetDevice(ctx->device));
  TRY(ensure(ctx, ctx->rs_d_keys, sizeof(float) * n));
  TRY(h2d(ctx, ctx->rs_d_keys.p, ctx->rs_keys.data(), sizeof(float) * n));
  if (ties)
  {
    TRY(ensure(ctx, ctx->rs_d_order, sizeof(uint32_t) * n));
    TRY(h2d(ctx, ctx->rs_d_order.p, ctx->rs_order.data(), sizeof(uint32_t) * n));
  }
  TRY(sync_stream(ctx));
  return 0;
}

int mcl3dl_hip_resample_plan(mcl3dl_hip_ctx* ctx, int mode, float initial_p, uint32_t* out_source,
                             uint8_t* out_duplicate, size_t* out_n_duplicates)
{
  if (!ctx)
    return -1;
  if (ctx->rs_n == 0)
    return ctx->fail(-5, "resample_plan before resample_begin");
  if (mode != 0 && mode != 1)
    return ctx->fail(-3, "mode must be 0 (resample) or 1 (resizeParticle)");
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t n = ctx->rs_n, n_out = ctx->rs_n_out;
  const int ni = static_cast<int>(n), no = static_cast<int>(n_out);
  TRY(ensure(ctx, ctx->rs_d_it, sizeof(uint32_t) * (n_out + 1)));  // [n_out] = the last search result below n
  TRY(ensure(ctx, ctx->rs_d_flag, sizeof(uint32_t) * (n_out + 1)));
  TRY(ensure(ctx, ctx->rs_d_ws, sizeof(uint32_t) * (n_out / 1023 + 8)));
  TRY(ensure(ctx, ctx->rs_d_source, sizeof(uint32_t) * n_out));
  TRY(ensure(ctx, ctx->rs_d_slot, sizeof(uint32_t) * n_out));
  uint32_t* d_it = ctx->rs_d_it.as<uint32_t>();
  uint32_t* d_flag = ctx->rs_d_flag.as<uint32_t>();
  HIP_TRY(hipMemsetAsync(d_it + n_out, 0, sizeof(uint32_t), ctx->stream));
  const float* d_pscan = nullptr;
  if (mode == 1)
  {
    // pscan += pstep (pf.h:421): another float recurrence, host side
    std::vector<float> pscan(n_out);
    float acc = 0;
    for (size_t i = 0; i < n_out; ++i)
      pscan[i] = (acc += ctx->rs_pstep);
    TRY(ensure(ctx, ctx->rs_d_pscan, sizeof(float) * n_out));
    TRY(h2d(ctx, ctx->rs_d_pscan.p, pscan.data(), sizeof(float) * n_out));
    TRY(sync_stream(ctx));  // pscan dies at the end of this block
    d_pscan = ctx->rs_d_pscan.as<float>();
  }
  // n_out lower_bound searches (pscan = pstep * i + initial_p computed in the kernel for mode 0, pf.h:209); pscan never
  // decreases, so the search the reference starts at the previous `it` lands where the global one does and the
  // it / it_prev walk of pf.h:204-223 / 414-434 becomes a neighbour comparison + an exclusive scan.
  hipLaunchKernelGGL(resample_lower_bound_kernel, dim3((no + 255) / 256), dim3(256), 0, ctx->stream,
                     ctx->rs_d_keys.as<float>(), ni, d_pscan, ctx->rs_pstep, initial_p, no, d_it, d_it + n_out);
  hipLaunchKernelGGL(resample_walk_kernel, dim3((no + 255) / 256), dim3(256), 0, ctx->stream, d_it, ni,
                     ctx->rs_sorted ? ctx->rs_d_order.as<uint32_t>() : static_cast<const uint32_t*>(nullptr), mode, no,
                     ctx->rs_d_source.as<uint32_t>(), d_flag);
  HIP_TRY(hipMemcpyAsync(ctx->rs_d_slot.p, d_flag, sizeof(uint32_t) * n_out, hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(hipMemsetAsync(d_flag + n_out, 0, sizeof(uint32_t), ctx->stream));
  TRY(device_exclusive_scan_ws(ctx, d_flag, static_cast<long long>(n_out) + 1, ctx->rs_d_ws.as<uint32_t>()));
  if (out_duplicate)
  {
    TRY(ensure(ctx, ctx->rs_d_dup8, n_out));
    hipLaunchKernelGGL(resample_slot_kernel, dim3((no + 255) / 256), dim3(256), 0, ctx->stream, d_flag, no,
                       ctx->rs_d_slot.as<uint32_t>(), ctx->rs_d_dup8.as<uint8_t>());
  }
  else
    hipLaunchKernelGGL(resample_slot_kernel, dim3((no + 255) / 256), dim3(256), 0, ctx->stream, d_flag, no,
                       ctx->rs_d_slot.as<uint32_t>(), static_cast<uint8_t*>(nullptr));
  HIP_TRY(hipGetLastError());
  uint32_t n_dup = 0;
  TRY(d2h(ctx, &n_dup, d_flag + n_out, sizeof(uint32_t)));
  if (out_source)
    TRY(d2h(ctx, out_source, ctx->rs_d_source.p, sizeof(uint32_t) * n_out));
  if (out_duplicate)
    TRY(d2h(ctx, out_duplicate, ctx->rs_d_dup8.p, n_out));
  TRY(sync_stream(ctx));
  ctx->rs_n_dup = n_dup;
  ctx->rs_planned = true;
  if (out_n_duplicates)
    *out_n_duplicates = n_dup;
  return 0;
}

int mcl3dl_hip_resample_begin_device(mcl3dl_hip_ctx* ctx, const float* d_weight, size_t n, size_t n_out, float* out_pstep)
{
  if (!ctx)
    return -1;
  if (!d_weight || n == 0 || n > 0x7fffffffu)
    return ctx->fail(-3, "bad arguments to resample_begin_device");
  HIP_TRY(hipSetDevice(ctx->device));
  // the prefix sums are a float recurrence in particle order (pf.h:193-197): 4 bytes per particle come to the host
  std::vector<float> w(n);
  TRY(d2h(ctx, w.data(), d_weight, sizeof(float) * n));
  TRY(sync_stream(ctx));
  return mcl3dl_hip_resample_begin(ctx, w.data(), n, n_out, out_pstep);
}

int mcl3dl_hip_resample_apply_slice_device(mcl3dl_hip_ctx* ctx, const float* d_state13_in, const float* noise13,
                                           size_t n_noise, size_t out_begin, size_t out_count, float* d_state13_out)
{
  if (!ctx)
    return -1;
  if (!ctx->rs_planned)
    return ctx->fail(-5, "resample_apply before resample_plan");
  if (!d_state13_in || !d_state13_out || d_state13_in == d_state13_out)
    return ctx->fail(-3, "resample_apply needs distinct input and output state arrays");
  if (out_begin > ctx->rs_n_out || out_count > ctx->rs_n_out - out_begin)
    return ctx->fail(-3, "resample_apply: slice [%zu, %zu) is outside the %zu planned slots", out_begin,
                     out_begin + out_count, ctx->rs_n_out);
  if (n_noise < ctx->rs_n_dup || (ctx->rs_n_dup && !noise13))
    return ctx->fail(-3, "resample_apply: %zu duplicated particles need noise, %zu given", ctx->rs_n_dup, n_noise);
  if (out_count == 0)
    return 0;
  HIP_TRY(hipSetDevice(ctx->device));
  TRY(ensure(ctx, ctx->rs_d_noise, sizeof(float) * 13 * ctx->rs_n_dup));
  TRY(h2d(ctx, ctx->rs_d_noise.p, noise13, sizeof(float) * 13 * ctx->rs_n_dup));
  const int no = static_cast<int>(out_count);
  hipLaunchKernelGGL(resample_apply_kernel, dim3((no + 255) / 256), dim3(256), 0, ctx->stream, d_state13_in,
                     ctx->rs_d_source.as<uint32_t>() + out_begin, ctx->rs_d_slot.as<uint32_t>() + out_begin,
                     ctx->rs_d_noise.as<float>(), no, d_state13_out);
  HIP_TRY(hipGetLastError());
  TRY(sync_stream(ctx));  // noise13 is the caller's host buffer
  return 0;
}

int mcl3dl_hip_resample_apply_device(mcl3dl_hip_ctx* ctx, const float* d_state13_in, const float* noise13,
                                     size_t n_noise, float* d_state13_out)
{
  if (!ctx)
    return -1;
  return mcl3dl_hip_resample_apply_slice_device(ctx, d_state13_in, noise13, n_noise, 0, ctx->rs_n_out, d_state13_out);
}

int mcl3dl_hip_resample_apply(mcl3dl_hip_ctx* ctx, const float* state13_in, const float* noise13, size_t n_noise,
                              float* state13_out)
{
  if (!ctx)
    return -1;
  if (!ctx->rs_planned)
    return ctx->fail(-5, "resample_apply before resample_plan");
  if (!state13_in || !state13_out)
    return ctx->fail(-3, "null state array");
  HIP_TRY(hipSetDevice(ctx->device));
  TRY(ensure(ctx, ctx->rs_d_in, sizeof(float) * 13 * ctx->rs_n));
  TRY(ensure(ctx, ctx->rs_d_out, sizeof(float) * 13 * ctx->rs_n_out));
  TRY(h2d(ctx, ctx->rs_d_in.p, state13_in, sizeof(float) * 13 * ctx->rs_n));
  TRY(mcl3dl_hip_resample_apply_device(ctx, ctx->rs_d_in.as<float>(), noise13, n_noise, ctx->rs_d_out.as<float>()));
  TRY(d2h(ctx, state13_out, ctx->rs_d_out.p, sizeof(float) * 13 * ctx->rs_n_out));
  TRY(sync_stream(ctx));
  return 0;
}

// ---- measurement support ---------------------------------------------------------------------------------------
int mcl3dl_hip_set_kernel_timing(mcl3dl_hip_ctx* ctx, int enable)
{
  if (!ctx)
    return -1;
  ++ctx->generation;
  TRY(timing_collect(ctx));
  ctx->timing = enable != 0;
  return 0;
}

int mcl3dl_hip_get_kernel_time(mcl3dl_hip_ctx* ctx, int kernel_id, double* total_ms, uint64_t* launches)
{
  if (!ctx)
    return -1;
  if (kernel_id < 0 || kernel_id >= MCL3DL_KERNEL_COUNT)
    return ctx->fail(-3, "bad kernel id");
  TRY(timing_collect(ctx));
  if (total_ms)
    *total_ms = ctx->kernel_ms[kernel_id];
  if (launches)
    *launches = ctx->kernel_launches[kernel_id];
  return 0;
}

int mcl3dl_hip_reset_kernel_time(mcl3dl_hip_ctx* ctx)
{
  if (!ctx)
    return -1;
  TRY(timing_collect(ctx));
  for (int k = 0; k < MCL3DL_KERNEL_COUNT; ++k)
  {
    ctx->kernel_ms[k] = 0;
    ctx->kernel_launches[k] = 0;
  }
  return 0;
}

int mcl3dl_hip_memory_footprint(mcl3dl_hip_ctx* ctx, uint64_t* bytes8)
{
  if (!ctx || !bytes8)
    return -1;
  for (int i = 0; i < 8; ++i)
    bytes8[i] = ctx->footprint[i];
  return 0;
}

int mcl3dl_hip_set_option(mcl3dl_hip_ctx* ctx, const char* name, double value)
{
  if (!ctx || !name)
    return -1;
  ++ctx->generation;
  const std::string key(name);
  if (key == "lik_index")
  {
    if (value != 0.0 && value != 1.0 && value != 2.0)
      return ctx->fail(-3, "lik_index must be 0 (27-cell scan), 1 (candidate runs) or 2 (candidate records)");
    if ((value == 0.0) != (ctx->lik_index == 0) || static_cast<int>(value) != ctx->lik_index)
      ctx->cand_dirty = true;
    ctx->lik_index = static_cast<int>(value);
    return 0;
  }
  if (key == "cand_voxel_ratio")
  {
    if (!(value >= 0.125 && value <= 2.0))
      return ctx->fail(-3, "cand_voxel_ratio must be in [0.125, 2]");
    if (value != ctx->cand_voxel_ratio)
      ctx->cand_dirty = true;
    ctx->cand_voxel_ratio = value;
    return 0;
  }
  if (key == "strict_order")
  {
    ctx->strict_order = value != 0.0;
    return 0;
  }
  if (key == "timing_mask")
  {
    ctx->timing_mask = static_cast<unsigned>(value);
    return 0;
  }
  if (key == "use_graph")
  {
    ctx->use_graph = value != 0.0;
    return 0;
  }
  if (key == "overlap_models")
  {
    ctx->overlap_models = value != 0.0;
    return 0;
  }
  if (key == "lik_small")
  {
    ctx->lik_small = value != 0.0;
    return 0;
  }
  if (key == "lik_tiled")
  {
    ctx->lik_tiled = value != 0.0;
    return 0;
  }
  if (key == "lik_group")
  {
    if (value != 8.0 && value != 16.0 && value != 32.0)
      return ctx->fail(-3, "lik_group must be 8, 16 or 32");
    ctx->lik_group = static_cast<int>(value);
    return 0;
  }
  if (key == "cand_phase")
  {
    if (!(value >= 0.0 && value < 1.0))
      return ctx->fail(-3, "cand_phase must be in [0, 1)");
    if (value != ctx->cand_phase)
      ctx->cand_dirty = true;
    ctx->cand_phase = value;
    return 0;
  }
  return ctx->fail(-3, "unknown option '%s'", name);
}

int mcl3dl_hip_index_stats(mcl3dl_hip_ctx* ctx, double* stats4)
{
  if (!ctx || !stats4)
    return -1;
  for (int i = 0; i < 4; ++i)
    stats4[i] = ctx->cand_stats[i];
  return 0;
}
}  // extern "C"
