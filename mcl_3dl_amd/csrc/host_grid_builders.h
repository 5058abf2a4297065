// host_grid_builders.h — part of the single translation unit mcl3dl_hip.hip: the cell-sorted exact-NN grid and the DDA
// occupancy / voxel index built ON THE DEVICE (grid_kernels.h). Both used to be sequential counting sorts on the host
// followed by an upload of the finished arrays (a 10 M-point map: seconds of host time, 1.1 GB of run delimiters over
// PCIe, at every map stamp and every map update); now the map is uploaded once as float4 {x, y, z, label} and everything
// else happens in HBM. Geometry (origin, extents) is computed on the host from the device-side min / max with the very
// expressions of the host builders, so the structures come out bit-identical.
#pragma once

namespace
{
// the map as a device cloud {x, y, z, label bits} in map order; re-uploaded only when the host copy changed
int ensure_map_dev(mcl3dl_hip_ctx* ctx)
{
  const size_t n = ctx->map_xyz.size() / 3;
  if (ctx->map_dev_valid && ctx->map_dev_n == n)
    return 0;
  TRY(upload_cloud(ctx, ctx->map_xyz.data(), ctx->map_label.data(), n, ctx->map_dev));
  ctx->map_dev_valid = true;
  ctx->map_dev_n = n;
  return 0;
}

int sort_bits(unsigned long long n_keys)
{
  int b = 1;
  while (b < 32 && (1ull << b) < n_keys)
    ++b;
  return b;
}

struct BuildTimer
{
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  ~BuildTimer()
  {
    if (ev0)
      (void)hipEventDestroy(ev0);
    if (ev1)
      (void)hipEventDestroy(ev1);
  }
};

// ---- exact-NN grid (replaces ChunkedKdtree::setInputCloud + pcl::KdTreeFLANN::setInputCloud, see host_map_compilers.h) ----
// The grid in use = the grid of the BASE map (built here: keys, histogram, stable sort, gather, scan — linear, but a sort of
// the whole map) merged with the current map update's points (three small kernels, no sort of the map: grid_kernels.h).
// The base grid is laid out for the bounds of base + update of the moment; a later update that stays inside those bounds
// only merges (0.36 -> ~0.07 ms at 1 M points), one that leaves them rebuilds the base for the new bounds.
int build_lik_grid_device(mcl3dl_hip_ctx* ctx)
{
  const size_t n = ctx->map_xyz.size() / 3;
  const size_t n_base = (ctx->n_base && ctx->n_base <= n) ? ctx->n_base : n;
  const size_t n_upd = n - n_base;
  const long long nn = static_cast<long long>(n);
  const float cell = ctx->match_dist_min * 1.01f;
  if (!(cell > 0.f) || !std::isfinite(cell))
    return ctx->fail(-3, "match_dist_min must be positive and finite");
  if (n == 0 || n > 0x7fffffffu)
    return ctx->fail(-3, "bad map size for the likelihood grid");
  const float inv = 1.0f / cell;
  BuildTimer tm;
  HIP_TRY(hipEventCreate(&tm.ev0));
  HIP_TRY(hipEventCreate(&tm.ev1));
  HIP_TRY(hipEventRecord(tm.ev0, ctx->stream));
  // can the base grid in place take this update? (same base map, same cell edge, update inside the bounds it was laid out for).
  // The update's points — a few thousand — are rescaled on the host (the same float product the device forms) and go up alone:
  // the whole map is neither uploaded nor touched by a kernel unless the base has to be rebuilt.
  bool keep_base = !ctx->lik_base_dirty && ctx->lik_base_n == n_base && ctx->lg.inv_cell == inv && ctx->lik_base_pts.p;
  std::vector<float4>& upd_host = ctx->lik_upd_host;
  upd_host.resize(n_upd);
  for (size_t k = 0; k < n_upd; ++k)
  {
    const size_t i = n_base + k;
    float v[3];
    for (int a = 0; a < 3; ++a)
    {
      v[a] = ctx->has_weight ? ctx->map_xyz[3 * i + a] * ctx->weight[a] : ctx->map_xyz[3 * i + a];
      if (!std::isfinite(v[a]))
        return ctx->fail(-3, "map point %zu is not finite", i);
      keep_base = keep_base && v[a] >= ctx->lik_base_lo[a] && v[a] <= ctx->lik_base_hi[a];
    }
    upd_host[k] = make_float4(v[0], v[1], v[2], bits_to_float(static_cast<uint32_t>(i)));
  }
  TempBuf sp;       // rescaled points of the whole map (only when the base is rebuilt)
  TempBuf sp_u;     // rescaled points of the update
  if (n_upd)
  {
    TRY(scratch_alloc(ctx, sp_u, sizeof(float4) * n_upd));
    TRY(h2d(ctx, sp_u.p, upd_host.data(), sizeof(float4) * n_upd));
  }
  if (!keep_base)
  {
    TRY(ensure_map_dev(ctx));
    TRY(scratch_alloc(ctx, sp, sizeof(float4) * n));
    hipLaunchKernelGGL(grid_rescale_kernel, dim3(blocks_for(nn)), dim3(256), 0, ctx->stream, ctx->map_dev.as<float4>(), nn,
                       ctx->weight[0], ctx->weight[1], ctx->weight[2], ctx->has_weight ? 1 : 0, static_cast<float4*>(sp.p));
    float mm[6];
    unsigned long long n_finite = 0;
    TRY(cloud_minmax(ctx, static_cast<const float4*>(sp.p), nn, mm, &n_finite));
    if (n_finite != n)
      return ctx->fail(-3, "%llu map point(s) are not finite", static_cast<unsigned long long>(n) - n_finite);
    float o[3];
    int dim[3];
    double total = 1;
    for (int a = 0; a < 3; ++a)
    {
      o[a] = mm[a] - 2.0f * cell;
      dim[a] = static_cast<int>(floorf((mm[3 + a] - o[a]) * inv)) + 3;
      total *= dim[a];
    }
    if (total > 3.0e9)
      return ctx->fail(-4, "likelihood grid would need %.3g cells (map extent too large for the dense index)", total);
    const size_t ncell = static_cast<size_t>(dim[0]) * dim[1] * dim[2];
    const long long nb = static_cast<long long>(n_base);
    TRY(ensure(ctx, ctx->lik_base_pts, sizeof(float4) * n_base));
    TRY(ensure(ctx, ctx->lik_base_cells, sizeof(uint32_t) * (ncell + 1)));
    TRY(ensure(ctx, ctx->cl_key[0], sizeof(uint32_t) * (n + 1)));
    TRY(ensure(ctx, ctx->cl_key[1], sizeof(uint32_t) * (n + 1)));
    TRY(ensure(ctx, ctx->cl_val[0], sizeof(uint32_t) * (n + 1)));
    TRY(ensure(ctx, ctx->cl_val[1], sizeof(uint32_t) * (n + 1)));
    HIP_TRY(hipMemsetAsync(ctx->lik_base_cells.p, 0, sizeof(uint32_t) * (ncell + 1), ctx->stream));
    const CellGeom g{ o[0], o[1], o[2], inv, dim[0], dim[1], dim[2] };
    hipLaunchKernelGGL(lik_cell_key_kernel, dim3(blocks_for(nb)), dim3(256), 0, ctx->stream, static_cast<const float4*>(sp.p),
                       nb, g, ctx->cl_key[0].as<uint32_t>(), ctx->cl_val[0].as<uint32_t>(), ctx->lik_base_cells.as<uint32_t>());
    TRY(sort_pairs(ctx, nb, sort_bits(ncell)));
    hipLaunchKernelGGL(grid_gather_kernel, dim3(blocks_for(nb)), dim3(256), 0, ctx->stream, static_cast<const float4*>(sp.p),
                       ctx->cl_val[1].as<uint32_t>(), nb, ctx->lik_base_pts.as<float4>());
    HIP_TRY(hipGetLastError());
    TRY(device_exclusive_scan(ctx, ctx->lik_base_cells.as<uint32_t>(), static_cast<long long>(ncell) + 1));
    ctx->lg.ox = o[0];
    ctx->lg.oy = o[1];
    ctx->lg.oz = o[2];
    ctx->lg.inv_cell = inv;
    ctx->lg.nx = dim[0];
    ctx->lg.ny = dim[1];
    ctx->lg.nz = dim[2];
    for (int a = 0; a < 3; ++a)
    {
      ctx->lik_base_lo[a] = mm[a];
      ctx->lik_base_hi[a] = mm[3 + a];
    }
    ctx->lik_base_n = n_base;
    ctx->lik_base_dirty = false;
    ++ctx->lik_grid_rebuilds;
  }
  const size_t ncell = static_cast<size_t>(ctx->lg.nx) * ctx->lg.ny * ctx->lg.nz;
  if (n_upd == 0)
  {
    ctx->lg.cell_start = ctx->lik_base_cells.as<uint32_t>();
    ctx->lg.pts = ctx->lik_base_pts.as<float4>();
  }
  else
  {
    const CellGeom g{ ctx->lg.ox, ctx->lg.oy, ctx->lg.oz, inv, ctx->lg.nx, ctx->lg.ny, ctx->lg.nz };
    const int nu = static_cast<int>(n_upd);
    TRY(ensure(ctx, ctx->lik_pts, sizeof(float4) * n));
    TRY(ensure(ctx, ctx->lik_cells, sizeof(uint32_t) * (ncell + 1)));
    TRY(ensure(ctx, ctx->cl_key[0], sizeof(uint32_t) * (n_upd + 1)));
    TRY(ensure(ctx, ctx->cl_key[1], sizeof(uint32_t) * (n_upd + 1)));
    TRY(ensure(ctx, ctx->cl_val[0], sizeof(uint32_t) * (n_upd + 1)));
    TRY(ensure(ctx, ctx->cl_val[1], sizeof(uint32_t) * (n_upd + 1)));
    const float4* sp_upd = static_cast<const float4*>(sp_u.p);
    hipLaunchKernelGGL(lik_update_key_kernel, dim3(blocks_for(nu)), dim3(256), 0, ctx->stream, sp_upd, nu, g,
                       ctx->cl_key[0].as<uint32_t>(), ctx->cl_val[0].as<uint32_t>());
    TRY(sort_pairs(ctx, nu, sort_bits(ncell)));
    hipLaunchKernelGGL(lik_merge_cells_kernel, dim3(blocks_for(static_cast<long long>(ncell) + 1)), dim3(256), 0, ctx->stream,
                       ctx->lik_base_cells.as<uint32_t>(), static_cast<long long>(ncell) + 1, ctx->cl_key[1].as<uint32_t>(),
                       static_cast<uint32_t>(n_upd), ctx->lik_cells.as<uint32_t>());
    hipLaunchKernelGGL(lik_merge_points_kernel, dim3(blocks_for(nn)), dim3(256), 0, ctx->stream,
                       ctx->lik_base_pts.as<float4>(), static_cast<long long>(n_base), ctx->lik_base_cells.as<uint32_t>(), g,
                       sp_upd, ctx->cl_key[1].as<uint32_t>(), ctx->cl_val[1].as<uint32_t>(), static_cast<uint32_t>(n_upd),
                       ctx->lik_pts.as<float4>());
    HIP_TRY(hipGetLastError());
    ctx->lg.cell_start = ctx->lik_cells.as<uint32_t>();
    ctx->lg.pts = ctx->lik_pts.as<float4>();
    if (keep_base)
      ++ctx->lik_grid_merges;
  }
  HIP_TRY(hipEventRecord(tm.ev1, ctx->stream));
  TRY(sync_stream(ctx));
  float ms = 0.f;
  HIP_TRY(hipEventSynchronize(tm.ev1));  // (the stream may have been waited for through the polled word: the event is past, the runtime has to look)
  HIP_TRY(hipEventElapsedTime(&ms, tm.ev0, tm.ev1));
  ctx->grid_build_ms[0] = ms;
  ctx->footprint[0] = sizeof(float4) * n;
  ctx->footprint[1] = sizeof(uint32_t) * (ncell + 1);
  ctx->lik_dirty = false;
  return 0;
}

// ---- DDA occupancy + voxel index (RaycastUsingDDA::updatePointCloud / setExists, raycast_using_dda.h:162-190,230-235) ----
// The map update as an overlay (DdaGrid::ov_*): the previous overlay's bits are withdrawn (a voxel without base points is
// empty again), the new points are keyed, sorted by voxel (stable: update order inside a voxel) and their bits set. A few
// small launches over the update's points — the base arrays are not touched, nothing is read back. upd = the update's
// points on the device {x, y, z, label bits}, all inside the grid's bounds (the caller checked).
int dda_overlay_apply(mcl3dl_hip_ctx* ctx, const float4* upd, size_t n_upd, size_t n_base)
{
  DdaGrid& d = ctx->dg;
  const DdaGeom g = ctx->dda_geom;
  if (d.ov_n > 0)
    hipLaunchKernelGGL(dda_overlay_clear_kernel, dim3(blocks_for(d.ov_n)), dim3(256), 0, ctx->stream, d.ov_key, d.ov_n, g,
                       d.vox_start, ctx->dda_bits.as<unsigned long long>());
  d.ov_n = 0;
  d.ov_base = static_cast<uint32_t>(n_base);
  if (n_upd == 0)
    return 0;
  if (n_upd > 0x7fffffffu)
    return ctx->fail(-3, "map update too large");
  const int n = static_cast<int>(n_upd);
  TRY(ensure(ctx, ctx->dda_ov_key, sizeof(uint32_t) * n_upd));  // (a reallocation waits for the stream: the clear kernel is through)
  TRY(ensure(ctx, ctx->dda_ov_pts, sizeof(float4) * n_upd));
  TRY(ensure(ctx, ctx->dda_ov_idx, sizeof(uint32_t) * n_upd));
  TRY(ensure(ctx, ctx->cl_key[0], sizeof(uint32_t) * (n_upd + 1)));
  TRY(ensure(ctx, ctx->cl_key[1], sizeof(uint32_t) * (n_upd + 1)));
  TRY(ensure(ctx, ctx->cl_val[0], sizeof(uint32_t) * (n_upd + 1)));
  TRY(ensure(ctx, ctx->cl_val[1], sizeof(uint32_t) * (n_upd + 1)));
  hipLaunchKernelGGL(dda_overlay_key_kernel, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, upd, n, g,
                     ctx->cl_key[0].as<uint32_t>(), ctx->cl_val[0].as<uint32_t>());
  TRY(sort_pairs(ctx, n, sort_bits(g.total)));
  hipLaunchKernelGGL(dda_overlay_set_kernel, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, upd, ctx->cl_key[1].as<uint32_t>(),
                     ctx->cl_val[1].as<uint32_t>(), n, g, ctx->dda_ov_key.as<uint32_t>(), ctx->dda_ov_pts.as<float4>(),
                     ctx->dda_ov_idx.as<uint32_t>(), ctx->dda_bits.as<unsigned long long>());
  HIP_TRY(hipGetLastError());
  d.ov_key = ctx->dda_ov_key.as<uint32_t>();
  d.ov_pts = ctx->dda_ov_pts.as<float4>();
  d.ov_idx = ctx->dda_ov_idx.as<uint32_t>();
  d.ov_n = n;
  return 0;
}

int build_dda_grid_device(mcl3dl_hip_ctx* ctx)
{
  const size_t n = ctx->map_xyz.size() / 3;
  const double grid = static_cast<double>(ctx->dda_grid_size);
  if (!(grid > 0))
    return ctx->fail(-3, "dda_grid_size must be positive");
  if (n == 0 || n > 0x7fffffffu)
    return ctx->fail(-3, "bad map size for the DDA grid");
  BuildTimer tm;
  HIP_TRY(hipEventCreate(&tm.ev0));
  HIP_TRY(hipEventCreate(&tm.ev1));
  TRY(ensure_map_dev(ctx));
  HIP_TRY(hipEventRecord(tm.ev0, ctx->stream));
  float mm[6];
  unsigned long long n_finite = 0;
  TRY(cloud_minmax(ctx, ctx->map_dev.as<float4>(), static_cast<long long>(n), mm, &n_finite));  // pcl::getMinMax3D
  if (n_finite != n)
    return ctx->fail(-3, "%llu map point(s) are not finite", static_cast<unsigned long long>(n) - n_finite);
  // A map update that stays inside the base map's bounds (the usual one: it is what the robot sees) does not change the
  // grid's geometry: the arrays are then built from the base map alone and the update rides on top as an overlay, which the
  // next update replaces without a rebuild.
  const size_t n_base = (ctx->n_base && ctx->n_base <= n) ? ctx->n_base : n;
  const size_t n_upd = n - n_base;
  bool overlay = false;
  if (n_upd > 0 && ctx->dda_overlay)
  {
    float mb[6];
    unsigned long long nf = 0;
    TRY(cloud_minmax(ctx, ctx->map_dev.as<float4>(), static_cast<long long>(n_base), mb, &nf));
    overlay = memcmp(mb, mm, sizeof(mm)) == 0;
  }
  const size_t n_csr = overlay ? n_base : n;
  const long long nn = static_cast<long long>(n_csr);
  const float* mn = mm;
  const float* mx = mm + 3;
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
  const int bdim[3] = { (dim[0] + 3) / 4, (dim[1] + 3) / 4, (dim[2] + 3) / 4 };
  const size_t n_bricks = static_cast<size_t>(bdim[0]) * bdim[1] * bdim[2];
  TRY(ensure(ctx, ctx->dda_bits, sizeof(unsigned long long) * n_bricks));
  TRY(ensure(ctx, ctx->dda_start, sizeof(uint32_t) * (total + 1)));
  TRY(ensure(ctx, ctx->dda_pts, sizeof(float4) * n_csr));
  TRY(ensure(ctx, ctx->dda_index, sizeof(uint32_t) * n_csr));
  TRY(ensure(ctx, ctx->cl_key[0], sizeof(uint32_t) * (n + 1)));
  TRY(ensure(ctx, ctx->cl_key[1], sizeof(uint32_t) * (n + 1)));
  TRY(ensure(ctx, ctx->cl_val[0], sizeof(uint32_t) * (n + 1)));
  TRY(ensure(ctx, ctx->cl_val[1], sizeof(uint32_t) * (n + 1)));
  TRY(ensure(ctx, ctx->cl_err, sizeof(int)));
  HIP_TRY(hipMemsetAsync(ctx->cl_err.p, 0, sizeof(int), ctx->stream));
  HIP_TRY(hipMemsetAsync(ctx->dda_bits.p, 0, sizeof(unsigned long long) * n_bricks, ctx->stream));
  HIP_TRY(hipMemsetAsync(ctx->dda_start.p, 0, sizeof(uint32_t) * (total + 1), ctx->stream));
  const DdaGeom g{ mn[0], mn[1], mn[2], grid, dim[0], dim[1], static_cast<unsigned long long>(total), bdim[0], bdim[1] };
  hipLaunchKernelGGL(dda_voxel_key_kernel, dim3(blocks_for(nn)), dim3(256), 0, ctx->stream, ctx->map_dev.as<float4>(), nn, g,
                     ctx->cl_key[0].as<uint32_t>(), ctx->cl_val[0].as<uint32_t>(), ctx->dda_start.as<uint32_t>(),
                     ctx->dda_bits.as<unsigned long long>(), ctx->cl_err.as<int>());
  TRY(sort_pairs(ctx, nn, sort_bits(total)));
  hipLaunchKernelGGL(dda_gather_kernel, dim3(blocks_for(nn)), dim3(256), 0, ctx->stream, ctx->map_dev.as<float4>(),
                     ctx->cl_val[1].as<uint32_t>(), nn, ctx->dda_pts.as<float4>(), ctx->dda_index.as<uint32_t>());
  HIP_TRY(hipGetLastError());
  TRY(device_exclusive_scan(ctx, ctx->dda_start.as<uint32_t>(), static_cast<long long>(total) + 1));
  int err = 0;
  TRY(d2h(ctx, &err, ctx->cl_err.p, sizeof(int)));
  DdaGrid& d = ctx->dg;
  d.bricks = ctx->dda_bits.as<unsigned long long>();
  d.bnx = bdim[0];
  d.bny = bdim[1];
  d.bnz = bdim[2];
  d.mul24_ok = (bdim[0] < (1 << 24) && static_cast<long long>(bdim[1]) * bdim[2] < (1ll << 24)) ? 1 : 0;
  d.vox_start = ctx->dda_start.as<uint32_t>();
  d.pts = ctx->dda_pts.as<float4>();
  d.pt_index = ctx->dda_index.as<uint32_t>();
  d.min_x = mn[0];
  d.min_y = mn[1];
  d.min_z = mn[2];
  d.max_x = mx[0];
  d.max_y = mx[1];
  d.max_z = mx[2];
  d.nx = dim[0];
  d.ny = dim[1];
  d.nz = dim[2];
  d.ov_n = 0;  // (a fresh bit array: nothing of an earlier overlay to withdraw)
  d.ov_base = static_cast<uint32_t>(n_base);
  ctx->dda_geom = g;
  if (overlay)
    TRY(dda_overlay_apply(ctx, ctx->map_dev.as<float4>() + n_base, n_upd, n_base));
  HIP_TRY(hipEventRecord(tm.ev1, ctx->stream));
  TRY(sync_stream(ctx));
  if (err)
    return ctx->fail(-3, "a map point falls outside its own DDA grid");
  float ms = 0.f;
  HIP_TRY(hipEventSynchronize(tm.ev1));  // (the stream may have been waited for through the polled word: the event is past, the runtime has to look)
  HIP_TRY(hipEventElapsedTime(&ms, tm.ev0, tm.ev1));
  ctx->grid_build_ms[1] = ms;
  ctx->dda_overlay_ok = ctx->dda_overlay && (n_upd == 0 || overlay);
  dda_ray_constants(ctx, d);
  ctx->footprint[2] = sizeof(unsigned long long) * n_bricks;
  ctx->footprint[3] = sizeof(uint32_t) * (total + 1);
  ctx->footprint[4] = sizeof(float4) * n + sizeof(uint32_t) * n;
  ctx->dda_dirty = false;
  return 0;
}

int build_lik_grid(mcl3dl_hip_ctx* ctx)
{
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = ctx->grid_build_host ? build_lik_grid_host(ctx) : build_lik_grid_device(ctx);
  ctx->grid_build_wall_ms[0] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return rc;
}

int build_dda_grid(mcl3dl_hip_ctx* ctx)
{
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = ctx->grid_build_host ? build_dda_grid_host(ctx) : build_dda_grid_device(ctx);
  ctx->grid_build_wall_ms[1] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return rc;
}
}  // namespace
