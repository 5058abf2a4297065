// host_cloud.h — part of the single translation unit mcl3dl_hip.hip: host-side drivers of cloud_kernels.h shared by the scan
// upload (api_core.inl) and the scan-preparation / map entry points (api_cloud.inl).
#pragma once

namespace
{
unsigned blocks_for(long long n)
{
  return static_cast<unsigned>((std::max<long long>(n, 1) + 255) / 256);
}

// stable ascending sort of (key, value) pairs on the context's stream; results in keys_out / vals_out
int sort_pairs(mcl3dl_hip_ctx* ctx, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
               long long n, int end_bit)
{
  if (n <= 0)
    return 0;
  if (n > 0x7fffffffLL)
    return ctx->fail(-3, "too many points to sort");
  size_t bytes = 0;
  HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, static_cast<int>(n), 0,
                                             end_bit, ctx->stream));
  TRY(ensure(ctx, ctx->sort_tmp, bytes));
  HIP_TRY(hipcub::DeviceRadixSort::SortPairs(ctx->sort_tmp.p, bytes, keys_in, keys_out, vals_in, vals_out,
                                             static_cast<int>(n), 0, end_bit, ctx->stream));
  return 0;
}

// min / max of the finite points of a device cloud -> ctx->cl_minmax (6 floats, device) and, on request, the host
int cloud_minmax(mcl3dl_hip_ctx* ctx, const float4* pts, long long n, float* host6, unsigned long long* host_cnt)
{
  const int nb = static_cast<int>(std::min<long long>((n + 255) / 256, 512));
  TRY(ensure(ctx, ctx->cl_blocks, sizeof(float) * 6 * std::max(nb, 1) + sizeof(unsigned) * std::max(nb, 1)));
  TRY(ensure(ctx, ctx->cl_minmax, sizeof(float) * 6 + sizeof(unsigned long long)));
  float* bo = ctx->cl_blocks.as<float>();
  unsigned* bc = reinterpret_cast<unsigned*>(bo + 6 * std::max(nb, 1));
  unsigned long long* cnt = reinterpret_cast<unsigned long long*>(ctx->cl_minmax.as<float>() + 6);
  hipLaunchKernelGGL(cloud_minmax_kernel, dim3(std::max(nb, 1)), dim3(256), 0, ctx->stream, pts, n, bo, bc);
  hipLaunchKernelGGL(cloud_minmax_final, dim3(1), dim3(64), 0, ctx->stream, bo, bc, std::max(nb, 1),
                     ctx->cl_minmax.as<float>(), cnt);
  HIP_TRY(hipGetLastError());
  if (host6 || host_cnt)
  {
    float h[8];
    TRY(d2h(ctx, h, ctx->cl_minmax.p, sizeof(float) * 6 + sizeof(unsigned long long)));
    TRY(sync_stream(ctx));
    if (host6)
      memcpy(host6, h, sizeof(float) * 6);
    if (host_cnt)
      memcpy(host_cnt, h + 6, sizeof(unsigned long long));
  }
  return 0;
}

// pcl::VoxelGrid<PointXYZIL> with only setLeafSize set (the node's three call sites): `in` (n points, device) ->
// `out` (one centroid per occupied leaf, in ascending leaf-index order). A leaf component <= 0 skips the filter.
int voxel_grid(mcl3dl_hip_ctx* ctx, const float4* in, size_t n, const float leaf[3], DevBuf& out, size_t* n_out)
{
  *n_out = 0;
  TRY(ensure(ctx, out, sizeof(float4) * std::max<size_t>(n, 1)));
  if (n == 0)
    return 0;
  if (!leaf || !(leaf[0] > 0.f && leaf[1] > 0.f && leaf[2] > 0.f))
  {
    HIP_TRY(hipMemcpyAsync(out.p, in, sizeof(float4) * n, hipMemcpyDeviceToDevice, ctx->stream));
    *n_out = n;
    return 0;
  }
  float mm[6];
  unsigned long long n_finite = 0;
  TRY(cloud_minmax(ctx, in, static_cast<long long>(n), mm, &n_finite));
  if (n_finite == 0)
    return 0;
  VoxelGridParams vp{};
  long long d[3];
  int max_b[3];
  for (int a = 0; a < 3; ++a)
  {
    vp.inv_leaf[a] = 1.0f / leaf[a];  // Eigen::Array4f::Ones() / leaf_size_.array()
    // voxel_grid.hpp: static_cast<std::int64_t>((max_p - min_p) * inverse_leaf_size) + 1
    d[a] = static_cast<long long>((mm[3 + a] - mm[a]) * vp.inv_leaf[a]) + 1;
    vp.min_b[a] = static_cast<int>(std::floor(mm[a] * vp.inv_leaf[a]));
    max_b[a] = static_cast<int>(std::floor(mm[3 + a] * vp.inv_leaf[a]));
  }
  if (d[0] * d[1] * d[2] > static_cast<long long>(std::numeric_limits<int32_t>::max()))
  {
    // "Leaf size is too small for the input dataset. Integer indices would overflow": PCL hands the input back
    HIP_TRY(hipMemcpyAsync(out.p, in, sizeof(float4) * n, hipMemcpyDeviceToDevice, ctx->stream));
    *n_out = n;
    return 0;
  }
  const int div0 = max_b[0] - vp.min_b[0] + 1, div1 = max_b[1] - vp.min_b[1] + 1;
  vp.mul[0] = 1;
  vp.mul[1] = div0;
  vp.mul[2] = div0 * div1;
  const long long nn = static_cast<long long>(n);
  TRY(ensure(ctx, ctx->cl_key[0], sizeof(uint32_t) * (n + 1)));
  TRY(ensure(ctx, ctx->cl_key[1], sizeof(uint32_t) * (n + 1)));
  TRY(ensure(ctx, ctx->cl_val[0], sizeof(uint32_t) * (n + 1)));
  TRY(ensure(ctx, ctx->cl_val[1], sizeof(uint32_t) * (n + 1)));
  TRY(ensure(ctx, ctx->cl_scan, sizeof(uint32_t) * (n + 2)));
  TRY(ensure(ctx, ctx->cl_scan_ws, sizeof(uint32_t) * (n / 1023 + 16)));
  hipLaunchKernelGGL(vg_key_kernel, dim3(blocks_for(nn)), dim3(256), 0, ctx->stream, in, nn, vp,
                     ctx->cl_key[0].as<uint32_t>(), ctx->cl_val[0].as<uint32_t>());
  TRY(sort_pairs(ctx, ctx->cl_key[0].as<uint32_t>(), ctx->cl_key[1].as<uint32_t>(), ctx->cl_val[0].as<uint32_t>(),
                 ctx->cl_val[1].as<uint32_t>(), nn, 32));
  const long long nf = static_cast<long long>(n_finite);  // the non-finite points carry key 0xffffffff: sorted last
  hipLaunchKernelGGL(vg_heads_kernel, dim3(blocks_for(nf + 1)), dim3(256), 0, ctx->stream, ctx->cl_key[1].as<uint32_t>(),
                     nf, ctx->cl_scan.as<uint32_t>());
  TRY(device_exclusive_scan_ws(ctx, ctx->cl_scan.as<uint32_t>(), nf + 1, ctx->cl_scan_ws.as<uint32_t>()));
  uint32_t n_leaves = 0;
  TRY(d2h(ctx, &n_leaves, ctx->cl_scan.as<uint32_t>() + nf, sizeof(uint32_t)));
  TRY(sync_stream(ctx));
  TRY(ensure(ctx, ctx->cl_start, sizeof(uint32_t) * (static_cast<size_t>(n_leaves) + 1)));
  hipLaunchKernelGGL(vg_starts_kernel, dim3(blocks_for(nf + 1)), dim3(256), 0, ctx->stream, ctx->cl_key[1].as<uint32_t>(),
                     ctx->cl_scan.as<uint32_t>(), nf, n_leaves, ctx->cl_start.as<uint32_t>());
  hipLaunchKernelGGL(vg_centroid_kernel, dim3(blocks_for(n_leaves)), dim3(256), 0, ctx->stream, in,
                     ctx->cl_val[1].as<uint32_t>(), ctx->cl_start.as<uint32_t>(), n_leaves, out.as<float4>());
  HIP_TRY(hipGetLastError());
  *n_out = n_leaves;
  return 0;
}

// clip predicate + order-preserving compaction: in (n) -> out. The number kept arrives in *kept32 at the caller's next
// sync_stream() (no synchronisation here: the two models' clips share one).
int clip_compact(mcl3dl_hip_ctx* ctx, const float4* in, size_t n, const float clip4[4], DevBuf& out, uint32_t* kept32,
                 int scan_slot)
{
  *kept32 = 0;
  TRY(ensure(ctx, out, sizeof(float4) * std::max<size_t>(n, 1)));
  if (n == 0)
    return 0;
  const long long nn = static_cast<long long>(n);
  // each clip has its own flag / scan arrays: the second one is enqueued while the first one's count is still in flight
  DevBuf& flags = ctx->cl_clip_scan[scan_slot];
  DevBuf& ws = ctx->cl_clip_ws[scan_slot];
  TRY(ensure(ctx, flags, sizeof(uint32_t) * (n + 2)));
  TRY(ensure(ctx, ws, sizeof(uint32_t) * (n / 1023 + 16)));
  // clip_near_sq_ = clip_near * clip_near etc. in float, like refreshParameters (likelihood.cpp:58-59, beam.cpp:60-61)
  const float near_sq = clip4[0] * clip4[0], far_sq = clip4[1] * clip4[1];
  hipLaunchKernelGGL(clip_flag_kernel, dim3(blocks_for(nn + 1)), dim3(256), 0, ctx->stream, in, nn, near_sq, far_sq, clip4[2],
                     clip4[3], flags.as<uint32_t>());
  TRY(device_exclusive_scan_ws(ctx, flags.as<uint32_t>(), nn + 1, ws.as<uint32_t>()));
  hipLaunchKernelGGL(compact_kernel, dim3(blocks_for(nn)), dim3(256), 0, ctx->stream, in, flags.as<uint32_t>(), nn,
                     out.as<float4>());
  TRY(d2h(ctx, kept32, flags.as<uint32_t>() + nn, sizeof(uint32_t)));
  return 0;
}

// host xyz (+ label) -> device float4 cloud
int upload_cloud(mcl3dl_hip_ctx* ctx, const float* xyz, const uint32_t* label, size_t n, DevBuf& out)
{
  TRY(ensure(ctx, out, sizeof(float4) * std::max<size_t>(n, 1)));
  if (n == 0)
    return 0;
  TRY(ensure(ctx, ctx->cl_in_xyz, sizeof(float) * 3 * n));
  TRY(h2d(ctx, ctx->cl_in_xyz.p, xyz, sizeof(float) * 3 * n));
  const uint32_t* d_label = nullptr;
  if (label)
  {
    TRY(ensure(ctx, ctx->cl_in_label, sizeof(uint32_t) * n));
    TRY(h2d(ctx, ctx->cl_in_label.p, label, sizeof(uint32_t) * n));
    d_label = ctx->cl_in_label.as<uint32_t>();
  }
  hipLaunchKernelGGL(cloud_pack_kernel, dim3(blocks_for(static_cast<long long>(n))), dim3(256), 0, ctx->stream,
                     ctx->cl_in_xyz.as<float>(), d_label, static_cast<long long>(n), out.as<float4>());
  HIP_TRY(hipGetLastError());
  return 0;
}

// PointCloud2 bytes -> device float4 cloud (mcl_3dl::fromROSMsg, point_conversion.h:64-92: x, y, z are required; a
// "label" field is used when present; "intensity" is not on the measurement path)
int decode_cloud(mcl3dl_hip_ctx* ctx, const uint8_t* data, size_t n_points, uint32_t point_step, int off_x, int off_y,
                 int off_z, int off_label, DevBuf& out)
{
  if (off_x < 0 || off_y < 0 || off_z < 0)
    return ctx->fail(-3, "Given PointCloud2 doesn't have x, y, z fields");
  if (n_points == 0)
    return ctx->fail(-3, "Given PointCloud2 is empty");
  const int offs[4] = { off_x, off_y, off_z, off_label };
  for (int k = 0; k < 4; ++k)
    if (offs[k] >= 0 && static_cast<uint32_t>(offs[k]) + 4 > point_step)
      return ctx->fail(-3, "field offset %d does not fit point_step %u", offs[k], point_step);
  if (!data)
    return ctx->fail(-3, "null PointCloud2 data");
  const size_t bytes = n_points * static_cast<size_t>(point_step);
  TRY(ensure(ctx, ctx->cl_in_xyz, bytes));
  TRY(ensure(ctx, out, sizeof(float4) * n_points));
  TRY(h2d(ctx, ctx->cl_in_xyz.p, data, bytes));
  hipLaunchKernelGGL(cloud_decode_kernel, dim3(blocks_for(static_cast<long long>(n_points))), dim3(256), 0, ctx->stream,
                     ctx->cl_in_xyz.as<uint8_t>(), static_cast<long long>(n_points), point_step, off_x, off_y, off_z,
                     off_label, out.as<float4>());
  HIP_TRY(hipGetLastError());
  return 0;
}

int download_cloud(mcl3dl_hip_ctx* ctx, const float4* src, size_t n, float* xyz, uint32_t* label)
{
  if (n == 0)
    return 0;
  TRY(ensure(ctx, ctx->cl_in_xyz, sizeof(float) * 3 * n));
  TRY(ensure(ctx, ctx->cl_in_label, sizeof(uint32_t) * n));
  hipLaunchKernelGGL(cloud_unpack_kernel, dim3(blocks_for(static_cast<long long>(n))), dim3(256), 0, ctx->stream, src,
                     static_cast<long long>(n), ctx->cl_in_xyz.as<float>(), label ? ctx->cl_in_label.as<uint32_t>() : nullptr);
  HIP_TRY(hipGetLastError());
  TRY(d2h(ctx, xyz, ctx->cl_in_xyz.p, sizeof(float) * 3 * n));
  if (label)
    TRY(d2h(ctx, label, ctx->cl_in_label.p, sizeof(uint32_t) * n));
  TRY(sync_stream(ctx));
  return 0;
}

int scan_begin_common(mcl3dl_hip_ctx* ctx, size_t n, const float* leaf3, const float* clip_lik4, const float* clip_beam4,
                      size_t* n_full, size_t* n_lik, size_t* n_beam)
{
  // ctx->sp_raw holds the accumulated cloud
  ctx->sp_ready = false;
  TRY(voxel_grid(ctx, ctx->sp_raw.as<float4>(), n, leaf3, ctx->sp_full, &ctx->sp_n_full));
  ctx->sp_kept32[0] = ctx->sp_kept32[1] = 0;
  if (clip_lik4)
    TRY(clip_compact(ctx, ctx->sp_full.as<float4>(), ctx->sp_n_full, clip_lik4, ctx->sp_clip[0], &ctx->sp_kept32[0], 0));
  if (clip_beam4)
    TRY(clip_compact(ctx, ctx->sp_full.as<float4>(), ctx->sp_n_full, clip_beam4, ctx->sp_clip[1], &ctx->sp_kept32[1], 1));
  TRY(sync_stream(ctx));  // ONE synchronisation delivers both counts
  ctx->sp_n_clip[0] = ctx->sp_kept32[0];
  ctx->sp_n_clip[1] = ctx->sp_kept32[1];
  ctx->sp_ready = true;
  if (n_full)
    *n_full = ctx->sp_n_full;
  if (n_lik)
    *n_lik = ctx->sp_n_clip[0];
  if (n_beam)
    *n_beam = ctx->sp_n_clip[1];
  return 0;
}

// The two sampled clouds in ctx->sp_samp[0] (likelihood, n_s points) and ctx->sp_samp[1] (beam, n_b points, w = origin id)
// -> the context's ordered scan buffers, on the device: the ordering api_core.inl:order_scan does on the host — 30-bit
// Morton key from the cloud's minimum corner resp. squared range from the scan origin, stable sort — so both entry paths
// produce the same order and with it bit-identical results. Raises error flag 2 (ctx->cl_err) for a bad origin id.
int device_order_scans(mcl3dl_hip_ctx* ctx, size_t n_s, size_t n_b, const float* origins, size_t n_o)
{
  const size_t n_max = std::max<size_t>(std::max(n_s, n_b), 1);
  TRY(ensure(ctx, ctx->cl_key[0], sizeof(uint32_t) * (n_max + 1)));
  TRY(ensure(ctx, ctx->cl_key[1], sizeof(uint32_t) * (n_max + 1)));
  TRY(ensure(ctx, ctx->cl_val[0], sizeof(uint32_t) * (n_max + 1)));
  TRY(ensure(ctx, ctx->cl_val[1], sizeof(uint32_t) * (n_max + 1)));
  TRY(ensure(ctx, ctx->scan_perm, sizeof(uint32_t) * n_s));
  TRY(ensure(ctx, ctx->scan_lik, sizeof(float4) * n_s));
  TRY(ensure(ctx, ctx->scan_beam, sizeof(float4) * n_b));
  TRY(ensure(ctx, ctx->origins, sizeof(float4) * n_o));
  if (n_s)
  {
    const long long ns = static_cast<long long>(n_s);
    TRY(cloud_minmax(ctx, ctx->sp_samp[0].as<float4>(), ns, nullptr, nullptr));
    hipLaunchKernelGGL(order_morton_key_kernel, dim3(blocks_for(ns)), dim3(256), 0, ctx->stream,
                       ctx->sp_samp[0].as<float4>(), ns, ctx->cl_minmax.as<float>(), ctx->cl_key[0].as<uint32_t>(),
                       ctx->cl_val[0].as<uint32_t>());
    TRY(sort_pairs(ctx, ctx->cl_key[0].as<uint32_t>(), ctx->cl_key[1].as<uint32_t>(), ctx->cl_val[0].as<uint32_t>(),
                   ctx->cl_val[1].as<uint32_t>(), ns, 30));
    hipLaunchKernelGGL(order_apply_kernel, dim3(blocks_for(ns)), dim3(256), 0, ctx->stream, ctx->sp_samp[0].as<float4>(),
                       ctx->cl_val[1].as<uint32_t>(), ns, 1, ctx->scan_lik.as<float4>(), ctx->scan_perm.as<uint32_t>());
  }
  if (n_o)
  {
    ctx->h_scan.origins.resize(n_o);
    for (size_t i = 0; i < n_o; ++i)
      ctx->h_scan.origins[i] = make_float4(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2], 0.f);
    TRY(h2d(ctx, ctx->origins.p, ctx->h_scan.origins.data(), sizeof(float4) * n_o));
  }
  if (n_b)
  {
    const long long nb = static_cast<long long>(n_b);
    hipLaunchKernelGGL(order_range_key_kernel, dim3(blocks_for(nb)), dim3(256), 0, ctx->stream,
                       ctx->sp_samp[1].as<float4>(), nb, ctx->origins.as<float4>(), static_cast<uint32_t>(n_o),
                       ctx->cl_key[0].as<uint32_t>(), ctx->cl_val[0].as<uint32_t>(), ctx->cl_err.as<int>());
    TRY(sort_pairs(ctx, ctx->cl_key[0].as<uint32_t>(), ctx->cl_key[1].as<uint32_t>(), ctx->cl_val[0].as<uint32_t>(),
                   ctx->cl_val[1].as<uint32_t>(), nb, 32));
    hipLaunchKernelGGL(order_apply_kernel, dim3(blocks_for(nb)), dim3(256), 0, ctx->stream, ctx->sp_samp[1].as<float4>(),
                       ctx->cl_val[1].as<uint32_t>(), nb, 0, ctx->scan_beam.as<float4>(), static_cast<uint32_t*>(nullptr));
  }
  HIP_TRY(hipGetLastError());
  return 0;
}
}  // namespace
