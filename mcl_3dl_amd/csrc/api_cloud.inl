// api_cloud.inl — included inside the extern "C" block of mcl3dl_hip.hip: SURVEY.md §8f-2 (scan preparation on the GPU)
// and §8f-4 (map from the wire format, map updates, matched / unmatched output). Device code: cloud_kernels.h.

int mcl3dl_hip_scan_begin(mcl3dl_hip_ctx* ctx, const float* xyz, const uint32_t* label, size_t n, const float* leaf3,
                          const float* clip_lik4, const float* clip_beam4, size_t* n_full, size_t* n_lik_clipped,
                          size_t* n_beam_clipped)
{
  if (!ctx)
    return -1;
  if (n && !xyz)
    return ctx->fail(-3, "null cloud");
  if (n > 0x7fffffffu)
    return ctx->fail(-3, "cloud too large");
  HIP_TRY(hipSetDevice(ctx->device));
  TRY(upload_cloud(ctx, xyz, label, n, ctx->sp_raw, true));
  return scan_begin_common(ctx, n, leaf3, clip_lik4, clip_beam4, n_full, n_lik_clipped, n_beam_clipped, true);
}

int mcl3dl_hip_scan_begin_pointcloud2(mcl3dl_hip_ctx* ctx, const uint8_t* data, size_t n_points, uint32_t point_step,
                                      int off_x, int off_y, int off_z, int off_label, uint32_t label_override,
                                      const float* leaf3, const float* clip_lik4, const float* clip_beam4, size_t* n_full,
                                      size_t* n_lik_clipped, size_t* n_beam_clipped)
{
  if (!ctx)
    return -1;
  if (n_points > 0x7fffffffu)
    return ctx->fail(-3, "cloud too large");
  HIP_TRY(hipSetDevice(ctx->device));
  // accumCloud overwrites every label with the index of the accumulated cloud (src/mcl_3dl.cpp:295-298): with one message
  // per update that index is label_override; pass 0xffffffff to keep the message's own labels
  (void)label_override;
  TRY(decode_cloud(ctx, data, n_points, point_step, off_x, off_y, off_z, label_override == 0xffffffffu ? off_label : -1,
                   ctx->sp_raw, true));
  if (label_override != 0xffffffffu && label_override != 0u)
    return ctx->fail(-3, "label_override must be 0 (single accumulated cloud) or 0xffffffff (keep the message's labels)");
  return scan_begin_common(ctx, n_points, leaf3, clip_lik4, clip_beam4, n_full, n_lik_clipped, n_beam_clipped, true);
}

int mcl3dl_hip_scan_finish(mcl3dl_hip_ctx* ctx, const uint32_t* idx_lik, size_t n_s, const uint32_t* idx_beam, size_t n_b,
                           const float* origins, size_t n_o)
{
  if (!ctx)
    return -1;
  if (!ctx->sp_ready)
    return ctx->fail(-5, "no prepared scan: call mcl3dl_hip_scan_begin first");
  if ((n_s && !idx_lik) || (n_b && (!idx_beam || !origins || n_o == 0)))
    return ctx->fail(-3, "null index / origin array");
  if (n_s > 0x7fffffffu || n_b > 0x7fffffffu)
    return ctx->fail(-3, "scan too large");
  // the sampler draws from a non-empty cloud only (point_cloud_uniform_sampler.h:63-64 returns an empty cloud otherwise)
  if ((n_s && ctx->sp_n_clip[0] == 0) || (n_b && ctx->sp_n_clip[1] == 0))
    return ctx->fail(-3, "indices given for an empty clipped cloud");
  HIP_TRY(hipSetDevice(ctx->device));
  // ---- gather the drawn points of both models (one launch, with the likelihood sample's min corner), then order both
  // scans on the device (same keys and stable order as the host path). ONE host-to-device copy carries the (zeroed) error
  // word and both index arrays: [error, pad x3][idx_lik n_s][idx_beam n_b]
  TRY(ensure(ctx, ctx->sp_samp[0], sizeof(float4) * std::max<size_t>(n_s, 1)));
  TRY(ensure(ctx, ctx->sp_samp[1], sizeof(float4) * std::max<size_t>(n_b, 1)));
  const size_t idx_bytes = 16 + sizeof(uint32_t) * (n_s + n_b);
  TRY(ensure(ctx, ctx->cl_idx, idx_bytes));
  int* d_err = ctx->cl_idx.as<int>();
  const uint32_t* d_idx_lik = ctx->cl_idx.as<uint32_t>() + 4;
  const uint32_t* d_idx_beam = d_idx_lik + n_s;
  {
    char* st = idx_bytes <= STAGE_MAX_COPY ? static_cast<char*>(stage_alloc(ctx, idx_bytes)) : nullptr;
    if (st)
    {
      memset(st, 0, 16);
      if (n_s)
        memcpy(st + 16, idx_lik, sizeof(uint32_t) * n_s);
      if (n_b)
        memcpy(st + 16 + sizeof(uint32_t) * n_s, idx_beam, sizeof(uint32_t) * n_b);
      HIP_TRY(hipMemcpyAsync(ctx->cl_idx.p, st, idx_bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    else
    {
      HIP_TRY(hipMemsetAsync(ctx->cl_idx.p, 0, 16, ctx->stream));
      if (n_s)
        TRY(h2d(ctx, const_cast<uint32_t*>(d_idx_lik), idx_lik, sizeof(uint32_t) * n_s));
      if (n_b)
        TRY(h2d(ctx, const_cast<uint32_t*>(d_idx_beam), idx_beam, sizeof(uint32_t) * n_b));
    }
  }
  if (n_s + n_b)
  {
    const long long na = static_cast<long long>(n_s), nb = static_cast<long long>(n_b);
    const unsigned blocks = minmax_blocks(na + nb);
    MinMaxOut mm;
    TRY(minmax_out(ctx, blocks, &mm));
    hipLaunchKernelGGL(gather2_minmax_kernel, dim3(blocks), dim3(256), 0, ctx->stream, ctx->sp_clip[0].as<float4>(),
                       static_cast<long long>(ctx->sp_n_clip[0]), d_idx_lik, na, ctx->sp_samp[0].as<float4>(),
                       ctx->sp_clip[1].as<float4>(), static_cast<long long>(ctx->sp_n_clip[1]), d_idx_beam, nb,
                       ctx->sp_samp[1].as<float4>(), d_err, mm);
  }
  TRY(device_order_scans(ctx, n_s, n_b, origins, n_o, true, d_err));
  HIP_TRY(hipGetLastError());
  int err = 0;
  TRY(d2h(ctx, &err, d_err, sizeof(int)));
  TRY(sync_stream(ctx));
  if (err == 1)
    return ctx->fail(-3, "a sample index is outside the clipped cloud");
  if (err == 2)
    return ctx->fail(-3, "a beam point names an origin that was not given");
  ctx->sp_n_samp[0] = n_s;
  ctx->sp_n_samp[1] = n_b;
  if (n_b > ctx->pow_table_len)
    ctx->pow_table_dirty = true;
  if (n_s != ctx->n_s || n_b != ctx->n_b || n_o != ctx->n_o || !ctx->has_scan)
    ++ctx->generation;
  ctx->n_s = n_s;
  ctx->n_b = n_b;
  ctx->n_o = n_o;
  ctx->has_scan = true;
  return 0;
}

int mcl3dl_hip_scan_download(mcl3dl_hip_ctx* ctx, int which, float* xyz, uint32_t* label, size_t capacity, size_t* n)
{
  if (!ctx)
    return -1;
  if (!ctx->sp_ready)
    return ctx->fail(-5, "no prepared scan");
  HIP_TRY(hipSetDevice(ctx->device));
  const float4* src = nullptr;
  size_t cnt = 0;
  switch (which)
  {
    case 0: src = ctx->sp_full.as<float4>(); cnt = ctx->sp_n_full; break;
    case 1: src = ctx->sp_clip[0].as<float4>(); cnt = ctx->sp_n_clip[0]; break;
    case 2: src = ctx->sp_clip[1].as<float4>(); cnt = ctx->sp_n_clip[1]; break;
    case 3: src = ctx->sp_samp[0].as<float4>(); cnt = ctx->sp_n_samp[0]; break;
    case 4: src = ctx->sp_samp[1].as<float4>(); cnt = ctx->sp_n_samp[1]; break;
    default: return ctx->fail(-3, "which must be 0..4");
  }
  if (n)
    *n = cnt;
  if (!xyz)
    return 0;
  if (capacity < cnt)
    return ctx->fail(-3, "capacity %zu < %zu points", capacity, cnt);
  return download_cloud(ctx, src, cnt, xyz, label);
}

int mcl3dl_hip_sort_pairs(mcl3dl_hip_ctx* ctx, const uint32_t* keys, const uint32_t* vals, size_t n, int end_bit,
                          uint32_t* out_keys, uint32_t* out_vals)
{
  if (!ctx)
    return -1;
  if (n == 0)
    return 0;
  if (!keys || !out_keys || !out_vals || end_bit < 1 || end_bit > 32 || n > 0x7fffffffu)
    return ctx->fail(-3, "bad arguments to sort_pairs");
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t bytes = sizeof(uint32_t) * n;
  for (int k = 0; k < 2; ++k)
  {
    TRY(ensure(ctx, ctx->cl_key[k], bytes + 4));
    TRY(ensure(ctx, ctx->cl_val[k], bytes + 4));
  }
  HIP_TRY(hipMemcpyAsync(ctx->cl_key[0].p, keys, bytes, hipMemcpyHostToDevice, ctx->stream));
  if (vals)
  {
    HIP_TRY(hipMemcpyAsync(ctx->cl_val[0].p, vals, bytes, hipMemcpyHostToDevice, ctx->stream));
    TRY(sort_pairs(ctx, static_cast<long long>(n), end_bit));
  }
  else
  {
    RsKeyGen kg{};
    kg.keys = ctx->cl_key[0].as<uint32_t>();
    TRY(radix_sort<RS_KEY_ARRAY>(ctx, kg, static_cast<long long>(n), end_bit, nullptr));
  }
  HIP_TRY(hipMemcpyAsync(out_keys, ctx->cl_key[1].p, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(out_vals, ctx->cl_val[1].p, bytes, hipMemcpyDeviceToHost, ctx->stream));
  return sync_stream(ctx);
}

// ---- §8f-4: the map from the wire format, map updates, matched / unmatched output ------------------------------------
namespace
{
// device cloud (n points) -> the context's host copy of the map, appended behind `keep` existing points
int map_from_device(mcl3dl_hip_ctx* ctx, const float4* src, size_t n, size_t keep)
{
  std::vector<float> xyz(3 * n);
  std::vector<uint32_t> lab(n);
  TRY(download_cloud(ctx, src, n, xyz.data(), lab.data()));
  ctx->map_dev_valid = false;
  ctx->map_xyz.resize(3 * keep);
  ctx->map_label.resize(keep);
  ctx->map_xyz.insert(ctx->map_xyz.end(), xyz.begin(), xyz.end());
  ctx->map_label.insert(ctx->map_label.end(), lab.begin(), lab.end());
  return 0;
}

int install_base_map(mcl3dl_hip_ctx* ctx, size_t n_in, const float* leaf3, uint64_t stamp, const float* dist_weight,
                     size_t* n_map)
{
  size_t n_out = 0;
  TRY(voxel_grid_now(ctx, ctx->sp_raw.as<float4>(), n_in, leaf3, ctx->sp_full, &n_out));
  if (n_out == 0)
    return ctx->fail(-3, "empty map");
  if (n_out > 0xfffffff0u)
    return ctx->fail(-3, "map too large (index must fit 32 bits)");
  TRY(map_from_device(ctx, ctx->sp_full.as<float4>(), n_out, 0));
  ++ctx->generation;
  ctx->stamp = stamp;
  ctx->has_weight = dist_weight != nullptr;
  for (int a = 0; a < 3; ++a)
    ctx->weight[a] = dist_weight ? dist_weight[a] : 1.0f;
  ctx->has_map = true;
  ctx->lik_dirty = ctx->cand_dirty = ctx->dda_dirty = true;
  ctx->lik_base_dirty = true;
  ctx->n_base = n_out;
  ctx->sp_ready = false;  // sp_full was borrowed
  if (n_map)
    *n_map = n_out;
  return 0;
}

int install_map_update(mcl3dl_hip_ctx* ctx, size_t n_in, const float* leaf3, uint64_t stamp, size_t* n_map, double* stats5)
{
  if (!ctx->has_map)
    return ctx->fail(-5, "no map: call mcl3dl_hip_set_map first");
  const size_t n_base = ctx->n_base ? ctx->n_base : ctx->map_xyz.size() / 3;
  // the update that is being replaced, rescaled like the index holds it
  std::vector<float4> old_update;
  const size_t n_old_total = ctx->map_xyz.size() / 3;
  TRY(rescaled_points(ctx, n_base, n_old_total - n_base, old_update, nullptr, nullptr));
  size_t n_out = 0;
  TRY(voxel_grid_now(ctx, ctx->sp_raw.as<float4>(), n_in, leaf3, ctx->sp_full, &n_out));
  ctx->sp_ready = false;
  if (n_base + n_out > 0xfffffff0u)
    return ctx->fail(-3, "map too large (index must fit 32 bits)");
  TRY(map_from_device(ctx, ctx->sp_full.as<float4>(), n_out, n_base));  // pc_map2 = pc_map + pc_update
  ++ctx->generation;
  ctx->n_base = n_base;
  ctx->stamp = stamp;
  ctx->lik_dirty = true;  // the cell grid (matched / unmatched, lik_index 0) is rebuilt on next use: a linear-time build
  // the DDA grid keeps its arrays when the update stays inside the bounds it was laid out for: the update's points replace the
  // previous overlay (host_grid_builders.h:dda_overlay_apply). Otherwise it is rebuilt on next use.
  bool dda_kept = false;
  if (!ctx->dda_dirty && ctx->dda_overlay && ctx->dda_overlay_ok)
  {
    const DdaGrid& d = ctx->dg;
    const float* u = ctx->map_xyz.data() + 3 * n_base;
    bool inside = true;
    for (size_t i = 0; i < n_out && inside; ++i)
      inside = u[3 * i] >= d.min_x && u[3 * i] <= d.max_x && u[3 * i + 1] >= d.min_y && u[3 * i + 1] <= d.max_y &&
               u[3 * i + 2] >= d.min_z && u[3 * i + 2] <= d.max_z;  // (a NaN fails every comparison)
    if (inside)
    {
      TRY(dda_overlay_apply(ctx, ctx->sp_full.as<float4>(), n_out, n_base));
      ++ctx->dda_overlay_updates;
      dda_kept = true;
    }
  }
  if (!dda_kept)
    ctx->dda_dirty = true;
  TRY(update_cand_grid(ctx, n_base, old_update, stats5));
  if (n_map)
    *n_map = n_base + n_out;
  return 0;
}
}  // namespace

int mcl3dl_hip_set_map_pointcloud2(mcl3dl_hip_ctx* ctx, const uint8_t* data, size_t n_points, uint32_t point_step, int off_x,
                                   int off_y, int off_z, int off_label, const float* leaf3, uint64_t stamp,
                                   const float* dist_weight, size_t* n_map)
{
  if (!ctx)
    return -1;
  HIP_TRY(hipSetDevice(ctx->device));
  TRY(decode_cloud(ctx, data, n_points, point_step, off_x, off_y, off_z, off_label, ctx->sp_raw));
  return install_base_map(ctx, n_points, leaf3, stamp, dist_weight, n_map);
}

int mcl3dl_hip_set_map_downsampled(mcl3dl_hip_ctx* ctx, const float* xyz, const uint32_t* label, size_t n,
                                   const float* leaf3, uint64_t stamp, const float* dist_weight, size_t* n_map)
{
  if (!ctx)
    return -1;
  if (!xyz || n == 0)
    return ctx->fail(-3, "empty map");
  HIP_TRY(hipSetDevice(ctx->device));
  TRY(upload_cloud(ctx, xyz, label, n, ctx->sp_raw));
  return install_base_map(ctx, n, leaf3, stamp, dist_weight, n_map);
}

int mcl3dl_hip_map_update(mcl3dl_hip_ctx* ctx, const float* xyz, const uint32_t* label, size_t n, const float* leaf3,
                          uint64_t stamp, size_t* n_map, double* stats5)
{
  if (!ctx)
    return -1;
  if (n && !xyz)
    return ctx->fail(-3, "null cloud");
  HIP_TRY(hipSetDevice(ctx->device));
  TRY(upload_cloud(ctx, xyz, label, n, ctx->sp_raw));
  return install_map_update(ctx, n, leaf3, stamp, n_map, stats5);
}

int mcl3dl_hip_map_update_pointcloud2(mcl3dl_hip_ctx* ctx, const uint8_t* data, size_t n_points, uint32_t point_step,
                                      int off_x, int off_y, int off_z, int off_label, const float* leaf3, uint64_t stamp,
                                      size_t* n_map, double* stats5)
{
  if (!ctx)
    return -1;
  HIP_TRY(hipSetDevice(ctx->device));
  TRY(decode_cloud(ctx, data, n_points, point_step, off_x, off_y, off_z, off_label, ctx->sp_raw));
  return install_map_update(ctx, n_points, leaf3, stamp, n_map, stats5);
}

int mcl3dl_hip_map_download(mcl3dl_hip_ctx* ctx, float* xyz, uint32_t* label, size_t capacity, size_t* n)
{
  if (!ctx)
    return -1;
  const size_t cnt = ctx->map_xyz.size() / 3;
  if (n)
    *n = cnt;
  if (!xyz)
    return 0;
  if (capacity < cnt)
    return ctx->fail(-3, "capacity %zu < %zu points", capacity, cnt);
  memcpy(xyz, ctx->map_xyz.data(), sizeof(float) * 3 * cnt);
  if (label)
    memcpy(label, ctx->map_label.data(), sizeof(uint32_t) * cnt);
  return 0;
}

int mcl3dl_hip_match_split(mcl3dl_hip_ctx* ctx, const float* pose7, const float* xyz, size_t n, float unmatch_dist,
                           double match_dist, float* out_matched_xyz, size_t cap_matched, size_t* n_matched,
                           float* out_unmatched_xyz, size_t cap_unmatched, size_t* n_unmatched)
{
  if (!ctx)
    return -1;
  if (!pose7 || !(unmatch_dist > 0.f))
    return ctx->fail(-3, "bad arguments to match_split");
  HIP_TRY(hipSetDevice(ctx->device));
  const float4* src = nullptr;
  if (xyz)
  {
    TRY(upload_cloud(ctx, xyz, nullptr, n, ctx->sp_raw));
    src = ctx->sp_raw.as<float4>();
  }
  else
  {
    if (!ctx->sp_ready)
      return ctx->fail(-5, "no prepared scan: pass points or call mcl3dl_hip_scan_begin first");
    src = ctx->sp_full.as<float4>();  // pc_local_full, src/mcl_3dl.cpp:771-772
    n = ctx->sp_n_full;
  }
  if (n_matched)
    *n_matched = 0;
  if (n_unmatched)
    *n_unmatched = 0;
  if (n == 0)
    return 0;
  TRY(ensure_structures(ctx, true, false, true));  // the cell-sorted map
  const float cell = 1.0f / ctx->lg.inv_cell;
  const int reach = static_cast<int>(std::ceil(unmatch_dist / cell)) + 1;
  if (reach > 64)
    return ctx->fail(-3, "radius %.3g is more than 64 cells of the map index", unmatch_dist);
  const long long nn = static_cast<long long>(n);
  TRY(ensure(ctx, ctx->ms_xyz, sizeof(float4) * n));
  TRY(ensure(ctx, ctx->ms_out, sizeof(float4) * n));
  TRY(ensure(ctx, ctx->ms_flag[0], sizeof(uint32_t) * (n + 2)));
  TRY(ensure(ctx, ctx->ms_flag[1], sizeof(uint32_t) * (n + 2)));
  TRY(ensure(ctx, ctx->cl_scan_ws, sizeof(uint32_t) * (n / 1023 + 16)));
  const Quat rot = qnormalized(Quat{ pose7[3], pose7[4], pose7[5], pose7[6] });  // state_6dof.h:217
  const float r2 = static_cast<float>(static_cast<double>(unmatch_dist) * static_cast<double>(unmatch_dist));
  const double match_sq = match_dist * match_dist;  // src/mcl_3dl.cpp:778
  hipLaunchKernelGGL(match_split_kernel, dim3(blocks_for(nn + 1)), dim3(256), 0, ctx->stream, src, nn,
                     Vec3f{ pose7[0], pose7[1], pose7[2] }, rot, ctx->lg, lik_params(ctx), r2, reach, match_sq,
                     ctx->ms_xyz.as<float4>(), ctx->ms_flag[0].as<uint32_t>(), ctx->ms_flag[1].as<uint32_t>());
  size_t counts[2] = { 0, 0 };
  float* outs[2] = { out_matched_xyz, out_unmatched_xyz };
  const size_t caps[2] = { cap_matched, cap_unmatched };
  // Both compactions by one kernel that writes the xyz triples where the caller reads them — page-locked memory: the
  // caller's own arrays when they come from mcl3dl_hip_host_alloc, a staging block otherwise — and ONE polled completion
  // (round 3: two rounds of scan + count read-back + compaction + unpack kernel + D2H copy, four synchronisations: 0.35 ms
  // for 35 k points; the kernels are ~20 us of it)
  if (ctx->zero_copy() && ctx->poll_mode())
  {
    const size_t want[2] = { outs[0] ? std::min(caps[0], n) : 0, outs[1] ? std::min(caps[1], n) : 0 };
    const bool own[2] = { want[0] && ctx->is_pinned(outs[0], 12 * want[0]), want[1] && ctx->is_pinned(outs[1], 12 * want[1]) };
    const size_t need = 64 + (own[0] ? 0 : 12 * want[0]) + (own[1] ? 0 : 12 * want[1]) + 32;
    char* blk = need <= STAGE_MAX_COPY ? static_cast<char*>(stage_alloc(ctx, need)) : nullptr;
    if (blk)
    {
      uint32_t* h_counts = reinterpret_cast<uint32_t*>(blk);
      float* dst[2] = { nullptr, nullptr };
      size_t off = 64;
      for (int k = 0; k < 2; ++k)
        if (want[k])
        {
          dst[k] = own[k] ? outs[k] : reinterpret_cast<float*>(blk + off);
          if (!own[k])
            off += (12 * want[k] + 15) & ~static_cast<size_t>(15);
        }
      uint32_t* f0 = ctx->ms_flag[0].as<uint32_t>();
      uint32_t* f1 = ctx->ms_flag[1].as<uint32_t>();
      TRY(device_exclusive_scan_ws(ctx, f0, nn + 1, ctx->cl_scan_ws.as<uint32_t>()));
      TRY(device_exclusive_scan_ws(ctx, f1, nn + 1, ctx->cl_scan_ws.as<uint32_t>()));
      hipLaunchKernelGGL(match_emit_kernel, dim3(blocks_for(nn)), dim3(256), 0, ctx->stream, ctx->ms_xyz.as<float4>(), f0, f1, nn,
                         static_cast<unsigned long long>(want[0]), static_cast<unsigned long long>(want[1]), dst[0], dst[1],
                         h_counts);
      HIP_TRY(hipGetLastError());
      // (the staged triples are handed over by hand below: their size is only known once the counts are here)
      const uint32_t* h_counts_c = h_counts;
      const float* src[2] = { dst[0], dst[1] };
      TRY(sync_stream(ctx, true));   // (recycles the staging block for later calls; nothing overwrites it before we return)
      for (int k = 0; k < 2; ++k)
      {
        counts[k] = h_counts_c[k];
        if (outs[k] && caps[k] < counts[k])
          return ctx->fail(-3, "output capacity %zu < %zu points", caps[k], counts[k]);
        if (outs[k] && !own[k] && counts[k])
          memcpy(outs[k], src[k], 12 * counts[k]);
      }
      if (n_matched)
        *n_matched = counts[0];
      if (n_unmatched)
        *n_unmatched = counts[1];
      return 0;
    }
  }
  for (int k = 0; k < 2; ++k)
  {
    uint32_t* flag = ctx->ms_flag[k].as<uint32_t>();
    TRY(device_exclusive_scan_ws(ctx, flag, nn + 1, ctx->cl_scan_ws.as<uint32_t>()));
    uint32_t cnt = 0;
    TRY(d2h(ctx, &cnt, flag + nn, sizeof(uint32_t)));
    hipLaunchKernelGGL(compact_kernel, dim3(blocks_for(nn)), dim3(256), 0, ctx->stream, ctx->ms_xyz.as<float4>(), flag, nn,
                       ctx->ms_out.as<float4>());
    TRY(sync_stream(ctx));
    counts[k] = cnt;
    if (outs[k])
    {
      if (caps[k] < cnt)
        return ctx->fail(-3, "output capacity %zu < %u points", caps[k], cnt);
      TRY(download_cloud(ctx, ctx->ms_out.as<float4>(), cnt, outs[k], nullptr));
    }
  }
  if (n_matched)
    *n_matched = counts[0];
  if (n_unmatched)
    *n_unmatched = counts[1];
  return 0;
}
