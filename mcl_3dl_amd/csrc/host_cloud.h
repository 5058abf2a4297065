// host_cloud.h — part of the single translation unit mcl3dl_hip.hip: host-side drivers of cloud_kernels.h shared by the scan
// upload (api_core.inl) and the scan-preparation / map entry points (api_cloud.inl).
#pragma once

namespace
{
unsigned blocks_for(long long n)
{
  return static_cast<unsigned>((std::max<long long>(n, 1) + 255) / 256);
}

// ---- min / max --------------------------------------------------------------------------------------------------------
unsigned minmax_blocks(long long n)
{
  return static_cast<unsigned>(std::min<long long>(std::max<long long>((n + 1023) / 1024, 1), 256));
}

// where a min / max launch of `nb` work-groups leaves its partials and its result (ctx->cl_minmax: 6 floats + the number of
// finite points as uint64); the ticket the last work-group resets is zeroed once, when it is allocated
int minmax_out(mcl3dl_hip_ctx* ctx, unsigned nb, MinMaxOut* o)
{
  TRY(ensure(ctx, ctx->cl_blocks, (sizeof(float) * 6 + sizeof(unsigned)) * nb));
  TRY(ensure(ctx, ctx->cl_minmax, sizeof(float) * 6 + sizeof(unsigned long long)));
  if (!ctx->cl_ticket.p)
  {
    TRY(ensure(ctx, ctx->cl_ticket, 64));
    HIP_TRY(hipMemsetAsync(ctx->cl_ticket.p, 0, 64, ctx->stream));
  }
  o->block_out = ctx->cl_blocks.as<float>();
  o->block_cnt = reinterpret_cast<unsigned*>(o->block_out + 6 * static_cast<size_t>(nb));
  o->ticket = ctx->cl_ticket.as<unsigned>();
  o->out6 = ctx->cl_minmax.as<float>();
  o->out_cnt = reinterpret_cast<unsigned long long*>(o->out6 + 6);
  return 0;
}

// ctx->cl_minmax -> the host (one synchronisation)
int minmax_to_host(mcl3dl_hip_ctx* ctx, float* host6, unsigned long long* host_cnt)
{
  float h[8];
  TRY(d2h(ctx, h, ctx->cl_minmax.p, sizeof(float) * 6 + sizeof(unsigned long long)));
  TRY(sync_stream(ctx));
  if (host6)
    memcpy(host6, h, sizeof(float) * 6);
  if (host_cnt)
    memcpy(host_cnt, h + 6, sizeof(unsigned long long));
  return 0;
}

// min / max of the finite points of a device cloud -> ctx->cl_minmax (device) and, on request, the host. One launch.
int cloud_minmax(mcl3dl_hip_ctx* ctx, const float4* pts, long long n, float* host6, unsigned long long* host_cnt)
{
  const unsigned nb = minmax_blocks(n);
  MinMaxOut mm;
  TRY(minmax_out(ctx, nb, &mm));
  hipLaunchKernelGGL(cloud_minmax_kernel, dim3(nb), dim3(256), 0, ctx->stream, pts, n, mm);
  HIP_TRY(hipGetLastError());
  if (host6 || host_cnt)
    TRY(minmax_to_host(ctx, host6, host_cnt));
  return 0;
}

// ---- stable radix sort (sort_kernels.h) ---------------------------------------------------------------------------------
int sort_passes(int end_bit)
{
  return std::max(1, (end_bit + 7) / 8);
}

// Sorts n elements by the key `kg` describes (bits [0, end_bit)). fin == nullptr: the sorted (key, value) pairs land in
// ctx->cl_key[1] / ctx->cl_val[1]; otherwise the last pass writes fin->src_pts in sorted order to fin->out_pts (and the
// permutation to fin->out_perm). ctx->cl_key[0..1] / cl_val[0..1] are the work arrays (kg.keys / kg.vals may be
// cl_key[0] / cl_val[0]). One launch up to 2048 elements, two per pass up to 524 288, rocprim beyond.
template <int KEYMODE>
int radix_sort(mcl3dl_hip_ctx* ctx, const RsKeyGen& kg, long long n, int end_bit, const RsFinal* fin)
{
  if (n <= 0)
    return 0;
  if (n > 0x7fffffffLL)
    return ctx->fail(-3, "too many points to sort");
  const size_t cap = sizeof(uint32_t) * (static_cast<size_t>(n) + 1);
  for (int k = 0; k < 2; ++k)
  {
    TRY(ensure(ctx, ctx->cl_key[k], cap));
    TRY(ensure(ctx, ctx->cl_val[k], cap));
  }
  RsKeyGen g = kg;  // (ensure() may have moved the work arrays the caller named before growing them: callers size them first)
  uint32_t* key[2] = { ctx->cl_key[0].as<uint32_t>(), ctx->cl_key[1].as<uint32_t>() };
  uint32_t* val[2] = { ctx->cl_val[0].as<uint32_t>(), ctx->cl_val[1].as<uint32_t>() };
  const int n_pass = sort_passes(end_bit);
  const uint32_t mask = end_bit >= 32 ? 0xffffffffu : ((1u << end_bit) - 1u);  // only bits [0, end_bit) order the pairs
  const RsFinal f = fin ? *fin : RsFinal{ nullptr, nullptr, nullptr, 0 };
  const int ni = static_cast<int>(n);
  if (n <= RS_ONE_LAUNCH_MAX)
  {
    if (fin)
      hipLaunchKernelGGL((rs_sort_block_kernel<KEYMODE, true>), dim3(1), dim3(RS_THREADS), 0, ctx->stream, g, f, key[1], val[1],
                         key[0], val[0], ni, n_pass, mask);
    else
      hipLaunchKernelGGL((rs_sort_block_kernel<KEYMODE, false>), dim3(1), dim3(RS_THREADS), 0, ctx->stream, g, f, key[1], val[1],
                         key[0], val[0], ni, n_pass, mask);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (n <= RS_MAX_ELEMS)
  {
    // 1024 elements per work-group up to 65 536 (every CU of a quarter of the chip ranks one round), 4096 above
    const int rounds = n <= 65536 ? 1 : 4;
    const int elems = RS_THREADS * rounds;
    const unsigned nb = static_cast<unsigned>((n + elems - 1) / elems);
    TRY(ensure(ctx, ctx->rs_table, sizeof(uint32_t) * 256 * nb * n_pass));
    uint32_t* table = ctx->rs_table.as<uint32_t>();
    // the pairs ping-pong between the two array sets; start in the one that makes the LAST pass write set 1
    int cur = (n_pass & 1) ? 0 : 1;
    hipLaunchKernelGGL((rs_keygen_count_kernel<KEYMODE>), dim3(nb), dim3(RS_THREADS), 0, ctx->stream, g, key[cur], val[cur],
                       table, ni, elems, mask);
    for (int p = 0; p < n_pass; ++p)
    {
      const bool apply = fin && p + 1 == n_pass;
      if (p)
        hipLaunchKernelGGL(rs_hist_kernel, dim3(nb), dim3(RS_THREADS), 0, ctx->stream, key[cur], table, ni, elems, p, mask);
#define RS_LAUNCH_PASS(AP, RR)                                                                                               \
  hipLaunchKernelGGL((rs_pass_kernel<AP, RR>), dim3(nb), dim3(RS_THREADS), 0, ctx->stream, key[cur], val[cur], key[cur ^ 1], \
                     val[cur ^ 1], f, table, ni, p, n_pass, mask)
      if (rounds == 1)
      {
        if (apply)
          RS_LAUNCH_PASS(true, 1);
        else
          RS_LAUNCH_PASS(false, 1);
      }
      else
      {
        if (apply)
          RS_LAUNCH_PASS(true, 4);
        else
          RS_LAUNCH_PASS(false, 4);
      }
#undef RS_LAUNCH_PASS
      cur ^= 1;
    }
    HIP_TRY(hipGetLastError());
    return 0;
  }
  // whole maps: rocprim's device-wide radix sort (stable), keys made by a kernel of ours, result into set 1
  if (!(KEYMODE == RS_KEY_ARRAY && g.keys == key[0] && (g.vals == val[0])))
    hipLaunchKernelGGL((rs_keygen_kernel<KEYMODE>), dim3(blocks_for(n)), dim3(256), 0, ctx->stream, g, key[0], val[0], n);
  size_t bytes = 0;
  HIP_TRY(rocprim::radix_sort_pairs(nullptr, bytes, key[0], key[1], val[0], val[1], static_cast<size_t>(n), 0u,
                                    static_cast<unsigned>(std::min(end_bit, 32)), ctx->stream));
  TRY(ensure(ctx, ctx->sort_tmp, bytes));
  HIP_TRY(rocprim::radix_sort_pairs(ctx->sort_tmp.p, bytes, key[0], key[1], val[0], val[1], static_cast<size_t>(n), 0u,
                                    static_cast<unsigned>(std::min(end_bit, 32)), ctx->stream));
  if (fin)
    hipLaunchKernelGGL(rs_apply_kernel, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, f, val[1], n);
  HIP_TRY(hipGetLastError());
  return 0;
}

// stable ascending sort of the (key, value) pairs in ctx->cl_key[0] / cl_val[0] (n entries) -> ctx->cl_key[1] / cl_val[1]
int sort_pairs(mcl3dl_hip_ctx* ctx, long long n, int end_bit)
{
  RsKeyGen kg{};
  kg.keys = ctx->cl_key[0].as<uint32_t>();
  kg.vals = ctx->cl_val[0].as<uint32_t>();
  return radix_sort<RS_KEY_ARRAY>(ctx, kg, n, end_bit, nullptr);
}

int bits_for(unsigned long long max_value)
{
  int b = 1;
  while (b < 64 && (max_value >> b))
    ++b;
  return b;
}

// host xyz (+ label) -> device float4 cloud; with_minmax: its min / max / finite count land in ctx->cl_minmax (same launch)
int upload_cloud(mcl3dl_hip_ctx* ctx, const float* xyz, const uint32_t* label, size_t n, DevBuf& out, bool with_minmax = false)
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
  const long long nn = static_cast<long long>(n);
  if (with_minmax)
  {
    const unsigned nb = minmax_blocks(nn);
    MinMaxOut mm;
    TRY(minmax_out(ctx, nb, &mm));
    hipLaunchKernelGGL(cloud_pack_minmax_kernel, dim3(nb), dim3(256), 0, ctx->stream, ctx->cl_in_xyz.as<float>(), d_label, nn,
                       out.as<float4>(), mm);
  }
  else
  {
    hipLaunchKernelGGL(cloud_pack_kernel, dim3(blocks_for(nn)), dim3(256), 0, ctx->stream, ctx->cl_in_xyz.as<float>(), d_label,
                       nn, out.as<float4>());
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// PointCloud2 bytes -> device float4 cloud (mcl_3dl::fromROSMsg, point_conversion.h:64-92: x, y, z are required; a
// "label" field is used when present; "intensity" is not on the measurement path). The buffer must be tightly packed
// little-endian rows: n_points * point_step bytes (row_step == width * point_step; include/mcl3dl_hip.h).
int decode_cloud(mcl3dl_hip_ctx* ctx, const uint8_t* data, size_t n_points, uint32_t point_step, int off_x, int off_y,
                 int off_z, int off_label, DevBuf& out, bool with_minmax = false)
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
  const long long nn = static_cast<long long>(n_points);
  if (with_minmax)
  {
    const unsigned nb = minmax_blocks(nn);
    MinMaxOut mm;
    TRY(minmax_out(ctx, nb, &mm));
    hipLaunchKernelGGL(cloud_decode_minmax_kernel, dim3(nb), dim3(256), 0, ctx->stream, ctx->cl_in_xyz.as<uint8_t>(), nn,
                       point_step, off_x, off_y, off_z, off_label, out.as<float4>(), mm);
  }
  else
  {
    hipLaunchKernelGGL(cloud_decode_kernel, dim3(blocks_for(nn)), dim3(256), 0, ctx->stream, ctx->cl_in_xyz.as<uint8_t>(), nn,
                       point_step, off_x, off_y, off_z, off_label, out.as<float4>());
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// pcl::VoxelGrid<PointXYZIL> with only setLeafSize set (the node's three call sites): `in` (n points, device) ->
// `out` (one centroid per occupied leaf, in ascending leaf-index order). A leaf component <= 0 skips the filter.
// have_minmax: ctx->cl_minmax already holds the min / max / finite count of `in` (the kernel that produced it computed them).
// The number of leaves stays on the device (ctx->cl_counts, word 0) when *deferred comes back true — the caller reads it
// with its next synchronisation — and is in *n_out otherwise (pass-through cases).
int voxel_grid(mcl3dl_hip_ctx* ctx, const float4* in, size_t n, const float leaf[3], DevBuf& out, size_t* n_out, bool* deferred,
               bool have_minmax = false)
{
  *n_out = 0;
  *deferred = false;
  TRY(ensure(ctx, out, sizeof(float4) * std::max<size_t>(n, 1)));
  TRY(ensure(ctx, ctx->cl_counts, sizeof(uint32_t) * 4));
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
  if (have_minmax)
    TRY(minmax_to_host(ctx, mm, &n_finite));
  else
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
  const long long div0 = max_b[0] - vp.min_b[0] + 1, div1 = max_b[1] - vp.min_b[1] + 1, div2 = max_b[2] - vp.min_b[2] + 1;
  vp.mul[0] = 1;
  vp.mul[1] = static_cast<int>(div0);
  vp.mul[2] = static_cast<int>(div0 * div1);
  // leaf indices are below div0 div1 div2 (<= 2^31 by the check above, up to rounding of the two extent formulas);
  // non-finite points sort behind all of them
  const unsigned long long cells = static_cast<unsigned long long>(div0) * div1 * div2;
  int end_bit = 32;
  vp.nonfinite_key = 0xffffffffu;
  if (cells < 0x7fffffffULL)
  {
    vp.nonfinite_key = static_cast<uint32_t>(cells);
    end_bit = bits_for(cells);
  }
  const long long nn = static_cast<long long>(n), nf = static_cast<long long>(n_finite);
  RsKeyGen kg{};
  kg.pts = in;
  kg.vp = vp;
  TRY(radix_sort<RS_KEY_LEAF>(ctx, kg, nn, end_bit, nullptr));
  const unsigned nb = static_cast<unsigned>((nf + CP_BLOCK - 1) / CP_BLOCK);
  TRY(ensure(ctx, ctx->cl_scan_ws, sizeof(uint32_t) * (2 * static_cast<size_t>((nn + CP_BLOCK - 1) / CP_BLOCK) + 16)));
  hipLaunchKernelGGL(vg_head_count_kernel, dim3(nb), dim3(CP_THREADS), 0, ctx->stream, ctx->cl_key[1].as<uint32_t>(), nf,
                     ctx->cl_scan_ws.as<uint32_t>());
  hipLaunchKernelGGL(vg_centroid_compact_kernel, dim3(nb), dim3(CP_THREADS), 0, ctx->stream, in, ctx->cl_key[1].as<uint32_t>(),
                     ctx->cl_val[1].as<uint32_t>(), nf, ctx->cl_scan_ws.as<uint32_t>(), out.as<float4>(),
                     ctx->cl_counts.as<uint32_t>());
  HIP_TRY(hipGetLastError());
  *deferred = true;
  return 0;
}

// the same, with the leaf count brought to the host (one more synchronisation)
int voxel_grid_now(mcl3dl_hip_ctx* ctx, const float4* in, size_t n, const float leaf[3], DevBuf& out, size_t* n_out,
                   bool have_minmax = false)
{
  bool deferred = false;
  TRY(voxel_grid(ctx, in, n, leaf, out, n_out, &deferred, have_minmax));
  if (deferred)
  {
    uint32_t n_leaves = 0;
    TRY(d2h(ctx, &n_leaves, ctx->cl_counts.p, sizeof(uint32_t)));
    TRY(sync_stream(ctx));
    *n_out = n_leaves;
  }
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
                      size_t* n_full, size_t* n_lik, size_t* n_beam, bool have_minmax)
{
  // ctx->sp_raw holds the accumulated cloud. VoxelGrid -> both clips, the three counts delivered by ONE synchronisation
  // (the VoxelGrid's own min / max round trip is the only other one).
  ctx->sp_ready = false;
  bool deferred = false;
  TRY(voxel_grid(ctx, ctx->sp_raw.as<float4>(), n, leaf3, ctx->sp_full, &ctx->sp_n_full, &deferred, have_minmax));
  const size_t n_upper = deferred ? n : ctx->sp_n_full;  // what the clips may have to look at
  TRY(ensure(ctx, ctx->sp_clip[0], sizeof(float4) * std::max<size_t>(n_upper, 1)));
  TRY(ensure(ctx, ctx->sp_clip[1], sizeof(float4) * std::max<size_t>(n_upper, 1)));
  uint32_t counts[3] = { 0, 0, 0 };
  uint32_t* d_counts = ctx->cl_counts.as<uint32_t>();
  if (n_upper && (clip_lik4 || clip_beam4))
  {
    // clip_near_sq_ = clip_near * clip_near etc. in float, like refreshParameters (likelihood.cpp:58-59, beam.cpp:60-61)
    auto params = [](const float* c4) {
      ClipParams c{};
      if (c4)
      {
        c.near_sq = c4[0] * c4[0];
        c.far_sq = c4[1] * c4[1];
        c.z_min = c4[2];
        c.z_max = c4[3];
        c.enabled = 1;
      }
      return c;
    };
    const ClipParams c0 = params(clip_lik4), c1 = params(clip_beam4);
    const unsigned nb = static_cast<unsigned>((n_upper + CP_BLOCK - 1) / CP_BLOCK);
    DevBuf& cnt = ctx->cl_clip_scan[0];
    TRY(ensure(ctx, cnt, sizeof(uint32_t) * 2 * nb));
    const uint32_t* n_dev = deferred ? d_counts : nullptr;
    hipLaunchKernelGGL(clip2_count_kernel, dim3(nb), dim3(CP_THREADS), 0, ctx->stream, ctx->sp_full.as<float4>(), n_dev,
                       static_cast<long long>(n_upper), c0, c1, cnt.as<uint32_t>());
    hipLaunchKernelGGL(clip2_compact_kernel, dim3(nb), dim3(CP_THREADS), 0, ctx->stream, ctx->sp_full.as<float4>(), n_dev,
                       static_cast<long long>(n_upper), c0, c1, cnt.as<uint32_t>(), ctx->sp_clip[0].as<float4>(),
                       ctx->sp_clip[1].as<float4>(), d_counts + 1);
    HIP_TRY(hipGetLastError());
    TRY(d2h(ctx, counts, d_counts, sizeof(uint32_t) * 3));
    TRY(sync_stream(ctx));
  }
  else if (deferred)
  {
    TRY(d2h(ctx, counts, d_counts, sizeof(uint32_t)));
    TRY(sync_stream(ctx));
  }
  if (deferred)
    ctx->sp_n_full = counts[0];
  ctx->sp_n_clip[0] = counts[1];
  ctx->sp_n_clip[1] = counts[2];
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
// produce the same order and with it bit-identical results. Keys are made inside the sort's first pass and the points are
// written by its last one. have_minmax: ctx->cl_minmax holds the min corner of sp_samp[0] already. Raises error flag 2
// (*d_err, device memory) for a bad origin id.
__global__ void scan_install_kernel(const float4* __restrict__ src, long long n, float4* __restrict__ dst,
                                    uint32_t* __restrict__ perm)
{
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  float4 q = src[i];
  q.w = 0.f;
  dst[i] = q;
  perm[i] = static_cast<uint32_t>(i);
}

int device_order_scans(mcl3dl_hip_ctx* ctx, size_t n_s, size_t n_b, const float* origins, size_t n_o, bool have_minmax,
                       int* d_err)
{
  const size_t n_max = std::max<size_t>(std::max(n_s, n_b), 1);
  for (int k = 0; k < 2; ++k)
  {
    TRY(ensure(ctx, ctx->cl_key[k], sizeof(uint32_t) * (n_max + 1)));
    TRY(ensure(ctx, ctx->cl_val[k], sizeof(uint32_t) * (n_max + 1)));
  }
  TRY(ensure_scan_block(ctx, n_s, n_b, n_o));
  if (n_s)
  {
    const long long ns = static_cast<long long>(n_s);
    if (!have_minmax)
      TRY(cloud_minmax(ctx, ctx->sp_samp[0].as<float4>(), ns, nullptr, nullptr));
    // a scan whose terms will be replayed in the caller's order (strict_order 1, or 2 from strict_auto_min points) and that is
    // at least two chunks long: ordered chunk by chunk of the caller's order (same keys, same bounding box for every chunk)
    const size_t chunk = (ctx->strict_chunk >= 1024 && (ctx->strict_order == 1 || (ctx->strict_order == 2 && ns >= ctx->strict_auto_min)) &&
                          n_s >= 2 * static_cast<size_t>(ctx->strict_chunk)) ? static_cast<size_t>(ctx->strict_chunk) & ~static_cast<size_t>(255) : 0;
    if (ctx->scan_presorted)
    {
      // option scan_presorted: the caller holds its scans in the engine's order (mcl3dl_hip_scan_order_host) — installed as
      // they are, permutation = identity, no ordering launches
      hipLaunchKernelGGL(scan_install_kernel, dim3(blocks_for(ns)), dim3(256), 0, ctx->stream, ctx->sp_samp[0].as<float4>(), ns,
                         ctx->scan_lik.as<float4>(), ctx->scan_perm.as<uint32_t>());
    }
    else
    for (size_t first = 0; first < n_s; first += chunk ? chunk : n_s)
    {
      const size_t n = chunk ? std::min(chunk, n_s - first) : n_s;
      RsKeyGen kg{};
      kg.pts = ctx->sp_samp[0].as<float4>() + first;
      kg.min3 = ctx->cl_minmax.as<float>();
      const RsFinal fin{ ctx->sp_samp[0].as<float4>() + first, ctx->scan_lik.as<float4>() + first, ctx->scan_perm.as<uint32_t>() + first, 1 };
      TRY(radix_sort<RS_KEY_MORTON>(ctx, kg, static_cast<long long>(n), MCL3DL_MORTON_BITS, &fin));
    }
    ctx->scan_chunk = ctx->scan_presorted ? 0 : chunk;
  }
  if (n_o && origins)  // (origins == nullptr: the caller's kernel has put them into ctx->origins already)
  {
    ctx->h_scan.origins.resize(n_o);
    for (size_t i = 0; i < n_o; ++i)
      ctx->h_scan.origins[i] = make_float4(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2], 0.f);
    TRY(h2d(ctx, ctx->origins.p, ctx->h_scan.origins.data(), sizeof(float4) * n_o));
  }
  if (n_b)
  {
    RsKeyGen kg{};
    kg.pts = ctx->sp_samp[1].as<float4>();
    kg.origins = ctx->origins.as<float4>();
    kg.n_o = static_cast<uint32_t>(n_o);
    kg.error = d_err;
    const RsFinal fin{ ctx->sp_samp[1].as<float4>(), ctx->scan_beam.as<float4>(), nullptr, 0 };
    TRY(radix_sort<RS_KEY_RANGE>(ctx, kg, static_cast<long long>(n_b), 32, &fin));
  }
  HIP_TRY(hipGetLastError());
  return 0;
}
}  // namespace
