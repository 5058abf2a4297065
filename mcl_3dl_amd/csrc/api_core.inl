// api_core.inl — included inside the extern "C" block of mcl3dl_hip.hip: lifecycle, map / parameters, scan upload,
// the device-resident and host-buffer forms of the measurement update, ray and nearest-neighbour queries.
int mcl3dl_hip_abi_version(void)
{
  return MCL3DL_HIP_ABI_VERSION;
}

int mcl3dl_hip_device_count(void)
{
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess)
  {
    (void)hipGetLastError();
    return 0;
  }
  return count > 0 ? count : 0;
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
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0)
      ctx->n_cus = cus;
  }
  beam_refresh(ctx);
  // MCL3DL_HIP_OPTIONS="name=value,name=value": tuning knobs (mcl3dl_hip_set_option) for a deployment that cannot change the
  // code that creates the context — and for running the whole test suite on a non-default kernel selection. A bad entry
  // fails the creation (a silently ignored knob is worse than no context).
  if (const char* env = getenv("MCL3DL_HIP_OPTIONS"))
  {
    std::string all(env);
    size_t pos = 0;
    while (pos < all.size())
    {
      size_t end = all.find_first_of(",;", pos);
      if (end == std::string::npos)
        end = all.size();
      std::string item = all.substr(pos, end - pos);
      pos = end + 1;
      const size_t first = item.find_first_not_of(" \t"), last = item.find_last_not_of(" \t");
      if (first == std::string::npos)
        continue;
      item = item.substr(first, last - first + 1);
      const size_t eq = item.find('=');
      char* tail = nullptr;
      const double v = eq == std::string::npos ? 0.0 : strtod(item.c_str() + eq + 1, &tail);
      if (eq == std::string::npos || eq == 0 || tail == item.c_str() + eq + 1 || *tail != '\0' ||
          mcl3dl_hip_set_option(ctx, item.substr(0, eq).c_str(), v) != 0)
      {
        fprintf(stderr, "mcl3dl_hip_create: MCL3DL_HIP_OPTIONS entry '%s' rejected (%s)\n", item.c_str(), ctx->err.c_str());
        mcl3dl_hip_destroy(ctx);
        return -3;
      }
    }
  }
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
  // (on THIS context's device and behind its own synchronisation: in a device group the current device can be another
  // rank's, and replay kernels may still read the term buffers freed below — ADVICE round 5)
  if (ctx->replay_stream)
  {
    (void)hipStreamSynchronize(ctx->replay_stream);
    (void)hipStreamDestroy(ctx->replay_stream);
  }
  for (int k = 0; k < 2; ++k)
  {
    if (ctx->ev_tiled[k])
      (void)hipEventDestroy(ctx->ev_tiled[k]);
    if (ctx->ev_replay[k])
      (void)hipEventDestroy(ctx->ev_replay[k]);
  }
  for (const EventPair& ep : ctx->pending)
  {
    (void)hipEventDestroy(ep.start);
    (void)hipEventDestroy(ep.stop);
  }
  for (hipEvent_t e : ctx->free_events)
    (void)hipEventDestroy(e);
  for (const mcl3dl_hip_ctx::ScratchBlk& b : ctx->scratch)
    (void)hipFree(b.p);
  for (const mcl3dl_hip_ctx::StageChunk& ch : ctx->stage)
    (void)hipHostFree(ch.p);
  for (const mcl3dl_hip_ctx::PinnedBlock& b : ctx->pinned)
    (void)hipHostFree(b.p);
  if (ctx->done_flag)
    (void)hipHostFree(const_cast<unsigned*>(ctx->done_flag));
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
  ctx->map_dev_valid = false;
  if (label)
    ctx->map_label.assign(label, label + n_m);
  else
    ctx->map_label.assign(n_m, 0u);
  ctx->stamp = stamp;
  ctx->has_weight = dist_weight != nullptr;
  for (int a = 0; a < 3; ++a)
    ctx->weight[a] = dist_weight ? dist_weight[a] : 1.0f;
  ctx->has_map = true;
  ctx->n_base = n_m;
  ctx->lik_dirty = true;
  ctx->lik_base_dirty = true;
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
    ctx->lik_dirty = ctx->cand_dirty = ctx->lik_base_dirty = true;  // cell / voxel edges follow the search radius
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
  if (!(dda_grid_size > 0.f))
    return ctx->fail(-3, "dda_grid_size must be > 0");
  // The adapter pushes its parameter block before every batched launch and every getBeamStatus call: only a value that
  // really changed may invalidate anything. The raycaster is re-created (reference: refreshParameters, beam.cpp:69-79)
  // only when one of ITS constructor arguments moved; the reference caches the voxel map by stamp the same way
  // (raycast_using_dda.h:164-171).
  const bool dda_changed = map_grid_x != ctx->map_grid[0] || map_grid_y != ctx->map_grid[1] ||
                           map_grid_z != ctx->map_grid[2] || dda_grid_size != ctx->dda_grid_size ||
                           ray_angle_half != ctx->ray_angle_half || hit_range != ctx->hit_range;
  const bool derived_changed = hit_range != ctx->hit_range || beam_likelihood_min != ctx->beam_likelihood_min ||
                               num_points != ctx->beam_num_points || ang_total_ref != ctx->ang_total_ref;
  const bool other_changed = filter_label_max != ctx->filter_label_max ||
                             (add_penalty_short_only_mode ? 1 : 0) != ctx->short_only;
  if (!dda_changed && !derived_changed && !other_changed)
    return 0;
  ++ctx->generation;
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
  if (dda_changed)
    ctx->dda_dirty = true;
  if (derived_changed)
    beam_refresh(ctx);
  return 0;
}

// Host-side ordering of one update's scans (no device work): shared by the single-context upload and by a device group,
// which orders once and pushes the same arrays to every GPU. Returns 0, or -3 with `err` filled.
static int order_scan(std::string& err, const float* scan_lik_xyz, size_t n_s, const float* scan_beam_xyz,
                      const uint32_t* scan_beam_origin, size_t n_b, const float* origins, size_t n_o, OrderedScan& o,
                      bool presorted = false)
{
  if ((n_s && !scan_lik_xyz) || (n_b && (!scan_beam_xyz || !origins || n_o == 0)))
  {
    err = "null scan array";
    return -3;
  }
  if (n_s > 0x7fffffffu || n_b > 0x7fffffffu)
  {
    err = "scan too large";
    return -3;
  }
  // likelihood scan: spatial (Morton) order. The score is a sum, so the order only changes which lanes work together.
  std::vector<float4>& lik = o.lik;
  lik.resize(n_s);
  o.perm.resize(n_s);
  if (n_s && presorted)
  {
    // option scan_presorted: the caller's order IS the engine's order
    for (size_t i = 0; i < n_s; ++i)
    {
      lik[i] = make_float4(scan_lik_xyz[3 * i], scan_lik_xyz[3 * i + 1], scan_lik_xyz[3 * i + 2], 0.f);
      o.perm[i] = static_cast<uint32_t>(i);
    }
  }
  else if (n_s)
  {
    // min / max over the finite points (like the device's cloud_minmax)
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (size_t i = 0; i < n_s; ++i)
    {
      const float* q = scan_lik_xyz + 3 * i;
      if (!(std::isfinite(q[0]) && std::isfinite(q[1]) && std::isfinite(q[2])))
        continue;
      for (int a = 0; a < 3; ++a)
      {
        mn[a] = std::min(mn[a], q[a]);
        mx[a] = std::max(mx[a], q[a]);
      }
    }
    // Morton key (10 bits per axis, 0.25 m cells; the MCL3DL_MORTON_BITS most significant bits the extent can set: cloud_keys.h) + LSD radix sort
    uint32_t cells = 0;
    for (int a = 0; a < 3; ++a)
    {
      const float e = (mx[a] - mn[a]) * 4.0f;
      const uint32_t c = (e >= 0.f) ? (e < 1023.f ? static_cast<uint32_t>(e) : 1023u) : 0u;
      cells = std::max(cells, c);
    }
    uint32_t bits = 0;
    while (bits < 32 && (cells >> bits) != 0)
      ++bits;
    const uint32_t drop = 3u * bits > MCL3DL_MORTON_BITS ? 3u * bits - MCL3DL_MORTON_BITS : 0u;
    std::vector<uint32_t>& idx = o.perm;
    std::vector<uint32_t>&key = o.key, &key2 = o.key2, &idx2 = o.idx2;
    key.resize(n_s);
    key2.resize(n_s);
    idx2.resize(n_s);
    for (size_t i = 0; i < n_s; ++i)
    {
      uint32_t c[3];
      for (int a = 0; a < 3; ++a)
      {
        const float f = (scan_lik_xyz[3 * i + a] - mn[a]) * 4.0f;
        c[a] = (f >= 0.f) ? (f < 1023.f ? static_cast<uint32_t>(f) : 1023u) : 0u;
      }
      key[i] = static_cast<uint32_t>(morton3(c[0], c[1], c[2])) >> drop;
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
  }
  // beam scan: ordered by range from its scan origin. A ray walks ~range/dda_grid voxels and (its end point being a
  // measured surface) ends near its last voxel, so the 64 rays of a wavefront finish together instead of idling behind the
  // longest one. The beam score is a count of penalised rays, so the order is free.
  std::vector<float4>& beam = o.beam;
  beam.resize(n_b);
  if (n_b)
  {
    std::vector<std::pair<float, uint32_t>>& keys = o.beam_keys;
    keys.resize(n_b);
    for (size_t i = 0; i < n_b; ++i)
    {
      const uint32_t og = scan_beam_origin ? scan_beam_origin[i] : 0u;
      if (og >= n_o)
      {
        char buf[160];
        snprintf(buf, sizeof(buf), "beam point %zu names origin %u but only %zu origins were given", i, og, n_o);
        err = buf;
        return -3;
      }
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
  std::vector<float4>& org = o.origins;
  org.resize(n_o);
  for (size_t i = 0; i < n_o; ++i)
    org[i] = make_float4(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2], 0.f);
  return 0;
}

// Ordered scans -> this context's device buffers. `o` must stay alive until the stream has been synchronised (copies above
// the staging limit read it directly). sync_at_end = false: the caller synchronises the stream itself before it returns.
static int push_scan(mcl3dl_hip_ctx* ctx, const OrderedScan& o, bool sync_at_end)
{
  const size_t n_s = o.lik.size(), n_b = o.beam.size(), n_o = o.origins.size();
  HIP_TRY(hipSetDevice(ctx->device));
  size_t off[4];
  TRY(ensure_scan_block(ctx, n_s, n_b, n_o, off));
  const void* src[4] = { o.perm.data(), o.lik.data(), o.beam.data(), o.origins.data() };
  const size_t bytes[4] = { sizeof(uint32_t) * n_s, sizeof(float4) * n_s, sizeof(float4) * n_b, sizeof(float4) * n_o };
  const size_t total = off[3] + bytes[3];
  void* staged = total <= STAGE_MAX_COPY ? stage_alloc(ctx, total) : nullptr;
  if (staged)
  {
    // the four arrays in the block's own layout, ONE copy (the gaps between the parts travel too: < 1 KB)
    for (int k = 0; k < 4; ++k)
      if (bytes[k])
        memcpy(static_cast<char*>(staged) + off[k], src[k], bytes[k]);
    HIP_TRY(hipMemcpyAsync(ctx->scan_block.p, staged, total, hipMemcpyHostToDevice, ctx->stream));
    ctx->stage_pending += total;
  }
  else
  {
    DevBuf* dst[4] = { &ctx->scan_perm, &ctx->scan_lik, &ctx->scan_beam, &ctx->origins };
    for (int k = 0; k < 4; ++k)
      TRY(h2d(ctx, dst[k]->p, src[k], bytes[k]));
  }
  if (sync_at_end)
    TRY(sync_stream(ctx));
  if (n_b > ctx->pow_table_len)
    ctx->pow_table_dirty = true;  // table[k] = beam_likelihood^k is a prefix property: a shorter scan reuses it
  if (n_s != ctx->n_s || n_b != ctx->n_b || n_o != ctx->n_o || !ctx->has_scan)
    ++ctx->generation;
  ctx->n_s = n_s;
  ctx->n_b = n_b;
  ctx->n_o = n_o;
  ctx->has_scan = true;
  return 0;
}

static int upload_scan_impl(mcl3dl_hip_ctx* ctx, const float* scan_lik_xyz, size_t n_s, const float* scan_beam_xyz,
                            const uint32_t* scan_beam_origin, size_t n_b, const float* origins, size_t n_o,
                            bool sync_at_end)
{
  if (!ctx)
    return -1;
  // Large scans are ordered on the device (host_cloud.h:device_order_scans: raw points up, min / key / stable radix sort /
  // gather there — same keys, same order, same bits as the host ordering, which costs ~0.1 ms of one core at 16 k points);
  // small ones on the host, where a dozen launches would cost more than the sort.
  if (ctx->scan_order_device > 0 && n_s + n_b >= static_cast<size_t>(ctx->scan_order_device))
  {
    if ((n_s && !scan_lik_xyz) || (n_b && (!scan_beam_xyz || !origins || n_o == 0)))
      return ctx->fail(-3, "null scan array");
    if (n_s > 0x7fffffffu || n_b > 0x7fffffffu)
      return ctx->fail(-3, "scan too large");
    if (scan_beam_origin)
      for (size_t i = 0; i < n_b; ++i)
        if (scan_beam_origin[i] >= n_o)
          return ctx->fail(-3, "beam point %zu names origin %u but only %zu origins were given", i, scan_beam_origin[i], n_o);
    HIP_TRY(hipSetDevice(ctx->device));
    TRY(ensure(ctx, ctx->cl_err, sizeof(int)));
    TRY(upload_cloud(ctx, scan_lik_xyz, nullptr, n_s, ctx->sp_samp[0], true));  // + the min corner of the Morton keys
    TRY(upload_cloud(ctx, scan_beam_xyz, scan_beam_origin, n_b, ctx->sp_samp[1]));
    TRY(device_order_scans(ctx, n_s, n_b, origins, n_o, n_s != 0, ctx->cl_err.as<int>()));
    if (sync_at_end)
      TRY(sync_stream(ctx));
    if (n_b > ctx->pow_table_len)
      ctx->pow_table_dirty = true;
    if (n_s != ctx->n_s || n_b != ctx->n_b || n_o != ctx->n_o || !ctx->has_scan)
      ++ctx->generation;
    ctx->n_s = n_s;
    ctx->n_b = n_b;
    ctx->n_o = n_o;
    ctx->has_scan = true;
    ctx->sp_n_samp[0] = n_s;
    ctx->sp_n_samp[1] = n_b;
    return 0;
  }
  std::string err;
  if (order_scan(err, scan_lik_xyz, n_s, scan_beam_xyz, scan_beam_origin, n_b, origins, n_o, ctx->h_scan, ctx->scan_presorted != 0) != 0)
    return ctx->fail(-3, "%s", err.c_str());
  return push_scan(ctx, ctx->h_scan, sync_at_end);
}

int mcl3dl_hip_upload_scan(mcl3dl_hip_ctx* ctx, const float* scan_lik_xyz, size_t n_s, const float* scan_beam_xyz,
                           const uint32_t* scan_beam_origin, size_t n_b, const float* origins, size_t n_o)
{
  return upload_scan_impl(ctx, scan_lik_xyz, n_s, scan_beam_xyz, scan_beam_origin, n_b, origins, n_o, true);
}

int mcl3dl_hip_scan_order_host(const float* scan_lik_xyz, size_t n_s, uint32_t* order)
{
  if (n_s == 0)
    return 0;
  if (!scan_lik_xyz || !order)
    return -3;
  std::string err;
  OrderedScan o;
  if (order_scan(err, scan_lik_xyz, n_s, nullptr, nullptr, 0, nullptr, 0, o) != 0)
    return -3;
  memcpy(order, o.perm.data(), sizeof(uint32_t) * n_s);
  return 0;
}

int mcl3dl_hip_scan_order(mcl3dl_hip_ctx* ctx, uint32_t* order, size_t n_s)
{
  if (!ctx)
    return -1;
  if (!ctx->has_scan)
    return ctx->fail(-5, "no scan installed");
  if (n_s != ctx->n_s)
    return ctx->fail(-3, "the likelihood scan holds %zu points, not %zu", ctx->n_s, n_s);
  if (n_s == 0)
    return 0;
  if (!order)
    return ctx->fail(-3, "null order array");
  HIP_TRY(hipSetDevice(ctx->device));
  TRY(d2h(ctx, order, ctx->scan_perm.p, sizeof(uint32_t) * n_s));
  TRY(sync_stream(ctx));
  if (ctx->scan_chunk)  // (ordered in chunks of the caller's order: the device holds indices relative to each chunk)
    for (size_t k = 0; k < n_s; ++k)
      order[k] += static_cast<uint32_t>((k / ctx->scan_chunk) * ctx->scan_chunk);
  return 0;
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
  if (world == 1 && pf_float_order(ctx, n_p))
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
bool pf_is_split(const mcl3dl_hip_ctx* ctx, size_t n_p)
{
  return !(n_p <= static_cast<size_t>(std::min(ctx->pf_fused_max, PF_FUSED_MAX)) && ctx->pf_fused);
}
// is pf::measure of n_p particles on this GPU the split form with the fp64 sum of the weights? Then launch_measure may leave the
// sum over the tiled kernel's per-tile partials to lik_pf_partial_kernel (LikTail).
bool pf_takes_tiles(const mcl3dl_hip_ctx* ctx, size_t n_p)
{
  const bool fused = n_p <= static_cast<size_t>(std::min(ctx->pf_fused_max, PF_FUSED_MAX)) && ctx->pf_fused;
  return !fused && !pf_float_order(ctx, n_p) && n_p <= static_cast<size_t>(1024) * PF_BLOCK;  // (one particle per thread of the grid)
}

// pf::measure on one GPU: the fused single-work-group kernel up to pf_fused_max particles (default 1024; the kernel takes up
// to PF_FUSED_MAX = 4096 — same bits as the split form), the split form beyond.
// ho (optional): page-locked arrays the last kernel writes the results to as well (PfEmit).
// The split form on one GPU with the fp64 sum of the weights is TWO launches since round 6 (five with lik_finalize_kernel in
// front until then): lik_pf_partial_kernel / pf_partial_kernel, then pf_apply_kernel whose every work-group runs pf_reduce_kernel's
// reduction itself. Same arithmetic in the same association as the launches apart (the multi-GPU protocol still runs them apart,
// the all-reduce between them): the same bits. Measured on one box, C2: 0.2344 -> 0.2303 ms per update, C3 0.3448 -> 0.3409
// (profiles/r06p_tail_ab.txt); the earlier forms of the idea that LOST are in profiles/r06o_pf_two_launch_ab.txt.
int pf_measure_single(mcl3dl_hip_ctx* ctx, float* d_weight, float* d_lik, float* d_beam, const float* d_extra,
                      float* d_ratio, size_t n_p, float* d_stats4, const PfEmit* ho = nullptr, const LikTail* tail = nullptr)
{
  const PfEmit emit = ho ? *ho : PfEmit{};
  const bool float_w = pf_float_order(ctx, n_p);
  // the beam score from the penalty counts on the way (beam_finalize_kernel's step; the kernel zeroes every counter it reads)
  const bool counts = tail && tail->beam_pending && (tail->pending || !float_w || !pf_is_split(ctx, n_p));
  const BeamCounts bc = counts ? BeamCounts{ ctx->penalty.as<unsigned>(), ctx->pow_table.as<float>(), ctx->beam_likelihood_min, d_beam }
                               : BeamCounts{ nullptr, nullptr, 0.0f, nullptr };
  if (tail && tail->pending)
  {
    // launch_measure left the tiled kernel's per-tile partials where they are: lik_finalize_kernel's sum and pf_partial_kernel's
    // product in one launch, one wavefront of pf_partial_kernel's blocks per work-group (pf_kernels.h)
    const int n_waves = static_cast<int>((n_p + 63) / 64), nb = pf_blocks(n_p);
    TRY(ensure(ctx, ctx->wnew, sizeof(float) * n_p));
    TRY(ensure(ctx, ctx->block_partials, sizeof(double) * 16 * nb));  // (whole blocks of four wavefront partials)
    const LikTiles lt{ ctx->lik_partial_sum.as<double>(), ctx->lik_partial_cnt.as<unsigned>(), tail->n_tiles, static_cast<int>(ctx->n_s),
                       d_lik, d_ratio, tail->beam_fill ? d_beam : static_cast<float*>(nullptr), bc };
    EventPair ep{};
    TRY(timing_begin(ctx, MCL3DL_KERNEL_PF, &ep));
    hipLaunchKernelGGL(lik_pf_partial_kernel, dim3(n_waves), dim3(256), 0, ctx->stream, lt, d_weight, d_beam, d_extra,
                       static_cast<int>(n_p), ctx->wnew.as<float>(), ctx->block_partials.as<double>());
    hipLaunchKernelGGL(pf_apply_kernel, dim3(nb), dim3(PF_BLOCK), 0, ctx->stream, d_weight, ctx->wnew.as<float>(),
                       static_cast<int>(n_p), 1, static_cast<const double*>(nullptr), d_stats4, emit, d_lik, d_ratio, d_beam,
                       ctx->block_partials.as<double>(), nb, n_waves, ctx->partial4.as<double>());
    if (counts)
      ctx->penalty_clean_n = n_p;  // (launched: the kernel zeroes every counter it reads)
    TRY(timing_end(ctx, ep));
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (tail && tail->beam_pending && !counts)  // (the float-order split form: the beam model's last step as a launch after all)
    hipLaunchKernelGGL(beam_finalize_kernel, dim3((static_cast<unsigned>(n_p) + 255) / 256), dim3(256), 0, ctx->stream,
                       ctx->penalty.as<unsigned>(), ctx->pow_table.as<float>(), ctx->beam_likelihood_min, d_beam,
                       static_cast<int>(n_p));
  if (n_p <= static_cast<size_t>(std::min(ctx->pf_fused_max, PF_FUSED_MAX)) && ctx->pf_fused)
  {
    TRY(ensure(ctx, ctx->wnew, sizeof(float) * n_p));
    EventPair ep{};
    TRY(timing_begin(ctx, MCL3DL_KERNEL_PF, &ep));
    hipLaunchKernelGGL(pf_fused_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_weight, d_lik, d_beam, d_extra, d_ratio,
                       static_cast<int>(n_p), ctx->wnew.as<float>(), ctx->partial4.as<double>(), d_stats4, emit, float_w ? 1 : 0, bc);
    if (counts)
      ctx->penalty_clean_n = n_p;  // (launched: the kernel zeroes every counter it reads)
    TRY(timing_end(ctx, ep));
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (!float_w)
  {
    const int nb = pf_blocks(n_p);
    TRY(ensure(ctx, ctx->wnew, sizeof(float) * n_p));
    TRY(ensure(ctx, ctx->block_partials, sizeof(double) * 4 * nb));
    EventPair ep{};
    TRY(timing_begin(ctx, MCL3DL_KERNEL_PF, &ep));
    hipLaunchKernelGGL(pf_partial_kernel, dim3(nb), dim3(PF_BLOCK), 0, ctx->stream, d_weight, d_lik, d_beam, d_extra, d_ratio,
                       static_cast<int>(n_p), ctx->wnew.as<float>(), ctx->block_partials.as<double>(), bc);
    hipLaunchKernelGGL(pf_apply_kernel, dim3(nb), dim3(PF_BLOCK), 0, ctx->stream, d_weight, ctx->wnew.as<float>(),
                       static_cast<int>(n_p), 1, static_cast<const double*>(nullptr), d_stats4, emit, d_lik, d_ratio, d_beam,
                       ctx->block_partials.as<double>(), nb, 0, ctx->partial4.as<double>());
    if (counts)
      ctx->penalty_clean_n = n_p;  // (launched: the kernel zeroes every counter it reads)
    TRY(timing_end(ctx, ep));
    HIP_TRY(hipGetLastError());
    return 0;
  }
  // the reference's float recurrence over the weights between the two (pf_strict_sum_kernel replaces the sum in `packed`)
  TRY(mcl3dl_hip_pf_partial_device(ctx, d_weight, d_lik, d_beam, d_extra, d_ratio, n_p, 0, 1, ctx->partial4.as<double>()));
  if (!ho)
    return mcl3dl_hip_pf_apply_device(ctx, d_weight, n_p, 1, ctx->partial4.as<double>(), d_stats4);
  EventPair ep{};
  TRY(timing_begin(ctx, MCL3DL_KERNEL_PF, &ep));
  hipLaunchKernelGGL(pf_apply_kernel, dim3(pf_blocks(n_p)), dim3(PF_BLOCK), 0, ctx->stream, d_weight, ctx->wnew.as<float>(),
                     static_cast<int>(n_p), 1, ctx->partial4.as<double>(), d_stats4, emit, d_lik, d_ratio, d_beam);
  TRY(timing_end(ctx, ep));
  HIP_TRY(hipGetLastError());
  return 0;
}

// The first half of pf::measure behind launch_measure for a rank of a device group (rank / world: the packed layout of the update's
// one all-reduce): what mcl3dl_hip_pf_partial_device launches, with the two steps launch_measure may have left to it (tail):
// lik_finalize_kernel's sum over the tiled kernel's per-tile partials (lik_pf_partial_kernel, wavefront partials) and the beam
// model's penalty count -> score (BeamCounts). Same arithmetic in the same association: the same bits.
int pf_partial_behind_measure(mcl3dl_hip_ctx* ctx, const float* d_weight, float* d_lik, float* d_beam, const float* d_extra,
                              float* d_ratio, size_t n_p, int rank, int world, double* d_packed, const LikTail& tail)
{
  if (!tail.pending && !tail.beam_pending)
    return mcl3dl_hip_pf_partial_device(ctx, d_weight, d_lik, d_beam, d_extra, d_ratio, n_p, rank, world, d_packed);
  const int nb = pf_blocks(n_p), n_waves = static_cast<int>((n_p + 63) / 64);
  TRY(ensure(ctx, ctx->wnew, sizeof(float) * n_p));
  TRY(ensure(ctx, ctx->block_partials, sizeof(double) * 16 * nb));
  const BeamCounts bc = tail.beam_pending ? BeamCounts{ ctx->penalty.as<unsigned>(), ctx->pow_table.as<float>(), ctx->beam_likelihood_min, d_beam }
                                          : BeamCounts{ nullptr, nullptr, 0.0f, nullptr };
  EventPair ep{};
  TRY(timing_begin(ctx, MCL3DL_KERNEL_PF, &ep));
  if (tail.pending)
  {
    const LikTiles lt{ ctx->lik_partial_sum.as<double>(), ctx->lik_partial_cnt.as<unsigned>(), tail.n_tiles, static_cast<int>(ctx->n_s),
                       d_lik, d_ratio, tail.beam_fill ? d_beam : static_cast<float*>(nullptr), bc };
    hipLaunchKernelGGL(lik_pf_partial_kernel, dim3(n_waves), dim3(256), 0, ctx->stream, lt, d_weight, d_beam, d_extra,
                       static_cast<int>(n_p), ctx->wnew.as<float>(), ctx->block_partials.as<double>());
    hipLaunchKernelGGL(pf_reduce_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->block_partials.as<double>(), nb, rank, world,
                       d_packed, n_waves);
  }
  else
  {
    hipLaunchKernelGGL(pf_partial_kernel, dim3(nb), dim3(PF_BLOCK), 0, ctx->stream, d_weight, d_lik, d_beam, d_extra, d_ratio,
                       static_cast<int>(n_p), ctx->wnew.as<float>(), ctx->block_partials.as<double>(), bc);
    hipLaunchKernelGGL(pf_reduce_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->block_partials.as<double>(), nb, rank, world,
                       d_packed);
  }
  if (world == 1 && pf_float_order(ctx, n_p))
    hipLaunchKernelGGL(pf_strict_sum_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->wnew.as<float>(), static_cast<int>(n_p),
                       d_packed);
  if (tail.beam_pending)
    ctx->penalty_clean_n = n_p;  // (launched: the kernel zeroes every counter it reads)
  TRY(timing_end(ctx, ep));
  HIP_TRY(hipGetLastError());
  return 0;
}

// ho (optional): page-locked arrays for the results; *host_written comes back true when the update's last kernel wrote
// them — otherwise the caller copies the device arrays home.
int enqueue_update(mcl3dl_hip_ctx* ctx, const float* d_pose, size_t n_p, float* d_weight, const float* d_extra,
                   float* d_lik, float* d_ratio, float* d_beam, float* d_stats4, const HostOut* ho = nullptr,
                   bool* host_written = nullptr)
{
  if (host_written)
    *host_written = ho != nullptr;  // every path below writes them
  const int one = launch_update_small(ctx, d_pose, n_p, d_weight, d_extra, d_lik, d_ratio, d_beam, d_stats4, ho);
  if (one != 0)
    return one < 0 ? one : 0;
  LikTail tail;
  tail.want = pf_takes_tiles(ctx, n_p) && d_lik && d_ratio && d_beam;
  tail.want_beam = d_beam != nullptr;
  TRY(launch_measure(ctx, d_pose, n_p, d_lik, d_ratio, d_beam, false, nullptr, &tail));
  TRY(pf_measure_single(ctx, d_weight, d_lik, d_beam, d_extra, d_ratio, n_p, d_stats4, ho, &tail));
  return 0;
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
  // (round 5: the captured-hipGraph form of this call is gone — replaying the 3-4 kernel update cost a fixed 10-16 us on
  // ROCm 7.2 against 3.3-3.8 us per plain launch and measured slower at every size for two rounds: DESIGN.md section 3.3)
  return enqueue_update(ctx, d_pose, n_p, d_weight_inout, d_extra, d_lik, d_match_ratio, d_beam, d_stats4);
}


int mcl3dl_hip_upload_poses(mcl3dl_hip_ctx* ctx, const float* pose, size_t n_p)
{
  if (!ctx)
    return -1;
  if (!pose || n_p == 0 || n_p > 0x7fffffffu)
    return ctx->fail(-3, "bad pose array");
  HIP_TRY(hipSetDevice(ctx->device));
  ctx->poses_set(0);
  TRY(ensure(ctx, ctx->pose, sizeof(float) * 7 * n_p));
  bool staged = false;
  TRY(h2d(ctx, ctx->pose.p, pose, sizeof(float) * 7 * n_p, &staged));
  // a staged copy is done with the caller's array already: no need to wait for the stream (the call that uses the poses
  // synchronises, and recycles the staging memory); a direct copy from the caller's memory — or a pile of unsynchronised
  // uploads — is waited for here
  if (!staged || ctx->stage_pending > (8u << 20))
    TRY(sync_stream(ctx));
  ctx->poses_set(n_p);
  return 0;
}

namespace
{
// slice = STAGE_FRONT_ONLY: only the front half of measure_update_staged — inputs taken over and scans ordered by the staging
// launch(es) — and return 3 (0: not eligible, nothing done). What a rank of a device group runs ahead of its kernels.
constexpr size_t STAGE_FRONT_ONLY = ~static_cast<size_t>(0);
int measure_update_staged(mcl3dl_hip_ctx* ctx, const float* pose, const float* extra, float* weight_inout, size_t n_p,
                          const float* scan_lik_xyz, size_t n_s, const float* scan_beam_xyz, const uint32_t* scan_beam_origin,
                          size_t n_b, const float* origins, size_t n_o, float* out_lik, float* out_match_ratio,
                          float* out_beam, float* st4, bool with_pf = true, size_t slice = 0);
// where the staging launch left the weights it took over
inline float* staged_weights(mcl3dl_hip_ctx* ctx)
{
  return reinterpret_cast<float*>(ctx->upd_block.as<char>() + 64);
}
}  // namespace

int mcl3dl_hip_measure_batch(mcl3dl_hip_ctx* ctx, const float* pose, size_t n_p, const float* scan_lik_xyz, size_t n_s,
                             const float* scan_beam_xyz, const uint32_t* scan_beam_origin, size_t n_b,
                             const float* origins, size_t n_o, float* out_lik, float* out_match_ratio, float* out_beam)
{
  if (!ctx)
    return -1;
  if (n_p == 0)
    return 0;
  if (!pose && ctx->n_pose_uploaded != n_p)
    return ctx->fail(-3, "null pose array (and mcl3dl_hip_upload_poses holds %zu poses, not %zu)", ctx->n_pose_uploaded,
                     n_p);
  HIP_TRY(hipSetDevice(ctx->device));
  {
    // scans (and poses) taken over by one launch, results written into page-locked memory (stage_kernels.h)
    const int staged = measure_update_staged(ctx, pose, nullptr, nullptr, n_p, scan_lik_xyz, n_s, scan_beam_xyz,
                                             scan_beam_origin, n_b, origins, n_o, out_lik, out_match_ratio, out_beam, nullptr,
                                             false);
    if (staged != 0)
      return staged < 0 ? staged : 0;
  }
  TRY(upload_scan_impl(ctx, scan_lik_xyz, n_s, scan_beam_xyz, scan_beam_origin, n_b, origins, n_o, false));
  TRY(ensure(ctx, ctx->lik, sizeof(float) * n_p));
  TRY(ensure(ctx, ctx->ratio, sizeof(float) * n_p));
  TRY(ensure(ctx, ctx->beam, sizeof(float) * n_p));
  if (pose)
  {
    ctx->poses_set(0);
    TRY(ensure(ctx, ctx->pose, sizeof(float) * 7 * n_p));
    TRY(h2d(ctx, ctx->pose.p, pose, sizeof(float) * 7 * n_p));
    ctx->poses_set(n_p);
  }
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

// ---- the same batch, delivered in particle slices ------------------------------------------------------------------------
int mcl3dl_hip_measure_batch_begin(mcl3dl_hip_ctx* ctx, const float* pose, size_t n_p, const float* scan_lik_xyz, size_t n_s,
                                   const float* scan_beam_xyz, const uint32_t* scan_beam_origin, size_t n_b,
                                   const float* origins, size_t n_o, float* out_lik, float* out_match_ratio, float* out_beam,
                                   size_t slice_particles)
{
  if (!ctx)
    return -1;
  HIP_TRY(hipSetDevice(ctx->device));
  TRY(progress_end(ctx));
  // prog.n_p stays 0 until the batch has been accepted: after a failed _begin, _wait refuses every index (-3) instead of
  // reporting n_p results nobody delivered
  ctx->prog = mcl3dl_hip_ctx::BatchProgress();
  if (n_p == 0)
    return 0;
  if (!pose && ctx->n_pose_uploaded != n_p)
    return ctx->fail(-3, "null pose array (and mcl3dl_hip_upload_poses holds %zu poses, not %zu)", ctx->n_pose_uploaded,
                     n_p);
  size_t slice = slice_particles ? slice_particles : static_cast<size_t>(ctx->batch_slice);
  if (slice == 0)
    slice = n_p >= 1024 ? std::max<size_t>(512, ((n_p + 3) / 4 + 63) & ~static_cast<size_t>(63)) : n_p;  // four slices
  slice = (slice + 15) & ~static_cast<size_t>(15);
  if (slice < n_p)
  {
    const int staged = measure_update_staged(ctx, pose, nullptr, nullptr, n_p, scan_lik_xyz, n_s, scan_beam_xyz,
                                             scan_beam_origin, n_b, origins, n_o, out_lik, out_match_ratio, out_beam, nullptr,
                                             false, slice);
    if (staged < 0)
    {
      // (slices already enqueued write page-locked staging memory or the caller's page-locked arrays: drained before the
      // caller gets control back, then forgotten)
      (void)hipStreamSynchronize(ctx->stream);
      ctx->prog = mcl3dl_hip_ctx::BatchProgress();
      return staged;
    }
    if (staged != 0)
    {
      ctx->prog.n_p = n_p;  // (staged == 1: the path delivered everything at once)
      return 0;
    }
  }
  // not eligible for slices: the whole batch now
  const int rc = mcl3dl_hip_measure_batch(ctx, pose, n_p, scan_lik_xyz, n_s, scan_beam_xyz, scan_beam_origin, n_b, origins, n_o,
                                          out_lik, out_match_ratio, out_beam);
  ctx->prog = mcl3dl_hip_ctx::BatchProgress();
  if (rc == 0)
    ctx->prog.n_p = n_p;
  return rc;
}

int mcl3dl_hip_measure_batch_wait(mcl3dl_hip_ctx* ctx, size_t particle, size_t* n_ready)
{
  if (!ctx)
    return -1;
  if (particle >= ctx->prog.n_p)
  {
    // (not ctx->fail(): a wrong index does not abandon the batch in flight)
    char buf[160];
    snprintf(buf, sizeof(buf), "particle %zu is not part of the batch (%zu particles)", particle, ctx->prog.n_p);
    ctx->err = buf;
    return -3;
  }
  return progress_wait(ctx, particle, n_ready);
}

int mcl3dl_hip_measure_batch_end(mcl3dl_hip_ctx* ctx)
{
  if (!ctx)
    return -1;
  HIP_TRY(hipSetDevice(ctx->device));
  return progress_end(ctx);
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
  TRY(pf_measure_single(ctx, ctx->weightb.as<float>(), ctx->lik.as<float>(), beam ? ctx->beam.as<float>() : nullptr,
                        extra ? ctx->extra.as<float>() : nullptr, match_ratio ? ctx->ratio.as<float>() : nullptr, n_p,
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

namespace
{
// The host-buffer update with scan_stage_kernel in front (one launch takes over scans, poses and weights: ordering included)
// and the results written straight into page-locked memory by the kernel that normalises the weights. Returns 1 when the update was run
// this way (results delivered, stream synchronised), 0 when it is not eligible (the caller runs the general path), < 0 on
// error.
// with_pf = false: mcl3dl_hip_measure_batch — the two models only (no weights, no pf::measure); pose may then be null (the
// poses mcl3dl_hip_upload_poses left on the device).
int measure_update_staged(mcl3dl_hip_ctx* ctx, const float* pose, const float* extra, float* weight_inout, size_t n_p,
                          const float* scan_lik_xyz, size_t n_s, const float* scan_beam_xyz, const uint32_t* scan_beam_origin,
                          size_t n_b, const float* origins, size_t n_o, float* out_lik, float* out_match_ratio,
                          float* out_beam, float* st4, bool with_pf, size_t slice)
{
  if (!ctx->update_stage || n_s > 0x0fffffffu || n_b > 0x0fffffffu || n_o > 4096 || n_p > 0x7fffffffu / 8)
    return 0;
  if ((n_s && !scan_lik_xyz) || (n_b && (!scan_beam_xyz || !origins || n_o == 0)))
    return ctx->fail(-3, "null scan array");
  if (scan_beam_origin)
    for (size_t i = 0; i < n_b; ++i)
      if (scan_beam_origin[i] >= n_o)
        return ctx->fail(-3, "beam point %zu names origin %u but only %zu origins were given", i, scan_beam_origin[i], n_o);
  if (!origins)
    n_o = 0;
  // Whatever has to be (re)built lazily — the index after a map change, the DDA grid, the penalty table — is built NOW: a
  // build synchronises the stream, recycles the staging memory and (polled) uses up completion sequence numbers, none of which
  // may happen between the allocations below and the kernels that read and write them.
  // (test hook "test_late_structures", MCL3DL_HIP_TEST_HOOKS=1 only: leave the builds to launch_measure as round 4's last but
  // one commit did — the hazard tests/test_gpu_api_fuzz.py is asked to find again)
  if (!ctx->test_late_structures)
  {
    TRY(ensure_structures(ctx, n_s > 0, n_b > 0));
    if (n_b > 0)
      TRY(ensure_pow_table(ctx, n_b));
  }
  const size_t fb = sizeof(float) * n_p;
  // ---- the input block: { poses | weights | odometry factor | likelihood xyz | beam xyz | beam origin ids | origins }
  struct Part
  {
    const void* src;
    size_t bytes;
    const void* dev;  // where the kernel reads it
  };
  Part part[7] = { { pose, pose ? 7 * fb : 0, nullptr },
                   { weight_inout, weight_inout ? fb : 0, nullptr },
                   { extra, extra ? fb : 0, nullptr },
                   { scan_lik_xyz, sizeof(float) * 3 * n_s, nullptr },
                   { scan_beam_xyz, sizeof(float) * 3 * n_b, nullptr },
                   { scan_beam_origin, scan_beam_origin ? sizeof(uint32_t) * n_b : 0, nullptr },
                   { origins, sizeof(float) * 3 * n_o, nullptr } };
  const bool zero_copy = ctx->zero_copy();
  const auto up = [](size_t b) { return (b + 255) & ~static_cast<size_t>(255); };
  size_t staged_bytes = 0, off[7];
  for (int k = 0; k < 7; ++k)
  {
    off[k] = staged_bytes;
    if (part[k].bytes && !(zero_copy && ctx->is_pinned(part[k].src, part[k].bytes)))
      staged_bytes += up(part[k].bytes);
  }
  char* staged = nullptr;
  if (staged_bytes)
  {
    if (staged_bytes > STAGE_MAX_COPY)
      return 0;
    staged = static_cast<char*>(stage_alloc(ctx, staged_bytes));
    if (!staged)
      return 0;
  }
  // ---- device arrays
  const size_t rpart = (fb + 63) & ~static_cast<size_t>(63);
  TRY(ensure(ctx, ctx->pose, 7 * fb));
  TRY(ensure(ctx, ctx->upd_block, 64 + 4 * rpart));
  TRY(ensure(ctx, ctx->extra, fb));
  TRY(ensure(ctx, ctx->partial4, sizeof(double) * 4));
  TRY(ensure(ctx, ctx->sp_samp[0], sizeof(float4) * std::max<size_t>(n_s, 1)));
  TRY(ensure(ctx, ctx->sp_samp[1], sizeof(float4) * std::max<size_t>(n_b, 1)));
  TRY(ensure(ctx, ctx->cl_minmax, sizeof(float) * 6 + sizeof(unsigned long long)));
  TRY(ensure(ctx, ctx->cl_err, sizeof(int)));
  TRY(ensure_scan_block(ctx, n_s, n_b, n_o));
  if (!zero_copy && staged_bytes)
    TRY(ensure(ctx, ctx->stage_in_dev, staged_bytes));
  for (int k = 0; k < 7; ++k)
  {
    if (!part[k].bytes)
      continue;
    if (zero_copy && ctx->is_pinned(part[k].src, part[k].bytes))
    {
      part[k].dev = part[k].src;
      continue;
    }
    memcpy(staged + off[k], part[k].src, part[k].bytes);
    part[k].dev = zero_copy ? staged + off[k] : ctx->stage_in_dev.as<char>() + off[k];
  }
  if (!zero_copy && staged_bytes)
  {
    HIP_TRY(hipMemcpyAsync(ctx->stage_in_dev.p, staged, staged_bytes, hipMemcpyHostToDevice, ctx->stream));
    ctx->stage_pending += staged_bytes;
  }
  char* blk = ctx->upd_block.as<char>();
  float* d_stats = reinterpret_cast<float*>(blk);
  float* d_w = reinterpret_cast<float*>(blk + 64);
  float* d_lik = reinterpret_cast<float*>(blk + 64 + rpart);
  float* d_ratio = reinterpret_cast<float*>(blk + 64 + 2 * rpart);
  float* d_beam = reinterpret_cast<float*>(blk + 64 + 3 * rpart);
  StageArgs a{};
  a.presorted = ctx->scan_presorted;
  a.in_pose = static_cast<const float*>(part[0].dev);
  a.in_w = static_cast<const float*>(part[1].dev);
  a.in_extra = static_cast<const float*>(part[2].dev);
  a.d_pose = ctx->pose.as<float>();
  a.d_w = d_w;
  a.d_extra = ctx->extra.as<float>();
  a.n_p = static_cast<int>(n_p);
  a.in_lik_xyz = static_cast<const float*>(part[3].dev);
  a.n_s = static_cast<int>(n_s);
  a.raw_lik = ctx->sp_samp[0].as<float4>();
  a.mm6 = ctx->cl_minmax.as<float>();
  a.mm_cnt = reinterpret_cast<unsigned long long*>(ctx->cl_minmax.as<float>() + 6);
  a.out_lik = ctx->scan_lik.as<float4>();
  a.out_perm = ctx->scan_perm.as<uint32_t>();
  a.in_beam_xyz = static_cast<const float*>(part[4].dev);
  a.in_beam_origin = static_cast<const uint32_t*>(part[5].dev);
  a.n_b = static_cast<int>(n_b);
  a.raw_beam = ctx->sp_samp[1].as<float4>();
  a.out_beam = ctx->scan_beam.as<float4>();
  a.in_origins = static_cast<const float*>(part[6].dev);
  a.n_o = static_cast<int>(n_o);
  a.d_origins = ctx->origins.as<float4>();
  a.d_err = ctx->cl_err.as<int>();
  const size_t n_copy = 9 * n_p;
  const size_t n_max = std::max(n_s, n_b);
  EventPair ep{};
  TRY(timing_begin(ctx, MCL3DL_KERNEL_STAGE, &ep));
  if (n_max <= static_cast<size_t>(ST_MAX_POINTS))
  {
    // everything in one launch: one work-group orders each scan
    const unsigned grid = 2u + static_cast<unsigned>(std::min<size_t>(std::max<size_t>((n_copy + RS_THREADS - 1) / RS_THREADS, 1), 64));
    hipLaunchKernelGGL((scan_stage_kernel<ST_MAX_ROUNDS>), dim3(grid), dim3(RS_THREADS), 0, ctx->stream, a);
    HIP_TRY(hipGetLastError());
  }
  else
  {
    // larger scans: one launch brings the arrays over (+ the min / max the Morton keys need), the chip-wide sort orders them
    // (256 points per work-group and round; the copies 16 bytes per thread: stage_kernels.h)
    const unsigned nb_lik = n_s ? static_cast<unsigned>(std::min<size_t>((n_s + 255) / 256, 256)) : 0u;
    const unsigned nb_beam = n_b ? static_cast<unsigned>(std::min<size_t>((n_b + 255) / 256, 64)) : (n_o ? 1u : 0u);
    const unsigned nb_copy = static_cast<unsigned>(std::min<size_t>(std::max<size_t>((7 * n_p + 1023) / 1024, 1), 64));
    MinMaxOut mm{};
    if (nb_lik)
      TRY(minmax_out(ctx, nb_lik, &mm));
    hipLaunchKernelGGL(stage_pack_kernel, dim3(nb_lik + nb_beam + nb_copy), dim3(256), 0, ctx->stream, a, mm, nb_lik, nb_beam);
    HIP_TRY(hipGetLastError());
    TRY(device_order_scans(ctx, n_s, n_b, nullptr, n_o, true, ctx->cl_err.as<int>()));
  }
  TRY(timing_end(ctx, ep));
  // the context's scan state, as upload_scan_impl leaves it
  if (n_b > ctx->pow_table_len)
    ctx->pow_table_dirty = true;
  if (n_s != ctx->n_s || n_b != ctx->n_b || n_o != ctx->n_o || !ctx->has_scan)
    ++ctx->generation;
  ctx->n_s = n_s;
  ctx->n_b = n_b;
  ctx->n_o = n_o;
  ctx->has_scan = true;
  ctx->sp_n_samp[0] = n_s;
  ctx->sp_n_samp[1] = n_b;
  if (pose)
    ctx->poses_set(n_p);
  if (slice == STAGE_FRONT_ONLY)
    return 3;  // the caller (a device group's rank) goes on from here: poses in ctx->pose, weights at staged_weights(ctx)
  if (!with_pf)
  {
    // the two models only: their per-particle results go home through a copy kernel into page-locked memory (or one D2H copy)
    const bool lik_wanted = out_lik || out_match_ratio;
    float* const user[3] = { out_lik, out_match_ratio, out_beam };
    const float* const dev[3] = { d_lik, d_ratio, d_beam };
    char* blk3 = zero_copy ? static_cast<char*>(stage_alloc(ctx, 3 * rpart)) : nullptr;
    if (blk3 && slice > 0 && slice < n_p && ctx->poll_mode() && ensure_done_flag(ctx))
    {
      // progressive delivery (mcl3dl_hip_measure_batch_begin): the particles are evaluated slice by slice, each slice's
      // results leave for page-locked memory as soon as its kernels are through and a completion word follows them, so the
      // caller's per-particle loop (the reference's pf::measure, pf.h:255-260) runs while the later slices are still on the GPU
      mcl3dl_hip_ctx::BatchProgress& pg = ctx->prog;
      pg = mcl3dl_hip_ctx::BatchProgress();
      pg.n_p = n_p;
      pg.slice = slice;
      pg.n_slices = (n_p + slice - 1) / slice;
      pg.seq0 = ctx->done_seq;
      PfEmit e{};
      float** slot[3] = { &e.lik, &e.ratio, &e.beam };
      for (int k = 0; k < 3; ++k)
        if (user[k])
        {
          *slot[k] = ctx->is_pinned(user[k], fb) ? user[k] : reinterpret_cast<float*>(blk3 + k * rpart);
          pg.user[k] = user[k];
          pg.host[k] = *slot[k];
        }
      // both models of a slice on ONE stream: the second stream's fork / join events cost ~35 us of host time per launch
      // (host_measure.h), which the caller's thread would pay once per slice before it can start on the results
      struct OverlapOff
      {
        mcl3dl_hip_ctx* c;
        int saved;
        ~OverlapOff()
        {
          c->overlap_models = saved;
        }
      } overlap_off{ ctx, ctx->overlap_models };
      ctx->overlap_models = 0;
      for (size_t lo = 0; lo < n_p; lo += slice)
      {
        const size_t n = std::min(slice, n_p - lo);
        TRY(launch_measure(ctx, ctx->pose.as<float>() + 7 * lo, n, lik_wanted ? d_lik + lo : nullptr,
                           lik_wanted ? d_ratio + lo : nullptr, out_beam ? d_beam + lo : nullptr, false, nullptr));
        PfEmit es{};
        es.lik = e.lik ? e.lik + lo : nullptr;
        es.ratio = e.ratio ? e.ratio + lo : nullptr;
        es.beam = e.beam ? e.beam + lo : nullptr;
        hipLaunchKernelGGL(emit3_kernel, dim3(pf_blocks(n)), dim3(PF_BLOCK), 0, ctx->stream, es, d_lik + lo, d_ratio + lo,
                           d_beam + lo, static_cast<int>(n));
        hipLaunchKernelGGL(done_flag_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->done_flag, ++ctx->done_seq);
        HIP_TRY(hipGetLastError());
        ++ctx->batch_slices_run;
      }
      pg.active = true;
      return 2;
    }
    TRY(launch_measure(ctx, ctx->pose.as<float>(), n_p, lik_wanted ? d_lik : nullptr, lik_wanted ? d_ratio : nullptr,
                       out_beam ? d_beam : nullptr, false, nullptr));
    if (blk3)
    {
      PfEmit e{};
      float** slot[3] = { &e.lik, &e.ratio, &e.beam };
      for (int k = 0; k < 3; ++k)
        if (user[k])
        {
          *slot[k] = ctx->is_pinned(user[k], fb) ? user[k] : reinterpret_cast<float*>(blk3 + k * rpart);
          if (*slot[k] != user[k])
            ctx->stage_out.push_back({ user[k], *slot[k], fb });
        }
      hipLaunchKernelGGL(emit3_kernel, dim3(pf_blocks(n_p)), dim3(PF_BLOCK), 0, ctx->stream, e, d_lik, d_ratio, d_beam,
                         static_cast<int>(n_p));
      HIP_TRY(hipGetLastError());
      TRY(sync_stream(ctx, true));
    }
    else
    {
      for (int k = 0; k < 3; ++k)
        if (user[k])
          TRY(d2h(ctx, user[k], dev[k], fb));
      TRY(sync_stream(ctx));
    }
    return 1;
  }
  // ---- results: written by the update's last kernel into page-locked memory (the caller's own arrays where they are
  // page-locked), or copied home in one block
  HostOut ho{};
  char* out_blk = zero_copy ? static_cast<char*>(stage_alloc(ctx, 64 + 4 * rpart)) : nullptr;
  struct Res
  {
    float* user;
    size_t offset, bytes;
    float** slot;
  };
  const Res res[5] = { { st4, 0, 4 * sizeof(float), &ho.stats4 },
                       { weight_inout, 64, fb, &ho.w },
                       { out_lik, 64 + rpart, fb, &ho.lik },
                       { out_match_ratio, 64 + 2 * rpart, fb, &ho.ratio },
                       { out_beam, 64 + 3 * rpart, fb, &ho.beam } };
  if (out_blk)
    for (int k = 0; k < 5; ++k)
      if (res[k].user)
        *res[k].slot = (k > 0 && ctx->is_pinned(res[k].user, res[k].bytes)) ? res[k].user
                                                                            : reinterpret_cast<float*>(out_blk + res[k].offset);
  bool host_written = false;
  TRY(enqueue_update(ctx, ctx->pose.as<float>(), n_p, d_w, extra ? ctx->extra.as<float>() : nullptr, d_lik, d_ratio, d_beam,
                     d_stats, out_blk ? &ho : nullptr, &host_written));
  if (host_written)
  {
    for (int k = 0; k < 5; ++k)
      if (res[k].user && *res[k].slot != res[k].user)
        ctx->stage_out.push_back({ res[k].user, *res[k].slot, res[k].bytes });
  }
  else
  {
    const D2hPiece pieces[5] = { { st4, 0, 4 * sizeof(float) },
                                 { weight_inout, 64, fb },
                                 { out_lik, 64 + rpart, fb },
                                 { out_match_ratio, 64 + 2 * rpart, fb },
                                 { out_beam, 64 + 3 * rpart, fb } };
    const size_t upto = out_beam ? 64 + 3 * rpart + fb : out_match_ratio ? 64 + 2 * rpart + fb : out_lik ? 64 + rpart + fb : 64 + fb;
    TRY(d2h_block(ctx, blk, upto, pieces, 5));
  }
  TRY(sync_stream(ctx, host_written));  // (results behind a D2H copy: the stream itself is waited for)
  return 1;
}
}  // namespace

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
  {
    float st[4] = { 0.f, 0.f, 0.f, 0.f };
    const int staged = measure_update_staged(ctx, pose, extra, weight_inout, n_p, scan_lik_xyz, n_s, scan_beam_xyz,
                                             scan_beam_origin, n_b, origins, n_o, out_lik, out_match_ratio, out_beam, st);
    if (staged < 0)
      return staged;
    if (staged == 1)
    {
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
  }
  TRY(upload_scan_impl(ctx, scan_lik_xyz, n_s, scan_beam_xyz, scan_beam_origin, n_b, origins, n_o, false));
  // { stats4 | weights (in / out) | lik | ratio | beam }, each part on a 64-byte boundary, in ONE allocation: the results go
  // home in one copy instead of five (~5 us each at C2's sizes)
  const size_t part = (fb + 63) & ~static_cast<size_t>(63);
  TRY(ensure(ctx, ctx->pose, sizeof(float) * 7 * n_p));
  TRY(ensure(ctx, ctx->upd_block, 64 + 4 * part));
  TRY(ensure(ctx, ctx->extra, fb));
  TRY(ensure(ctx, ctx->partial4, sizeof(double) * 4));
  char* blk = ctx->upd_block.as<char>();
  float* d_stats = reinterpret_cast<float*>(blk);
  float* d_w = reinterpret_cast<float*>(blk + 64);
  float* d_lik = reinterpret_cast<float*>(blk + 64 + part);
  float* d_ratio = reinterpret_cast<float*>(blk + 64 + 2 * part);
  float* d_beam = reinterpret_cast<float*>(blk + 64 + 3 * part);
  ctx->poses_set(0);
  TRY(h2d(ctx, ctx->pose.p, pose, sizeof(float) * 7 * n_p));
  ctx->poses_set(n_p);
  TRY(h2d(ctx, d_w, weight_inout, fb));
  if (extra)
    TRY(h2d(ctx, ctx->extra.p, extra, fb));
  TRY(enqueue_update(ctx, ctx->pose.as<float>(), n_p, d_w, extra ? ctx->extra.as<float>() : nullptr, d_lik, d_ratio, d_beam,
                     d_stats));
  float st[4];
  const D2hPiece pieces[5] = { { st, 0, sizeof(st) },
                               { weight_inout, 64, fb },
                               { out_lik, 64 + part, fb },
                               { out_match_ratio, 64 + 2 * part, fb },
                               { out_beam, 64 + 3 * part, fb } };
  const size_t upto = out_beam ? 64 + 3 * part + fb : out_match_ratio ? 64 + 2 * part + fb : out_lik ? 64 + part + fb : 64 + fb;
  TRY(d2h_block(ctx, blk, upto, pieces, 5));
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

