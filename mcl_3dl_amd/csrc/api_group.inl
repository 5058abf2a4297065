// api_group.inl — included inside the extern "C" block of mcl3dl_hip.hip: the device group (host_group.h) behind the C ABI.
namespace
{
int group_comms(mcl3dl_hip_group* g)
{
  if (g->collective == 1 || !g->comms.empty())
    return 0;
  if (!g->devices_distinct)
    return g->fail(-3, "a device id is listed more than once: RCCL needs one GPU per rank — set option \"collective\" to 1 "
                       "(host combine) for several contexts on one GPU");
  const std::string why = g->rccl.load();
  if (!why.empty())
    return g->fail(-7, "%s; set option \"collective\" to 1 (host combine) or MCL3DL_HIP_RCCL_LIB", why.c_str());
  g->comms.assign(g->n(), nullptr);
  const ncclResult_t rc = g->rccl.CommInitAll(g->comms.data(), g->n(), g->devices.data());
  if (rc != ncclSuccess)
  {
    g->comms.clear();
    return g->fail(-7, "ncclCommInitAll over %d devices failed: %s", g->n(), g->rccl.GetErrorString(rc));
  }
  return 0;
}

// the shard's share of the 2 + 2N-double record when it holds no particle: sums 0, max ratio 0, -min ratio -1
int pack_empty(mcl3dl_hip_ctx* ctx, int rank, int world, double* d_packed)
{
  hipLaunchKernelGGL(pf_reduce_kernel, dim3(1), dim3(64), 0, ctx->stream, static_cast<const double*>(nullptr), 0, rank, world,
                     d_packed);
  HIP_TRY(hipGetLastError());
  return 0;
}
}  // namespace

int mcl3dl_hip_group_create(mcl3dl_hip_group** out, const int* device_ids, int n_devices)
{
  if (!out)
    return -1;
  *out = nullptr;
  if (!device_ids || n_devices < 1 || n_devices > 64)
    return -3;
  mcl3dl_hip_group* g = new mcl3dl_hip_group;
  for (int r = 0; r < n_devices; ++r)
  {
    mcl3dl_hip_ctx* c = nullptr;
    const int rc = mcl3dl_hip_create(&c, device_ids[r]);
    if (rc != 0)
    {
      for (mcl3dl_hip_ctx* p : g->ctx)
        mcl3dl_hip_destroy(p);
      delete g;
      return rc;
    }
    for (int q = 0; q < r; ++q)
      if (device_ids[q] == device_ids[r])
        g->devices_distinct = false;
    g->ctx.push_back(c);
    g->devices.push_back(device_ids[r]);
  }
  const char* env = getenv("MCL3DL_HIP_COLLECTIVE");
  if (env && std::string(env) == "host")
    g->collective = 1;
  g->host_packed.resize(n_devices);
  if (n_devices > 1)
    g->pool.start(n_devices);
  g->vote.resize(n_devices);
  *out = g;
  return 0;
}

void mcl3dl_hip_group_destroy(mcl3dl_hip_group* g)
{
  if (!g)
    return;
  g->pool.stop();
  for (mcl3dl_hip_ctx* c : g->ctx)
  {
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
  }
  for (ncclComm_t c : g->comms)
    if (c)
      (void)g->rccl.CommDestroy(c);
  g->comms.clear();
  for (mcl3dl_hip_ctx* c : g->ctx)
    mcl3dl_hip_destroy(c);
  delete g;
}

const char* mcl3dl_hip_group_last_error(const mcl3dl_hip_group* g)
{
  return g ? g->err.c_str() : "null group";
}

int mcl3dl_hip_group_size(const mcl3dl_hip_group* g)
{
  return g ? g->n() : 0;
}

mcl3dl_hip_ctx* mcl3dl_hip_group_context(mcl3dl_hip_group* g, int rank)
{
  return (g && rank >= 0 && rank < g->n()) ? g->ctx[rank] : nullptr;
}

int mcl3dl_hip_group_shard(size_t n_p, int n_devices, int rank, size_t* begin, size_t* count)
{
  if (n_devices < 1 || rank < 0 || rank >= n_devices || !begin || !count)
    return -3;
  size_t lo, hi;
  shard_bounds(n_p, n_devices, rank, &lo, &hi);
  *begin = lo;
  *count = hi - lo;
  return 0;
}

int mcl3dl_hip_group_set_map(mcl3dl_hip_group* g, const float* xyz, const uint32_t* label, size_t n_m, uint64_t stamp,
                             const float* dist_weight)
{
  if (!g)
    return -1;
  int bad = 0;
  const int rc = g->pool.run_all([&](int r) { return mcl3dl_hip_set_map(g->ctx[r], xyz, label, n_m, stamp, dist_weight); },
                                 &bad);
  return rc ? g->fail_rank(rc, bad) : 0;
}

int mcl3dl_hip_group_set_likelihood_params(mcl3dl_hip_group* g, float match_dist_min, float match_dist_flat,
                                           float match_weight)
{
  if (!g)
    return -1;
  for (int r = 0; r < g->n(); ++r)
  {
    const int rc = mcl3dl_hip_set_likelihood_params(g->ctx[r], match_dist_min, match_dist_flat, match_weight);
    if (rc)
      return g->fail_rank(rc, r);
  }
  return 0;
}

int mcl3dl_hip_group_set_beam_params(mcl3dl_hip_group* g, float map_grid_x, float map_grid_y, float map_grid_z,
                                     float dda_grid_size, float ray_angle_half, float hit_range,
                                     float beam_likelihood_min, uint32_t num_points, float ang_total_ref,
                                     uint32_t filter_label_max, int add_penalty_short_only_mode)
{
  if (!g)
    return -1;
  for (int r = 0; r < g->n(); ++r)
  {
    const int rc = mcl3dl_hip_set_beam_params(g->ctx[r], map_grid_x, map_grid_y, map_grid_z, dda_grid_size, ray_angle_half,
                                              hit_range, beam_likelihood_min, num_points, ang_total_ref, filter_label_max,
                                              add_penalty_short_only_mode);
    if (rc)
      return g->fail_rank(rc, r);
  }
  return 0;
}

int mcl3dl_hip_group_set_option(mcl3dl_hip_group* g, const char* name, double value)
{
  if (!g || !name)
    return -1;
  if (std::string(name) == "collective")
  {
    if (value != 0.0 && value != 1.0)
      return g->fail(-3, "collective must be 0 (RCCL all-reduce) or 1 (host combine)");
    g->collective = static_cast<int>(value);
    return 0;
  }
  if (std::string(name) == "direct_single")
  {
    g->direct_single = value != 0.0;
    return 0;
  }
  if (std::string(name) == "inject_failure_rank")
  {
    // test hook (only with MCL3DL_HIP_TEST_HOOKS=1 in the environment): the next sharded update fails on that rank after its
    // kernels are enqueued and before the collective
    const char* hooks = getenv("MCL3DL_HIP_TEST_HOOKS");
    if (!hooks || std::string(hooks) != "1")
      return g->fail(-3, "inject_failure_rank is a test hook: set MCL3DL_HIP_TEST_HOOKS=1 in the environment to enable it");
    g->inject_failure_rank = static_cast<int>(value);
    return 0;
  }
  for (int r = 0; r < g->n(); ++r)
  {
    const int rc = mcl3dl_hip_set_option(g->ctx[r], name, value);
    if (rc)
      return g->fail_rank(rc, r);
  }
  return 0;
}

int mcl3dl_hip_group_collective_stats(const mcl3dl_hip_group* g, uint64_t* rccl_all_reduces, uint64_t* host_combines)
{
  if (!g)
    return -1;
  if (rccl_all_reduces)
    *rccl_all_reduces = g->collectives_rccl;
  if (host_combines)
    *host_combines = g->collectives_host;
  return 0;
}

int mcl3dl_hip_group_upload_poses(mcl3dl_hip_group* g, const float* pose, size_t n_p)
{
  if (!g)
    return -1;
  if (!pose || n_p == 0 || n_p > 0x7fffffffu)
    return g->fail(-3, "bad pose array");
  g->n_pose_uploaded = 0;
  int bad = 0;
  const int N = g->n();
  const int rc = g->pool.run_all(
      [&](int r) -> int
      {
        mcl3dl_hip_ctx* ctx = g->ctx[r];
        size_t lo, hi;
        shard_bounds(n_p, N, r, &lo, &hi);
        ctx->poses_set(0);
        if (hi == lo)
          return 0;
        return mcl3dl_hip_upload_poses(ctx, pose + 7 * lo, hi - lo);
      },
      &bad);
  if (rc)
    return g->fail_rank(rc, bad);
  g->n_pose_uploaded = n_p;
  return 0;
}

int mcl3dl_hip_group_measure_batch(mcl3dl_hip_group* g, const float* pose, size_t n_p, const float* scan_lik_xyz,
                                   size_t n_s, const float* scan_beam_xyz, const uint32_t* scan_beam_origin, size_t n_b,
                                   const float* origins, size_t n_o, float* out_lik, float* out_match_ratio,
                                   float* out_beam)
{
  if (!g)
    return -1;
  if (n_p == 0)
    return 0;
  if (!pose && g->n_pose_uploaded != n_p)
    return g->fail(-3, "null pose array (and mcl3dl_hip_group_upload_poses holds %zu poses, not %zu)", g->n_pose_uploaded,
                   n_p);
  if (g->n() == 1 && g->direct_single)
  {
    const int rc = mcl3dl_hip_measure_batch(g->ctx[0], pose, n_p, scan_lik_xyz, n_s, scan_beam_xyz, scan_beam_origin, n_b,
                                            origins, n_o, out_lik, out_match_ratio, out_beam);
    return rc ? g->fail_rank(rc, 0) : 0;
  }
  const bool device_order = g->ctx[0]->scan_order_device > 0 && n_s + n_b >= static_cast<size_t>(g->ctx[0]->scan_order_device);
  std::string err;
  if (!device_order &&
      order_scan(err, scan_lik_xyz, n_s, scan_beam_xyz, scan_beam_origin, n_b, origins, n_o, g->scan, g->ctx[0]->scan_presorted != 0) != 0)
    return g->fail(-3, "%s", err.c_str());
  if (pose)
    g->n_pose_uploaded = 0;
  int bad = 0;
  const int N = g->n();
  const int rc = g->pool.run_all(
      [&](int r) -> int
      {
        mcl3dl_hip_ctx* ctx = g->ctx[r];
        size_t lo, hi;
        shard_bounds(n_p, N, r, &lo, &hi);
        const size_t n = hi - lo;
        if (device_order)
          TRY(upload_scan_impl(ctx, scan_lik_xyz, n_s, scan_beam_xyz, scan_beam_origin, n_b, origins, n_o, false));
        else
          TRY(push_scan(ctx, g->scan, false));
        if (n == 0)
          return sync_stream(ctx);
        if (pose)
        {
          ctx->poses_set(0);
          TRY(ensure(ctx, ctx->pose, sizeof(float) * 7 * n));
          TRY(h2d(ctx, ctx->pose.p, pose + 7 * lo, sizeof(float) * 7 * n));
          ctx->poses_set(n);
        }
        else if (ctx->n_pose_uploaded != n)
          return ctx->fail(-3, "uploaded pose shard holds %zu poses, not %zu", ctx->n_pose_uploaded, n);
        TRY(ensure(ctx, ctx->lik, sizeof(float) * n));
        TRY(ensure(ctx, ctx->ratio, sizeof(float) * n));
        TRY(ensure(ctx, ctx->beam, sizeof(float) * n));
        const bool lik_wanted = out_lik || out_match_ratio;
        TRY(launch_measure(ctx, ctx->pose.as<float>(), n, lik_wanted ? ctx->lik.as<float>() : nullptr,
                           lik_wanted ? ctx->ratio.as<float>() : nullptr, out_beam ? ctx->beam.as<float>() : nullptr,
                           false, nullptr));
        if (out_lik)
          TRY(d2h(ctx, out_lik + lo, ctx->lik.p, sizeof(float) * n));
        if (out_match_ratio)
          TRY(d2h(ctx, out_match_ratio + lo, ctx->ratio.p, sizeof(float) * n));
        if (out_beam)
          TRY(d2h(ctx, out_beam + lo, ctx->beam.p, sizeof(float) * n));
        return sync_stream(ctx);
      },
      &bad);
  if (rc)
    return g->fail_rank(rc, bad);
  if (pose)
    g->n_pose_uploaded = n_p;
  return 0;
}

// Progressive form. One device: the context's own slices (api_core.inl). N devices (round 5): every
// rank takes its shard of the batch as a progressive batch of its own — poses, scans and kernels enqueued by its worker
// thread, nothing waited for — and the shards' results land in the caller's arrays shard by shard, slice by slice, through
// each rank's page-locked emission + polled completion word. _wait(particle) is answered by the rank that owns the particle
// (after the ranks in front of it: the reference's loop consumes the particles in order, pf.h:255-260), from the caller's
// thread; the other ranks keep computing. Results are those of mcl3dl_hip_group_measure_batch bit for bit.
int mcl3dl_hip_group_measure_batch_begin(mcl3dl_hip_group* g, const float* pose, size_t n_p, const float* scan_lik_xyz,
                                         size_t n_s, const float* scan_beam_xyz, const uint32_t* scan_beam_origin,
                                         size_t n_b, const float* origins, size_t n_o, float* out_lik,
                                         float* out_match_ratio, float* out_beam, size_t slice_particles)
{
  if (!g)
    return -1;
  TRY(mcl3dl_hip_group_measure_batch_end(g));  // (a batch still open is ended first, like the context form does)
  g->prog_n_p = 0;
  g->prog_direct = false;
  g->prog_sharded = false;
  if (!pose && n_p && g->n_pose_uploaded != n_p)
    return g->fail(-3, "null pose array (and mcl3dl_hip_group_upload_poses holds %zu poses, not %zu)", g->n_pose_uploaded,
                   n_p);
  if (g->n() == 1 && g->direct_single)
  {
    const int rc = mcl3dl_hip_measure_batch_begin(g->ctx[0], pose, n_p, scan_lik_xyz, n_s, scan_beam_xyz, scan_beam_origin, n_b,
                                                  origins, n_o, out_lik, out_match_ratio, out_beam, slice_particles);
    if (rc)
      return g->fail_rank(rc, 0);
    g->prog_n_p = n_p;
    g->prog_direct = true;
    return 0;
  }
  if (n_p == 0)
    return 0;
  const int N = g->n();
  if (pose)
    g->n_pose_uploaded = 0;
  int bad = 0;
  const int rc = g->pool.run_all(
      [&](int r) -> int
      {
        mcl3dl_hip_ctx* ctx = g->ctx[r];
        size_t lo, hi;
        shard_bounds(n_p, N, r, &lo, &hi);
        const size_t n = hi - lo;
        HIP_TRY(hipSetDevice(ctx->device));
        if (n == 0)
          return 0;
        // (the context form slices a shard of >= 1024 particles in four by default; shorter shards, and shards whose shape the
        // staged path does not take, are evaluated whole before this returns — they are short)
        return mcl3dl_hip_measure_batch_begin(ctx, pose ? pose + 7 * lo : nullptr, n, scan_lik_xyz, n_s, scan_beam_xyz,
                                              scan_beam_origin, n_b, origins, n_o, out_lik ? out_lik + lo : nullptr,
                                              out_match_ratio ? out_match_ratio + lo : nullptr,
                                              out_beam ? out_beam + lo : nullptr, slice_particles);
      },
      &bad);
  g->prog_done.assign(static_cast<size_t>(N), false);
  if (rc)
  {
    for (int r = 0; r < N; ++r)  // the other ranks' batches are drained before the caller gets control back
    {
      (void)hipSetDevice(g->ctx[r]->device);
      (void)mcl3dl_hip_measure_batch_end(g->ctx[r]);
    }
    return g->fail_rank(rc, bad);
  }
  if (pose)
    g->n_pose_uploaded = n_p;
  g->prog_n_p = n_p;
  g->prog_sharded = true;
  return 0;
}

int mcl3dl_hip_group_measure_batch_wait(mcl3dl_hip_group* g, size_t particle, size_t* n_ready)
{
  if (!g)
    return -1;
  if (particle >= g->prog_n_p)
    return g->fail(-3, "particle %zu is not part of the batch (%zu particles)", particle, g->prog_n_p);
  if (g->prog_direct)
  {
    const int rc = mcl3dl_hip_measure_batch_wait(g->ctx[0], particle, n_ready);
    return rc ? g->fail_rank(rc, 0) : 0;
  }
  if (!g->prog_sharded)
  {
    if (n_ready)
      *n_ready = g->prog_n_p;
    return 0;
  }
  const int N = g->n();
  for (int r = 0; r < N; ++r)
  {
    size_t lo, hi;
    shard_bounds(g->prog_n_p, N, r, &lo, &hi);
    if (hi == lo)
      continue;
    mcl3dl_hip_ctx* ctx = g->ctx[r];
    if (particle >= hi)
    {
      // a rank in front of the owner: all of it must have arrived for the particles up to `particle` to be a leading run
      if (!g->prog_done[static_cast<size_t>(r)])
      {
        (void)hipSetDevice(ctx->device);
        const int rc = mcl3dl_hip_measure_batch_wait(ctx, hi - lo - 1, nullptr);
        if (rc)
          return g->fail_rank(rc, r);
        g->prog_done[static_cast<size_t>(r)] = true;
      }
      continue;
    }
    (void)hipSetDevice(ctx->device);
    size_t n = 0;
    const int rc = mcl3dl_hip_measure_batch_wait(ctx, particle - lo, &n);
    if (rc)
      return g->fail_rank(rc, r);
    if (n >= hi - lo)
      g->prog_done[static_cast<size_t>(r)] = true;
    if (n_ready)
      *n_ready = lo + n;
    return 0;
  }
  return g->fail(-3, "particle %zu belongs to no shard", particle);
}

int mcl3dl_hip_group_measure_batch_end(mcl3dl_hip_group* g)
{
  if (!g)
    return -1;
  if (g->prog_direct)
  {
    g->prog_direct = false;
    const int rc = mcl3dl_hip_measure_batch_end(g->ctx[0]);
    if (rc)
      return g->fail_rank(rc, 0);
  }
  if (g->prog_sharded)
  {
    g->prog_sharded = false;
    int first_rc = 0, first_r = 0;
    for (int r = 0; r < g->n(); ++r)
    {
      (void)hipSetDevice(g->ctx[r]->device);
      const int rc = mcl3dl_hip_measure_batch_end(g->ctx[r]);
      if (rc && !first_rc)
      {
        first_rc = rc;
        first_r = r;
      }
    }
    if (first_rc)
      return g->fail_rank(first_rc, first_r);
  }
  return 0;
}

namespace
{
int resident_poses(mcl3dl_hip_ctx* ctx);  // api_group_state.inl

// One update over the group's shards. resident = false: mcl3dl_hip_group_measure_update (poses and prior weights come from the
// host, the weights go back). resident = true: the particles mcl3dl_hip_group_upload_state / _resample_apply left on the
// devices (pose = first 7 floats of each 13-float state, kept as ctx->pose; weights in ctx->gs_weight, updated in place);
// weight_inout is then an optional OUTPUT (may be null: nothing but four scalars comes back).
int group_update_impl(mcl3dl_hip_group* g, bool resident, const float* pose, const float* extra, float* weight_inout,
                      size_t n_p, const float* scan_lik_xyz, size_t n_s, const float* scan_beam_xyz,
                      const uint32_t* scan_beam_origin, size_t n_b, const float* origins, size_t n_o, float* out_lik,
                      float* out_match_ratio, float* out_beam, float* entropy, float* match_ratio_min, float* match_ratio_max,
                      int* restored)
{
  if (resident && g->n_resident == 0)
    return g->fail(-5, "no resident particles (mcl3dl_hip_group_upload_state first)");
  if (n_p == 0)
    return g->fail(-3, "no particles");
  if (!resident && (!pose || !weight_inout))
    return g->fail(-3, "null pose / weight array");
  if (resident && g->n_resident != n_p)
    return g->fail(-5, "%zu particles are resident on the group's devices, not %zu (mcl3dl_hip_group_upload_state first)",
                   g->n_resident, n_p);
  if (!resident && g->n() == 1 && g->direct_single)
  {
    const int rc = mcl3dl_hip_measure_update(g->ctx[0], pose, extra, weight_inout, n_p, scan_lik_xyz, n_s, scan_beam_xyz,
                                             scan_beam_origin, n_b, origins, n_o, out_lik, out_match_ratio, out_beam,
                                             entropy, match_ratio_min, match_ratio_max, restored);
    return rc ? g->fail_rank(rc, 0) : 0;
  }
  // a group of one device that may call its context directly needs no collective (and never loads RCCL)
  const bool no_collective = g->n() == 1 && g->direct_single;
  if (!no_collective)
    TRY(group_comms(g));
  // Large scans are ordered by every device for itself (upload_scan_impl: raw points up, keys / stable radix sort / gather
  // there — the same order as the host's, bit for bit): N redundant sorts of ~0.05 ms that run side by side, instead of
  // 0.15 ms of one host core at 16 k points ahead of any GPU work. Small scans are ordered once on the host and pushed.
  const bool device_order = g->ctx[0]->scan_order_device > 0 && n_s + n_b >= static_cast<size_t>(g->ctx[0]->scan_order_device);
  // (a rank whose staging launch is not eligible and whose scan is small pushes a copy ordered ONCE on the host, by whichever
  // rank needs it first)
  std::string host_order_error;
  std::once_flag host_order_once;
  bool host_ordered = false;
  const int N = g->n();
  const size_t n_pack = 2 + 2 * static_cast<size_t>(N);
  const bool host_combine = g->collective == 1 && !no_collective;
  std::vector<float> stats(4 * static_cast<size_t>(N), 0.f);
  std::vector<float*> rank_weights(static_cast<size_t>(N), nullptr);  // where each rank's prior weights are on its device
  if (!resident)
    g->n_pose_uploaded = 0;

  // phase A: upload the shard, measure, partial sums, (RCCL) all-reduce, and — with RCCL — straight on to phase B
  const auto phase_b = [&](mcl3dl_hip_ctx* ctx, int r, size_t lo, size_t n) -> int
  {
    const size_t fb = sizeof(float) * n;
    TRY(ensure(ctx, ctx->stats4, sizeof(float) * 4));
    if (n)
    {
      float* d_w = rank_weights[r];
      // the normalising kernel writes the shard's results straight into page-locked memory (the caller's arrays where they
      // are page-locked, a staging block otherwise) and the rank learns of its completion from a polled word — instead of up
      // to five D2H copies and a hipStreamSynchronize per rank
      const size_t rpart = (fb + 63) & ~static_cast<size_t>(63);
      char* blk = (ctx->zero_copy() && ctx->poll_mode() && 64 + 4 * rpart <= STAGE_MAX_COPY) ?
                      static_cast<char*>(stage_alloc(ctx, 64 + 4 * rpart)) : nullptr;
      if (blk)
      {
        PfEmit e{};
        e.stats4 = reinterpret_cast<float*>(blk);
        ctx->stage_out.push_back({ &stats[4 * r], e.stats4, sizeof(float) * 4 });
        float* const user[4] = { weight_inout, out_lik, out_match_ratio, out_beam };
        float** const slot[4] = { &e.w, &e.lik, &e.ratio, &e.beam };
        for (int k = 0; k < 4; ++k)
          if (user[k])
          {
            float* u = user[k] + lo;
            *slot[k] = ctx->is_pinned(u, fb) ? u : reinterpret_cast<float*>(blk + 64 + k * rpart);
            if (*slot[k] != u)
              ctx->stage_out.push_back({ u, *slot[k], fb });
          }
        HIP_TRY(hipSetDevice(ctx->device));
        EventPair ep{};
        TRY(timing_begin(ctx, MCL3DL_KERNEL_PF, &ep));
        hipLaunchKernelGGL(pf_apply_kernel, dim3(pf_blocks(n)), dim3(PF_BLOCK), 0, ctx->stream, d_w, ctx->wnew.as<float>(),
                           static_cast<int>(n), N, ctx->packed.as<double>(), ctx->stats4.as<float>(), e, ctx->lik.as<float>(),
                           ctx->ratio.as<float>(), ctx->beam.as<float>());
        TRY(timing_end(ctx, ep));
        HIP_TRY(hipGetLastError());
        return sync_stream(ctx, true);
      }
      TRY(mcl3dl_hip_pf_apply_device(ctx, d_w, n, N, ctx->packed.as<double>(), ctx->stats4.as<float>()));
      if (weight_inout)
        TRY(d2h(ctx, weight_inout + lo, d_w, fb));
      TRY(d2h(ctx, &stats[4 * r], ctx->stats4.p, sizeof(float) * 4));
      if (out_lik)
        TRY(d2h(ctx, out_lik + lo, ctx->lik.p, fb));
      if (out_match_ratio)
        TRY(d2h(ctx, out_match_ratio + lo, ctx->ratio.p, fb));
      if (out_beam)
        TRY(d2h(ctx, out_beam + lo, ctx->beam.p, fb));
    }
    return sync_stream(ctx);
  };
  int bad = 0;
  std::vector<int> rcs(N, 0);
  const int inject = g->inject_failure_rank;
  g->inject_failure_rank = -1;
  constexpr int RC_ABANDONED = -8;
  int rc = g->pool.run_all(
      [&](int r) -> int
      {
        mcl3dl_hip_ctx* ctx = g->ctx[r];
        size_t lo, hi;
        shard_bounds(n_p, N, r, &lo, &hi);
        const size_t n = hi - lo, fb = sizeof(float) * n;
        // everything up to the collective; nothing in here waits for another rank
        const auto phase_a = [&]() -> int
        {
          HIP_TRY(hipSetDevice(ctx->device));
          if (resident && n)
          {
            if (ctx->gs_n != n)
              return ctx->fail(-5, "this device holds %zu resident particles, its shard has %zu", ctx->gs_n, n);
            // ctx->pose is shared with the non-resident calls (uploads, mcl3dl_hip_group_measure_update, moments of explicit
            // states): whoever wrote it last, the poses evaluated here are those of the resident states
            TRY(resident_poses(ctx));
          }
          // ONE launch takes the rank's inputs over — the raw scans (every rank orders them for itself, side by side), its
          // pose / weight / odometry-factor shard — out of page-locked memory (stage_kernels.h); where that form is not
          // eligible: uploads + the ordering launches, or the scans ordered once on the host and pushed
          bool staged = false;
          if (n)
          {
            const int st = measure_update_staged(ctx, resident ? nullptr : pose + 7 * lo, extra ? extra + lo : nullptr,
                                                 resident ? nullptr : weight_inout + lo, n, scan_lik_xyz, n_s, scan_beam_xyz,
                                                 scan_beam_origin, n_b, origins, n_o, nullptr, nullptr, nullptr, nullptr, true,
                                                 STAGE_FRONT_ONLY);
            if (st < 0)
              return st;
            staged = st == 3;
          }
          if (!staged)
          {
            if (device_order)
              TRY(upload_scan_impl(ctx, scan_lik_xyz, n_s, scan_beam_xyz, scan_beam_origin, n_b, origins, n_o, false));
            else
            {
              std::call_once(host_order_once,
                             [&]
                             {
                               host_ordered = order_scan(host_order_error, scan_lik_xyz, n_s, scan_beam_xyz, scan_beam_origin, n_b,
                                                         origins, n_o, g->scan, g->ctx[0]->scan_presorted != 0) == 0;
                             });
              if (!host_ordered)
                return ctx->fail(-3, "%s", host_order_error.c_str());
              TRY(push_scan(ctx, g->scan, false));
            }
          }
          TRY(ensure(ctx, ctx->packed, sizeof(double) * n_pack));
          if (!resident && !staged)
            ctx->poses_set(0);
          if (n)
          {
            TRY(ensure(ctx, ctx->lik, fb));
            TRY(ensure(ctx, ctx->ratio, fb));
            TRY(ensure(ctx, ctx->beam, fb));
            TRY(ensure(ctx, ctx->extra, fb));
            if (!resident && !staged)
            {
              TRY(ensure(ctx, ctx->pose, sizeof(float) * 7 * n));
              TRY(ensure(ctx, ctx->weightb, fb));
              TRY(h2d(ctx, ctx->pose.p, pose + 7 * lo, sizeof(float) * 7 * n));
              ctx->poses_set(n);
              TRY(h2d(ctx, ctx->weightb.p, weight_inout + lo, fb));
            }
            float* d_w = resident ? ctx->gs_weight.as<float>() : (staged ? staged_weights(ctx) : ctx->weightb.as<float>());
            rank_weights[r] = d_w;
            if (extra && !staged)
              TRY(h2d(ctx, ctx->extra.p, extra + lo, fb));
            // (the sum over the tiled kernel's per-tile partials and the beam model's last step ride in the first pf::measure kernel,
            // as on one GPU: one launch less per model and rank)
            LikTail tail;
            tail.want = n <= static_cast<size_t>(1024) * PF_BLOCK;
            tail.want_beam = true;
            TRY(launch_measure(ctx, ctx->pose.as<float>(), n, ctx->lik.as<float>(), ctx->ratio.as<float>(),
                               ctx->beam.as<float>(), false, nullptr, &tail));
            TRY(pf_partial_behind_measure(ctx, d_w, ctx->lik.as<float>(), ctx->beam.as<float>(),
                                          extra ? ctx->extra.as<float>() : nullptr, ctx->ratio.as<float>(), n, r, N,
                                          ctx->packed.as<double>(), tail));
          }
          else
            TRY(pack_empty(ctx, r, N, ctx->packed.as<double>()));
          if (r == inject)
            return ctx->fail(-9, "injected failure ahead of the collective (test hook)");
          return 0;
        };
        int rc_a = phase_a();
        // the vote: the collective is entered by all ranks or by none
        const bool all_ok = g->vote.vote(rc_a == 0);
        if (rc_a == 0 && !all_ok)
          rc_a = ctx->fail(RC_ABANDONED, "update abandoned: another rank failed ahead of the collective");
        if (rc_a != 0)
        {
          (void)hipStreamSynchronize(ctx->stream);  // drain what this rank enqueued; its results are discarded
          ctx->stage_out.clear();
          ctx->stage_cur = 0;
          ctx->stage_off = 0;
          ctx->stage_pending = 0;
          return rcs[r] = rc_a;
        }
        if (no_collective)
          return rcs[r] = phase_b(ctx, r, lo, n);
        if (host_combine)
        {
          g->host_packed[r].resize(n_pack);
          TRY(d2h(ctx, g->host_packed[r].data(), ctx->packed.p, sizeof(double) * n_pack));
          return rcs[r] = sync_stream(ctx);
        }
        // the update's single collective: 16 + 16 N bytes over xGMI, on this device's stream
        const ncclResult_t nrc = g->rccl.AllReduce(ctx->packed.p, ctx->packed.p, n_pack, ncclDouble, ncclSum, g->comms[r],
                                                   ctx->stream);
        // second vote: a collective that one rank could not enqueue never completes on the ranks that did — nobody may
        // wait for its stream then; the communicators are aborted below
        const bool enqueued = g->vote.vote(nrc == ncclSuccess);
        if (nrc != ncclSuccess)
          return rcs[r] = ctx->fail(-7, "ncclAllReduce failed: %s", g->rccl.GetErrorString(nrc));
        if (!enqueued)
        {
          ctx->stage_out.clear();
          return rcs[r] = ctx->fail(RC_ABANDONED, "update abandoned: another rank could not enqueue the all-reduce");
        }
        return rcs[r] = phase_b(ctx, r, lo, n);
      },
      &bad);
  if (rc)
  {
    // report the rank that actually failed, not one that merely stood down
    for (int r = 0; r < N; ++r)
      if (rcs[r] != 0 && rcs[r] != RC_ABANDONED)
      {
        rc = rcs[r];
        bad = r;
        break;
      }
    // communicators that saw a failed or abandoned update are rebuilt on next use
    if (!host_combine && !g->comms.empty())
      g->drop_comms();
    return g->fail_rank(rc, bad);
  }
  if (host_combine)
  {
    // sum the N records in rank order (deterministic) and hand the total back to every device
    std::vector<double> total(n_pack, 0.0);
    for (int r = 0; r < N; ++r)
      for (size_t i = 0; i < n_pack; ++i)
        total[i] += g->host_packed[r][i];
    rc = g->pool.run_all(
        [&](int r) -> int
        {
          mcl3dl_hip_ctx* ctx = g->ctx[r];
          size_t lo, hi;
          shard_bounds(n_p, N, r, &lo, &hi);
          HIP_TRY(hipSetDevice(ctx->device));
          TRY(h2d(ctx, ctx->packed.p, total.data(), sizeof(double) * n_pack));
          return phase_b(ctx, r, lo, hi - lo);
        },
        &bad);
    if (rc)
      return g->fail_rank(rc, bad);
    ++g->collectives_host;
  }
  else if (!no_collective)
    ++g->collectives_rccl;
  g->n_pose_uploaded = n_p;
  // every rank computed the same four scalars from the same all-reduced record: take the first non-empty shard's
  int src = 0;
  for (int r = 0; r < N; ++r)
  {
    size_t lo, hi;
    shard_bounds(n_p, N, r, &lo, &hi);
    if (hi > lo)
    {
      src = r;
      break;
    }
  }
  if (entropy)
    *entropy = stats[4 * src + 0];
  if (match_ratio_min)
    *match_ratio_min = stats[4 * src + 1];
  if (match_ratio_max)
    *match_ratio_max = stats[4 * src + 2];
  if (restored)
    *restored = stats[4 * src + 3] != 0.0f;
  return 0;
}
}  // namespace

int mcl3dl_hip_group_measure_update(mcl3dl_hip_group* g, const float* pose, const float* extra, float* weight_inout,
                                    size_t n_p, const float* scan_lik_xyz, size_t n_s, const float* scan_beam_xyz,
                                    const uint32_t* scan_beam_origin, size_t n_b, const float* origins, size_t n_o,
                                    float* out_lik, float* out_match_ratio, float* out_beam, float* entropy,
                                    float* match_ratio_min, float* match_ratio_max, int* restored)
{
  if (!g)
    return -1;
  return group_update_impl(g, false, pose, extra, weight_inout, n_p, scan_lik_xyz, n_s, scan_beam_xyz, scan_beam_origin, n_b,
                           origins, n_o, out_lik, out_match_ratio, out_beam, entropy, match_ratio_min, match_ratio_max,
                           restored);
}
