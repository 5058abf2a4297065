// api_group_state.inl — included inside the extern "C" block of mcl3dl_hip.hip: a device group's particles RESIDENT on its
// GPUs between updates, and the steps either side of pf::measure over the shards (SURVEY.md 8f-1, 8f-3 for N GPUs behind the
// C ABI — the reference's process is C++ and cannot bind the torch.distributed helpers of mcl_3dl_amd/distributed.py):
//
//   mcl3dl_hip_group_upload_state / _download_state    std::vector<Particle<State6DOF>> <-> shards (include/mcl_3dl/pf.h:457)
//   mcl3dl_hip_group_update_resident                   pf::measure over the resident particles: nothing but the scan goes up
//   mcl3dl_hip_group_expectation / _covariance         pf.h:294-303,361-390 / :304-360: one record per shard, combined on the host
//   mcl3dl_hip_group_resample_begin / _plan / _apply   pf.h:187-225 / 399-436: the float prefix recurrence runs ONCE, in particle
//                                                      order (it is not associative: per-shard prefixes plus offsets would not
//                                                      reproduce the reference's accum_probability_), every rank plans all slots
//                                                      and writes its slice of the new generation from the all-gathered states
//                                                      (RCCL ncclAllGather over xGMI, or through the host with collective = 1)
namespace
{
__global__ void state13_to_pose7_kernel(const float* __restrict__ state13, int n, float* __restrict__ pose7)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
#pragma unroll
  for (int k = 0; k < 7; ++k)
    pose7[7 * static_cast<size_t>(i) + k] = state13[13 * static_cast<size_t>(i) + k];
}

// padded all-gather layout [rank][max_count][13] -> flat particle order (shards differ in size by at most one)
__global__ void unpad_states_kernel(const float* __restrict__ padded, size_t max_count, size_t base, size_t rem, size_t n,
                                    float* __restrict__ flat)
{
  const size_t t = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= 13 * n)
    return;
  const size_t i = t / 13, k = t % 13;
  // owner of particle i under shard_bounds: the first `rem` shards hold base + 1 particles
  const size_t split = rem * (base + 1);
  const size_t r = i < split ? i / (base + 1) : rem + (base ? (i - split) / base : 0);
  const size_t lo = r * base + (r < rem ? r : rem);
  flat[t] = padded[(r * max_count + (i - lo)) * 13 + k];
}

int rebuild_pose(mcl3dl_hip_ctx* ctx, size_t n)
{
  ctx->poses_set(0);
  if (n == 0)
    return 0;
  TRY(ensure(ctx, ctx->pose, sizeof(float) * 7 * n));
  const int ni = static_cast<int>(n);
  hipLaunchKernelGGL(state13_to_pose7_kernel, dim3((ni + 255) / 256), dim3(256), 0, ctx->stream,
                     ctx->gs_state[ctx->gs_cur].as<float>(), ni, ctx->pose.as<float>());
  HIP_TRY(hipGetLastError());
  ctx->poses_set(n);
  ctx->pose_resident = true;
  return 0;
}

// the resident calls read the poses from ctx->pose: re-derived from the resident states whenever another call (an upload, a
// non-resident update, a moments call with its own states) has written that buffer since
int resident_poses(mcl3dl_hip_ctx* ctx)
{
  if (ctx->gs_n && (!ctx->pose_resident || ctx->n_pose_uploaded != ctx->gs_n))
    return rebuild_pose(ctx, ctx->gs_n);
  return 0;
}
}  // namespace

int mcl3dl_hip_group_upload_state(mcl3dl_hip_group* g, const float* state13, const float* weight, size_t n_p)
{
  if (!g)
    return -1;
  if (!state13 || n_p == 0 || n_p > 0x7fffffffu / 16)
    return g->fail(-3, "bad state array");
  g->n_resident = 0;
  g->rs_begun = g->rs_planned = false;
  int bad = 0;
  const int N = g->n();
  const int rc = g->pool.run_all(
      [&](int r) -> int
      {
        mcl3dl_hip_ctx* ctx = g->ctx[r];
        size_t lo, hi;
        shard_bounds(n_p, N, r, &lo, &hi);
        const size_t n = hi - lo;
        HIP_TRY(hipSetDevice(ctx->device));
        ctx->gs_n = 0;
        ctx->gs_cur = 0;
        if (n == 0)
          return 0;
        // (room for the LARGEST shard on every rank: the all-gather of the resampling step sends that many states)
        const size_t cap_count = (n_p + N - 1) / N;
        TRY(ensure(ctx, ctx->gs_state[0], sizeof(float) * 13 * cap_count));
        TRY(ensure(ctx, ctx->gs_weight, sizeof(float) * cap_count));
        TRY(h2d(ctx, ctx->gs_state[0].p, state13 + 13 * lo, sizeof(float) * 13 * n));
        if (weight)
          TRY(h2d(ctx, ctx->gs_weight.p, weight + lo, sizeof(float) * n));
        else
          hipLaunchKernelGGL(fill_kernel, dim3((static_cast<int>(n) + 255) / 256), dim3(256), 0, ctx->stream,
                             ctx->gs_weight.as<float>(), 1.0f / static_cast<float>(n_p), static_cast<float*>(nullptr), 0.0f,
                             static_cast<int>(n));
        TRY(rebuild_pose(ctx, n));
        ctx->gs_n = n;
        return sync_stream(ctx);
      },
      &bad);
  if (rc)
    return g->fail_rank(rc, bad);
  g->n_resident = n_p;
  g->n_pose_uploaded = n_p;
  return 0;
}

int mcl3dl_hip_group_download_state(mcl3dl_hip_group* g, float* state13, float* weight, size_t n_p)
{
  if (!g)
    return -1;
  if (g->n_resident == 0 || g->n_resident != n_p)
    return g->fail(-5, "%zu particles are resident, %zu asked for", g->n_resident, n_p);
  int bad = 0;
  const int N = g->n();
  const int rc = g->pool.run_all(
      [&](int r) -> int
      {
        mcl3dl_hip_ctx* ctx = g->ctx[r];
        size_t lo, hi;
        shard_bounds(n_p, N, r, &lo, &hi);
        const size_t n = hi - lo;
        if (n == 0)
          return 0;
        HIP_TRY(hipSetDevice(ctx->device));
        if (state13)
          TRY(d2h(ctx, state13 + 13 * lo, ctx->gs_state[ctx->gs_cur].p, sizeof(float) * 13 * n));
        if (weight)
          TRY(d2h(ctx, weight + lo, ctx->gs_weight.p, sizeof(float) * n));
        return sync_stream(ctx);
      },
      &bad);
  return rc ? g->fail_rank(rc, bad) : 0;
}

size_t mcl3dl_hip_group_resident(const mcl3dl_hip_group* g)
{
  return g ? g->n_resident : 0;
}

int mcl3dl_hip_group_update_resident(mcl3dl_hip_group* g, const float* extra, const float* scan_lik_xyz, size_t n_s,
                                     const float* scan_beam_xyz, const uint32_t* scan_beam_origin, size_t n_b,
                                     const float* origins, size_t n_o, float* out_weight, float* out_lik,
                                     float* out_match_ratio, float* out_beam, float* entropy, float* match_ratio_min,
                                     float* match_ratio_max, int* restored)
{
  if (!g)
    return -1;
  return group_update_impl(g, true, nullptr, extra, out_weight, g->n_resident, scan_lik_xyz, n_s, scan_beam_xyz, scan_beam_origin,
                           n_b, origins, n_o, out_lik, out_match_ratio, out_beam, entropy, match_ratio_min, match_ratio_max,
                           restored);
}

int mcl3dl_hip_group_expectation(mcl3dl_hip_group* g, const float* bias, float* out_mean7, float* out_total,
                                 int64_t* out_max_index, int64_t* out_max_biased_index)
{
  if (!g)
    return -1;
  const size_t n_p = g->n_resident;
  if (n_p == 0)
    return g->fail(-5, "no resident particles (mcl3dl_hip_group_upload_state first)");
  const int N = g->n();
  g->h_parts.assign(16 * static_cast<size_t>(N), 0.0);
  std::vector<uint64_t> offsets(N, 0);
  int bad = 0;
  const int rc = g->pool.run_all(
      [&](int r) -> int
      {
        mcl3dl_hip_ctx* ctx = g->ctx[r];
        size_t lo, hi;
        shard_bounds(n_p, N, r, &lo, &hi);
        const size_t n = hi - lo;
        offsets[r] = lo;
        double* rec = &g->h_parts[16 * static_cast<size_t>(r)];
        if (n == 0)
        {
          rec[10] = -1.0;  // an empty shard never holds the maximum
          rec[12] = -1.0;
          return 0;
        }
        HIP_TRY(hipSetDevice(ctx->device));
        TRY(ensure(ctx, ctx->gs_rec, sizeof(double) * 32));
        const float* d_bias = nullptr;
        if (bias)
        {
          TRY(ensure(ctx, ctx->extra, sizeof(float) * n));
          TRY(h2d(ctx, ctx->extra.p, bias + lo, sizeof(float) * n));
          d_bias = ctx->extra.as<float>();
        }
        TRY(resident_poses(ctx));
        TRY(mcl3dl_hip_moments_partial_device(ctx, ctx->pose.as<float>(), ctx->gs_weight.as<float>(), d_bias, n,
                                              ctx->gs_rec.as<double>()));
        TRY(d2h(ctx, rec, ctx->gs_rec.p, sizeof(double) * 16));
        return sync_stream(ctx);
      },
      &bad);
  if (rc)
    return g->fail_rank(rc, bad);
  if (mcl3dl_hip_moments_finish(g->h_parts.data(), N, offsets.data(), out_mean7, out_total, out_max_index,
                                out_max_biased_index) != 0)
    return g->fail(-3, "moments_finish rejected the records");
  return 0;
}

int mcl3dl_hip_group_covariance(mcl3dl_hip_group* g, const float* mean7, float* out_cov36)
{
  if (!g)
    return -1;
  const size_t n_p = g->n_resident;
  if (n_p == 0)
    return g->fail(-5, "no resident particles (mcl3dl_hip_group_upload_state first)");
  if (!mean7 || !out_cov36)
    return g->fail(-3, "null mean / covariance array");
  const int N = g->n();
  g->h_parts.assign(22 * static_cast<size_t>(N), 0.0);
  int bad = 0;
  const int rc = g->pool.run_all(
      [&](int r) -> int
      {
        mcl3dl_hip_ctx* ctx = g->ctx[r];
        size_t lo, hi;
        shard_bounds(n_p, N, r, &lo, &hi);
        const size_t n = hi - lo;
        if (n == 0)
          return 0;
        HIP_TRY(hipSetDevice(ctx->device));
        TRY(ensure(ctx, ctx->gs_rec, sizeof(double) * 32));
        TRY(resident_poses(ctx));
        TRY(mcl3dl_hip_covariance_partial_device(ctx, ctx->pose.as<float>(), ctx->gs_weight.as<float>(), n, nullptr, 0, mean7,
                                                 ctx->gs_rec.as<double>()));
        TRY(d2h(ctx, &g->h_parts[22 * static_cast<size_t>(r)], ctx->gs_rec.p, sizeof(double) * 22));
        return sync_stream(ctx);
      },
      &bad);
  if (rc)
    return g->fail_rank(rc, bad);
  double total[22] = { 0 };
  for (int r = 0; r < N; ++r)  // rank order: deterministic
    for (int k = 0; k < 22; ++k)
      total[k] += g->h_parts[22 * static_cast<size_t>(r) + k];
  return mcl3dl_hip_covariance_finish(total, out_cov36);
}

int mcl3dl_hip_group_resample_begin(mcl3dl_hip_group* g, size_t n_out, float* out_pstep)
{
  if (!g)
    return -1;
  const size_t n_p = g->n_resident;
  if (n_p == 0)
    return g->fail(-5, "no resident particles (mcl3dl_hip_group_upload_state first)");
  if (n_out == 0)
    n_out = n_p;
  g->rs_begun = g->rs_planned = false;
  // every weight comes to the host once (4 bytes per particle): the prefix recurrence of pf.h:193-197 is sequential
  g->h_weight.resize(n_p);
  TRY(mcl3dl_hip_group_download_state(g, nullptr, g->h_weight.data(), n_p));
  const int N = g->n();
  std::vector<float> pstep(N, 0.f);
  int bad = 0;
  const int rc = g->pool.run_all(
      [&](int r) -> int { return mcl3dl_hip_resample_begin(g->ctx[r], g->h_weight.data(), n_p, n_out, &pstep[r]); }, &bad);
  if (rc)
    return g->fail_rank(rc, bad);
  g->rs_n_out = n_out;
  g->rs_begun = true;
  if (out_pstep)
    *out_pstep = pstep[0];
  return 0;
}

int mcl3dl_hip_group_resample_plan(mcl3dl_hip_group* g, int mode, float initial_p, uint32_t* out_source,
                                   uint8_t* out_duplicate, size_t* out_n_duplicates)
{
  if (!g)
    return -1;
  if (!g->rs_begun)
    return g->fail(-5, "group_resample_plan before group_resample_begin");
  const int N = g->n();
  std::vector<size_t> n_dup(N, 0);
  int bad = 0;
  const int rc = g->pool.run_all(
      [&](int r) -> int
      {
        return mcl3dl_hip_resample_plan(g->ctx[r], mode, initial_p, r == 0 ? out_source : nullptr,
                                        r == 0 ? out_duplicate : nullptr, &n_dup[r]);
      },
      &bad);
  if (rc)
    return g->fail_rank(rc, bad);
  for (int r = 1; r < N; ++r)
    if (n_dup[r] != n_dup[0])
      return g->fail(-4, "the ranks disagree on the resampling plan (%zu and %zu duplicated particles)", n_dup[0], n_dup[r]);
  g->rs_n_dup = n_dup[0];
  g->rs_planned = true;
  if (out_n_duplicates)
    *out_n_duplicates = n_dup[0];
  return 0;
}

int mcl3dl_hip_group_resample_apply(mcl3dl_hip_group* g, const float* noise13, size_t n_noise)
{
  if (!g)
    return -1;
  if (!g->rs_planned)
    return g->fail(-5, "group_resample_apply before group_resample_plan");
  if (n_noise < g->rs_n_dup || (g->rs_n_dup && !noise13))
    return g->fail(-3, "group_resample_apply: %zu duplicated particles need noise, %zu given", g->rs_n_dup, n_noise);
  const size_t n_p = g->n_resident, n_out = g->rs_n_out;
  const int N = g->n();
  const bool single = N == 1 && g->direct_single;
  const bool host_gather = g->collective == 1 && !single;
  if (!single && !host_gather)
    TRY(group_comms(g));
  if (host_gather)
  {
    g->h_state.resize(13 * n_p);
    TRY(mcl3dl_hip_group_download_state(g, g->h_state.data(), nullptr, n_p));
  }
  const size_t base = n_p / N, rem = n_p % N, max_count = base + (rem ? 1 : 0);
  int bad = 0;
  std::vector<int> rcs(N, 0);
  constexpr int RC_ABANDONED = -8;
  int rc = g->pool.run_all(
      [&](int r) -> int
      {
        mcl3dl_hip_ctx* ctx = g->ctx[r];
        size_t lo, hi, olo, ohi;
        shard_bounds(n_p, N, r, &lo, &hi);
        shard_bounds(n_out, N, r, &olo, &ohi);
        const size_t n = hi - lo, n_new = ohi - olo;
        const int other = ctx->gs_cur ^ 1;
        const float* d_all = nullptr;
        // everything ahead of the collective; the all-gather is entered by all ranks or by none
        const auto prepare = [&]() -> int
        {
          HIP_TRY(hipSetDevice(ctx->device));
          const size_t cap_new = std::max<size_t>((n_out + N - 1) / N, 1);
          TRY(ensure(ctx, ctx->gs_state[other], sizeof(float) * 13 * cap_new));
          TRY(ensure(ctx, ctx->gs_weight, sizeof(float) * cap_new));
          if (single)
            return 0;
          TRY(ensure(ctx, ctx->gs_all, sizeof(float) * 13 * n_p));
          if (host_gather)
            return h2d(ctx, ctx->gs_all.p, g->h_state.data(), sizeof(float) * 13 * n_p);
          TRY(ensure(ctx, ctx->gs_pad, sizeof(float) * 13 * max_count * static_cast<size_t>(N)));
          if (ctx->gs_state[ctx->gs_cur].cap < sizeof(float) * 13 * max_count)  // (every shard buffer holds the largest shard)
            return ctx->fail(-4, "internal: shard buffer smaller than the all-gather's send count");
          return 0;
        };
        int rc_p = prepare();
        const bool all_ok = g->vote.vote(rc_p == 0);
        if (rc_p == 0 && !all_ok)
          rc_p = ctx->fail(RC_ABANDONED, "resampling abandoned: another rank failed ahead of the all-gather");
        if (rc_p != 0)
        {
          (void)hipStreamSynchronize(ctx->stream);
          ctx->stage_out.clear();
          ctx->stage_cur = 0;
          ctx->stage_off = 0;
          ctx->stage_pending = 0;
          return rcs[r] = rc_p;
        }
        if (single)
          d_all = ctx->gs_state[ctx->gs_cur].as<float>();
        else if (host_gather)
          d_all = ctx->gs_all.as<float>();
        else
        {
          // 13 floats x max_count per rank over xGMI into the padded layout, then into particle order
          const ncclResult_t nrc = g->rccl.AllGather(ctx->gs_state[ctx->gs_cur].p, ctx->gs_pad.p, 13 * max_count, ncclFloat,
                                                     g->comms[r], ctx->stream);
          const bool enqueued = g->vote.vote(nrc == ncclSuccess);
          if (nrc != ncclSuccess)
            return rcs[r] = ctx->fail(-7, "ncclAllGather failed: %s", g->rccl.GetErrorString(nrc));
          if (!enqueued)
            return rcs[r] = ctx->fail(RC_ABANDONED, "resampling abandoned: another rank could not enqueue the all-gather");
          const size_t total = 13 * n_p;
          hipLaunchKernelGGL(unpad_states_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, ctx->stream,
                             ctx->gs_pad.as<float>(), max_count, base, rem, n_p, ctx->gs_all.as<float>());
          HIP_TRY(hipGetLastError());
          d_all = ctx->gs_all.as<float>();
        }
        (void)n;
        if (n_new)
        {
          TRY(mcl3dl_hip_resample_apply_slice_device(ctx, d_all, noise13, n_noise, olo, n_new, ctx->gs_state[other].as<float>()));
          // pf.h:207 / 417: every particle of the new generation weighs 1 / n
          hipLaunchKernelGGL(fill_kernel, dim3((static_cast<int>(n_new) + 255) / 256), dim3(256), 0, ctx->stream,
                             ctx->gs_weight.as<float>(), 1.0f / static_cast<float>(n_out), static_cast<float*>(nullptr), 0.0f,
                             static_cast<int>(n_new));
          HIP_TRY(hipGetLastError());
        }
        ctx->gs_cur = other;
        ctx->gs_n = n_new;
        TRY(rebuild_pose(ctx, n_new));
        return rcs[r] = sync_stream(ctx);
      },
      &bad);
  g->rs_begun = g->rs_planned = false;
  if (rc)
  {
    for (int r = 0; r < N; ++r)
      if (rcs[r] != 0 && rcs[r] != RC_ABANDONED)
      {
        rc = rcs[r];
        bad = r;
        break;
      }
    if (!single && !host_gather && !g->comms.empty())
      g->drop_comms();
    g->n_resident = 0;  // the shards may be half way into the new generation
    return g->fail_rank(rc, bad);
  }
  if (!single && !host_gather)
    ++g->collectives_rccl;
  else if (host_gather)
    ++g->collectives_host;
  g->n_resident = n_out;
  g->n_pose_uploaded = n_out;
  return 0;
}
