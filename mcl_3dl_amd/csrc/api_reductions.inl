// api_reductions.inl — included inside the extern "C" block of mcl3dl_hip.hip.
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
  ctx->poses_set(0);  // the pose buffer is about to hold these states, not an uploaded particle set
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
  ctx->poses_set(0);  // the pose buffer is about to hold these states, not an uploaded particle set
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

