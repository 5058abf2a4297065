// api_resample.inl — included inside the extern "C" block of mcl3dl_hip.hip.
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
  if (n_out == 0 || n_out > 0x7fffffffu)
    return ctx->fail(-3, "bad arguments to resample_begin_device");
  // the prefix sums are a float recurrence in particle order (pf.h:193-197): 4 bytes per particle come to the host (0.3 ms at
  // 262 144 particles; one device lane running the recurrence took 3.9 ms — that form and its option went in round 6)
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

