// api_support.inl — included inside the extern "C" block of mcl3dl_hip.hip: kernel timing, footprint, options.
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
  // any option may change which kernels an update enqueues: a captured update graph (use_graph) is re-captured after every
  // call, whichever option it names (options are set once per deployment, captures cost microseconds)
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
    if (value != 0.0 && !(value >= 0.125 && value <= 2.0))
      return ctx->fail(-3, "cand_voxel_ratio must be 0 (chosen per map) or in [0.125, 2]");
    if (value != ctx->cand_voxel_ratio)
      ctx->cand_dirty = true;
    ctx->cand_voxel_ratio = value;
    return 0;
  }
  if (key == "cand_aniso")
  {
    if (value != 0.0 && value != 1.0 && value != 2.0)
      return ctx->fail(-3, "cand_aniso must be 0 (cubes), 1 (boxes that follow the dist_weight) or 2 (boxes when cubes exceed the budget)");
    if (static_cast<int>(value) != ctx->cand_aniso)
      ctx->cand_dirty = true;
    ctx->cand_aniso = static_cast<int>(value);
    return 0;
  }
  if (key == "cand_aniso_max")
  {
    if (!(value >= 1.0 && value <= 64.0))
      return ctx->fail(-3, "cand_aniso_max must be in [1, 64]");
    if (value != ctx->cand_aniso_max)
      ctx->cand_dirty = true;
    ctx->cand_aniso_max = value;
    return 0;
  }
  if (key == "index_budget_bytes")
  {
    if (!(value >= 0.0 || value == -1.0))
      return ctx->fail(-3, "index_budget_bytes must be >= 0 (0 = no budget) or -1 (a quarter of the device's memory)");
    if (value != ctx->index_budget_opt)
      ctx->cand_dirty = true;
    ctx->index_budget_opt = value;
    return 0;
  }
  if (key == "strict_order")
  {
    if (!(value == 0.0 || value == 1.0 || value == 2.0 || value == 3.0))
      return ctx->fail(-3, "strict_order must be 0 (never), 1 (always, weights too), 2 (large scans only) or 3 (always, in the engine's scan order)");
    if (static_cast<int>(value) != ctx->strict_order)
      ++ctx->generation;  // a captured update graph holds the other kernel selection
    ctx->strict_order = static_cast<int>(value);
    return 0;
  }
  if (key == "overlap_min_rays")
  {
    if (!(value >= 0.0 && value <= 9.0e18))
      return ctx->fail(-3, "overlap_min_rays must be >= 0");
    ctx->overlap_min_rays = static_cast<long long>(value);
    return 0;
  }
  if (key == "resample_prefix_device")
  {
    ctx->resample_prefix_device = value != 0.0;
    return 0;
  }
  if (key == "update_small")
  {
    ctx->update_small = value != 0.0;
    return 0;
  }
  if (key == "update_stage")
  {
    ctx->update_stage = value != 0.0;
    return 0;
  }
  if (key == "update_zero_copy")
  {
    ctx->update_zero_copy = value != 0.0;
    return 0;
  }
  if (key == "pf_tail")
  {
    ctx->pf_tail = value != 0.0;
    return 0;
  }
  if (key == "poll_sync")
  {
    if (value != 0.0 && value != 1.0 && value != 2.0)
      return ctx->fail(-3, "poll_sync must be 0, 1 or 2");
    ctx->poll_sync = static_cast<int>(value);
    return 0;
  }
  if (key == "test_late_structures")
  {
    const char* hooks = getenv("MCL3DL_HIP_TEST_HOOKS");
    if (!hooks || std::string(hooks) != "1")
      return ctx->fail(-3, "test_late_structures is a test hook: set MCL3DL_HIP_TEST_HOOKS=1 in the environment to enable it");
    ctx->test_late_structures = value != 0.0;
    return 0;
  }
  if (key == "poll_spin_us")
  {
    if (!(value >= 0.0 && value <= 1e9))
      return ctx->fail(-3, "poll_spin_us must be >= 0");
    ctx->poll_spin_us = value;
    return 0;
  }
  if (key == "chain_ppl")
  {
    if (value != 0.0 && value != 1.0 && value != 4.0)
      return ctx->fail(-3, "chain_ppl must be 0 (by size), 1 or 4");
    ctx->chain_ppl = static_cast<int>(value);
    return 0;
  }
  if (key == "chain_multi_max")
  {
    if (!(value >= 0.0 && value <= 1e9))
      return ctx->fail(-3, "chain_multi_max must be >= 0");
    ctx->chain_multi_max = static_cast<int>(value);
    return 0;
  }
  if (key == "update_fold_done")
  {
    ctx->fold_done_opt = value != 0.0;
    return 0;
  }
  if (key == "poll_query_us")
  {
    if (!(value >= 100.0 && value <= 1e9))
      return ctx->fail(-3, "poll_query_us must be >= 100");
    ctx->poll_query_us = value;
    return 0;
  }
  if (key == "strict_rows")
  {
    ctx->strict_rows = value != 0.0;
    return 0;
  }
  if (key == "dda_overlay")
  {
    ctx->dda_overlay = value != 0.0;
    ctx->dda_dirty = true;
    return 0;
  }
  if (key == "cand_prune_coop")
  {
    ctx->cand_prune_coop = value != 0.0;
    return 0;
  }
  if (key == "batch_slice")
  {
    if (value < 0.0 || value > 1.0e9)
      return ctx->fail(-3, "batch_slice must be a particle count (0 = automatic)");
    ctx->batch_slice = static_cast<int>(value);
    return 0;
  }
  if (key == "update_particle")
  {
    ctx->update_particle = value != 0.0;
    return 0;
  }
  if (key == "update_small_conformant")
  {
    ctx->update_small_conformant = value != 0.0;
    return 0;
  }
  if (key == "update_small_max")
  {
    if (!(value >= 1.0 && value <= 65536.0))
      return ctx->fail(-3, "update_small_max must be in [1, 65536]");
    ctx->update_small_max = static_cast<int>(value);
    return 0;
  }
  if (key == "scan_presorted")
  {
    ctx->scan_presorted = value != 0.0 ? 1 : 0;
    return 0;
  }
  if (key == "strict_chunk")
  {
    if (!(value == 0.0 || (value >= 1024.0 && value <= 1e9)))
      return ctx->fail(-3, "strict_chunk must be 0 (replay the scan in one piece) or a point count >= 1024");
    ctx->strict_chunk = static_cast<int>(value);
    return 0;
  }
  if (key == "strict_auto_min")
  {
    if (!(value >= 1.0 && value <= 2147483647.0))
      return ctx->fail(-3, "strict_auto_min must be a positive point count");
    if (static_cast<int>(value) != ctx->strict_auto_min)
      ++ctx->generation;
    ctx->strict_auto_min = static_cast<int>(value);
    return 0;
  }
  if (key == "strict_skew")
  {
    ctx->strict_skew = value != 0.0;
    return 0;
  }
  if (key == "strict_gpw")
  {
    if (value != 0.0 && value != 1.0 && value != 2.0)
      return ctx->fail(-3, "strict_gpw must be 0 (chosen per launch), 1 or 2");
    ctx->strict_gpw = static_cast<int>(value);
    return 0;
  }
  if (key == "strict_auto_max_bytes")
  {
    if (!(value >= 0.0))
      return ctx->fail(-3, "strict_auto_max_bytes must be >= 0");
    ctx->strict_auto_max_bytes = value;
    return 0;
  }
  if (key == "timing_mask")
  {
    ctx->timing_mask = static_cast<unsigned>(value);
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
  if (key == "lik_tiled_min")
  {
    if (!(value >= 1.0 && value <= 1e9))
      return ctx->fail(-3, "lik_tiled_min must be >= 1");
    ctx->lik_tiled_min = static_cast<int>(value);
    return 0;
  }
  if (key == "lik_group")
  {
    if (value != 0.0 && value != 4.0 && value != 8.0 && value != 16.0 && value != 32.0)
      return ctx->fail(-3, "lik_group must be 0 (chosen per launch), 4, 8, 16 or 32");
    ctx->lik_group = static_cast<int>(value);
    return 0;
  }
  if (key == "scan_order_device")
  {
    if (!(value >= 0.0 && value <= 2e9))
      return ctx->fail(-3, "scan_order_device must be >= 0");
    ctx->scan_order_device = static_cast<int>(value);
    return 0;
  }
  if (key == "pf_fused")
  {
    ctx->pf_fused = value != 0.0;
    return 0;
  }
  if (key == "sort_full_pass")
  {
    ctx->sort_full_pass = value != 0.0;
    return 0;
  }
  if (key == "sort_one_launch")
  {
    ctx->sort_one_launch = value != 0.0;
    return 0;
  }
  if (key == "pf_fused_max")
  {
    if (!(value >= 1.0 && value <= static_cast<double>(PF_FUSED_MAX)))
      return ctx->fail(-3, "pf_fused_max must be 1..%d", PF_FUSED_MAX);
    ctx->pf_fused_max = static_cast<int>(value);
    ++ctx->generation;
    return 0;
  }
  if (key == "lik_coop")
  {
    ctx->lik_coop = value != 0.0;
    return 0;
  }
  if (key == "lik_defer")
  {
    if (!(value == 0.0 || value == 1.0 || value == 2.0))
      return ctx->fail(-3, "lik_defer must be 0 (never), 1 (whenever the records allow it) or 2 (crowded maps only)");
    if (static_cast<int>(value) != ctx->lik_defer)
    {
      ctx->cand_dirty = true;  // the record size of a crowded map follows it (host_map_compilers.h:build_cand_grid)
      ++ctx->generation;       // a captured update graph holds the other kernel
    }
    ctx->lik_defer = static_cast<int>(value);
    return 0;
  }
  if (key == "lik_defer_min_frac")
  {
    if (!(value >= 0.0 && value <= 1.0))
      return ctx->fail(-3, "lik_defer_min_frac must be in [0, 1]");
    ctx->lik_defer_min_frac = value;
    ++ctx->generation;
    return 0;
  }
  if (key == "cand_bound")
  {
    if ((value != 0.0) != (ctx->cand_bound != 0))
      ctx->cand_dirty = true;
    ctx->cand_bound = value != 0.0 ? 1 : 0;
    return 0;
  }
  if (key == "cand_packed")
  {
    if ((value != 0.0) != (ctx->cand_packed != 0))
      ctx->cand_dirty = true;
    ctx->cand_packed = value != 0.0 ? 1 : 0;
    return 0;
  }
  if (key == "beam_prepare")
  {
    ctx->beam_prepare = value != 0.0;
    ++ctx->generation;
    return 0;
  }
  if (key == "lik_wide_max_particles")
  {
    if (!(value >= 0.0 && value <= 2e9))
      return ctx->fail(-3, "lik_wide_max_particles must be >= 0");
    ctx->lik_wide_max_particles = static_cast<int>(value);
    ++ctx->generation;
    return 0;
  }
  if (key == "grid_build_host")
  {
    const int v = value != 0.0;
    if (v != ctx->grid_build_host)
    {
      ctx->grid_build_host = v;
      ctx->lik_dirty = ctx->dda_dirty = ctx->lik_base_dirty = true;
      ++ctx->generation;
    }
    return 0;
  }
  if (key == "cand_record_parts")
  {
    if (value != 0.0 && value != 4.0 && value != 8.0)
      return ctx->fail(-3, "cand_record_parts must be 0 (chosen per map), 4 (64-byte records) or 8 (128-byte records)");
    if (static_cast<int>(value) != ctx->cand_record_parts)
    {
      ctx->cand_dirty = true;
      ++ctx->generation;
    }
    ctx->cand_record_parts = static_cast<int>(value);
    return 0;
  }
  if (key == "cand_refine" || key == "cand_refine_above")
  {
    const bool which = key == "cand_refine";
    if (!(value >= (which ? 1.0 : 0.0) && value <= (which ? 4.0 : 32.0)))
      return ctx->fail(-3, "cand_refine must be in [1, 4], cand_refine_above in [0, 32]");
    int& field = which ? ctx->cand_refine : ctx->cand_refine_above;
    if (static_cast<int>(value) != field)
      ctx->cand_dirty = true;
    field = static_cast<int>(value);
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

int mcl3dl_hip_get_option(mcl3dl_hip_ctx* ctx, const char* name, double* value)
{
  if (!ctx || !name || !value)
    return -1;
  const std::string key(name);
  if (key == "lik_index") *value = ctx->lik_index;
  else if (key == "cand_voxel_ratio") *value = ctx->cand_voxel_ratio;
  else if (key == "cand_phase") *value = ctx->cand_phase;
  else if (key == "cand_aniso") *value = ctx->cand_aniso;
  else if (key == "cand_aniso_max") *value = ctx->cand_aniso_max;
  else if (key == "index_budget_bytes") *value = ctx->index_budget_opt;
  else if (key == "index_budget_in_use") *value = ctx->index_budget_bytes;
  else if (key == "cand_aniso_active") *value = ctx->cand_aniso_active ? 1.0 : 0.0;
  else if (key == "cand_edge_ratio_x") *value = ctx->cand_edge_ratio[0];
  else if (key == "cand_edge_ratio_y") *value = ctx->cand_edge_ratio[1];
  else if (key == "cand_edge_ratio_z") *value = ctx->cand_edge_ratio[2];
  else if (key == "index_record_bytes") *value = static_cast<double>(ctx->footprint[6]);
  else if (key == "index_note") *value = ctx->index_note.empty() ? 0.0 : 1.0;
  else if (key == "cand_record_parts") *value = ctx->cand_record_parts;
  else if (key == "cand_record_parts_in_use") *value = ctx->cand_parts;
  else if (key == "cand_voxels_over8") *value = ctx->cand_over8;
  else if (key == "cand_ovf_compactions") *value = static_cast<double>(ctx->cand_ovf_compactions);
  else if (key == "cand_ovf_leaked") *value = ctx->cand_ovf_leaked;
  else if (key == "strict_order") *value = ctx->strict_order;
  else if (key == "strict_auto_min") *value = ctx->strict_auto_min;
  else if (key == "strict_chunk") *value = ctx->strict_chunk;
  else if (key == "scan_presorted") *value = ctx->scan_presorted;
  else if (key == "scan_chunk_in_use") *value = static_cast<double>(ctx->scan_chunk);
  else if (key == "lik_grid_merges") *value = static_cast<double>(ctx->lik_grid_merges);
  else if (key == "lik_grid_rebuilds") *value = static_cast<double>(ctx->lik_grid_rebuilds);
  else if (key == "strict_gpw") *value = ctx->strict_gpw;
  else if (key == "strict_skew") *value = ctx->strict_skew;
  else if (key == "strict_auto_max_bytes") *value = ctx->strict_auto_max_bytes;
  else if (key == "strict_auto_skipped") *value = static_cast<double>(ctx->strict_auto_skipped);
  else if (key == "overlap_min_rays") *value = static_cast<double>(ctx->overlap_min_rays);
  else if (key == "resample_prefix_device") *value = ctx->resample_prefix_device;
  else if (key == "cand_refine") *value = ctx->cand_refine;
  else if (key == "cand_refine_above") *value = ctx->cand_refine_above;
  else if (key == "update_small") *value = ctx->update_small;
  else if (key == "update_stage") *value = ctx->update_stage;
  else if (key == "update_zero_copy") *value = ctx->update_zero_copy;
  else if (key == "pf_tail") *value = ctx->pf_tail;
  else if (key == "update_particle") *value = ctx->update_particle;
  else if (key == "poll_sync") *value = ctx->poll_sync;
  else if (key == "poll_spin_us") *value = ctx->poll_spin_us;
  else if (key == "poll_query_us") *value = ctx->poll_query_us;
  else if (key == "update_fold_done") *value = ctx->fold_done_opt ? 1.0 : 0.0;
  else if (key == "chain_ppl") *value = ctx->chain_ppl;
  else if (key == "chain_multi_max") *value = ctx->chain_multi_max;
  else if (key == "batch_slice") *value = ctx->batch_slice;
  else if (key == "cand_prune_coop") *value = ctx->cand_prune_coop;
  else if (key == "dda_overlay") *value = ctx->dda_overlay;
  else if (key == "strict_rows") *value = ctx->strict_rows;
  else if (key == "dda_overlay_updates") *value = static_cast<double>(ctx->dda_overlay_updates);
  else if (key == "dda_overlay_points") *value = ctx->dda_dirty ? 0.0 : static_cast<double>(ctx->dg.ov_n);
  else if (key == "batch_slices_run") *value = static_cast<double>(ctx->batch_slices_run);
  else if (key == "update_small_max") *value = ctx->update_small_max;
  else if (key == "update_small_conformant") *value = ctx->update_small_conformant;
  else if (key == "timing_mask") *value = ctx->timing_mask;
  else if (key == "overlap_models") *value = ctx->overlap_models;
  else if (key == "lik_small") *value = ctx->lik_small;
  else if (key == "lik_tiled") *value = ctx->lik_tiled;
  else if (key == "lik_group") *value = ctx->lik_group;
  else if (key == "lik_tiled_min") *value = ctx->lik_tiled_min;
  else if (key == "lik_coop") *value = ctx->lik_coop;
  else if (key == "lik_defer") *value = ctx->lik_defer;
  else if (key == "lik_defer_min_frac") *value = ctx->lik_defer_min_frac;
  else if (key == "lik_defer_active") *value = lik_defer_active(ctx) ? 1.0 : 0.0;
  else if (key == "cand_packed") *value = ctx->cand_packed;
  else if (key == "cand_packed_active") *value = ctx->rg.packed;
  else if (key == "cand_bound") *value = ctx->cand_bound;
  else if (key == "cand_bound_active") *value = ctx->rg.bound_step > 0.0f ? 1.0 : 0.0;
  else if (key == "beam_prepare") *value = ctx->beam_prepare;
  else if (key == "lik_wide_max_particles") *value = ctx->lik_wide_max_particles;
  else if (key == "grid_build_host") *value = ctx->grid_build_host;
  else if (key == "lik_grid_build_ms") *value = ctx->grid_build_ms[0];
  else if (key == "dda_grid_build_ms") *value = ctx->grid_build_ms[1];
  else if (key == "lik_grid_build_wall_ms") *value = ctx->grid_build_wall_ms[0];
  else if (key == "dda_grid_build_wall_ms") *value = ctx->grid_build_wall_ms[1];
  else if (key == "pf_fused") *value = ctx->pf_fused;
  else if (key == "sort_full_pass") *value = ctx->sort_full_pass;
  else if (key == "sort_one_launch") *value = ctx->sort_one_launch;
  else if (key == "pf_fused_max") *value = ctx->pf_fused_max;
  else if (key == "scan_order_device") *value = ctx->scan_order_device;
  else
    return ctx->fail(-3, "unknown option '%s'", name);
  return 0;
}

// ---- page-locked host memory for the caller's arrays ---------------------------------------------------------------------
int mcl3dl_hip_host_alloc(mcl3dl_hip_ctx* ctx, size_t bytes, void** out)
{
  if (!ctx || !out)
    return -1;
  *out = nullptr;
  if (bytes == 0)
    return ctx->fail(-3, "mcl3dl_hip_host_alloc of 0 bytes");
  HIP_TRY(hipSetDevice(ctx->device));
  void* p = pinned_alloc(ctx, bytes);
  if (!p)
    return ctx->fail(-2, "hipHostMalloc of %zu bytes failed", bytes);
  ctx->pinned.push_back({ static_cast<char*>(p), bytes });
  *out = p;
  return 0;
}

int mcl3dl_hip_host_free(mcl3dl_hip_ctx* ctx, void* p)
{
  if (!ctx)
    return -1;
  if (!p)
    return 0;
  for (size_t k = 0; k < ctx->pinned.size(); ++k)
    if (ctx->pinned[k].p == p)
    {
      HIP_TRY(hipSetDevice(ctx->device));
      TRY(sync_stream(ctx));  // nothing in flight reads or writes it
      HIP_TRY(hipHostFree(p));
      ctx->pinned.erase(ctx->pinned.begin() + static_cast<long>(k));
      return 0;
    }
  return ctx->fail(-3, "mcl3dl_hip_host_free: not a block of mcl3dl_hip_host_alloc");
}

int mcl3dl_hip_index_stats(mcl3dl_hip_ctx* ctx, double* stats8)
{
  if (!ctx || !stats8)
    return -1;
  for (int i = 0; i < 8; ++i)
    stats8[i] = ctx->cand_stats[i];
  return 0;
}
