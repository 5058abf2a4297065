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
    if (value != 0.0 && value != 2.0)
      return ctx->fail(-3, "lik_index must be 0 (27-cell scan) or 2 (candidate records)");
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
  if (key == "strict_exact_max")
  {
    // strict_order 2: scans of at most this many points are added up in the caller's order, as floats (0 = none)
    if (!(value >= 0.0 && value <= 2147483647.0))
      return ctx->fail(-3, "strict_exact_max must be a point count >= 0");
    if (static_cast<int>(value) != ctx->strict_exact_max)
      ++ctx->generation;
    ctx->strict_exact_max = static_cast<int>(value);
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
  if (key == "cand_bound")
  {
    if ((value != 0.0) != (ctx->cand_bound != 0))
      ctx->cand_dirty = true;
    ctx->cand_bound = value != 0.0 ? 1 : 0;
    return 0;
  }
  if (key == "beam_prepare")
  {
    ctx->beam_prepare = value != 0.0;
    return 0;
  }
  if (key == "cand_packed")
  {
    if ((value != 0.0) != (ctx->cand_packed != 0))
      ctx->cand_dirty = true;
    ctx->cand_packed = value != 0.0 ? 1 : 0;
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

// Every option that can be set can be read back, next to the read-only diagnostics (what the index in place was built with, the
// counters of the map path, build times): ONE table, name -> getter.
namespace
{
struct OptionGetter
{
  const char* name;
  double (*get)(const mcl3dl_hip_ctx*);
};
const OptionGetter kOptionGetters[] = {
    { "lik_index", [](const mcl3dl_hip_ctx* c) -> double { return c->lik_index; } },
    { "cand_voxel_ratio", [](const mcl3dl_hip_ctx* c) -> double { return c->cand_voxel_ratio; } },
    { "cand_phase", [](const mcl3dl_hip_ctx* c) -> double { return c->cand_phase; } },
    { "cand_aniso", [](const mcl3dl_hip_ctx* c) -> double { return c->cand_aniso; } },
    { "index_budget_bytes", [](const mcl3dl_hip_ctx* c) -> double { return c->index_budget_opt; } },
    { "index_budget_in_use", [](const mcl3dl_hip_ctx* c) -> double { return c->index_budget_bytes; } },
    { "cand_aniso_active", [](const mcl3dl_hip_ctx* c) -> double { return c->cand_aniso_active ? 1.0 : 0.0; } },
    { "cand_edge_ratio_x", [](const mcl3dl_hip_ctx* c) -> double { return c->cand_edge_ratio[0]; } },
    { "cand_edge_ratio_y", [](const mcl3dl_hip_ctx* c) -> double { return c->cand_edge_ratio[1]; } },
    { "cand_edge_ratio_z", [](const mcl3dl_hip_ctx* c) -> double { return c->cand_edge_ratio[2]; } },
    { "index_record_bytes", [](const mcl3dl_hip_ctx* c) -> double { return static_cast<double>(c->footprint[6]); } },
    { "index_note", [](const mcl3dl_hip_ctx* c) -> double { return c->index_note.empty() ? 0.0 : 1.0; } },
    { "cand_record_parts", [](const mcl3dl_hip_ctx* c) -> double { return c->cand_record_parts; } },
    { "cand_record_parts_in_use", [](const mcl3dl_hip_ctx* c) -> double { return c->cand_parts; } },
    { "cand_voxels_over8", [](const mcl3dl_hip_ctx* c) -> double { return c->cand_over8; } },
    { "cand_ovf_compactions", [](const mcl3dl_hip_ctx* c) -> double { return static_cast<double>(c->cand_ovf_compactions); } },
    { "cand_ovf_leaked", [](const mcl3dl_hip_ctx* c) -> double { return c->cand_ovf_leaked; } },
    { "strict_order", [](const mcl3dl_hip_ctx* c) -> double { return c->strict_order; } },
    { "strict_auto_min", [](const mcl3dl_hip_ctx* c) -> double { return c->strict_auto_min; } },
    { "strict_exact_max", [](const mcl3dl_hip_ctx* c) -> double { return c->strict_exact_max; } },
    { "lik_exact", [](const mcl3dl_hip_ctx* c) -> double { return c->lik_exact ? 1.0 : 0.0; } },
    { "strict_chunk", [](const mcl3dl_hip_ctx* c) -> double { return c->strict_chunk; } },
    { "scan_presorted", [](const mcl3dl_hip_ctx* c) -> double { return c->scan_presorted; } },
    { "scan_chunk_in_use", [](const mcl3dl_hip_ctx* c) -> double { return static_cast<double>(c->scan_chunk); } },
    { "lik_grid_merges", [](const mcl3dl_hip_ctx* c) -> double { return static_cast<double>(c->lik_grid_merges); } },
    { "lik_grid_rebuilds", [](const mcl3dl_hip_ctx* c) -> double { return static_cast<double>(c->lik_grid_rebuilds); } },
    { "strict_auto_max_bytes", [](const mcl3dl_hip_ctx* c) -> double { return c->strict_auto_max_bytes; } },
    { "strict_auto_skipped", [](const mcl3dl_hip_ctx* c) -> double { return static_cast<double>(c->strict_auto_skipped); } },
    { "update_small", [](const mcl3dl_hip_ctx* c) -> double { return c->update_small; } },
    { "update_stage", [](const mcl3dl_hip_ctx* c) -> double { return c->update_stage; } },
    { "update_zero_copy", [](const mcl3dl_hip_ctx* c) -> double { return c->update_zero_copy; } },
    { "poll_sync", [](const mcl3dl_hip_ctx* c) -> double { return c->poll_sync; } },
    { "poll_spin_us", [](const mcl3dl_hip_ctx* c) -> double { return c->poll_spin_us; } },
    { "chain_ppl", [](const mcl3dl_hip_ctx* c) -> double { return c->chain_ppl; } },
    { "batch_slice", [](const mcl3dl_hip_ctx* c) -> double { return c->batch_slice; } },
    { "cand_prune_coop", [](const mcl3dl_hip_ctx* c) -> double { return c->cand_prune_coop; } },
    { "dda_overlay", [](const mcl3dl_hip_ctx* c) -> double { return c->dda_overlay; } },
    { "dda_overlay_updates", [](const mcl3dl_hip_ctx* c) -> double { return static_cast<double>(c->dda_overlay_updates); } },
    { "dda_overlay_points", [](const mcl3dl_hip_ctx* c) -> double { return c->dda_dirty ? 0.0 : static_cast<double>(c->dg.ov_n); } },
    { "batch_slices_run", [](const mcl3dl_hip_ctx* c) -> double { return static_cast<double>(c->batch_slices_run); } },
    { "update_small_max", [](const mcl3dl_hip_ctx* c) -> double { return c->update_small_max; } },
    { "update_small_conformant", [](const mcl3dl_hip_ctx* c) -> double { return c->update_small_conformant; } },
    { "timing_mask", [](const mcl3dl_hip_ctx* c) -> double { return c->timing_mask; } },
    { "overlap_models", [](const mcl3dl_hip_ctx* c) -> double { return c->overlap_models; } },
    { "lik_small", [](const mcl3dl_hip_ctx* c) -> double { return c->lik_small; } },
    { "lik_tiled", [](const mcl3dl_hip_ctx* c) -> double { return c->lik_tiled; } },
    { "lik_group", [](const mcl3dl_hip_ctx* c) -> double { return c->lik_group; } },
    { "lik_tiled_min", [](const mcl3dl_hip_ctx* c) -> double { return c->lik_tiled_min; } },
    { "lik_coop", [](const mcl3dl_hip_ctx* c) -> double { return c->lik_coop; } },
    { "lik_defer", [](const mcl3dl_hip_ctx* c) -> double { return c->lik_defer; } },
    { "lik_defer_active", [](const mcl3dl_hip_ctx* c) -> double { return lik_defer_active(c) ? 1.0 : 0.0; } },
    { "beam_prepare", [](const mcl3dl_hip_ctx* c) -> double { return c->beam_prepare; } },
    { "cand_aniso_max", [](const mcl3dl_hip_ctx* c) -> double { return c->cand_aniso_max; } },
    { "cand_packed", [](const mcl3dl_hip_ctx* c) -> double { return c->cand_packed; } },
    { "cand_packed_active", [](const mcl3dl_hip_ctx* c) -> double { return c->rg.packed; } },
    { "cand_bound", [](const mcl3dl_hip_ctx* c) -> double { return c->cand_bound; } },
    { "cand_bound_active", [](const mcl3dl_hip_ctx* c) -> double { return c->rg.bound_step > 0.0f ? 1.0 : 0.0; } },
    { "grid_build_host", [](const mcl3dl_hip_ctx* c) -> double { return c->grid_build_host; } },
    { "lik_grid_build_ms", [](const mcl3dl_hip_ctx* c) -> double { return c->grid_build_ms[0]; } },
    { "dda_grid_build_ms", [](const mcl3dl_hip_ctx* c) -> double { return c->grid_build_ms[1]; } },
    { "lik_grid_build_wall_ms", [](const mcl3dl_hip_ctx* c) -> double { return c->grid_build_wall_ms[0]; } },
    { "dda_grid_build_wall_ms", [](const mcl3dl_hip_ctx* c) -> double { return c->grid_build_wall_ms[1]; } },
    { "pf_fused", [](const mcl3dl_hip_ctx* c) -> double { return c->pf_fused; } },
    { "scan_order_device", [](const mcl3dl_hip_ctx* c) -> double { return c->scan_order_device; } },
};
}  // namespace

int mcl3dl_hip_get_option(mcl3dl_hip_ctx* ctx, const char* name, double* value)
{
  if (!ctx || !name || !value)
    return -1;
  for (const OptionGetter& g : kOptionGetters)
    if (strcmp(g.name, name) == 0)
    {
      *value = g.get(ctx);
      return 0;
    }
  return ctx->fail(-3, "unknown option '%s'", name);
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
