// host_context.h — part of the single translation unit mcl3dl_hip.hip (included there, nowhere else): the context
// object behind mcl3dl_hip_ctx, error macros, device-buffer / staging helpers, hipEvent kernel timing.
#pragma once

namespace
{
// A device allocation owned by the context (grown by ensure(), released with the context: mcl3dl_hip_destroy selects the
// device before it deletes the context object).
struct DevBuf
{
  void* p = nullptr;
  size_t cap = 0;
  bool view = false;  // p points into another DevBuf's allocation (ensure_scan_block): never freed, never grown by ensure()
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf()
  {
    if (p && !view)
      (void)hipFree(p);
  }
  template <typename T>
  T* as() const
  {
    return static_cast<T*>(p);
  }
};

// One update's scans after the host-side ordering (api_core.inl:order_scan): Morton-ordered likelihood points + the
// permutation back to the caller's order, range-ordered beam points {x, y, z, origin id}, origins. The scratch vectors
// are kept so that a steady stream of scans allocates nothing.
struct OrderedScan
{
  std::vector<float4> lik, beam, origins;
  std::vector<uint32_t> perm;
  std::vector<uint32_t> key, key2, idx2;
  std::vector<std::pair<float, uint32_t>> beam_keys;
};

struct EventPair
{
  hipEvent_t start, stop;
  int kernel;
};
}  // namespace

struct mcl3dl_hip_ctx
{
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  // The two LiDAR models are independent until pf::measure: the beam kernels run on a second stream, forked from and
  // joined back into `stream` with events, so their (VALU-heavy, memory-light) waves fill the slots the likelihood
  // kernel leaves idle while it waits on L2.
  hipStream_t aux_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int overlap_models = 1;
  long long overlap_min_rays = 262144;  // launches below this many rays keep both models on one stream
  std::string err;

  // host copy of the map (kept to rebuild the device structures when parameters change)
  std::vector<float> map_xyz;
  std::vector<uint32_t> map_label;
  uint64_t stamp = 0;
  bool has_map = false;
  bool has_weight = false;
  float weight[3] = { 1.f, 1.f, 1.f };

  // LidarMeasurementModelLikelihoodParameters defaults, include/mcl_3dl/parameters.h:74-76
  float match_dist_min = 0.2f, match_dist_flat = 0.05f, match_weight = 5.0f;
  // LidarMeasurementModelBeamParameters defaults, include/mcl_3dl/parameters.h:96-112
  float map_grid[3] = { 0.1f, 0.1f, 0.1f };
  float dda_grid_size = 0.2f;
  float ray_angle_half = static_cast<float>(0.25 * M_PI / 180.0);
  float hit_range = 0.3f;
  float beam_likelihood_min = 0.2f;
  uint32_t beam_num_points = 3;
  float ang_total_ref = static_cast<float>(M_PI / 6.0);
  uint32_t filter_label_max = 0xFFFFFFFFu;
  int short_only = 1;
  // derived, src/lidar_measurement_model_beam.cpp:65-67
  float hit_range_sq = 0, beam_likelihood = 0, sin_total_ref = 0;

  bool lik_dirty = true, dda_dirty = true, cand_dirty = true;
  DevBuf lik_pts, lik_cells;
  // the cell grid of the BASE map alone (host_grid_builders.h): after a map update the grid in use (lik_pts / lik_cells) is the
  // merge of this one with the update's points — no sort of the map, no histogram (round 5)
  DevBuf lik_base_pts, lik_base_cells;
  bool lik_base_dirty = true;
  size_t lik_base_n = 0;
  float lik_base_lo[3] = { 0, 0, 0 }, lik_base_hi[3] = { 0, 0, 0 };  // rescaled bounds the base grid's geometry was laid out for
  uint64_t lik_grid_merges = 0, lik_grid_rebuilds = 0;
  std::vector<float4> lik_upd_host;  // the update's rescaled points (host staging of the merge)
  LikGrid lg{};
  // lik_index 2 = the candidate-voxel records (map_compiler.h) serve measure(), 0 = 27-cell scan of the cell grid (the canonical
  // structure of SURVEY.md 8d, also what STATS counts on)
  int lik_index = 2;
  int lik_small = 1;       // 1 = several particles share a wavefront when the scan has <= 32 points
  int lik_tiled = 1;       // 1 = tile-major XCD-aware kernel for large scans, 0 = one work-group per particle always
  int lik_tiled_min = 1024;  // scans of at least this many points take the tiled kernel
  int lik_group = 0;       // particles per work-group of the tiled kernel: 0 = chosen per launch, or 4 / 8 / 16 / 32
  // up to this many particles a scan of > 512 points gets 1024 threads per particle: 512 x 16 wavefronts are ONE round of the
  // chip's 8192 wavefront slots (profiles/r06m_wide_threshold.txt: 64 x 4096 27 -> 16 us, 300 x 3000 27 -> 22, 512 x 4096 34 -> 29;
  // 1024 particles and more: 1.2 x SLOWER). Round 2 had set 64 from scans of ~1000 points; since round 6 this kernel serves
  // every default-mode scan up to 4096 points below 2048 particles (caller-order rows)
  int lik_wide_max_particles = 512;
  int lik_coop = 1;        // tiled kernel: 1 = quad-cooperative record fetch + VALU-trimmed evaluation (same results)
  DevBuf lik_partial_sum, lik_partial_cnt;
  int pf_fused = 1;        // 1 = pf::measure as ONE kernel up to pf_fused_max particles on one GPU (same bits, two launches fewer)
  int pf_fused_max = 1024;  // measured: one work-group beats three launches up to 1024 particles, ties at 2048, loses at 4096
  // 1 = add the likelihood terms AND the weights in the reference's float order (single GPU; bit-identical results);
  // 2 (default) = replay the likelihood terms in that order for scans of at least strict_auto_min points, where the
  // reference's own float rounding (a random walk of n_s roundings) reaches the 1e-5 tolerance of north_star; 0 = never
  int strict_order = 2;
  // strict_order 2: exact caller-order float sums for scans of at most strict_exact_max points and of at least strict_auto_min.
  // strict_auto_min comes from a bound, not from a measurement: the reference's float recurrence over n positive terms
  // differs from the exact sum by a random walk of n roundings, each within half an ulp of the running sum — relative
  // standard deviation <= 2^-24 sqrt(n) / 3 (ulp(s_i) <= 2^-23 s_i, s_i ~ (i / n) S) — while the fp64 tree is the exact sum
  // rounded once. Three standard deviations stay inside north_star's 1e-5 up to n = (1e-5 x 2^24)^2 = 28 147 points; from
  // there on the default replays the reference's own order. (A bound on the WORST case, n x 2^-24, would put that limit at
  // 168 points: sums of equal terms can drift systematically — which is what strict_exact_max and strict_order 1 are for.)
  int strict_auto_min = 28147;
  int strict_exact_max = 4096;
  int strict_rows_max_particles = 2048;  // exact sums by the per-particle kernels' LDS rows below this many particles, by the tiled kernel + replay from it
  // the likelihoods the last launch_measure / one-launch update enqueued are the reference's floats bit for bit (caller-order
  // rows, replay, in-kernel chain): pf::measure on one GPU then adds the weights in the reference's float order as well
  bool lik_exact = false;
  double strict_auto_max_bytes = 0.0;  // > 0: the automatic replay is also skipped when its buffer would exceed this many bytes
  uint64_t strict_auto_skipped = 0;  // launches of the automatic mode that summed in fp64 because the replay buffer did not fit
  DevBuf scan_block;  // { perm | lik scan | beam scan | origins } of the current update in ONE allocation (ensure_scan_block)
  DevBuf scan_perm, strict_terms;
  // Option strict_chunk > 0: a scan whose terms will be replayed in the caller's order is ORDERED in chunks of that order
  // (strict_chunk points each, Morton order inside a chunk; scan_perm then holds indices relative to the chunk): chunk c + 1 is
  // evaluated while chunk c is replayed on a stream of its own, and the term buffer holds two chunks instead of the whole scan
  // (host_measure.h). Measured and OFF by default (profiles/r05g_chunked_replay.txt): the term buffer of C5 shrinks from 17 GB to
  // 4.3 GB, but the update takes 27.9 instead of 25.6 ms — the replay's 1024-thread, 128 KB work-groups find no room next to
  // the tiled kernel's (eight 19 KB work-groups per CU), so the two do not overlap and the shorter launches cost their tails.
  // scan_chunk = chunk size the scan in place was ordered with (0 = one piece).
  size_t scan_chunk = 0;
  int strict_chunk = 0;
  int scan_presorted = 0;  // option: likelihood scans arrive in the engine's order already — no ordering pass (see include/mcl3dl_hip.h)
  hipStream_t replay_stream = nullptr;
  hipEvent_t ev_tiled[2] = { nullptr, nullptr }, ev_replay[2] = { nullptr, nullptr };
  // the whole update as one launch (update_kernels.h) up to update_small_max particles when the per-particle likelihood
  // kernel would run anyway: same bits, two to four launches fewer. Measured (profiles/r03*_update_small.txt): ahead of the
  // separate kernels up to ~500 particles (64 x 96 + 3: 26.9 -> 22.4 us, 64 x 1000: 13.7 -> 10.3), behind from 1024 on —
  // every work-group's arrival is an atomic the memory side serialises, and there are as many as particles
  int update_small = 1;
  int update_small_max = 512;
  int update_small_conformant = 0;  // 1 = acq_rel arrival tickets at agent scope (update_kernels.h:last_arrival)
  DevBuf us_tickets;
  // host-buffer updates (mcl3dl_hip_measure_update): update_stage = 1: the caller's scans / poses / weights are taken over by
  // ONE launch (stage_kernels.h:scan_stage_kernel — ordering included) for scans up to ST_MAX_POINTS points per model;
  // update_zero_copy = 1: that kernel reads them where they lie in page-locked host memory and the last kernel of the update
  // writes the results there (no DMA copy either way), 0 = one H2D copy of the staged block, one D2H copy of the results.
  // (Rounds 4-5 also carried a two-launch tail — pf_norm_kernel, every work-group recomputing the reduction — and a
  // per-particle-only form of the one-launch update; both measured slower than the split kernels and are gone: HISTORY.md.)
  int update_stage = 1;
  int update_zero_copy = 1;
  // host-buffer updates end by POLLING a word in page-locked memory that a one-thread kernel behind the update's last kernel
  // writes, instead of hipStreamSynchronize: 6.3 against 12.2 us for launch + completion of one kernel on this part
  // (profiles/r04e_launch_cost.txt). 2 (default) = every synchronisation of the context's stream is that word (scan
  // preparation -5..10 %, the post-update reductions -10 %, a whole filter iteration -8 %: profiles/r04ad_poll_all.txt),
  // 1 = only the host-buffer update and its relatives, 0 = hipStreamSynchronize everywhere.
  int poll_sync = 2;
  // what the PLATFORM can do, set by pinned_alloc alone: page-locked memory visible to the device at the host's address. The
  // two options above are what the caller asked for; what runs is option && capability (an option could
  // switch a mode back on that the platform had ruled out, and update_zero_copy = 0 used to switch the polled word off for good)
  bool zero_copy_supported = true;
  bool zero_copy() const
  {
    return update_zero_copy != 0 && zero_copy_supported;
  }
  int poll_mode() const
  {
    return zero_copy_supported ? poll_sync : 0;
  }
  // completion waits: the polled word is spun on for at most poll_spin_us microseconds (covers every update up to a few
  // thousand particles), then polled between naps that grow with the time already waited (a 25 ms update of 65 536 particles
  // costs its caller ~2 ms of CPU, not 25), with hipStreamQuery looked at every few milliseconds so that a faulted queue
  // comes back as an error instead of an endless wait
  double poll_spin_us = 2000.0;
  double poll_query_us = 5000.0;  // interval of the hipStreamQuery health checks while napping (option "poll_query_us")
  bool test_late_structures = false;  // fault injection for the API-sequence fuzz (option of the same name, test hooks only)
  volatile unsigned* done_flag = nullptr;
  unsigned done_seq = 0;
  // strict_order = 3 (likelihood_kernels.h: LikChain): hand-off words, their tag counter, the page-locked error word
  DevBuf chain_carry, chain_lik;
  uint32_t chain_tag = 1;
  volatile unsigned* chain_err = nullptr;
  // a measure_batch delivered in particle slices (mcl3dl_hip_measure_batch_begin / _wait / _end): slice k is in the host
  // arrays once the completion word has reached seq0 + k + 1
  struct BatchProgress
  {
    bool active = false;
    size_t n_p = 0, slice = 0, n_slices = 0, delivered = 0;  // delivered = slices already handed to the caller's arrays
    unsigned seq0 = 0;
    float* user[3] = { nullptr, nullptr, nullptr };
    const float* host[3] = { nullptr, nullptr, nullptr };  // page-locked source of each (== user when that is page-locked)
  } prog;
  int cand_prune_coop = 1;  // option "cand_prune_coop": 16 lanes per voxel in the map compiler's pruning pass (0 = one thread)
  int batch_slice = 0;  // option "batch_slice": particles per slice of a progressive batch (0 = automatic)
  uint64_t batch_slices_run = 0;
  DevBuf stage_in_dev;
  // page-locked host memory handed out by mcl3dl_hip_host_alloc: arrays inside it are read / written in place
  struct PinnedBlock
  {
    char* p;
    size_t bytes;
  };
  std::vector<PinnedBlock> pinned;
  bool is_pinned(const void* q, size_t bytes) const
  {
    const char* c = static_cast<const char*>(q);
    for (const PinnedBlock& b : pinned)
      if (c >= b.p && c + bytes <= b.p + b.bytes)
        return true;
    return false;
  }
  double cand_voxel_ratio = 0.0;  // voxel edge / match_dist_min; 0 = chosen per map (host_map_compilers.h:build_cand_grid)
  double cand_phase = 0.5;        // grid origin shifted by this fraction of a voxel (see build_cand_grid)
  int cand_aniso = 2;             // option: voxel edges follow the dist_weight axis by axis (host_map_compilers.h:cand_axis_stretch): 0 never, 1 always, 2 when cubes exceed the budget
  bool cand_aniso_active = false; // ... what the index in place was built with
  double cand_aniso_max = 8.0;    // ... up to this factor over the base edge
  double index_budget_opt = -1.0;   // option index_budget_bytes: upper bound of the candidate records; -1 = a quarter of the device's memory, 0 = none
  double index_budget_bytes = 0.0;  // ... resolved at build time
  double cand_need_bytes = 0.0;     // record bytes the last priced voxel edge needs
  double cand_edge_ratio[3] = { 0.5, 0.5, 0.5 };  // voxel edge / match_dist_min of the index in place, per axis
  int cand_record_parts = 0;      // inline candidates per voxel record: 4 (64 bytes), 8 (128 bytes), 0 = chosen per map
  uint32_t cand_parts = 4;        // what the current index was built with
  int n_cus = 256;                // compute units of the device (hipDeviceAttributeMultiprocessorCount)
  int cand_packed = 1;            // option: packed w words in the voxel records when the map allows it (map_compiler.h)
  int cand_bound = 1;             // option: ... with the skip bound of the overflow candidates (the bounded form) when the map allows it
  int lik_defer = 1;              // option: overflow rounds of the tiled kernel deferred and run densely: 0 never, 1 always
                                  // (packed 64-byte records), 2 = when more than lik_defer_min_frac of the voxels overflow
  double lik_defer_min_frac = 0.03;
  double cand_over8 = 0;          // voxels with more than eight candidates (index statistics)
  DevBuf cand_table, cand_start, cand_pts, cand_rec, cand_ovf;
  // kept for map updates (host_map_compilers.h:update_cand_grid): geometry, sizes, every rescaled map point
  DevBuf cand_all_pts;
  CompileParams cand_cp{};
  long long cand_n_table = 0;
  uint32_t cand_n_bricks = 0, cand_n_ovf = 0, cand_ovf_leaked = 0;
  uint64_t cand_ovf_compactions = 0;  // times the orphaned overflow records were reclaimed in place (compact_overflow)
  size_t cand_n_points = 0;
  RecGrid rg{};
  // bricks, preliminary candidates, candidates kept, build ms, voxels with candidates, voxels with overflow, overflow
  // records, voxel edge / match_dist_min actually used
  double cand_stats[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  DevBuf dda_bits, dda_start, dda_pts, dda_index;
  // the map update as an overlay of the DDA grid (DdaGrid::ov_*): option "dda_overlay"; dda_overlay_ok = the arrays above
  // hold the base map only and the grid's bounds are the base map's, so an update inside them needs no rebuild
  DevBuf dda_ov_key, dda_ov_pts, dda_ov_idx;
  int dda_overlay = 1;
  bool dda_overlay_ok = false;
  uint64_t dda_overlay_updates = 0;
  DdaGeom dda_geom{};
  // the map as a device cloud (host_grid_builders.h); 1 = build the cell grid / the DDA grid on the host instead
  DevBuf map_dev;
  bool map_dev_valid = false;
  size_t map_dev_n = 0;
  int grid_build_host = 0;
  double grid_build_ms[2] = { 0, 0 };       // device time of the last cell-grid / DDA-grid build (device builders)
  double grid_build_wall_ms[2] = { 0, 0 };  // host wall time of the last build, either builder (upload of the map included)
  DdaGrid dg{};
  uint64_t footprint[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };

  // scans of the current update
  DevBuf scan_lik, scan_beam, origins, pow_table;
  // per-(particle, origin) ray constants (beam_origin_kernel) for launches of at least beam_prepare_min_rays rays
  DevBuf beam_origin;
  int beam_prepare = 1;
  long long beam_prepare_min_rays = 32768;
  size_t n_s = 0, n_b = 0, n_o = 0;
  bool has_scan = false;
  bool pow_table_dirty = true;
  size_t pow_table_len = 0;  // penalty counts 0..pow_table_len the device table covers
  size_t penalty_clean_n = 0;  // the first so many penalty counters are known to be zero (the update's tail kernel zeroes what it reads)

  // work buffers
  size_t n_pose_uploaded = 0;  // poses `pose` holds from mcl3dl_hip_upload_poses / the last host-buffer call
  // `pose` holds the 7-float poses derived from this rank's RESIDENT states (api_group_state.inl:rebuild_pose). Every other
  // writer of `pose` goes through poses_set(), which withdraws the mark: the resident calls then re-derive the poses from
  // gs_state instead of evaluating somebody else's
  bool pose_resident = false;
  void poses_set(size_t n)
  {
    n_pose_uploaded = n;
    pose_resident = false;
  }
  size_t stage_pending = 0;       // bytes staged for H2D copies since the last sync_stream
  DevBuf upd_block;  // measure_update: { stats4 | weights | lik | ratio | beam } in one allocation, so that the results go home in ONE copy
  DevBuf pose, lik, ratio, beam, weightb, wnew, extra, penalty, block_partials, partial4, stats4, ray_stats,
      tested, ray_begin, ray_end, ray_status, ray_hit, mom_blocks, mom_arg, mom_out, mom_idx, subset,
      packed;  // 2 + 2N doubles: this rank's record of a device group's all-reduce (host_group.h)

  // point-cloud preparation on the device (api_cloud.inl, cloud_kernels.h)
  DevBuf sort_tmp, cl_blocks, cl_minmax, cl_key[2], cl_val[2], cl_scan, cl_scan_ws, cl_start, cl_in_xyz, cl_in_label, cl_idx,
      cl_idx2, cl_err;
  DevBuf cl_ticket;  // arrival counter of the fused min / max reductions (zero between launches)
  DevBuf cl_counts;  // {VoxelGrid leaves, points the likelihood clip keeps, points the beam clip keeps}: one D2H for the three
  DevBuf rs_table;   // per-pass, per-work-group digit totals of the multi-work-group radix sort (sort_kernels.h)
  DevBuf sp_raw, sp_full, sp_clip[2], sp_samp[2];   // accumulated cloud, voxel-filtered, clipped (lik / beam), sampled
  DevBuf cl_clip_scan[2], cl_clip_ws[2];            // flag / scan arrays of the two clips (enqueued back to back)
  uint32_t sp_kept32[2] = { 0, 0 };                 // their counts, delivered by one synchronisation
  size_t sp_n_full = 0, sp_n_clip[2] = { 0, 0 }, sp_n_samp[2] = { 0, 0 };
  bool sp_ready = false;
  int chain_ppl = 0;          // strict_order = 3: tiles per work-group (0 = by size, 1, 4: likelihood_chain_multi.h)
  int chain_multi_max = 1536; // ... the four-tile form up to this many particles (profiles/r05r_chain_multi.txt: slower from 2048)
  int scan_order_device = 4096;  // scans of at least this many points (both models together) are ordered on the device; 0 = never
  size_t n_base = 0;  // points of the base map; anything behind them in map_xyz is the current map update
  DevBuf ms_xyz, ms_out, ms_flag[2];

  // this rank's shard of a device group's resident particles (api_group_state.inl): 13-float states (ping-pong), weights;
  // the 7-float poses the measurement kernels read are kept in `pose`
  DevBuf gs_state[2], gs_weight, gs_all, gs_pad, gs_rec;
  int gs_cur = 0;
  size_t gs_n = 0;

  // resampling plan
  std::vector<float> rs_keys;        // accumulated probabilities, in particles_dup_ order after std::sort
  std::vector<uint32_t> rs_order;    // which particle sits at each position of particles_dup_
  std::vector<uint32_t> rs_source, rs_slot;
  size_t rs_n = 0, rs_n_out = 0, rs_n_dup = 0;
  float rs_pstep = 0.f;
  bool rs_planned = false;
  DevBuf rs_d_keys, rs_d_pscan, rs_d_it, rs_d_source, rs_d_slot, rs_d_noise, rs_d_in, rs_d_out, rs_d_order, rs_d_flag,
      rs_d_ws, rs_d_dup8;
  bool rs_sorted = false;  // std::sort had ties to order: rs_order is not the identity

  // counts everything that can change what an update enqueues (parameters, options, map, stream, scan sizes, reallocated
  // buffers); kept as a cheap change stamp for diagnostics
  uint64_t generation = 0;
  std::string index_note;  // why the map compiler fell back to a coarser / plainer index form (diagnostics)

  // Pinned staging for the host-buffer entry points: small copies go through page-locked memory so that
  // hipMemcpyAsync really is asynchronous (a pageable copy costs a driver-side staging round trip each); results are
  // handed to the caller's arrays when the stream is synchronised (sync_stream).
  struct StageChunk
  {
    char* p;
    size_t cap;
  };
  struct StagedResult
  {
    void* user;
    const void* staged;
    size_t bytes;
  };
  std::vector<StageChunk> stage;
  size_t stage_cur = 0, stage_off = 0;
  std::vector<StagedResult> stage_out;
  // host-side scan staging (kept in the context so that it outlives the asynchronous copies)
  OrderedScan h_scan;

  // scratch memory of the map compilers (host_map_compilers.h:TempBuf): recycled instead of hipMalloc / hipFree per use
  struct ScratchBlk
  {
    void* p;
    size_t cap;
    bool used;
  };
  std::vector<ScratchBlk> scratch;

  // timing
  bool timing = false;
  unsigned timing_mask = 0xffffffffu;  // bit k = time kernel group k (MCL3DL_KERNEL_*); each timed group costs two event records
  std::vector<EventPair> pending;
  std::vector<hipEvent_t> free_events;
  double kernel_ms[MCL3DL_KERNEL_COUNT] = {};
  uint64_t kernel_launches[MCL3DL_KERNEL_COUNT] = {};

  int fail(int code, const char* fmt, ...)
  {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    err = buf;
    stage_out.clear();  // results of a failed call are not delivered (their destinations may be gone)
    // (a progressive batch in flight is NOT abandoned by another call's failure: its output arrays belong to it until
    // mcl3dl_hip_measure_batch_end by contract, and _wait must never report results it has not delivered)
    return code;
  }
};

namespace
{
void scratch_release(mcl3dl_hip_ctx* ctx, void* p)
{
  for (mcl3dl_hip_ctx::ScratchBlk& b : ctx->scratch)
    if (b.p == p)
    {
      b.used = false;
      return;
    }
  (void)hipFree(p);
}

// frees idle scratch blocks until at most keep_bytes of them stay parked (largest first)
void scratch_trim(mcl3dl_hip_ctx* ctx, size_t keep_bytes)
{
  size_t idle = 0;
  for (const mcl3dl_hip_ctx::ScratchBlk& b : ctx->scratch)
    idle += b.used ? 0 : b.cap;
  bool synced = false;
  while (idle > keep_bytes)
  {
    int big = -1;
    for (size_t k = 0; k < ctx->scratch.size(); ++k)
      if (!ctx->scratch[k].used && (big < 0 || ctx->scratch[k].cap > ctx->scratch[static_cast<size_t>(big)].cap))
        big = static_cast<int>(k);
    if (big < 0)
      break;
    if (!synced)
    {
      (void)hipStreamSynchronize(ctx->stream);  // enqueued kernels may still read an idle block
      synced = true;
    }
    (void)hipFree(ctx->scratch[static_cast<size_t>(big)].p);
    idle -= ctx->scratch[static_cast<size_t>(big)].cap;
    ctx->scratch.erase(ctx->scratch.begin() + big);
  }
}

#define HIP_TRY(expr)                                                                            \
  do                                                                                             \
  {                                                                                              \
    const hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                        \
      return ctx->fail(-2, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

#define TRY(expr)        \
  do                     \
  {                      \
    const int r_ = (expr); \
    if (r_ != 0)         \
      return r_;         \
  } while (0)

inline float bits_to_float(uint32_t u)
{
  float f;
  memcpy(&f, &u, sizeof(f));
  return f;
}

int ensure(mcl3dl_hip_ctx* ctx, DevBuf& b, size_t bytes)
{
  if (bytes == 0)
    bytes = 16;
  if (b.cap >= bytes)
    return 0;
  if (b.view)
    return ctx->fail(-2, "internal: ensure() on a view buffer");
  if (b.p)
  {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  const size_t cap = bytes + bytes / 4;
  HIP_TRY(hipMalloc(&b.p, cap));
  b.cap = cap;
  ++ctx->generation;  // a captured update graph holds the old address
  return 0;
}

// The per-update scan arrays as views into ONE device block, laid out for the current sizes: { permutation (n_s uint32) |
// likelihood scan (n_s float4) | beam scan (n_b float4) | origins (n_o float4) }, each part on a 256-byte boundary — a
// host-ordered scan then goes up in ONE copy instead of four (each small copy costs ~4 us on the device: at the reference's
// own sizes the four of them were a third of a host-buffer update). off4 (optional) = the parts' byte offsets.
int ensure_scan_block(mcl3dl_hip_ctx* ctx, size_t n_s, size_t n_b, size_t n_o, size_t* off4 = nullptr)
{
  const auto up = [](size_t b) { return (std::max<size_t>(b, 16) + 255) & ~static_cast<size_t>(255); };
  const size_t part[4] = { up(sizeof(uint32_t) * n_s), up(sizeof(float4) * n_s), up(sizeof(float4) * n_b), up(sizeof(float4) * n_o) };
  const size_t total = part[0] + part[1] + part[2] + part[3];
  TRY(ensure(ctx, ctx->scan_block, total));
  ctx->scan_chunk = 0;  // (device_order_scans sets it again when it orders the scan in chunks)
  DevBuf* view[4] = { &ctx->scan_perm, &ctx->scan_lik, &ctx->scan_beam, &ctx->origins };
  size_t off = 0;
  bool moved = false;
  for (int k = 0; k < 4; ++k)
  {
    void* at = ctx->scan_block.as<char>() + off;
    moved = moved || view[k]->p != at;
    view[k]->p = at;
    view[k]->cap = part[k];
    view[k]->view = true;
    if (off4)
      off4[k] = off;
    off += part[k];
  }
  if (moved)
    ++ctx->generation;  // a captured update graph holds the old addresses
  return 0;
}

constexpr size_t STAGE_MAX_COPY = 4u << 20;  // larger copies go straight from / to the caller's (pageable) memory

// Page-locked host memory that kernels can read and write in place (mapped, coherent: a kernel's stores are visible to the
// host once the stream has been synchronised). The device must see it at the host's address — the zero-copy update passes
// host pointers straight to its kernels; a platform where it does not gets update_zero_copy switched off.
void* pinned_alloc(mcl3dl_hip_ctx* ctx, size_t bytes)
{
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) != hipSuccess)
  {
    (void)hipGetLastError();
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess)
    {
      (void)hipGetLastError();
      return nullptr;
    }
  }
  void* dp = nullptr;
  if (hipHostGetDevicePointer(&dp, p, 0) != hipSuccess || dp != p)
  {
    (void)hipGetLastError();
    ctx->zero_copy_supported = false;
  }
  return p;
}

// bump allocation in page-locked chunks; everything is released for reuse by sync_stream. nullptr = allocation failed
// (the caller then falls back to a direct copy).
void* stage_alloc(mcl3dl_hip_ctx* ctx, size_t bytes)
{
  bytes = (bytes + 255) & ~static_cast<size_t>(255);
  while (ctx->stage_cur < ctx->stage.size())
  {
    mcl3dl_hip_ctx::StageChunk& ch = ctx->stage[ctx->stage_cur];
    if (ctx->stage_off + bytes <= ch.cap)
    {
      void* p = ch.p + ctx->stage_off;
      ctx->stage_off += bytes;
      return p;
    }
    ++ctx->stage_cur;
    ctx->stage_off = 0;
  }
  const size_t last = ctx->stage.empty() ? (512u << 10) : ctx->stage.back().cap;
  const size_t cap = std::max(bytes, 2 * last);
  void* p = pinned_alloc(ctx, cap);
  if (!p)
    return nullptr;
  ctx->stage.push_back({ static_cast<char*>(p), cap });
  ctx->stage_cur = ctx->stage.size() - 1;
  ctx->stage_off = bytes;
  return p;
}

// *staged (optional) = the caller's array has been copied into page-locked staging memory: it may be reused as soon as this
// returns, whatever the stream is doing (the staging memory itself is recycled by the next sync_stream).
int h2d(mcl3dl_hip_ctx* ctx, void* dst, const void* src, size_t bytes, bool* staged = nullptr)
{
  if (staged)
    *staged = bytes == 0;
  if (bytes == 0)
    return 0;
  if (bytes <= STAGE_MAX_COPY)
  {
    if (void* p = stage_alloc(ctx, bytes))
    {
      memcpy(p, src, bytes);
      HIP_TRY(hipMemcpyAsync(dst, p, bytes, hipMemcpyHostToDevice, ctx->stream));
      ctx->stage_pending += bytes;
      if (staged)
        *staged = true;
      return 0;
    }
  }
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  return 0;
}

// The data is in `dst` only after sync_stream().
int d2h(mcl3dl_hip_ctx* ctx, void* dst, const void* src, size_t bytes)
{
  if (bytes == 0)
    return 0;
  if (bytes <= STAGE_MAX_COPY)
  {
    if (void* p = stage_alloc(ctx, bytes))
    {
      HIP_TRY(hipMemcpyAsync(p, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
      ctx->stage_out.push_back({ dst, p, bytes });
      return 0;
    }
  }
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  return 0;
}

// One D2H copy of `bytes` from `src`, handed out in pieces at sync_stream(): piece = { caller's array (may be null: skipped),
// offset into the block, bytes }. Falls back to one copy per piece when the block does not fit the staging memory.
struct D2hPiece
{
  void* user;
  size_t offset, bytes;
};

int d2h_block(mcl3dl_hip_ctx* ctx, const void* src, size_t bytes, const D2hPiece* pieces, int n_pieces)
{
  void* p = bytes <= STAGE_MAX_COPY ? stage_alloc(ctx, bytes) : nullptr;
  if (!p)
  {
    for (int i = 0; i < n_pieces; ++i)
      if (pieces[i].user)
        TRY(d2h(ctx, pieces[i].user, static_cast<const char*>(src) + pieces[i].offset, pieces[i].bytes));
    return 0;
  }
  HIP_TRY(hipMemcpyAsync(p, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  for (int i = 0; i < n_pieces; ++i)
    if (pieces[i].user && pieces[i].bytes)
      ctx->stage_out.push_back({ pieces[i].user, static_cast<char*>(p) + pieces[i].offset, pieces[i].bytes });
  return 0;
}

__global__ void done_flag_kernel(volatile unsigned* flag, unsigned seq)
{
  __threadfence_system();
  *flag = seq;
}

// Completion of everything enqueued on ctx->stream so far, observed through page-locked memory: a one-thread kernel behind
// it writes a sequence number, the host spins on it (bounded: ~2 s, then hipStreamSynchronize decides). What the kernels in
// front wrote into page-locked memory left the device before the flag did (uncached stores, one ordered path to the host).
bool ensure_done_flag(mcl3dl_hip_ctx* ctx)
{
  if (!ctx->zero_copy_supported)
    return false;
  if (!ctx->done_flag)
  {
    volatile unsigned* f = static_cast<volatile unsigned*>(pinned_alloc(ctx, 64));
    if (!f || !ctx->zero_copy_supported)
      return false;  // (a block the device cannot write in place stays unused; it is released with the process)
    *f = 0u;
    ctx->done_flag = f;
  }
  return true;
}

inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  __asm__ __volatile__("yield");
#endif
}

inline double mono_us()
{
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return 1e6 * static_cast<double>(ts.tv_sec) + 1e-3 * static_cast<double>(ts.tv_nsec);
}

// Waits until the completion word has reached `seq` (sequence numbers only grow, the stream is in order). Bounded by wall
// time, not by iterations: a pure spin for poll_spin_us, then naps of 1/32 of the time already waited (at most 1 ms) between
// looks at the word — and a hipStreamQuery every ~5 ms: an error (a faulted queue) is returned within milliseconds, and a
// stream that has drained without the word having arrived (it cannot: the word's kernel is on it) ends the wait as well.
int spin_done_flag(mcl3dl_hip_ctx* ctx, unsigned seq)
{
  const auto arrived = [&]() { return static_cast<int>(*ctx->done_flag - seq) >= 0; };
  for (int spin = 0; spin < 256; ++spin)
  {
    if (arrived())
    {
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      return 0;
    }
    cpu_relax();
  }
  const double t0 = mono_us();
  double next_query = ctx->poll_query_us;
  for (;;)
  {
    for (int spin = 0; spin < 64; ++spin)
    {
      if (arrived())
      {
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        return 0;
      }
      cpu_relax();
    }
    const double waited = mono_us() - t0;
    if (waited < ctx->poll_spin_us)
      continue;
    if (waited >= next_query)
    {
      const hipError_t q = hipStreamQuery(ctx->stream);
      if (q != hipErrorNotReady)
      {
        if (q != hipSuccess)
          return ctx->fail(-2, "the stream failed while its completion was awaited: %s", hipGetErrorString(q));
        // drained: everything in front of the word's kernel — and the kernel — has run
        if (arrived())
        {
          __atomic_thread_fence(__ATOMIC_ACQUIRE);
          return 0;
        }
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return 0;
      }
      next_query = waited + ctx->poll_query_us;
    }
    const double nap_us = std::min(1000.0, waited / 32.0);
    timespec ts{ 0, static_cast<long>(nap_us * 1e3) };
    nanosleep(&ts, nullptr);
  }
}

int wait_done_flag(mcl3dl_hip_ctx* ctx)
{
  if (!ensure_done_flag(ctx))
  {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
  }
  const unsigned seq = ++ctx->done_seq;
  hipLaunchKernelGGL(done_flag_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->done_flag, seq);
  HIP_TRY(hipGetLastError());
  return spin_done_flag(ctx, seq);
}

// strict_order = 3: the page-locked error word of the in-kernel hand-offs, looked at (and cleared) after EVERY completed wait —
// sync_stream, but also the slices of a progressive batch, which never pass through sync_stream. Whatever
// was staged for the failed work is dropped, so that no later synchronisation copies stale results into the caller's arrays.
int chain_check(mcl3dl_hip_ctx* ctx)
{
  if (!ctx->chain_err || !*ctx->chain_err)
    return 0;
  *ctx->chain_err = 0u;
  ctx->stage_out.clear();
  ctx->stage_cur = 0;
  ctx->stage_off = 0;
  ctx->stage_pending = 0;
  ctx->prog = mcl3dl_hip_ctx::BatchProgress();
  return ctx->fail(-2, "strict_order = 3: a hand-off of the in-kernel float sum did not arrive (likelihoods of this update are invalid)");
}

// Progressive batch: waits until the slice holding `particle` is on the host, hands every slice that has arrived to the
// caller's arrays and returns the number of particles whose results are there.
int progress_wait(mcl3dl_hip_ctx* ctx, size_t particle, size_t* n_ready)
{
  mcl3dl_hip_ctx::BatchProgress& pg = ctx->prog;
  if (!pg.active)
  {
    if (n_ready)
      *n_ready = pg.n_p;
    return 0;
  }
  const size_t k = std::min(particle / pg.slice, pg.n_slices - 1);
  if (k >= pg.delivered)
  {
    const int rc = spin_done_flag(ctx, pg.seq0 + static_cast<unsigned>(k) + 1u);
    if (rc != 0)
    {
      pg = mcl3dl_hip_ctx::BatchProgress();  // the stream failed: there is no batch any more (later _wait calls are refused)
      return rc;
    }
  }
  TRY(chain_check(ctx));  // (a slice whose hand-off timed out is not delivered: the batch is gone, rc -2)
  const unsigned now = *ctx->done_flag;
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  const size_t arrived = std::min<size_t>(pg.n_slices, static_cast<size_t>(std::max(0, static_cast<int>(now - pg.seq0))));
  if (arrived > pg.delivered)
  {
    const size_t lo = pg.delivered * pg.slice, hi = std::min(pg.n_p, arrived * pg.slice);
    for (int a = 0; a < 3; ++a)
      if (pg.user[a] && pg.host[a] != pg.user[a])
        memcpy(pg.user[a] + lo, pg.host[a] + lo, sizeof(float) * (hi - lo));
    pg.delivered = arrived;
  }
  if (n_ready)
    *n_ready = std::min(pg.n_p, pg.delivered * pg.slice);
  return 0;
}

// hipStreamSynchronize (or, polled = true and the option on, the polled completion flag) + hand the staged results to the
// caller's arrays + recycle the staging memory.
int sync_stream(mcl3dl_hip_ctx* ctx, bool polled = false)
{
  if (ctx->prog.active)
  {
    // a progressive batch is still in flight: its remaining slices are delivered before the staging memory is recycled
    TRY(progress_wait(ctx, ctx->prog.n_p - 1, nullptr));
    ctx->prog.active = false;
  }
  // poll_sync = 2: EVERY synchronisation of the context's stream is the polled word (the copies and kernels in front of the
  // one-thread kernel are stream-ordered ahead of it, so what they wrote — staged D2H copies included — is there when the
  // word arrives); 1: only where the caller asked for it (the host-buffer update and its relatives)
  if ((polled || ctx->poll_mode() >= 2) && ctx->poll_mode())
    TRY(wait_done_flag(ctx));
  else
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  TRY(chain_check(ctx));
  for (const mcl3dl_hip_ctx::StagedResult& r : ctx->stage_out)
    memcpy(r.user, r.staged, r.bytes);
  ctx->stage_out.clear();
  ctx->stage_cur = 0;
  ctx->stage_off = 0;
  ctx->stage_pending = 0;
  return 0;
}

// End of a progressive batch: everything delivered, staging memory recycled. The batch's last completion word is the last
// thing on the stream unless another call enqueued copies behind it, in which case the stream is synchronised the usual way.
int progress_end(mcl3dl_hip_ctx* ctx)
{
  if (!ctx->prog.active)
    return 0;
  if (!ctx->stage_out.empty() || ctx->stage_pending != 0)
    return sync_stream(ctx, true);
  TRY(progress_wait(ctx, ctx->prog.n_p - 1, nullptr));
  ctx->prog.active = false;
  ctx->stage_cur = 0;
  ctx->stage_off = 0;
  return 0;
}

// ---- timing -----------------------------------------------------------------------------------------
int timing_begin(mcl3dl_hip_ctx* ctx, int kernel, EventPair* ep, hipStream_t on = nullptr)
{
  if (!on)
    on = ctx->stream;
  ep->start = nullptr;
  if (!ctx->timing || !(ctx->timing_mask & (1u << kernel)))
    return 0;
  hipEvent_t ev[2];
  for (int i = 0; i < 2; ++i)
  {
    if (!ctx->free_events.empty())
    {
      ev[i] = ctx->free_events.back();
      ctx->free_events.pop_back();
    }
    else
    {
      // timing only: no system-scope fence when the event is recorded. A default event writes the L2s back and invalidates
      // them, and the kernel bracketed by two of them then starts on cold L2s — the tiled likelihood kernel measured
      // 272 us between default events against 230 us without them (profiles/r03n_C2 kernel trace), i.e. the probe
      // changed what it measured.
      HIP_TRY(hipEventCreateWithFlags(&ev[i], hipEventDisableSystemFence));
    }
  }
  ep->start = ev[0];
  ep->stop = ev[1];
  ep->kernel = kernel;
  HIP_TRY(hipEventRecord(ep->start, on));
  return 0;
}

int timing_end(mcl3dl_hip_ctx* ctx, const EventPair& ep, hipStream_t on = nullptr)
{
  if (!ctx->timing || !ep.start)
    return 0;
  HIP_TRY(hipEventRecord(ep.stop, on ? on : ctx->stream));
  ctx->pending.push_back(ep);
  return 0;
}

int timing_collect(mcl3dl_hip_ctx* ctx)
{
  if (ctx->pending.empty())
    return 0;
  TRY(sync_stream(ctx));
  HIP_TRY(hipStreamSynchronize(ctx->aux_stream));
  for (const EventPair& ep : ctx->pending)
  {
    float ms = 0.f;
    HIP_TRY(hipEventSynchronize(ep.stop));  // (past already; after a polled wait the runtime has not looked yet)
    HIP_TRY(hipEventElapsedTime(&ms, ep.start, ep.stop));
    ctx->kernel_ms[ep.kernel] += ms;
    ctx->kernel_launches[ep.kernel] += 1;
    ctx->free_events.push_back(ep.start);
    ctx->free_events.push_back(ep.stop);
  }
  ctx->pending.clear();
  return 0;
}

}  // namespace
