// host_measure.h — part of the single translation unit mcl3dl_hip.hip: parameter structs for the kernels and
// launch_measure, the one place that enqueues the likelihood and beam kernels of an update.
#pragma once

namespace
{
int ensure_structures(mcl3dl_hip_ctx* ctx, bool need_lik, bool need_dda, bool need_cells = false)
{
  if (!ctx->has_map)
    return ctx->fail(-5, "no map: call mcl3dl_hip_set_map first");
  bool built = false;
  if (need_lik && (ctx->lik_index == 0 || need_cells) && ctx->lik_dirty)
  {
    TRY(build_lik_grid(ctx));
    built = true;
  }
  if (need_lik && ctx->lik_index >= 1 && !need_cells && ctx->cand_dirty)
  {
    TRY(build_cand_grid(ctx));
    built = true;
  }
  if (need_dda && ctx->dda_dirty)
  {
    TRY(build_dda_grid(ctx));
    built = true;
  }
  if (built)
    scratch_trim(ctx, 64u << 20);  // the temporaries of a whole-map build do not stay parked (map updates keep their small ones)
  return 0;
}

// Page-locked, device-mapped arrays the last kernel of an update writes its results to (each may be null).
using HostOut = PfEmit;

LikParams lik_params(const mcl3dl_hip_ctx* ctx)
{
  LikParams p;
  p.wx = ctx->weight[0];
  p.wy = ctx->weight[1];
  p.wz = ctx->weight[2];
  p.has_weight = ctx->has_weight ? 1 : 0;
  p.match_dist_min = ctx->match_dist_min;
  // pcl::KdTreeFLANN::radiusSearch: (float)(radius * radius) with radius widened to double
  p.r2 = static_cast<float>(static_cast<double>(ctx->match_dist_min) * static_cast<double>(ctx->match_dist_min));
  p.match_dist_flat = ctx->match_dist_flat;
  p.match_weight = ctx->match_weight;
  return p;
}

BeamParams beam_params(const mcl3dl_hip_ctx* ctx)
{
  BeamParams p;
  p.sin_total_ref = ctx->sin_total_ref;
  p.hit_range_sq = ctx->hit_range_sq;
  p.filter_label_max = ctx->filter_label_max;
  p.short_only = ctx->short_only;
  p.beam_likelihood_min = ctx->beam_likelihood_min;
  return p;
}

// LidarMeasurementModelBeam::refreshParameters, src/lidar_measurement_model_beam.cpp:65-67 (host libm, like the reference)
void beam_refresh(mcl3dl_hip_ctx* ctx)
{
  ctx->hit_range_sq = static_cast<float>(std::pow(static_cast<double>(ctx->hit_range), 2));
  ctx->beam_likelihood = static_cast<float>(
      std::pow(static_cast<double>(ctx->beam_likelihood_min), 1.0 / static_cast<float>(ctx->beam_num_points)));
  ctx->sin_total_ref = sinf(ctx->ang_total_ref);
  ctx->pow_table_dirty = true;
}

// 3-D Morton key of a scan point (robot frame), 0.25 m cells: neighbouring lanes of a wavefront then gather from
// neighbouring map cells.
uint64_t morton3(uint32_t x, uint32_t y, uint32_t z)
{
  auto spread = [](uint64_t v)
  {
    v &= 0x1fffff;
    v = (v | v << 32) & 0x1f00000000ffffULL;
    v = (v | v << 16) & 0x1f0000ff0000ffULL;
    v = (v | v << 8) & 0x100f00f00f00f00fULL;
    v = (v | v << 4) & 0x10c30c30c30c30c3ULL;
    v = (v | v << 2) & 0x1249249249249249ULL;
    return v;
  };
  return spread(x) | (spread(y) << 1) | (spread(z) << 2);
}

// table[k] = score_beam after k penalised rays: `score_beam *= beam_likelihood_` repeated k times (beam.cpp:148), float. Built
// for at least 1024 counts so that alternating scan sizes (the adapter launches the two models separately) do not rebuild it
// every update.
// n_b_coming: the beam scan size of an update whose scan is not in the context yet (0: the current one)
int ensure_pow_table(mcl3dl_hip_ctx* ctx, size_t n_b_coming = 0)
{
  const size_t n_b = std::max(ctx->n_b, n_b_coming);
  if (!(ctx->pow_table_dirty || n_b > ctx->pow_table_len))
    return 0;
  const size_t len = std::max<size_t>(n_b, 1024);
  std::vector<float> table(len + 1);
  table[0] = 1.0f;
  for (size_t k = 1; k <= len; ++k)
    table[k] = table[k - 1] * ctx->beam_likelihood;
  TRY(ensure(ctx, ctx->pow_table, sizeof(float) * table.size()));
  TRY(h2d(ctx, ctx->pow_table.p, table.data(), sizeof(float) * table.size()));
  TRY(sync_stream(ctx));
  ctx->pow_table_dirty = false;
  ctx->pow_table_len = len;
  ++ctx->generation;
  return 0;
}

// Which likelihood kernel an update of np particles x ns points runs, with every size check and every buffer the launch
// needs done HERE — before launch_measure forks the beam kernels onto the second stream, so that no error path can
// return with un-joined work in flight.
struct LikPlan
{
  bool tiled = false, small = false;
  int group_size = 16, W = 1, n_tiles = 0, n_groups = 0;
  long long blocks = 0;
  float* strict_terms = nullptr;
  size_t chunk = 0;    // > 0: the scan is ordered in chunks of the caller's order and replayed chunk by chunk (two term buffers)
  bool chain = false;  // the float sum in the scan array's order inside the tiled kernel (likelihood_kernels.h: LikChain)
  uint32_t chain_tag0 = 0;
  int chain_ppl = 1;   // > 1: likelihood_chain_multi_kernel, that many tiles per work-group (n_super super-tiles)
  int n_super = 0;
  bool rows = false;   // per-particle / small-scan kernel with the terms parked in LDS at their ORIGINAL scan indices and added up
                       // as the reference adds them (float, sequentially, caller's order): bit-identical likelihoods
};

// Scans up to this many points fit the caller-order term row of the per-particle kernels (48 KB of the 64 KB of LDS a
// work-group gets without asking for more).
constexpr int LIK_ROW_MAX = 12288;

// Which likelihood kernel family a launch of np particles x ns points takes, and how its terms are added (no allocation:
// plan_lik and the one-launch update both ask).
//   exact   the sum must be the reference's float, in the caller's order: strict_order 1 always; the default (2) for every
//           scan of at most strict_exact_max points (4096: the reference's operating range and far beyond — the fuzz's gate is
//           assert_array_equal there) and from strict_auto_min points up, where the reference's own float rounding is no longer
//           safely inside north_star's 1e-5 (host_context.h); in between, and with strict_order 0, the fixed-order fp64 tree
//   rows    ... by the per-particle kernels themselves: every term parked in LDS at its ORIGINAL scan index, one wavefront
//           runs the recurrence (lik_particle / likelihood_small_kernel) — no term array, no second launch
//   replay  ... by the tiled kernel + the N_s x N_p term array + lik_strict_sum_rows_kernel
// Measured (profiles/r06d_rows_vs_replay.txt): below ~2000 particles the rows are the cheaper exact form at every scan size
// that fits them (64 x 4096: 21 against 38 us; 1024 x 4096: 48 against 58), above it the replay (4096 x 4096: 116 against 126).
struct LikMode
{
  bool tiled = false, rows = false, replay = false;
};

LikMode lik_mode(const mcl3dl_hip_ctx* ctx, int np, int ns)
{
  LikMode m;
  const bool by_size = ctx->lik_tiled && np >= 4 &&
                       (ns >= ctx->lik_tiled_min || (np >= 256 && 4 * static_cast<long long>(ns) >= 3ll * ctx->lik_tiled_min));
  if (ctx->strict_order == 3)
  {
    m.tiled = true;  // the in-kernel chain, in the engine's scan order
    return m;
  }
  const bool exact = ctx->strict_order == 1 ||
                     (ctx->strict_order == 2 && (ns <= ctx->strict_exact_max || ns >= ctx->strict_auto_min));
  // (a scan ordered in chunks of the caller's order — option strict_chunk — carries chunk-relative indices: the replay's business)
  const bool rows_fit = ns <= LIK_ROW_MAX && ctx->scan_chunk == 0;
  if (!exact)
  {
    m.tiled = by_size;
    // where a per-particle kernel runs anyway the rows cost a few per cent: exact there too (unless told not to: strict_order 0)
    m.rows = !by_size && ctx->strict_order == 2 && rows_fit;
    return m;
  }
  if (rows_fit && (!by_size || np < ctx->strict_rows_max_particles))
    m.rows = true;
  else
    m.tiled = m.replay = true;
  return m;
}

// does an update over ns scan points replay the likelihood terms in the reference's float order?
// the tiled kernel with its overflow rounds deferred (likelihood_kernels.h): needs packed 64-byte records; mode 2 = only
// on maps where enough voxels overflow for a wavefront to meet one in nearly every round
bool lik_defer_active(const mcl3dl_hip_ctx* ctx)
{
  if (ctx->lik_defer == 0 || ctx->lik_index != 2 || !ctx->rg.packed || ctx->rg.rec_parts != 4)
    return false;
  if (ctx->lik_defer == 1)
    return true;
  const double with_cand = ctx->cand_stats[4], over4 = ctx->cand_stats[5];
  return with_cand > 0 && over4 / with_cand > ctx->lik_defer_min_frac;
}

// strict_order = 3: the reference's float recurrence over the scan in the ENGINE's order (mcl3dl_hip_scan_order), inside the
// tiled kernel — no term array, no replay (needs the page-locked error word of the hand-off: ensure_chain)
bool lik_chain(const mcl3dl_hip_ctx* ctx)
{
  return ctx->strict_order == 3;
}

// hand-off words + error word of the in-kernel chain for n_p particles and n_tiles tiles; *tag0 = tag of tile 0
int ensure_chain(mcl3dl_hip_ctx* ctx, size_t n_p, int n_tiles, uint32_t* tag0)
{
  if (!ctx->chain_err)
  {
    // (the word is kept only once the platform has shown that the device can write it in place: a context on a platform
    // without such memory fails HERE on every call instead of launching kernels that write through a pointer the device
    // cannot reach — ADVICE round 5)
    volatile unsigned* w = ctx->zero_copy_supported ? static_cast<volatile unsigned*>(pinned_alloc(ctx, 64)) : nullptr;
    if (!w || !ctx->zero_copy_supported)
      return ctx->fail(-2, "strict_order = 3 needs page-locked memory the device can write in place");
    *w = 0u;
    ctx->chain_err = w;
  }
  const size_t need = sizeof(unsigned long long) * 2 * n_p;
  const bool fresh = need > ctx->chain_carry.cap;
  TRY(ensure(ctx, ctx->chain_carry, need));
  // tags only grow; a fresh buffer (or a wrap of the 32-bit tag) starts from cleared words and tag 1
  if (fresh || ctx->chain_tag > 0xffffffffu - static_cast<uint32_t>(n_tiles) - 2u)
  {
    HIP_TRY(hipMemsetAsync(ctx->chain_carry.p, 0, ctx->chain_carry.cap, ctx->stream));
    ctx->chain_tag = 1u;
  }
  *tag0 = ctx->chain_tag;
  ctx->chain_tag += static_cast<uint32_t>(n_tiles) + 1u;
  return 0;
}

// strict_order = 3 launches of ONE device are serialised across contexts: the hand-off makes progress only while a
// consumer's producer is resident or done, which dispatch in block-index order guarantees inside one launch — but two chain kernels
// running side by side (two contexts, or two ranks of a device group on one GPU) could fill each other's XCDs with polling
// consumers (bounded: ~1 s, then error -2). A process-wide event per device: a chain launch waits for the previous one's.
struct ChainSerial
{
  std::mutex m;
  hipEvent_t ev = nullptr;
};
inline ChainSerial& chain_serial(int device)
{
  static ChainSerial per_device[64];
  return per_device[(device >= 0 && device < 64) ? device : 0];
}

// rows of G floats per particle group: what the float-order replay of ns points x n_p particles stores
size_t strict_terms_bytes(size_t n_p, int ns, int group_size)
{
  const size_t G = static_cast<size_t>(group_size);
  return ((n_p + G - 1) / G) * (sizeof(float) * static_cast<size_t>(ns) * G + sizeof(float4) * STRICT_SKEW4);
}

// device memory of the float-order replay: the whole scan's terms, or two chunks' when the scan was ordered in chunks
size_t strict_plan_bytes(const mcl3dl_hip_ctx* ctx, size_t n_p, int ns, int group_size)
{
  if (ctx->scan_chunk && static_cast<size_t>(ns) > ctx->scan_chunk)
    return 2 * strict_terms_bytes(n_p, static_cast<int>(ctx->scan_chunk), group_size);
  return strict_terms_bytes(n_p, ns, group_size);
}

int ensure_replay_stream(mcl3dl_hip_ctx* ctx)
{
  if (!ctx->replay_stream)
    HIP_TRY(hipStreamCreateWithFlags(&ctx->replay_stream, hipStreamNonBlocking));
  for (int k = 0; k < 2; ++k)
  {
    if (!ctx->ev_tiled[k])
      HIP_TRY(hipEventCreateWithFlags(&ctx->ev_tiled[k], hipEventDisableTiming));
    if (!ctx->ev_replay[k])
      HIP_TRY(hipEventCreateWithFlags(&ctx->ev_replay[k], hipEventDisableTiming));
  }
  return 0;
}

int plan_group_size(const mcl3dl_hip_ctx* ctx, int np, int ns)
{
  // particles per work-group of the tiled kernel: the largest of 16 / 8 / 4 that still gives the 256 CUs x 8
  // work-group slots something to do (few particles x a long scan would otherwise leave most of the GPU idle)
  int group_size = ctx->lik_group;
  if (group_size == 0)
  {
    const long long n_tiles_ll = (ns + 255) / 256;
    group_size = 4;
    for (int gg = 16; gg >= 4; gg >>= 1)
      if (n_tiles_ll * ((np + gg - 1) / gg) >= 2048)
      {
        group_size = gg;
        break;
      }
  }
  return group_size;
}

int plan_lik(mcl3dl_hip_ctx* ctx, size_t n_p, int ns, LikPlan* pl)
{
  const int np = static_cast<int>(n_p);
  const bool chain = lik_chain(ctx);
  const LikMode mode = lik_mode(ctx, np, ns);
  pl->tiled = mode.tiled;
  bool strict = mode.replay;
  int group_size = plan_group_size(ctx, np, ns);
  if (chain && group_size > 16)
    group_size = 16;
  if (strict && ctx->strict_order == 2)
  {
    // The AUTOMATIC replay (scans of at least strict_auto_min points) costs ns x n_p floats of device memory — 0.5 GB at
    // 32 768 x 4096, 26 GB at 65 536 x 100 000. It is a refinement (the fp64 sums are within ~1e-5 of the reference's
    // float at such sizes), so it must never make an update fail: when the buffer would take more than half of the free
    // device memory, or its allocation fails, this launch sums in fp64 like smaller scans do. (strict_order = 1 — asked for
    // explicitly — still fails loudly.)
    const size_t need = strict_plan_bytes(ctx, n_p, ns, group_size);
    if (ctx->strict_auto_max_bytes > 0.0 && static_cast<double>(need) > ctx->strict_auto_max_bytes)
    {
      strict = false;
      ++ctx->strict_auto_skipped;
    }
    else if (need > ctx->strict_terms.cap)
    {
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess)
      {
        (void)hipGetLastError();
        free_b = 0;
      }
      const bool over_cap = ctx->strict_auto_max_bytes > 0.0 && static_cast<double>(need) > ctx->strict_auto_max_bytes;
      if (over_cap || need + need / 4 > (free_b + ctx->strict_terms.cap) / 2 || ensure(ctx, ctx->strict_terms, need) != 0)
      {
        (void)hipGetLastError();
        ctx->err.clear();
        strict = false;
        ++ctx->strict_auto_skipped;
      }
    }
  }
  // tiled from lik_tiled_min points up (default 1024), and already from three quarters of that when there are enough particles
  // to fill the GPU with (tile, group) pairs (4096 x 1000: 30.6 us tiled against 34.9 us; 64 x 1000 and 4096 x 512: no gain)
  pl->rows = mode.rows;
  pl->group_size = group_size;
  pl->small = !pl->tiled && ns <= 32 && np >= 256 && ctx->lik_small;
  if (pl->small)
  {
    int W = 1;
    while (W < ns)
      W <<= 1;
    pl->W = W;
    pl->blocks = (static_cast<long long>(np) * W + 255) / 256;
    if (pl->blocks > 0x7fffffffLL)
      return ctx->fail(-3, "too many work-groups for the small-scan likelihood kernel");
  }
  else if (pl->tiled)
  {
    pl->n_tiles = (ns + 255) / 256;
    pl->n_groups = (np + group_size - 1) / group_size;
    // per XCD: the interleaved tiles of the largest multiple of eight, then an eighth of the remaining (tile, group) pairs
    const long long full_tiles = pl->n_tiles & ~7, rem_items = static_cast<long long>(pl->n_tiles - full_tiles) * pl->n_groups;
    pl->blocks = 8 * ((full_tiles / 8) * pl->n_groups + (rem_items + 7) / 8);
    if (chain)
    {
      // few particles on a long scan (default kernel family only): four tiles per work-group, a quarter of the hand-offs
      // (likelihood_chain_multi.h). chain_ppl: 0 = by size, 1 = never, 4 = whenever the family allows.
      const bool family = ctx->lik_coop && ctx->lik_index == 2 && ctx->match_dist_min > 1e-5f && lik_defer_active(ctx) && ctx->lik_group == 0;
      if (family && (ctx->chain_ppl == 4 || (ctx->chain_ppl == 0 && pl->n_tiles >= 16 && np <= ctx->chain_multi_max)))
      {
        pl->chain_ppl = 4;
        pl->group_size = 4;
        pl->n_groups = (np + 3) / 4;
        pl->n_super = (pl->n_tiles + 3) / 4;
        pl->blocks = 8ll * ((pl->n_super + 7) / 8) * pl->n_groups;
        if (pl->blocks > 0x7fffffffLL)
          return ctx->fail(-3, "too many work-groups for the tiled likelihood kernel");
        pl->chain = true;
        return ensure_chain(ctx, n_p, pl->n_super, &pl->chain_tag0);
      }
      // rows of eight tiles, no shared-out remainder (likelihood_tiled_kernel<..., CHAIN>)
      pl->blocks = 8ll * ((pl->n_tiles + 7) / 8) * pl->n_groups;
      if (pl->blocks > 0x7fffffffLL)
        return ctx->fail(-3, "too many work-groups for the tiled likelihood kernel");
      pl->chain = true;
      return ensure_chain(ctx, n_p, pl->n_tiles, &pl->chain_tag0);
    }
    if (pl->blocks > 0x7fffffffLL)
      return ctx->fail(-3, "too many work-groups for the tiled likelihood kernel");
    TRY(ensure(ctx, ctx->lik_partial_sum, sizeof(double) * static_cast<size_t>(pl->n_tiles) * n_p));
    TRY(ensure(ctx, ctx->lik_partial_cnt, sizeof(unsigned) * static_cast<size_t>(pl->n_tiles) * n_p));
    if (strict)
    {
      TRY(ensure(ctx, ctx->strict_terms, strict_plan_bytes(ctx, n_p, ns, group_size)));
      pl->strict_terms = ctx->strict_terms.as<float>();
      pl->chunk = (ctx->scan_chunk && static_cast<size_t>(ns) > ctx->scan_chunk) ? ctx->scan_chunk : 0;
      if (pl->chunk)
        TRY(ensure_replay_stream(ctx));
    }
  }
  return 0;
}

// lik_strict_sum_rows_kernel with as many particle groups per work-group as keep the launch in ONE round of work-groups, up to a
// full adder wavefront (64 lanes / GG particles per group)
template <int GG>
void launch_strict_sum(mcl3dl_hip_ctx* ctx, const float* strict_terms, int ns, int np, int n_groups, float* d_lik,
                       hipStream_t on = nullptr, int accumulate = 0)
{
  if (!on)
    on = ctx->stream;
  constexpr int MAX_GPW = 64 / GG >= 4 ? 4 : (64 / GG >= 2 ? 2 : 1);
  int gpw = n_groups <= ctx->n_cus ? 1 : (n_groups <= 2 * ctx->n_cus ? 2 : 4);
  gpw = std::min(gpw, MAX_GPW);
  const int skew = STRICT_SKEW4;
#define LAUNCH_STRICT(CHUNK, GPW, GRID)                                                                               \
  hipLaunchKernelGGL((lik_strict_sum_rows_kernel<GG, CHUNK, GPW>), dim3(GRID), dim3(1024), 0, on, strict_terms, ns, np, \
                     n_groups, d_lik, skew, accumulate)
  if constexpr (MAX_GPW >= 4)
    if (gpw == 4)
    {
      LAUNCH_STRICT(16384, 4, (n_groups + 3) / 4);
      return;
    }
  if constexpr (MAX_GPW >= 2)
    if (gpw >= 2)
    {
      LAUNCH_STRICT(32768, 2, (n_groups + 1) / 2);
      return;
    }
  LAUNCH_STRICT(65536, 1, n_groups);
#undef LAUNCH_STRICT
}

// What launch_measure leaves to the kernel behind it when the caller asks for it (`want`: the split pf::measure follows on the
// same stream): the sum over the tiled kernel's per-tile partials — d_lik / d_ratio (and d_beam's ones, `beam_fill`) are then NOT
// written by launch_measure but by lik_pf_partial_kernel (pf_kernels.h: LikTiles), one launch less per update.
struct LikTail
{
  bool want = false;
  bool pending = false, beam_fill = false;
  int n_tiles = 0;
  // the beam model's last step (penalty count -> score, beam_finalize_kernel) left to that kernel as well: ctx->penalty holds the
  // counts, d_beam is NOT written yet (pf_measure_single runs beam_finalize_kernel itself when its kernel cannot take the counts)
  bool want_beam = false, beam_pending = false;
};

int launch_measure(mcl3dl_hip_ctx* ctx, const float* d_pose, size_t n_p, float* d_lik, float* d_ratio, float* d_beam,
                   bool stats, double* stats6, LikTail* tail = nullptr)
{
  if (!ctx->has_scan)
    return ctx->fail(-5, "no scan uploaded: call mcl3dl_hip_upload_scan first");
  if (n_p == 0)
    return 0;
  if (n_p > 0x7fffffffu)
    return ctx->fail(-3, "too many particles");
  const bool want_lik = (d_lik || d_ratio || stats);
  const bool want_beam = (d_beam || stats);
  TRY(ensure_structures(ctx, want_lik && ctx->n_s > 0, want_beam && ctx->n_b > 0, stats));
  const int np = static_cast<int>(n_p);
  bool beam_forked = false, beam_ones_by_finalize = false;
  bool merged = false;  // the beam kernel's work-groups ride in the tiled likelihood kernel's launch (lik_beam_kernel)
  bool merged_particle = false;  // ... in the per-particle likelihood kernel's (lik_particle_beam_kernel)
  bool merged_chain = false;     // ... in the tiled kernel's in-kernel-chain form (strict_order 3)
  struct
  {
    long long n_rays = 0, blocks = 0;
    BeamParams bp{};
    const BeamOrigin* prepared = nullptr;
  } merged_beam;
  // any return between the fork and the join below (a failing HIP call) first waits for the second stream, so the caller
  // never gets control back with beam kernels still writing its buffers
  struct ForkGuard
  {
    mcl3dl_hip_ctx* c;
    bool armed = false;
    ~ForkGuard()
    {
      if (armed)
        (void)hipStreamSynchronize(c->aux_stream);
    }
  } fork_guard{ ctx };
  LikPlan plan;
  if (want_lik && !stats && ctx->n_s > 0)
    TRY(plan_lik(ctx, n_p, static_cast<int>(ctx->n_s), &plan));
  // are the likelihoods of this launch the reference's floats bit for bit (caller-order rows, the float-order replay, the
  // in-kernel chain, or no terms at all)? pf::measure then adds the weights as the reference does, too (pf_float_order)
  if (want_lik && !stats)
    ctx->lik_exact = ctx->n_s == 0 || plan.rows || plan.chain || plan.strict_terms != nullptr;
  if (stats && ctx->n_s > 0)
    TRY(ensure(ctx, ctx->tested, sizeof(double) * n_p));
  // ---- beam model (enqueued first: on its own stream when both models run, see mcl3dl_hip_ctx::aux_stream)
  if (want_beam)
  {
    if (ctx->n_b == 0)
    {
      // (1, 0) for every particle; with the tiled likelihood kernel behind it the per-particle finalize writes the ones
      beam_ones_by_finalize = !stats && d_beam && want_lik && ctx->n_s > 0 && plan.tiled;
      if (!stats && !beam_ones_by_finalize)
        hipLaunchKernelGGL(fill_kernel, dim3((np + 255) / 256), dim3(256), 0, ctx->stream, d_beam, 1.0f,
                           static_cast<float*>(nullptr), 0.0f, np);
    }
    else
    {
      TRY(ensure_pow_table(ctx));
      const BeamParams bp = beam_params(ctx);
      const long long n_rays = static_cast<long long>(n_p) * static_cast<long long>(ctx->n_b);
      const long long blocks = (n_rays + 255) / 256;
      if (blocks > 0x7fffffffLL)
        return ctx->fail(-3, "too many rays for one launch");
      {
        const size_t cap0 = ctx->penalty.cap;
        TRY(ensure(ctx, ctx->penalty, sizeof(unsigned) * n_p));
        if (ctx->penalty.cap != cap0)
          ctx->penalty_clean_n = 0;  // (a new allocation: nothing is known about its content)
      }
      const bool beam_prepared = !stats && ctx->beam_prepare && n_rays >= ctx->beam_prepare_min_rays &&
                                 static_cast<long long>(n_p) * static_cast<long long>(ctx->n_o) < 0x7fffffffLL;
      if (beam_prepared)
        TRY(ensure(ctx, ctx->beam_origin, sizeof(BeamOrigin) * n_p * ctx->n_o));
      if (stats)
        TRY(ensure(ctx, ctx->ray_stats, sizeof(RayStats)));
      // Both models in ONE launch (lik_beam_kernel, update_kernels.h: the two kernels' work-groups interleaved, so that every CU
      // hosts both all the way) whenever the likelihood side is the tiled kernel's cooperative fp64-tree form (G <= 16) and the beam side is large
      // enough to be worth interleaving: the beam kernel is NOT launched here but with the tiled kernel below.
      // Not in front of a LONG caller-order replay (replay_is_long below): on two streams that replay — memory-bound, VALU idle —
      // overlaps the rest of the beam kernel, which the lock-step interleave cannot offer (C5 shard: 3.09 against 3.14 ms).
      // Measured, C3: 0.3446 (two streams) -> 0.3313 ms; 4096 rays per particle: 1.0855 -> 1.0070 (profiles/r06s_lik_beam_one_launch.txt).
      // (the per-particle likelihood kernel's 256-thread form takes the beam kernel's work-groups along the same way:
      // lik_particle_beam_kernel)
      merged_particle = ctx->overlap_models && !stats && want_lik && ctx->n_s > 0 && !plan.tiled && !plan.small && !plan.chain &&
                        ctx->lik_index == 2 && !(np <= ctx->lik_wide_max_particles && ctx->n_s > 512) && blocks >= 16 &&
                        blocks < 0x0fffffffLL && ctx->dg.ov_n == 0;
      // (the in-kernel chain's single-tile form, strict_order 3, rides the same way: a consumer's producer keeps its lower block index)
      merged_chain = ctx->overlap_models && !stats && want_lik && ctx->n_s > 0 && plan.tiled && plan.chain && plan.chain_ppl != 4 &&
                     plan.group_size <= 16 && ctx->lik_coop && ctx->lik_index == 2 && ctx->match_dist_min > 1e-5f && blocks >= 64 &&
                     blocks < 0x3fffffffLL && plan.blocks < 0x3fffffffLL && ctx->dg.ov_n == 0;
      // (in front of the caller-order replay: two streams only where they would be used — from overlap_min_rays rays — AND the replay
      // is long enough to hide the beam kernel's tail behind: a term array of at least half a gigabyte. Measured, merged against
      // streams: 4096 x 4096 + 128 rays - 12 %, 16384 x 4096 + 512 (268 MB) - 3 %, 8192 x 32768 + 512 (1 GB) + 1 %, the C5 shard + 2 %:
      // profiles/r06s_lik_beam_one_launch.txt)
      const bool replay_is_long = plan.strict_terms != nullptr && n_rays >= ctx->overlap_min_rays &&
                                  strict_terms_bytes(n_p, static_cast<int>(ctx->n_s), plan.group_size) >= (static_cast<size_t>(512) << 20);
      merged = ctx->overlap_models && !stats && want_lik && ctx->n_s > 0 && plan.tiled && !plan.chain && !plan.chunk &&
               !replay_is_long && plan.group_size <= 16 && ctx->lik_coop && ctx->lik_index == 2 && ctx->match_dist_min > 1e-5f &&
               blocks >= 64 && blocks < 0x3fffffffLL && plan.blocks < 0x3fffffffLL &&
               ctx->dg.ov_n == 0;  // (the beam kernel's map-update-overlay form needs 66 VGPRs: it would spill inside the 64 of the merged launch)
      // the second stream pays only for large launches: the fork / join events cost ~35 us (64 particles x 96 + 3 points:
      // 52 us per update with them, 17 without), the overlap itself is worth ~5 % at C3 (2.1 M rays)
      const bool overlap = !merged && !merged_particle && !merged_chain && ctx->overlap_models && !stats && want_lik && ctx->n_s > 0 && n_rays >= ctx->overlap_min_rays;
      hipStream_t bs = overlap ? ctx->aux_stream : ctx->stream;
      if (overlap)
      {
        HIP_TRY(hipEventRecord(ctx->ev_fork, ctx->stream));
        HIP_TRY(hipStreamWaitEvent(bs, ctx->ev_fork, 0));
        fork_guard.armed = true;
      }
      EventPair ep{};
      if (!stats)
        TRY(timing_begin(ctx, MCL3DL_KERNEL_BEAM, &ep, bs));
      // (beam_origin_kernel zeroes the counters itself; the update's tail kernel leaves them zeroed behind itself)
      const bool counters_clean = ctx->penalty_clean_n >= n_p;
      ctx->penalty_clean_n = 0;  // ... and from here on they are in use
      if (!beam_prepared && !counters_clean)
      {
        // a kernel, not hipMemsetAsync: a memset node at the head of a single-stream captured update faulted on its third
        // replay (ROCm 7.2; 700 particles x 3 rays, 128 x 48), the same zeroing as a kernel node does not
        hipLaunchKernelGGL(fill_kernel, dim3((np + 255) / 256), dim3(256), 0, bs, reinterpret_cast<float*>(ctx->penalty.p), 0.0f,
                           static_cast<float*>(nullptr), 0.0f, np);
      }
      if (stats)
      {
        HIP_TRY(hipMemsetAsync(ctx->ray_stats.p, 0, sizeof(RayStats), bs));
        hipLaunchKernelGGL((beam_kernel<true>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, bs, d_pose,
                           ctx->scan_beam.as<float4>(), static_cast<int>(ctx->n_b), ctx->origins.as<float4>(), n_rays,
                           ctx->dg, bp, ctx->penalty.as<unsigned>(), ctx->ray_stats.as<RayStats>(),
                           static_cast<const BeamOrigin*>(nullptr), static_cast<int>(ctx->n_o));
      }
      else
      {
        // what depends only on (particle, origin) is computed once per pair when the launch is large enough to pay for
        // one more kernel
        const BeamOrigin* prepared = nullptr;
        if (beam_prepared)
        {
          const long long n_pairs = static_cast<long long>(n_p) * static_cast<long long>(ctx->n_o);
          hipLaunchKernelGGL(beam_origin_kernel, dim3(static_cast<unsigned>((n_pairs + 255) / 256)), dim3(256), 0, bs, d_pose,
                             np, ctx->origins.as<float4>(), static_cast<int>(ctx->n_o), ctx->dg,
                             ctx->beam_origin.as<BeamOrigin>(), ctx->penalty.as<unsigned>());
          prepared = ctx->beam_origin.as<BeamOrigin>();
        }
        if (merged || merged_particle || merged_chain)
        {
          // (the rays ride in the likelihood kernel's launch below; the beam model's last step comes behind that launch)
          merged_beam.n_rays = n_rays;
          merged_beam.blocks = blocks;
          merged_beam.bp = bp;
          merged_beam.prepared = prepared;
        }
        else
        {
        // (a map update rides on the DDA grid as an overlay: the kernel that looks it up is chosen only then)
        const auto kernel = ctx->dg.ov_n > 0 ? beam_kernel<false, true> : beam_kernel<false, false>;
        hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, bs, d_pose,
                           ctx->scan_beam.as<float4>(), static_cast<int>(ctx->n_b), ctx->origins.as<float4>(), n_rays,
                           ctx->dg, bp, ctx->penalty.as<unsigned>(), static_cast<RayStats*>(nullptr), prepared,
                           static_cast<int>(ctx->n_o));
        if (tail && tail->want_beam)
          tail->beam_pending = true;
        else
          hipLaunchKernelGGL(beam_finalize_kernel, dim3((np + 255) / 256), dim3(256), 0, bs,
                             ctx->penalty.as<unsigned>(), ctx->pow_table.as<float>(), ctx->beam_likelihood_min, d_beam,
                             np);
        }
        TRY(timing_end(ctx, ep, bs));
      }
      if (overlap)
      {
        HIP_TRY(hipEventRecord(ctx->ev_join, bs));
        beam_forked = true;
      }
    }
    HIP_TRY(hipGetLastError());
  }
  // ---- likelihood-field model
  if (want_lik)
  {
    if (ctx->n_s == 0)
    {
      if (!stats)
        hipLaunchKernelGGL(fill_kernel, dim3((np + 255) / 256), dim3(256), 0, ctx->stream, d_lik, 1.0f, d_ratio, 0.0f,
                           np);
    }
    else
    {
      const LikParams lp = lik_params(ctx);
      const int ns = static_cast<int>(ctx->n_s);
      EventPair ep{};
      if (stats)
      {
        hipLaunchKernelGGL((likelihood_kernel<256, 0, true>), dim3(np), dim3(256), 0, ctx->stream, d_pose,
                           ctx->scan_lik.as<float4>(), ns, ctx->lg, ctx->rg, lp, nullptr, nullptr,
                           ctx->tested.as<double>(), 0);
      }
      else
      {
        TRY(timing_begin(ctx, MCL3DL_KERNEL_LIKELIHOOD, &ep));
        const float4* scan = ctx->scan_lik.as<float4>();
        const bool tiled = plan.tiled, small = plan.small;
        // the cooperative form's sqrt needs match_dist_min > 1.2e-7 m (likelihood_kernels.h:sqrt_in_radius)
        const int coop_arg = (ctx->lik_coop && ctx->lik_index == 2 && ctx->match_dist_min > 1e-5f) ? 1 : 0;
        const int group_size = plan.group_size;
        float* strict_terms = plan.strict_terms;
        // caller-order float sums inside the per-particle kernels: the permutation (device index -> caller's index) and the
        // LDS row of chain_row_floats(ns) floats
        const uint32_t* row_perm = plan.rows ? ctx->scan_perm.as<uint32_t>() : nullptr;
        const size_t row_bytes = plan.rows ? sizeof(float) * static_cast<size_t>(chain_row_floats(ns)) : 0;
        if (small)
        {
          const int W = plan.W;
          const long long blocks = plan.blocks;
#define LAUNCH_SMALL(WW, MODE)                                                                                         \
  hipLaunchKernelGGL((likelihood_small_kernel<WW, MODE>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0,           \
                     ctx->stream, d_pose, np, scan, ns, ctx->lg, ctx->rg, lp, d_lik, d_ratio, coop_arg, row_perm)
#define LAUNCH_SMALL_W(MODE)       \
  switch (W)                       \
  {                                \
    case 1: LAUNCH_SMALL(1, MODE); break;   \
    case 2: LAUNCH_SMALL(2, MODE); break;   \
    case 4: LAUNCH_SMALL(4, MODE); break;   \
    case 8: LAUNCH_SMALL(8, MODE); break;   \
    case 16: LAUNCH_SMALL(16, MODE); break; \
    default: LAUNCH_SMALL(32, MODE); break; \
  }
          if (ctx->lik_index == 2)
          {
            LAUNCH_SMALL_W(2)
          }
          else
          {
            LAUNCH_SMALL_W(0)
          }
#undef LAUNCH_SMALL_W
#undef LAUNCH_SMALL
        }
        else if (tiled)
        {
          const int G = group_size;
          const int n_tiles = plan.n_tiles, n_groups = plan.n_groups;
          const long long blocks = plan.blocks;
#define LAUNCH_TILED(GG, MODE, WW, CC, DD)                                                                             \
  hipLaunchKernelGGL((likelihood_tiled_kernel<GG, MODE, WW, CC, DD>), dim3(static_cast<unsigned>(t_blocks)), dim3(256), 0, \
                     ctx->stream, d_pose, np, t_scan, t_ns, t_tiles, n_groups, ctx->lg, ctx->rg, lp, t_psum, t_pcnt, \
                     t_perm, t_terms, STRICT_SKEW4)
          const bool coop = coop_arg != 0;
          const bool defer = coop && lik_defer_active(ctx);
          // the launch of both models (lik_beam_kernel): the beam side and the interleave for a tiled grid of t_blocks work-groups —
          // rounds of beam8 x 8 beam work-groups + tiled8 x 8 tiled ones, beam8 : tiled8 ~ the two grids' ratio (each at most 8)
          const auto merged_args = [&](long long t_blocks, LikBeamArgs& a) -> long long
          {
            const long long nbb = merged_beam.blocks;
            uint32_t beam8 = 1, tiled8 = 1;
            if (nbb >= t_blocks)
              beam8 = static_cast<uint32_t>(std::min<long long>(8, (nbb + t_blocks / 2) / t_blocks));
            else
              tiled8 = static_cast<uint32_t>(std::min<long long>(8, (t_blocks + nbb / 2) / nbb));
            const long long rounds = std::max((nbb + 8 * beam8 - 1) / (8 * beam8), (t_blocks + 8 * tiled8 - 1) / (8 * tiled8));
            a.pose7 = d_pose;
            a.n_p = np;
            a.n_groups = plan.n_groups;
            a.g = ctx->lg;
            a.rg = ctx->rg;
            a.prm = lp;
            a.strict_skew4 = STRICT_SKEW4;
            a.scan_beam = ctx->scan_beam.as<float4>();
            a.n_b = static_cast<int>(ctx->n_b);
            a.origins = ctx->origins.as<float4>();
            a.n_rays = merged_beam.n_rays;
            a.dg = ctx->dg;
            a.bp = merged_beam.bp;
            a.penalty = ctx->penalty.as<unsigned>();
            a.prepared = merged_beam.prepared;
            a.n_o = static_cast<int>(ctx->n_o);
            a.beam8 = beam8;
            a.tiled8 = tiled8;
            a.n_beam_blocks = static_cast<uint32_t>(nbb);
            a.n_tiled_blocks = static_cast<uint32_t>(t_blocks);
            return rounds * 8 * (beam8 + tiled8);
          };
          // the beam model's last step behind such a launch: left to the update's tail kernel, or a launch of its own
          const auto merged_beam_done = [&]()
          {
            if (tail && tail->want_beam)
              tail->beam_pending = true;
            else
              hipLaunchKernelGGL(beam_finalize_kernel, dim3((np + 255) / 256), dim3(256), 0, ctx->stream, ctx->penalty.as<unsigned>(),
                                 ctx->pow_table.as<float>(), ctx->beam_likelihood_min, d_beam, np);
          };
          if (plan.chain)
          {
            ChainSerial& cs = chain_serial(ctx->device);
            std::lock_guard<std::mutex> chain_lock(cs.m);
            if (cs.ev)
              HIP_TRY(hipStreamWaitEvent(ctx->stream, cs.ev, 0));
            else
              HIP_TRY(hipEventCreateWithFlags(&cs.ev, hipEventDisableTiming));
            struct ChainDone  // (recorded behind the launch on every way out of this block)
            {
              hipEvent_t ev;
              hipStream_t st;
              ~ChainDone()
              {
                (void)hipEventRecord(ev, st);
              }
            } chain_done{ cs.ev, ctx->stream };
            float* lik_out = d_lik;
            if (!lik_out)  // (only the match ratio was asked for: the sum still has somewhere to go)
            {
              TRY(ensure(ctx, ctx->chain_lik, sizeof(float) * n_p));
              lik_out = ctx->chain_lik.as<float>();
            }
            const LikChain lc{ ctx->chain_carry.as<unsigned long long>(), plan.chain_tag0, lik_out, d_ratio,
                               beam_ones_by_finalize ? d_beam : static_cast<float*>(nullptr), ctx->chain_err };
            if (plan.chain_ppl == 4)
            {
              hipLaunchKernelGGL((likelihood_chain_multi_kernel<4, 4>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, ctx->stream,
                                 d_pose, np, scan, ns, n_tiles, plan.n_super, n_groups, ctx->rg, lp, lc);
            }
            else if (merged_chain)
            {
              LikBeamArgs a{};
              const long long grid = merged_args(blocks, a);
              a.scan = scan;
              a.n_s = ns;
              a.n_tiles = n_tiles;
              a.ch = lc;
#define LAUNCH_MERGED_CHAIN(GG)                                                                                                 \
  do                                                                                                                            \
  {                                                                                                                             \
    if (defer)                                                                                                                  \
      hipLaunchKernelGGL((lik_beam_kernel<GG, true, false, true>), dim3(static_cast<unsigned>(grid)), dim3(256), 0, ctx->stream, a);  \
    else                                                                                                                        \
      hipLaunchKernelGGL((lik_beam_kernel<GG, false, false, true>), dim3(static_cast<unsigned>(grid)), dim3(256), 0, ctx->stream, a); \
  } while (0)
              switch (G)
              {
                case 4:
                  LAUNCH_MERGED_CHAIN(4);
                  break;
                case 8:
                  LAUNCH_MERGED_CHAIN(8);
                  break;
                default:
                  LAUNCH_MERGED_CHAIN(16);
                  break;
              }
#undef LAUNCH_MERGED_CHAIN
              merged_beam_done();
            }
            else
            {
#define LAUNCH_CHAIN(GG, MODE, CC, DD)                                                                                  \
  hipLaunchKernelGGL((likelihood_tiled_kernel<GG, MODE, 8, CC, DD, true>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, \
                     ctx->stream, d_pose, np, scan, ns, n_tiles, n_groups, ctx->lg, ctx->rg, lp,              \
                     static_cast<double*>(nullptr), static_cast<unsigned*>(nullptr),                                   \
                     static_cast<const uint32_t*>(nullptr), static_cast<float*>(nullptr), 0, lc)
#define LAUNCH_CHAIN_G(GG)              \
  do                                    \
  {                                     \
    if (coop && defer)                  \
      LAUNCH_CHAIN(GG, 2, true, true);  \
    else if (coop)                      \
      LAUNCH_CHAIN(GG, 2, true, false); \
    else if (ctx->lik_index == 2)       \
      LAUNCH_CHAIN(GG, 2, false, false);\
    else                                \
      LAUNCH_CHAIN(GG, 0, false, false);\
  } while (0)
            switch (G)
            {
              case 4:
                LAUNCH_CHAIN_G(4);
                break;
              case 8:
                LAUNCH_CHAIN_G(8);
                break;
              default:
                LAUNCH_CHAIN_G(16);
                break;
            }
#undef LAUNCH_CHAIN_G
#undef LAUNCH_CHAIN
            }
          }
          else
          {
#define LAUNCH_TILED_G(GG, WW)         \
  do                                   \
  {                                    \
    if (coop && defer)                 \
      LAUNCH_TILED(GG, 2, WW, true, true);   \
    else if (coop)                     \
      LAUNCH_TILED(GG, 2, WW, true, false);  \
    else if (ctx->lik_index == 2)      \
      LAUNCH_TILED(GG, 2, WW, false, false); \
    else                               \
      LAUNCH_TILED(GG, 0, WW, false, false); \
  } while (0)
          // one launch of the tiled kernel over t_ns points starting at t_scan (the whole scan, or one chunk of it)
          const auto launch_tiled = [&](const float4* t_scan, int t_ns, int t_tiles, double* t_psum, unsigned* t_pcnt,
                                        const uint32_t* t_perm, float* t_terms)
          {
            const long long full = t_tiles & ~7, rem = static_cast<long long>(t_tiles - full) * n_groups;
            const long long t_blocks = 8 * ((full / 8) * n_groups + (rem + 7) / 8);  // (plan_lik's count, for this many tiles)
            if (merged)
            {
              LikBeamArgs a{};
              const long long grid = merged_args(t_blocks, a);
              a.scan = t_scan;
              a.n_s = t_ns;
              a.n_tiles = t_tiles;
              a.partial_sum = t_psum;
              a.partial_cnt = t_pcnt;
              a.scan_perm = t_perm;
              a.strict_terms = t_terms;
#define LAUNCH_MERGED_G(GG)                                                                                              \
  do                                                                                                                     \
  {                                                                                                                      \
    if (defer)                                                                                                           \
      hipLaunchKernelGGL((lik_beam_kernel<GG, true, false>), dim3(static_cast<unsigned>(grid)), dim3(256), 0, ctx->stream, a);  \
    else                                                                                                                 \
      hipLaunchKernelGGL((lik_beam_kernel<GG, false, false>), dim3(static_cast<unsigned>(grid)), dim3(256), 0, ctx->stream, a); \
  } while (0)
              switch (G)
              {
                case 4:
                  LAUNCH_MERGED_G(4);
                  break;
                case 8:
                  LAUNCH_MERGED_G(8);
                  break;
                default:
                  LAUNCH_MERGED_G(16);
                  break;
              }
#undef LAUNCH_MERGED_G
              merged_beam_done();
              return;
            }
            switch (G)
            {
              case 4:
                LAUNCH_TILED_G(4, 8);
                break;
              case 8:
                LAUNCH_TILED_G(8, 8);
                break;
              case 32:
                LAUNCH_TILED_G(32, 4);  // 33 KB of LDS per work-group: 4 wavefronts per SIMD at most
                break;
              default:
                LAUNCH_TILED_G(16, 8);
                break;
            }
          };
          const auto launch_replay = [&](const float* terms, int r_ns, hipStream_t on, int accumulate)
          {
            switch (G)
            {
              case 4:
                launch_strict_sum<4>(ctx, terms, r_ns, np, n_groups, d_lik, on, accumulate);
                break;
              case 8:
                launch_strict_sum<8>(ctx, terms, r_ns, np, n_groups, d_lik, on, accumulate);
                break;
              case 32:
                launch_strict_sum<32>(ctx, terms, r_ns, np, n_groups, d_lik, on, accumulate);
                break;
              default:
                launch_strict_sum<16>(ctx, terms, r_ns, np, n_groups, d_lik, on, accumulate);
                break;
            }
          };
          double* const psum = ctx->lik_partial_sum.as<double>();
          unsigned* const pcnt = ctx->lik_partial_cnt.as<unsigned>();
          if (strict_terms && plan.chunk && d_lik)
          {
            // The scan was ordered in chunks of the caller's order (host_cloud.h:device_order_scans). Chunk c is evaluated on the
            // context's stream into term buffer c % 2 while chunk c - 1 is replayed — the reference's float recurrence continued
            // from the sums chunk c - 2 ... left in d_lik — on a stream of its own: the replay streams its terms at memory speed,
            // the evaluation is bound by VALU issue, and the two share the GPU well; the buffer holds two chunks, not the scan.
            const size_t chunk = plan.chunk;
            const int n_chunks = static_cast<int>((static_cast<size_t>(ns) + chunk - 1) / chunk);
            float* const buf[2] = { strict_terms,
                                    strict_terms + strict_terms_bytes(n_p, static_cast<int>(chunk), G) / sizeof(float) };
            struct ReplayGuard
            {
              mcl3dl_hip_ctx* c;
              ~ReplayGuard()
              {
                (void)hipStreamSynchronize(c->replay_stream);
              }
            };
            bool failed = false;
            for (int c = 0; c < n_chunks && !failed; ++c)
            {
              const size_t first = static_cast<size_t>(c) * chunk;
              const int c_ns = static_cast<int>(std::min(chunk, static_cast<size_t>(ns) - first));
              const int c_tiles = (c_ns + 255) / 256, tile0 = static_cast<int>(first / 256);
              if (c >= 2)
                failed = failed || hipStreamWaitEvent(ctx->stream, ctx->ev_replay[c & 1], 0) != hipSuccess;  // its buffer is free again
              launch_tiled(scan + first, c_ns, c_tiles, psum + static_cast<size_t>(tile0) * n_p, pcnt + static_cast<size_t>(tile0) * n_p,
                           ctx->scan_perm.as<uint32_t>() + first, buf[c & 1]);
              failed = failed || hipEventRecord(ctx->ev_tiled[c & 1], ctx->stream) != hipSuccess ||
                       hipStreamWaitEvent(ctx->replay_stream, ctx->ev_tiled[c & 1], 0) != hipSuccess;
              launch_replay(buf[c & 1], c_ns, ctx->replay_stream, c > 0 ? 1 : 0);
              failed = failed || hipEventRecord(ctx->ev_replay[c & 1], ctx->replay_stream) != hipSuccess;
            }
            // match ratios (and nothing else: the likelihoods are the replay's) from the per-tile counts
            hipLaunchKernelGGL(lik_finalize_kernel, dim3((np + 31) / 32), dim3(256), 0, ctx->stream, psum, pcnt, n_tiles, np, ns,
                               static_cast<float*>(nullptr), d_ratio, beam_ones_by_finalize ? d_beam : static_cast<float*>(nullptr));
            if (failed || hipStreamWaitEvent(ctx->stream, ctx->ev_replay[(n_chunks - 1) & 1], 0) != hipSuccess)
            {
              ReplayGuard drain{ ctx };
              return ctx->fail(-2, "stream / event call failed in the chunked float-order replay: %s", hipGetErrorString(hipGetLastError()));
            }
          }
          else
          {
          // (a chunk-ordered scan whose likelihoods nobody asked for: its permutation is chunk-relative, no terms are kept)
          launch_tiled(scan, ns, n_tiles, psum, pcnt, ctx->scan_perm.as<uint32_t>(), plan.chunk ? nullptr : strict_terms);
          if (tail && tail->want && !strict_terms && d_lik && d_ratio)
          {
            tail->pending = true;  // lik_pf_partial_kernel adds the tiles up
            tail->n_tiles = n_tiles;
            tail->beam_fill = beam_ones_by_finalize;
          }
          else
          hipLaunchKernelGGL(lik_finalize_kernel, dim3((np + 31) / 32), dim3(256), 0, ctx->stream, psum, pcnt, n_tiles, np, ns,
                               d_lik, d_ratio, beam_ones_by_finalize ? d_beam : static_cast<float*>(nullptr));
          if (strict_terms && d_lik && !plan.chunk)
            launch_replay(strict_terms, ns, ctx->stream, 0);
          }
#undef LAUNCH_TILED_G
#undef LAUNCH_TILED
          }
        }
        else
        {
#define LAUNCH_LIK(BLOCK, MODE)                                                                                   \
  hipLaunchKernelGGL((likelihood_kernel<BLOCK, MODE, false>), dim3(np), dim3(BLOCK), row_bytes, ctx->stream, d_pose, scan, ns, \
                     ctx->lg, ctx->rg, lp, d_lik, d_ratio, nullptr, coop_arg, row_perm)
        if (merged_particle)
        {
          const int lik_block = ns <= 128 ? 64 : 256;  // (the work-group size LAUNCH_LIK below would take)
          const long long nbb = (merged_beam.n_rays + lik_block - 1) / lik_block, npl = np;
          uint32_t beam8 = 1, lik8 = 1;
          if (nbb >= npl)
            beam8 = static_cast<uint32_t>(std::min<long long>(8, (nbb + npl / 2) / npl));
          else
            lik8 = static_cast<uint32_t>(std::min<long long>(8, (npl + nbb / 2) / nbb));
          const long long rounds = std::max((nbb + 8 * beam8 - 1) / (8 * beam8), (npl + 8 * lik8 - 1) / (8 * lik8));
          LikParticleBeamArgs a{};
          a.pose7 = d_pose;
          a.n_p = np;
          a.scan = scan;
          a.n_s = ns;
          a.g = ctx->lg;
          a.rg = ctx->rg;
          a.prm = lp;
          a.out_lik = d_lik;
          a.out_ratio = d_ratio;
          a.coop = coop_arg;
          a.perm = row_perm;
          a.scan_beam = ctx->scan_beam.as<float4>();
          a.n_b = static_cast<int>(ctx->n_b);
          a.origins = ctx->origins.as<float4>();
          a.n_rays = merged_beam.n_rays;
          a.dg = ctx->dg;
          a.bp = merged_beam.bp;
          a.penalty = ctx->penalty.as<unsigned>();
          a.prepared = merged_beam.prepared;
          a.n_o = static_cast<int>(ctx->n_o);
          a.beam8 = beam8;
          a.lik8 = lik8;
          a.n_beam_blocks = static_cast<uint32_t>(nbb);
          if (lik_block == 64)
            hipLaunchKernelGGL(lik_particle_beam_kernel<64>, dim3(static_cast<unsigned>(rounds * 8 * (beam8 + lik8))), dim3(64), row_bytes,
                               ctx->stream, a);
          else
            hipLaunchKernelGGL(lik_particle_beam_kernel<256>, dim3(static_cast<unsigned>(rounds * 8 * (beam8 + lik8))), dim3(256), row_bytes,
                               ctx->stream, a);
          if (tail && tail->want_beam)
            tail->beam_pending = true;
          else
            hipLaunchKernelGGL(beam_finalize_kernel, dim3((np + 255) / 256), dim3(256), 0, ctx->stream, ctx->penalty.as<unsigned>(),
                               ctx->pow_table.as<float>(), ctx->beam_likelihood_min, d_beam, np);
        }
        else if (ctx->lik_index == 2)
        {
          if (ns <= 128)
            LAUNCH_LIK(64, 2);
          else if (np <= ctx->lik_wide_max_particles && ns > 512)
            LAUNCH_LIK(1024, 2);  // few particles: 16 wavefronts share a scan — a quarter of the dependent load chains per lane
          else
            LAUNCH_LIK(256, 2);
        }
        else
        {
          if (ns <= 128)
            LAUNCH_LIK(64, 0);
          else
            LAUNCH_LIK(256, 0);
        }
#undef LAUNCH_LIK
        }
        TRY(timing_end(ctx, ep));
      }
    }
    HIP_TRY(hipGetLastError());
  }
  if (beam_forked)
  {
    HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));  // later work on `stream` sees the beam scores
    fork_guard.armed = false;
  }
  if (stats)
  {
    std::vector<double> tested(ctx->n_s ? n_p : 0);
    RayStats rs{ 0, 0, 0 };
    if (ctx->n_s)
      TRY(d2h(ctx, tested.data(), ctx->tested.p, sizeof(double) * n_p));
    if (ctx->n_b)
      TRY(d2h(ctx, &rs, ctx->ray_stats.p, sizeof(RayStats)));
    TRY(sync_stream(ctx));
    stats6[0] = std::accumulate(tested.begin(), tested.end(), 0.0);
    stats6[1] = static_cast<double>(n_p) * static_cast<double>(ctx->n_s);
    stats6[2] = static_cast<double>(rs.steps);
    stats6[3] = static_cast<double>(rs.occupied);
    stats6[4] = static_cast<double>(rs.tested);
    stats6[5] = static_cast<double>(n_p) * static_cast<double>(ctx->n_b);
  }
  return 0;
}

int pf_blocks(size_t n)
{
  const size_t b = (n + PF_BLOCK - 1) / PF_BLOCK;
  return static_cast<int>(std::min<size_t>(std::max<size_t>(b, 1), 1024));
}


// Does pf::measure on ONE GPU add the un-normalised weights as the reference does (pf.h:255-260: float, sequentially, particle
// order; float_chain.h) instead of the fp64 tree? strict_order 1: always. The default (2): up to pf_fused_max = 1024 particles —
// the reference's operating range; pf::measure is then the reference's arithmetic bit for bit given its inputs, inside the
// fused kernel / the one-launch update at no extra launch. Beyond that the recurrence needs a launch of its own
// (pf_strict_sum_kernel: +10 us at 4096 particles, a third of a 4096 x 96 update, profiles/r06d_rows_vs_replay.txt) for weights
// that agree to ~1e-7 anyway. Independent of pf_fused, so that the fused and the split form give the same bits.
bool pf_float_order(const mcl3dl_hip_ctx* ctx, size_t n_p)
{
  return ctx->strict_order == 1 || (ctx->strict_order == 2 && n_p <= static_cast<size_t>(std::min(ctx->pf_fused_max, PF_FUSED_MAX)));
}

// The whole update — both models and pf::measure — as ONE launch (update_kernels.h) where the sizes are launch-bound:
// returns 1 when it was enqueued, 0 when this update is not eligible (the caller then runs the separate kernels), < 0 on
// error. Eligible: one GPU, per-particle likelihood kernel (not the tiled / small-scan forms), at most update_small_max
// particles and 256 beam points.
int launch_update_small(mcl3dl_hip_ctx* ctx, const float* d_pose, size_t n_p, float* d_weight, const float* d_extra,
                        float* d_lik, float* d_ratio, float* d_beam, float* d_stats4, const PfEmit* ho = nullptr)
{
  if (!ctx->update_small || n_p == 0 || ctx->n_b > 256 || ctx->strict_order == 3 || !ctx->has_scan || !d_lik || !d_ratio || !d_beam)
    return 0;
  if (n_p > static_cast<size_t>(ctx->update_small_max))
    return 0;
  const int ns = static_cast<int>(ctx->n_s);
  if (ns > 0)
  {
    // (decided without plan_lik's buffer allocations: the tiled form needs per-tile partials this path never touches)
    const int np = static_cast<int>(n_p);
    const bool tiled = lik_mode(ctx, np, ns).tiled;
    const bool small = !tiled && ns <= 32 && np >= 256 && ctx->lik_small;
    if (tiled || small)
      return 0;
  }
  TRY(ensure_structures(ctx, ns > 0, ctx->n_b > 0));
  if (ctx->n_b > 0 && ctx->dg.ov_n > 0)
    return 0;  // a map update rides on the DDA grid: the one-launch kernel is compiled without the overlay lookup
  if (ctx->n_b > 0)
    TRY(ensure_pow_table(ctx));
  const int nvb = static_cast<int>((n_p + PF_BLOCK - 1) / PF_BLOCK);
  TRY(ensure(ctx, ctx->wnew, sizeof(float) * n_p));
  TRY(ensure(ctx, ctx->block_partials, sizeof(double) * 4 * static_cast<size_t>(nvb)));
  TRY(ensure(ctx, ctx->partial4, sizeof(double) * 4));
  const size_t n_tickets = static_cast<size_t>(nvb) * 37 + static_cast<size_t>(ticket_tree_size(nvb)) + 1;
  if (ctx->us_tickets.cap < sizeof(unsigned) * n_tickets)
  {
    TRY(ensure(ctx, ctx->us_tickets, sizeof(unsigned) * n_tickets));
    // (a kernel, not hipMemsetAsync: see launch_measure — a memset node in a captured update faulted on replay)
    const int n_words = static_cast<int>(ctx->us_tickets.cap / sizeof(unsigned));
    hipLaunchKernelGGL(fill_kernel, dim3((n_words + 255) / 256), dim3(256), 0, ctx->stream,
                       reinterpret_cast<float*>(ctx->us_tickets.p), 0.0f, static_cast<float*>(nullptr), 0.0f,
                       n_words);  // the update kernel leaves them zero
  }
  UpdateSmallArgs a{};
  a.pose7 = d_pose;
  a.n_p = static_cast<int>(n_p);
  a.scan_lik = ctx->scan_lik.as<float4>();
  a.n_s = ns;
  a.g = ctx->lg;
  a.rg = ctx->rg;
  a.prm = lik_params(ctx);
  a.coop = (ctx->lik_coop && ctx->lik_index == 2 && ctx->match_dist_min > 1e-5f) ? 1 : 0;
  a.scan_beam = ctx->scan_beam.as<float4>();
  a.n_b = static_cast<int>(ctx->n_b);
  a.origins = ctx->origins.as<float4>();
  a.dg = ctx->dg;
  a.bp = beam_params(ctx);
  a.pow_table = ctx->pow_table.as<float>();
  a.w = d_weight;
  a.extra = d_extra;
  a.use_beam = 1;
  a.out_lik = d_lik;
  a.out_ratio = d_ratio;
  a.out_beam = d_beam;
  a.w_new = ctx->wnew.as<float>();
  a.vb_partials = ctx->block_partials.as<double>();
  a.tickets = ctx->us_tickets.as<unsigned>();
  a.packed = ctx->partial4.as<double>();
  a.stats4 = d_stats4;
  a.conformant = ctx->update_small_conformant;
  a.emit = ho ? *ho : PfEmit{};
  // the reference's float recurrences, in its own order: the likelihood terms over the caller's scan (lik_particle's row) and —
  // where this launch also finishes pf::measure — the weights over the particles (pf.h:255-260)
  const bool rows = ns > 0 && lik_mode(ctx, static_cast<int>(n_p), ns).rows;
  const bool float_w = pf_float_order(ctx, n_p);
  a.perm = rows ? ctx->scan_perm.as<uint32_t>() : nullptr;
  a.float_order_w = float_w ? 1 : 0;
  ctx->lik_exact = rows || ns == 0;
  const size_t row_floats = std::max<size_t>(rows ? chain_row_floats(ns) : 0, float_w ? chain_row_floats(static_cast<int>(n_p)) : 0);
  const size_t row_bytes = sizeof(float) * row_floats;
  EventPair ep{};
  TRY(timing_begin(ctx, MCL3DL_KERNEL_UPDATE, &ep));
  const unsigned grid = static_cast<unsigned>(n_p);
#define LAUNCH_US(BLOCK, MODE) \
  hipLaunchKernelGGL((update_small_kernel<BLOCK, MODE>), dim3(grid), dim3(BLOCK), row_bytes, ctx->stream, a)
  // the work-group size the separate likelihood kernel would get (launch_measure), so that the lanes add in the same order
  const bool narrow = ns <= 128 && ctx->n_b <= 128;
  if (ctx->lik_index == 2)
  {
    if (ns <= 128)
      LAUNCH_US(64, 2);
    else if (static_cast<int>(n_p) <= ctx->lik_wide_max_particles && ns > 512)
      LAUNCH_US(1024, 2);
    else
      LAUNCH_US(256, 2);
  }
  else
  {
    if (ns <= 128)
      LAUNCH_US(64, 0);
    else
      LAUNCH_US(256, 0);
  }
  (void)narrow;
#undef LAUNCH_US
  TRY(timing_end(ctx, ep));
  HIP_TRY(hipGetLastError());
  return 1;
}

}  // namespace
