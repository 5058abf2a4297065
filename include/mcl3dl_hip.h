/* mcl3dl_hip.h — C ABI of the MI355X-native LiDAR measurement-update engine for mcl_3dl.
 *
 * This is the drop-in boundary: plain C, plain pointers and sizes, no C++/torch types.
 * Each entry point names the reference interface it replaces (paths relative to the at-wat/mcl_3dl
 * v0.7.0 tree).  The C++ adapter classes in mcl_3dl_amd/cpp/include/ (same class names as the
 * reference's plugins) and the ctypes binding mcl_3dl_amd/capi.py both sit on top of exactly these symbols.
 *
 * Conventions
 *   - every function returning int: 0 = OK, negative = error; mcl3dl_hip_last_error(ctx) has the text.
 *   - the caller owns every host buffer; the context owns all device memory. One context = one GPU;
 *     a context is not thread-safe (the reference caller is single-threaded: src/mcl_3dl.cpp:1466).
 *   - arrays are contiguous, little-endian, float32 unless stated.  pose = 7 floats per particle:
 *     px,py,pz (State6DOF::pos_), qx,qy,qz,qw (State6DOF::rot_, NOT required to be normalised).
 *   - "host" entry points take host pointers and are synchronous; "device" entry points take device
 *     pointers, enqueue on the context's stream and return without synchronising.
 *   - There is no CPU fallback: without a usable gfx950 device mcl3dl_hip_create fails.
 */
#ifndef MCL3DL_HIP_H
#define MCL3DL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCL3DL_HIP_ABI_VERSION 3

typedef struct mcl3dl_hip_ctx mcl3dl_hip_ctx;

/* BeamStatus, include/mcl_3dl/lidar_measurement_models/lidar_measurement_model_beam.h:64-70 (same order). */
enum
{
  MCL3DL_BEAM_SHORT = 0,
  MCL3DL_BEAM_HIT = 1,
  MCL3DL_BEAM_LONG = 2,
  MCL3DL_BEAM_TOTAL_REFLECTION = 3
};

/* kernel ids for mcl3dl_hip_get_kernel_time */
enum
{
  MCL3DL_KERNEL_LIKELIHOOD = 0,
  MCL3DL_KERNEL_BEAM = 1,
  MCL3DL_KERNEL_PF = 2,
  MCL3DL_KERNEL_UPDATE = 3, /* the whole update as one launch (<= update_small_max particles): likelihood + beam + pf::measure */
  MCL3DL_KERNEL_STAGE = 4,  /* scan_stage_kernel of a host-buffer update: scan ordering + pose / weight take-over */
  MCL3DL_KERNEL_COUNT = 5
};

int mcl3dl_hip_abi_version(void);

/* ---- lifecycle ------------------------------------------------------------------------------------ */
/* Replaces: construction of the two LiDAR models + kd-tree in MCL3dlNode (src/mcl_3dl.cpp:1315-1329). */
/* Number of HIP devices this process sees (0 when there is none: the library has no CPU path). */
int mcl3dl_hip_device_count(void);
/* The environment variable MCL3DL_HIP_OPTIONS ("name=value,name=value", names of mcl3dl_hip_set_option) is applied to every
 * context at creation; an entry that mcl3dl_hip_set_option rejects fails the creation (-3, reason on stderr). */
int mcl3dl_hip_create(mcl3dl_hip_ctx** out, int device_id);
void mcl3dl_hip_destroy(mcl3dl_hip_ctx* ctx);
const char* mcl3dl_hip_last_error(const mcl3dl_hip_ctx* ctx);
/* Use an existing hipStream_t (e.g. the host framework's current stream); NULL = the context's own (non-blocking)
 * stream. To run on the legacy default stream pass hipStreamLegacy, not NULL. */
int mcl3dl_hip_set_stream(mcl3dl_hip_ctx* ctx, void* hip_stream);
void* mcl3dl_hip_get_stream(mcl3dl_hip_ctx* ctx);
int mcl3dl_hip_synchronize(mcl3dl_hip_ctx* ctx);

/* ---- map + parameters ------------------------------------------------------------------------------ */
/* Replaces: ChunkedKdtree::setInputCloud (include/mcl_3dl/chunked_kdtree.h:124-216, called at
 * src/mcl_3dl.cpp:1369) and RaycastUsingDDA::updatePointCloud (include/mcl_3dl/raycasts/raycast_using_dda.h:162-190).
 * dist_weight = the PointRepresentation rescale values (src/mcl_3dl.cpp:1270), NULL = none.
 * The device structures are (re)built lazily, keyed on `stamp` like raycast_using_dda.h:168. */
int mcl3dl_hip_set_map(mcl3dl_hip_ctx* ctx, const float* xyz /*n_m*3*/, const uint32_t* label /*n_m or NULL*/,
                       size_t n_m, uint64_t stamp, const float* dist_weight /*3 or NULL*/);
/* Replaces: LidarMeasurementModelLikelihoodParameters + refreshParameters
 * (include/mcl_3dl/parameters.h:64-89, src/lidar_measurement_model_likelihood.cpp:56-61). */
int mcl3dl_hip_set_likelihood_params(mcl3dl_hip_ctx* ctx, float match_dist_min, float match_dist_flat,
                                     float match_weight);
/* Replaces: LidarMeasurementModelBeamParameters + refreshParameters with use_raycast_using_dda = true
 * (include/mcl_3dl/parameters.h:91-132, src/lidar_measurement_model_beam.cpp:58-80). num_points is
 * num_points_default_ (it defines beam_likelihood_ = pow(beam_likelihood_min, 1/num_points)). */
int mcl3dl_hip_set_beam_params(mcl3dl_hip_ctx* ctx, float map_grid_x, float map_grid_y, float map_grid_z,
                               float dda_grid_size, float ray_angle_half, float hit_range, float beam_likelihood_min,
                               uint32_t num_points, float ang_total_ref, uint32_t filter_label_max,
                               int add_penalty_short_only_mode);

/* ---- host entry points (synchronous) ------------------------------------------------------------- */
/* Replaces: the N_p calls of LidarMeasurementModelLikelihood::measure
 * (src/lidar_measurement_model_likelihood.cpp:105-139) and LidarMeasurementModelBeam::measure
 * (src/lidar_measurement_model_beam.cpp:124-155) made by the measure lambda (src/mcl_3dl.cpp:409-415).
 * n_s == 0 -> out_lik = 1, out_match_ratio = 0; n_b == 0 -> out_beam = 1 (the (1,0) of an empty cloud).
 * scan_beam_origin[i] = PointXYZIL::label of beam point i = index into origins. Any out_* may be NULL. */
/* Upload the particle poses once per update; measure_batch calls with pose == NULL and the same n_p then use them. The
 * node asks each model separately (src/mcl_3dl.cpp:409-415): with this the drop-in classes send the poses once per
 * pf::measure, not once per model. Any host-buffer call that is given a pose array replaces the uploaded set. The pose array
 * may be reused as soon as the call returns (it does not wait for the stream unless the copy could not be staged). */
int mcl3dl_hip_upload_poses(mcl3dl_hip_ctx* ctx, const float* pose /*n_p*7*/, size_t n_p);
int mcl3dl_hip_measure_batch(mcl3dl_hip_ctx* ctx, const float* pose /*n_p*7 or NULL*/, size_t n_p,
                             const float* scan_lik_xyz /*n_s*3*/, size_t n_s, const float* scan_beam_xyz /*n_b*3*/,
                             const uint32_t* scan_beam_origin /*n_b*/, size_t n_b, const float* origins /*n_o*3*/,
                             size_t n_o, float* out_lik /*n_p*/, float* out_match_ratio /*n_p*/,
                             float* out_beam /*n_p*/);

/* measure_batch delivered in particle order while the GPU is still working — for a caller that consumes the results one
 * particle at a time on the host, which is what the reference's pf::measure does with the lambda of src/mcl_3dl.cpp:399-426
 * (include/mcl_3dl/pf.h:255-260: ~80 ns of host work per particle, 0.34 ms at 4096 particles — longer than the kernels).
 *   _begin  same arguments as measure_batch + slice_particles (0 = option "batch_slice", itself 0 = automatic: four slices,
 *           none below 512 particles; batches under 1024 particles are not sliced). Enqueues everything and returns; the
 *           INPUT arrays may be reused at once (they were copied or are page-locked memory the kernels read in place — the
 *           latter must stay untouched until _end), the OUTPUT arrays belong to the batch until _end.
 *   _wait   returns once out_*[particle] are valid; *n_ready (optional) = number of leading particles whose results are
 *           valid, so that a loop calls _wait once per slice, not once per particle.
 *   _end    all results delivered (also implied by any other call on the context that synchronises its stream).
 * Results are those of measure_batch bit for bit (a particle's result does not depend on which particles share its launch). */
int mcl3dl_hip_measure_batch_begin(mcl3dl_hip_ctx* ctx, const float* pose /*n_p*7 or NULL*/, size_t n_p,
                                   const float* scan_lik_xyz, size_t n_s, const float* scan_beam_xyz,
                                   const uint32_t* scan_beam_origin, size_t n_b, const float* origins, size_t n_o,
                                   float* out_lik, float* out_match_ratio, float* out_beam, size_t slice_particles);
int mcl3dl_hip_measure_batch_wait(mcl3dl_hip_ctx* ctx, size_t particle, size_t* n_ready);
int mcl3dl_hip_measure_batch_end(mcl3dl_hip_ctx* ctx);

/* Replaces: pf::ParticleFilter::measure (include/mcl_3dl/pf.h:252-279) given the per-particle factors of the
 * measure lambda (src/mcl_3dl.cpp:402-425): weight *= ((1*beam)*lik)*extra ; sum ; if sum > 0 normalise and
 * entropy = -sum(w ln w) over w > 0, else weights restored and *restored = 1 (entropy untouched -> NaN here).
 * beam / extra / match_ratio may be NULL (factor 1 / no min-max). */
int mcl3dl_hip_pf_measure(mcl3dl_hip_ctx* ctx, float* weight_inout /*n_p*/, const float* lik /*n_p*/,
                          const float* beam /*n_p or NULL*/, const float* extra /*n_p or NULL*/,
                          const float* match_ratio /*n_p or NULL*/, size_t n_p, float* entropy,
                          float* match_ratio_min, float* match_ratio_max, int* restored);

/* Replaces: the whole `pf_->measure(measure_func)` statement (src/mcl_3dl.cpp:398-426): measure_batch +
 * pf_measure with everything kept on the device in between.  extra = the odometry-error factor
 * NormalLikelihood(odom_err_integ_lin.norm()) computed by the caller (include/mcl_3dl/nd.h:41-58), or NULL.
 * out_lik / out_match_ratio / out_beam may be NULL. */
int mcl3dl_hip_measure_update(mcl3dl_hip_ctx* ctx, const float* pose /*n_p*7*/, const float* extra /*n_p or NULL*/,
                              float* weight_inout /*n_p*/, size_t n_p, const float* scan_lik_xyz, size_t n_s,
                              const float* scan_beam_xyz, const uint32_t* scan_beam_origin, size_t n_b,
                              const float* origins, size_t n_o, float* out_lik, float* out_match_ratio,
                              float* out_beam, float* entropy, float* match_ratio_min, float* match_ratio_max,
                              int* restored);

/* Page-locked host memory for the arrays of the host-buffer entry points. The reference keeps its particles in a
 * std::vector (include/mcl_3dl/pf.h:457) and its scans in pcl::PointCloud objects; a caller that packs poses / scan
 * points for this library anyway can pack them straight into a block from mcl3dl_hip_host_alloc: mcl3dl_hip_measure_update
 * then reads such arrays where they lie (no staging copy, no DMA copy) and writes weight_inout / out_* arrays that lie in
 * such a block directly from the update's last kernel. Arrays anywhere else keep working (staged through the library's
 * own page-locked block). Blocks belong to the context (freed with it); host_free waits for the context's stream. */
int mcl3dl_hip_host_alloc(mcl3dl_hip_ctx* ctx, size_t bytes, void** out);
int mcl3dl_hip_host_free(mcl3dl_hip_ctx* ctx, void* p);

/* Replaces: LidarMeasurementModelBeam::getBeamStatus (src/lidar_measurement_model_beam.cpp:157-192; used for the
 * debug markers at src/mcl_3dl.cpp:471-478) for n explicit rays. hit_index = map index of CastResult::point_
 * (-1 when the ray is exhausted). */
int mcl3dl_hip_beam_status(mcl3dl_hip_ctx* ctx, const float* begin_xyz /*n*3*/, const float* end_xyz /*n*3*/,
                           size_t n, int32_t* status /*n*/, int32_t* hit_index /*n or NULL*/);

/* Replaces: ChunkedKdtree::radiusSearch(p, radius, id, sqdist, 1) (include/mcl_3dl/chunked_kdtree.h:217-237) for n query
 * points (map frame) and ANY radius — besides the likelihood model the node searches with unmatch_output_dist
 * (matched / unmatched clouds, src/mcl_3dl.cpp:776-789) and with the global-localisation grid (:1058-1070).
 * out_index[i] = map index of the nearest point with d2 < (float)(radius*radius) in the dist_weight-rescaled metric, or -1;
 * out_sqdist[i] = that d2 (flann::L2_Simple float arithmetic), -1 when not found. Ties in d2 -> the lowest index. */
int mcl3dl_hip_radius_search(mcl3dl_hip_ctx* ctx, const float* query_xyz /*n*3*/, size_t n, float radius,
                             int32_t* out_index /*n*/, float* out_sqdist /*n or NULL*/);

/* Replaces: driving one RaycastUsingDDA by hand — setRay + getNextCastResult until the first collision
 * (include/mcl_3dl/raycast.h:45-77, include/mcl_3dl/raycasts/raycast_using_dda.h:66-159), the way the reference's
 * waypoint tests do (test/src/test_raycast_dda.cpp:157-183). Uses the caster of mcl3dl_hip_set_beam_params
 * (hit_range = hit_tolerance). out_xyz receives up to max_out voxel centres (CastResult::pos_); *n_visited may exceed
 * max_out. Introspection only: not on the per-update path. */
int mcl3dl_hip_dda_trace(mcl3dl_hip_ctx* ctx, const float* begin3, const float* end3, float* out_xyz /*max_out*3*/,
                         int max_out, int* n_visited, int* collided, int* hit_index);

/* ---- device entry points (asynchronous on the context's stream) ------------------------------------- */
/* Upload (and spatially order) the two filtered scans `pc_locals` of one update (src/mcl_3dl.cpp:377-383). */
int mcl3dl_hip_upload_scan(mcl3dl_hip_ctx* ctx, const float* scan_lik_xyz, size_t n_s, const float* scan_beam_xyz,
                           const uint32_t* scan_beam_origin, size_t n_b, const float* origins, size_t n_o);
/* The order in which the engine holds the likelihood scan it was last given (by mcl3dl_hip_upload_scan, _scan_finish or any
 * host-buffer call): position k holds the caller's point order[k] — a stable sort by the Morton key of 0.25 m cells, which is
 * what gives a wavefront neighbouring points. It matters to a caller in one case: with the option "strict_order" = 3 a
 * particle's likelihood is the reference's float recurrence `score_like += dist * match_weight`
 * (src/lidar_measurement_model_likelihood.cpp:124-134) over the scan IN THIS ORDER, i.e. bit for bit what
 * LidarMeasurementModelLikelihood::measure returns for the cloud { scan[order[0]], scan[order[1]], ... } — computed inside
 * the likelihood kernel, without the N_s x N_p term array and the replay pass that the caller's own order costs
 * ("strict_order" = 1 / 2). The order of a sampled cloud carries no information (PointCloudUniformSampler draws it,
 * include/mcl_3dl/point_cloud_random_samplers/point_cloud_uniform_sampler.h:56-74), but it does select which float
 * roundings happen: two orders of the same points differ by ~1e-6..1e-5 relative, like any two float summation orders. */
int mcl3dl_hip_scan_order(mcl3dl_hip_ctx* ctx, uint32_t* order /*n_s*/, size_t n_s /* must equal the installed scan's size */);
/* The same order computed on the host, without a context or a device (same keys, same stable sort): for a caller that wants
 * to HOLD its sampled cloud in the engine's order — a cloud permuted by `order` is left as it is by the engine (the sort is
 * stable and the keys depend only on the points and their bounding box), so "the caller's order" and "the engine's order"
 * coincide and strict_order = 3 is bit-identical to the reference's measure() on the caller's own cloud. The drop-in model
 * classes do exactly this in filter() when MCL3DL_HIP_ENGINE_ORDER=1. Returns 0, -3 on a null array. */
int mcl3dl_hip_scan_order_host(const float* scan_lik_xyz /*n_s*3*/, size_t n_s, uint32_t* order /*n_s*/);
/* measure_batch on device-resident poses against the uploaded scans; outputs are device arrays (may be NULL). */
int mcl3dl_hip_measure_device(mcl3dl_hip_ctx* ctx, const float* d_pose /*n_p*7*/, size_t n_p, float* d_lik,
                              float* d_match_ratio, float* d_beam);
/* pf::measure split for particle shards (one context per GPU, `world` shards, this one is `rank`):
 *   partial: w_new = w*((1*beam)*lik)*extra kept in the context; d_packed[2 + 2*world] (doubles) = { sum w_new,
 *            sum w_new*ln(w_new), then per rank r: max match_ratio, -min match_ratio } with only THIS rank's pair filled
 *            (the others 0)
 *   [one all-reduce(SUM) of d_packed across the shards — the update's only collective; nothing to do for world == 1]
 *   apply:   weights = w_new / packed[0] if packed[0] > 0, else untouched; d_stats4 = { entropy, match_ratio_min,
 *            match_ratio_max, restored } with entropy = ln S - T/S (== -sum p ln p). */
int mcl3dl_hip_pf_partial_device(mcl3dl_hip_ctx* ctx, const float* d_weight, const float* d_lik, const float* d_beam,
                                 const float* d_extra, const float* d_match_ratio, size_t n_p, int rank, int world,
                                 double* d_packed);
int mcl3dl_hip_pf_apply_device(mcl3dl_hip_ctx* ctx, float* d_weight_inout, size_t n_p, int world,
                               const double* d_packed, float* d_stats4);

/* ---- "next" row: the reductions right after the measurement update ---------------------- */
/* Replaces: pf::ParticleFilter::expectationBiased (include/mcl_3dl/pf.h:294-303, called at src/mcl_3dl.cpp:451) with
 * ParticleWeightedMeanQuat (include/mcl_3dl/state_6dof.h:316-355), and max() / maxBiased() (pf.h:361-390, :452).
 * bias NULL = probability_bias_ 1 (= expectation over all particles). out_mean7 = px,py,pz,qx,qy,qz,qw;
 * out_total = sum of weight*bias; the max indices are those of the FIRST maximum (strict < in the reference).
 * Per-particle products are the reference's float expressions; sums are fp64 trees (reference: float sequential). */
int mcl3dl_hip_expectation(mcl3dl_hip_ctx* ctx, const float* pose /*n*7*/, const float* weight /*n*/,
                           const float* bias /*n or NULL*/, size_t n, float* out_mean7, float* out_total,
                           int32_t* out_max_index, int32_t* out_max_biased_index);
/* Replaces: pf::ParticleFilter::covariance (pf.h:304-360, called at src/mcl_3dl.cpp:373,706) with
 * State6DOF::covElement (state_6dof.h:162-184) about `mean7` (the caller's expectation). subset = the particle indices
 * the caller's RNG drew for random_sample_ratio < 1 (pf.h:322-336), or NULL = all n particles. out_cov36 row-major. */
int mcl3dl_hip_covariance(mcl3dl_hip_ctx* ctx, const float* pose /*n*7*/, const float* weight /*n*/, size_t n,
                          const uint32_t* subset /*n_subset or NULL*/, size_t n_subset, const float* mean7,
                          float* out_cov36);
/* The same on device-resident poses / weights (results still returned to the host: they are a dozen scalars). */
int mcl3dl_hip_expectation_device(mcl3dl_hip_ctx* ctx, const float* d_pose, const float* d_weight, const float* d_bias,
                                  size_t n, float* out_mean7, float* out_total, int32_t* out_max_index,
                                  int32_t* out_max_biased_index);
int mcl3dl_hip_covariance_device(mcl3dl_hip_ctx* ctx, const float* d_pose, const float* d_weight, size_t n_particles,
                                 const uint32_t* d_subset, size_t n_subset, const float* mean7, float* out_cov36);

/* The same reductions over particle shards (multi-GPU hosts, mcl_3dl_amd/distributed.py:sharded_expectation /
 * sharded_covariance). *_partial_device leaves one record per shard in DEVICE memory:
 *   moments    16 doubles {sum w, sum w*pos[3], sum w*front[3], sum w*up[3], max w, its index in the shard,
 *              max w*bias, its index, 0, 0}      -> all-gather, then mcl3dl_hip_moments_finish (host arithmetic of
 *              ParticleWeightedMeanQuat::getMean, state_6dof.h:345-350; index_offset[r] = first particle of shard r)
 *   covariance 22 doubles {21 upper-triangular sums, sum w} -> all-reduce(SUM), then mcl3dl_hip_covariance_finish.
 * The *_finish functions are pure host functions (no context). */
int mcl3dl_hip_moments_partial_device(mcl3dl_hip_ctx* ctx, const float* d_pose, const float* d_weight,
                                      const float* d_bias /*or NULL*/, size_t n, double* d_out16);
int mcl3dl_hip_moments_finish(const double* parts16 /*world*16, rank order*/, int world,
                              const uint64_t* index_offset /*world or NULL*/, float* out_mean7, float* out_total,
                              int64_t* out_max_index, int64_t* out_max_biased_index);
int mcl3dl_hip_covariance_partial_device(mcl3dl_hip_ctx* ctx, const float* d_pose, const float* d_weight,
                                         size_t n_particles, const uint32_t* d_subset /*or NULL*/, size_t n_subset,
                                         const float* mean7 /*host*/, double* d_out22);
int mcl3dl_hip_covariance_finish(const double* sums22, float* out_cov36);

/* ---- "next" row: resampling ------------------------------------------------------------------ */
/* Replaces: pf::ParticleFilter::resample (include/mcl_3dl/pf.h:187-225, called at src/mcl_3dl.cpp:809) and
 * resizeParticle (pf.h:399-436, :880-884) on 13-dof states (State6DOF::operator[], state_6dof.h:80-149: pos, rot,
 * odom_err_integ_lin, odom_err_integ_ang). The caller keeps its random engine, so the sequence is three calls:
 *   begin : float prefix sums of the weights + the std::sort of the duplicate array (tie groups of weight-0 particles
 *           are ordered by libstdc++'s introsort exactly as in the reference); returns pstep = accum / n_out
 *   [resample only: the caller draws initial_p = uniform_real_distribution<float>(0, pstep)(engine_), pf.h:203]
 *   plan  : mode 0 = resample (pscan = pstep*i + initial_p), 1 = resizeParticle (pscan += pstep; initial_p ignored);
 *           n_out lower_bound searches on the device; out_source[i] = particle copied into slot i,
 *           out_duplicate[i] = 1 where the reference adds noise (it == it_prev, pf.h:217-221)
 *   [the caller draws one State6DOF::generateNoise per duplicated slot, in slot order, pf.h:216]
 *   apply : gather on the device; duplicated slots get state + noise (State6DOF::operator+, state_6dof.h:248-260)
 *           and normalize() (:150-153). New weights are 1/n_out (pf.h:207 / 419) — the caller's to set.
 * any out_* may be NULL. noise13 holds n_noise >= (number of duplicates) 13-float noise states. */
int mcl3dl_hip_resample_begin(mcl3dl_hip_ctx* ctx, const float* weight /*n*/, size_t n, size_t n_out, float* out_pstep);
int mcl3dl_hip_resample_plan(mcl3dl_hip_ctx* ctx, int mode, float initial_p, uint32_t* out_source /*n_out*/,
                             uint8_t* out_duplicate /*n_out*/, size_t* out_n_duplicates);
int mcl3dl_hip_resample_apply(mcl3dl_hip_ctx* ctx, const float* state13_in /*n*13*/, const float* noise13, size_t n_noise,
                              float* state13_out /*n_out*13*/);
int mcl3dl_hip_resample_apply_device(mcl3dl_hip_ctx* ctx, const float* d_state13_in, const float* noise13 /*host*/,
                                     size_t n_noise, float* d_state13_out);
/* Device-resident / sharded variants. begin_device takes the weights from device memory (they travel to the host: the
 * prefix sums are a sequential float recurrence). apply_slice_device fills only the output slots
 * [out_begin, out_begin + out_count) — a rank's own shard when particles are split over GPUs: every rank plans on the
 * all-gathered weights, gathers the states, and writes its slice (mcl_3dl_amd/distributed.py:sharded_resample).
 * d_state13_in holds all n input particles, d_state13_out out_count states, noise13 the noise of ALL duplicated slots. */
int mcl3dl_hip_resample_begin_device(mcl3dl_hip_ctx* ctx, const float* d_weight /*n*/, size_t n, size_t n_out,
                                     float* out_pstep);
int mcl3dl_hip_resample_apply_slice_device(mcl3dl_hip_ctx* ctx, const float* d_state13_in, const float* noise13 /*host*/,
                                           size_t n_noise, size_t out_begin, size_t out_count, float* d_state13_out);

/* One device-resident update in one call (single GPU): measure_device + pf_partial_device + pf_apply_device, i.e. the
 * statement pf_->measure(measure_func) of src/mcl_3dl.cpp:398-426 on device arrays. d_extra, d_lik, d_match_ratio,
 * d_beam may be NULL (no odometry factor / results kept internally); d_stats4 as in pf_apply_device.
 * Results are identical to the three separate calls. (Until ABI 2 an option "use_graph" replayed the sequence from a
 * captured hipGraph; on ROCm 7.2 / MI355X a replay cost a fixed 10-16 us against 3.3-3.8 us per plain launch and measured
 * slower at every size, so the option and its two introspection calls are gone in ABI 3.) */
int mcl3dl_hip_update_device(mcl3dl_hip_ctx* ctx, const float* d_pose, size_t n_p, float* d_weight_inout,
                             const float* d_extra, float* d_lik, float* d_match_ratio, float* d_beam, float* d_stats4);

/* ---- "next" row: scan preparation on the GPU ------------------------------------------------------
 * Replaces, for one measurement update: the pcl::VoxelGrid down-sampling of the accumulated cloud
 * (src/mcl_3dl.cpp:363-367), both models' clip filter (src/lidar_measurement_model_likelihood.cpp:79-103,
 * src/lidar_measurement_model_beam.cpp:98-122) and the sampler's gather
 * (include/mcl_3dl/point_cloud_random_samplers/point_cloud_uniform_sampler.h:56-74) — the random engine stays with the
 * caller, like resampling:
 *   begin  : cloud (robot frame; label = index of the accumulated cloud, src/mcl_3dl.cpp:295-298) -> VoxelGrid(leaf3;
 *            a component <= 0 or NULL skips it) -> pc_local_full; clipped copies for the likelihood and the beam model
 *            (clip = {clip_near, clip_far, clip_z_min, clip_z_max}; NULL = that model is not used). Returns the three
 *            sizes; the clouds stay on the device.
 *   [the caller draws n_s resp. n_b indices with std::uniform_int_distribution<size_t>(0, size - 1), sampler.h:66-71]
 *   finish : gathers the drawn points, orders them (the ordering mcl3dl_hip_upload_scan does on the host, same keys, same
 *            stable order) and installs them as the scans of the next mcl3dl_hip_measure_device / _update_device — no
 *            host round trip of the points. Results are bit-identical to uploading the same sampled clouds.
 * VoxelGrid follows pcl::VoxelGrid<PointXYZIL> as the node configures it (setLeafSize only): leaf index
 * int(floor(p * inv_leaf) - float(min_b)) . divb_mul, one centroid per occupied leaf in ascending leaf order, label = the
 * most frequent one (smallest on a tie); xyz are summed as floats in INPUT order (PCL's own std::sort leaves the order
 * inside a leaf unspecified; PCL is not vendored by the reference: parity at this boundary is against the restatement
 * in oracle/). Non-finite points are dropped by the filter. Intensity is not carried.
 * download: which = 0 pc_local_full, 1 / 2 clipped likelihood / beam cloud, 3 / 4 sampled clouds in the caller's index
 * order (xyz NULL: only *n). */
int mcl3dl_hip_scan_begin(mcl3dl_hip_ctx* ctx, const float* xyz /*n*3*/, const uint32_t* label /*n or NULL*/, size_t n,
                          const float* leaf3, const float* clip_lik4, const float* clip_beam4, size_t* n_full,
                          size_t* n_lik_clipped, size_t* n_beam_clipped);
/* The same from the wire format (mcl_3dl::fromROSMsg, include/mcl_3dl/point_conversion.h:64-92): little-endian
 * sensor_msgs/PointCloud2 data with FLOAT32 x / y / z at the given byte offsets and an optional UINT32 "label" field
 * (off_label < 0: none). label_override = 0 stamps every point with accumulated-cloud index 0 (what accumCloud does for
 * the first / only cloud), 0xffffffff keeps the message's labels.
 * LAYOUT REQUIRED by every *_pointcloud2 entry point: `data` is n_points * point_step bytes, i.e. rows without padding
 * (row_step == width * point_step), is_bigendian == false, and x / y / z (label) of datatype FLOAT32 (UINT32). The
 * reference's pcl::fromROSMsg also takes padded rows and big-endian data; a caller holding such a message checks it with
 * mcl3dl_hip_pointcloud2_layout_ok() below and repacks it (or passes plain arrays) when that returns 0 — the entry points
 * cannot see width / row_step / endianness and would decode shifted points without an error. */
static inline int mcl3dl_hip_pointcloud2_layout_ok(uint32_t width, uint32_t height, uint32_t point_step, uint32_t row_step,
                                                   int is_bigendian, int datatype_x, int datatype_y, int datatype_z,
                                                   int datatype_label /* -1: no label field */)
{
  /* sensor_msgs/PointField: FLOAT32 = 7, UINT32 = 6 */
  return !is_bigendian && (height <= 1 || row_step == width * point_step) && datatype_x == 7 && datatype_y == 7 &&
         datatype_z == 7 && (datatype_label < 0 || datatype_label == 6);
}
int mcl3dl_hip_scan_begin_pointcloud2(mcl3dl_hip_ctx* ctx, const uint8_t* data, size_t n_points, uint32_t point_step,
                                      int off_x, int off_y, int off_z, int off_label, uint32_t label_override,
                                      const float* leaf3, const float* clip_lik4, const float* clip_beam4, size_t* n_full,
                                      size_t* n_lik_clipped, size_t* n_beam_clipped);
int mcl3dl_hip_scan_finish(mcl3dl_hip_ctx* ctx, const uint32_t* idx_lik /*n_s*/, size_t n_s,
                           const uint32_t* idx_beam /*n_b*/, size_t n_b, const float* origins /*n_o*3*/, size_t n_o);
int mcl3dl_hip_scan_download(mcl3dl_hip_ctx* ctx, int which, float* xyz, uint32_t* label, size_t capacity, size_t* n);
/* Introspection: the stable radix sort the cloud path runs on the device (VoxelGrid leaf order, scan ordering; replaces the
 * std::sort of pcl::VoxelGrid and nothing else of the reference), exposed so that it can be checked on its own:
 * (keys, vals) sorted ascending by key bits [0, end_bit), equal keys in input order. One launch up to 16 384 pairs. */
int mcl3dl_hip_sort_pairs(mcl3dl_hip_ctx* ctx, const uint32_t* keys, const uint32_t* vals /*NULL: 0..n-1*/, size_t n,
                          int end_bit, uint32_t* out_keys, uint32_t* out_vals);

/* ---- "next" row: map from the wire format, map updates, matched / unmatched output --------------
 * set_map_*: replaces cbMapcloud + loadMapCloud (src/mcl_3dl.cpp:128-139, 1140-1158): decode, VoxelGrid(map_downsample),
 * then as mcl3dl_hip_set_map. *n_map = points kept.
 * map_update*: replaces cbMapcloudUpdate + cbMapUpdateTimer (src/mcl_3dl.cpp:141-153, 1355-1369): pc_map2 = pc_map +
 * VoxelGrid(update, update_downsample); a later update REPLACES the earlier one (n = 0 removes it). The candidate-voxel
 * index is not rebuilt: only the bricks within reach of a removed or an added point are compiled again (from the points
 * that can reach them) and installed over their old records — results identical to a fresh index of the merged map.
 * stats6 (may be NULL) = {bricks re-compiled, bricks added, points that took part, device milliseconds, overflow
 * records appended, outcome}; outcome 0 = incremental update done; otherwise the index is rebuilt as a whole on next use:
 * 1 nothing was built yet, 2 lik_index != 2, 3 the index was built for another point count, 4 a point outside the grid
 * the index was laid out for, 5 brick / overflow budget exhausted; 6 = no brick touched (nothing to do).
 * map_download: the map as the engine holds it (base, then update). */
int mcl3dl_hip_set_map_pointcloud2(mcl3dl_hip_ctx* ctx, const uint8_t* data, size_t n_points, uint32_t point_step, int off_x,
                                   int off_y, int off_z, int off_label, const float* leaf3, uint64_t stamp,
                                   const float* dist_weight, size_t* n_map);
int mcl3dl_hip_set_map_downsampled(mcl3dl_hip_ctx* ctx, const float* xyz, const uint32_t* label, size_t n,
                                   const float* leaf3, uint64_t stamp, const float* dist_weight, size_t* n_map);
int mcl3dl_hip_map_update(mcl3dl_hip_ctx* ctx, const float* xyz, const uint32_t* label, size_t n, const float* leaf3,
                          uint64_t stamp, size_t* n_map, double* stats6);
int mcl3dl_hip_map_update_pointcloud2(mcl3dl_hip_ctx* ctx, const uint8_t* data, size_t n_points, uint32_t point_step,
                                      int off_x, int off_y, int off_z, int off_label, const float* leaf3, uint64_t stamp,
                                      size_t* n_map, double* stats6);
int mcl3dl_hip_map_download(mcl3dl_hip_ctx* ctx, float* xyz, uint32_t* label, size_t capacity, size_t* n);
/* Replaces the matched / unmatched classification of src/mcl_3dl.cpp:761-805 as ONE device pass: every point of
 * pc_local_full (xyz NULL: the cloud mcl3dl_hip_scan_begin left on the device; otherwise n explicit robot-frame points)
 * is transformed by the pose (State6DOF::transform), searched with radiusSearch(p, unmatch_dist, ., ., 1); no neighbour ->
 * unmatched; sqdist < match_dist^2 (compared in double, :778-786) -> matched. Outputs are the transformed points in input
 * order (NULL: counts only). */
int mcl3dl_hip_match_split(mcl3dl_hip_ctx* ctx, const float* pose7, const float* xyz, size_t n, float unmatch_dist,
                           double match_dist, float* out_matched_xyz, size_t cap_matched, size_t* n_matched,
                           float* out_unmatched_xyz, size_t cap_unmatched, size_t* n_unmatched);

/* ---- device groups: N GPUs behind one handle, one host process ----------------------------------
 * The reference node is ONE C++ process (src/mcl_3dl.cpp:1466); a group lets that process use every GPU of the node with
 * src/mcl_3dl.cpp untouched: the drop-in model classes call the group_* forms of the entry points above.
 * A group owns one context per listed device and one worker thread per device. Particles are split into contiguous
 * shards (mcl3dl_hip_group_shard: sizes differ by at most one), map / parameters / scan are replicated on every device,
 * and one update needs exactly ONE collective — the all-reduce(sum) of the 2 + 2N doubles described at
 * mcl3dl_hip_pf_partial_device:
 *   "collective" 0 (default): ncclAllReduce on each device's stream (RCCL over xGMI; librccl is dlopen'ed the first time
 *                a group of more than one device runs an update — MCL3DL_HIP_RCCL_LIB overrides the library name)
 *   "collective" 1 (or environment MCL3DL_HIP_COLLECTIVE=host): the N records are brought to the host, summed in rank
 *                order and sent back — for several contexts on ONE GPU (RCCL needs one GPU per rank) and as a fallback.
 * With one device every group_* call is the plain call on its context (bit-identical results, no thread, no RCCL);
 * option "direct_single" = 0 sends a one-device group through the sharded path as well (tests: RCCL with one rank).
 * A group is not thread-safe (one caller at a time), like a context. */
typedef struct mcl3dl_hip_group mcl3dl_hip_group;
int mcl3dl_hip_group_create(mcl3dl_hip_group** out, const int* device_ids, int n_devices /* 1..64 */);
void mcl3dl_hip_group_destroy(mcl3dl_hip_group* g);
const char* mcl3dl_hip_group_last_error(const mcl3dl_hip_group* g);
int mcl3dl_hip_group_size(const mcl3dl_hip_group* g);
/* The context of one rank (options, introspection, per-device queries such as mcl3dl_hip_beam_status). */
mcl3dl_hip_ctx* mcl3dl_hip_group_context(mcl3dl_hip_group* g, int rank);
/* Shard bookkeeping (pure host function): particles [*begin, *begin + *count) belong to `rank`. */
int mcl3dl_hip_group_shard(size_t n_p, int n_devices, int rank, size_t* begin, size_t* count);
/* Broadcast forms of mcl3dl_hip_set_map / _set_likelihood_params / _set_beam_params / _set_option
 * ("collective" is the group's own option, everything else goes to every context). */
int mcl3dl_hip_group_set_map(mcl3dl_hip_group* g, const float* xyz, const uint32_t* label, size_t n_m, uint64_t stamp,
                             const float* dist_weight);
int mcl3dl_hip_group_set_likelihood_params(mcl3dl_hip_group* g, float match_dist_min, float match_dist_flat,
                                           float match_weight);
int mcl3dl_hip_group_set_beam_params(mcl3dl_hip_group* g, float map_grid_x, float map_grid_y, float map_grid_z,
                                     float dda_grid_size, float ray_angle_half, float hit_range,
                                     float beam_likelihood_min, uint32_t num_points, float ang_total_ref,
                                     uint32_t filter_label_max, int add_penalty_short_only_mode);
int mcl3dl_hip_group_set_option(mcl3dl_hip_group* g, const char* name, double value);
/* Sharded forms of mcl3dl_hip_upload_poses / _measure_batch / _measure_update: same arguments, same results; host arrays
 * are scattered to / gathered from the shards. measure_batch needs no collective; measure_update runs the one all-reduce. */
int mcl3dl_hip_group_upload_poses(mcl3dl_hip_group* g, const float* pose /*n_p*7*/, size_t n_p);
int mcl3dl_hip_group_measure_batch(mcl3dl_hip_group* g, const float* pose /*n_p*7 or NULL*/, size_t n_p,
                                   const float* scan_lik_xyz, size_t n_s, const float* scan_beam_xyz,
                                   const uint32_t* scan_beam_origin, size_t n_b, const float* origins, size_t n_o,
                                   float* out_lik, float* out_match_ratio, float* out_beam);
/* mcl3dl_hip_measure_batch_begin / _wait / _end for a group: one device delivers slice by slice; a sharded group evaluates
 * the whole batch inside _begin and _wait reports every particle ready. */
int mcl3dl_hip_group_measure_batch_begin(mcl3dl_hip_group* g, const float* pose /*n_p*7 or NULL*/, size_t n_p,
                                         const float* scan_lik_xyz, size_t n_s, const float* scan_beam_xyz,
                                         const uint32_t* scan_beam_origin, size_t n_b, const float* origins, size_t n_o,
                                         float* out_lik, float* out_match_ratio, float* out_beam, size_t slice_particles);
int mcl3dl_hip_group_measure_batch_wait(mcl3dl_hip_group* g, size_t particle, size_t* n_ready);
int mcl3dl_hip_group_measure_batch_end(mcl3dl_hip_group* g);
int mcl3dl_hip_group_measure_update(mcl3dl_hip_group* g, const float* pose, const float* extra, float* weight_inout,
                                    size_t n_p, const float* scan_lik_xyz, size_t n_s, const float* scan_beam_xyz,
                                    const uint32_t* scan_beam_origin, size_t n_b, const float* origins, size_t n_o,
                                    float* out_lik, float* out_match_ratio, float* out_beam, float* entropy,
                                    float* match_ratio_min, float* match_ratio_max, int* restored);
/* ---- the group's particles RESIDENT on its GPUs: a whole filter iteration without a per-update pose upload ----------------
 * Replaces, for N GPUs, what the node does with pf_ around the measurement (src/mcl_3dl.cpp:398-452, 706-709, 809):
 *   upload_state      the particle vector (include/mcl_3dl/pf.h:457) as 13 floats per particle {pos 3, rot 4 (x,y,z,w),
 *                     odom_err_integ lin 3, ang 3} (State6DOF, state_6dof.h:56-66) + probabilities (NULL: 1 / n_p each),
 *                     scattered into contiguous shards (mcl3dl_hip_group_shard); they stay on the devices until the next
 *                     upload. The motion model (prediction) is the caller's: download, predict, upload — or keep the states
 *                     where they are when nothing moved.
 *   update_resident   pf::measure (pf.h:252-279 with the lambda of src/mcl_3dl.cpp:402-425) over the resident particles:
 *                     only the scan (and the odometry factor `extra`, n_p floats or NULL) goes up, one all-reduce of
 *                     2 + 2N doubles, four scalars come back; out_weight / out_lik / out_match_ratio / out_beam (each may be
 *                     NULL) fetch per-particle results. Same results as mcl3dl_hip_group_measure_update on the same particles.
 *   expectation       pf::expectationBiased + max + maxBiased (pf.h:294-303, 361-390): one 16-double record per shard,
 *                     combined on the host (mcl3dl_hip_moments_finish); bias = probability_bias_ per particle or NULL.
 *   covariance        pf::covariance about mean7 (pf.h:304-360, all particles): 22 sums per shard, added in rank order.
 *   resample_begin    accum_probability_ (pf.h:193-197 / 401-405): the float recurrence is sequential over ALL particles, so
 *                     every weight comes to the host once (4 B per particle) and the prefixes go to every device; n_out = 0
 *                     keeps the particle count (resample), another value is resizeParticle (pf.h:399-436). *out_pstep for
 *                     the caller's uniform draw (pf.h:203).
 *   resample_plan     mode / initial_p / outputs as mcl3dl_hip_resample_plan (every rank plans all slots: n_out searches).
 *   resample_apply    all-gather of the 13-float states (ncclAllGather over xGMI; through the host with "collective" 1), every
 *                     rank writes ITS shard of the new generation (duplicates get noise13[slot], State6DOF::operator+ and
 *                     normalize(): pf.h:214-218), weights 1 / n_out (pf.h:207), poses refreshed. The particles are resident
 *                     again, n_out of them.
 * With one device and "direct_single" 1 none of this touches RCCL. */
int mcl3dl_hip_group_upload_state(mcl3dl_hip_group* g, const float* state13 /*n_p*13*/, const float* weight /*n_p or NULL*/,
                                  size_t n_p);
int mcl3dl_hip_group_download_state(mcl3dl_hip_group* g, float* state13 /*n_p*13 or NULL*/, float* weight /*n_p or NULL*/,
                                    size_t n_p);
size_t mcl3dl_hip_group_resident(const mcl3dl_hip_group* g);
int mcl3dl_hip_group_update_resident(mcl3dl_hip_group* g, const float* extra /*n_p or NULL*/, const float* scan_lik_xyz,
                                     size_t n_s, const float* scan_beam_xyz, const uint32_t* scan_beam_origin, size_t n_b,
                                     const float* origins, size_t n_o, float* out_weight, float* out_lik,
                                     float* out_match_ratio, float* out_beam, float* entropy, float* match_ratio_min,
                                     float* match_ratio_max, int* restored);
int mcl3dl_hip_group_expectation(mcl3dl_hip_group* g, const float* bias /*n_p or NULL*/, float* out_mean7, float* out_total,
                                 int64_t* out_max_index, int64_t* out_max_biased_index);
int mcl3dl_hip_group_covariance(mcl3dl_hip_group* g, const float* mean7, float* out_cov36);
int mcl3dl_hip_group_resample_begin(mcl3dl_hip_group* g, size_t n_out /*0 = as many as there are*/, float* out_pstep);
int mcl3dl_hip_group_resample_plan(mcl3dl_hip_group* g, int mode, float initial_p, uint32_t* out_source /*n_out or NULL*/,
                                   uint8_t* out_duplicate /*n_out or NULL*/, size_t* out_n_duplicates);
int mcl3dl_hip_group_resample_apply(mcl3dl_hip_group* g, const float* noise13 /*n_dup*13, host*/, size_t n_noise);
/* How many updates went through each kind of collective so far. */
int mcl3dl_hip_group_collective_stats(const mcl3dl_hip_group* g, uint64_t* rccl_all_reduces, uint64_t* host_combines);

/* ---- measurement support ------------------------------------------------------------------------------ */
/* Per-kernel hipEvent timing on the launch stream (off by default). */
int mcl3dl_hip_set_kernel_timing(mcl3dl_hip_ctx* ctx, int enable);
int mcl3dl_hip_get_kernel_time(mcl3dl_hip_ctx* ctx, int kernel_id, double* total_ms, uint64_t* launches);
int mcl3dl_hip_reset_kernel_time(mcl3dl_hip_ctx* ctx);
/* Exact workload counts of the last measure (re-runs the kernels in counting mode; not for timed regions):
 * stats[0] = sum over (particle, point) of K = map points in the 27-cell neighbourhood, [1] = evaluations,
 * [2] = DDA voxel steps, [3] = occupied voxels visited, [4] = map points tested, [5] = rays. */
int mcl3dl_hip_workload_stats(mcl3dl_hip_ctx* ctx, const float* d_pose, size_t n_p, double* stats6);
/* Sizes of the device-resident structures (bytes): [0] cell-grid points, [1] cell-grid index, [2] DDA occupancy bitmap,
 * [3] DDA voxel index, [4] DDA points, [5] candidate-voxel brick table, [6] candidate-voxel run delimiters,
 * [7] candidate points (with lik_index 2: [6] = the 64-byte voxel records, [7] = their overflow records). Structures that
 * were never needed are 0. */
int mcl3dl_hip_memory_footprint(mcl3dl_hip_ctx* ctx, uint64_t* bytes8);
/* Options (no reference counterpart). Round 6 pruned the switchboard: what lost its A/B is gone (the two-launch pf::measure tail,
 * the one-launch sorts, the completion word folded into the last kernel, the device-side resampling prefix, the CSR form of the
 * candidate index, sharper pruning of crowded voxels, the replay's A/B knobs, the dense record grid) and the thresholds nobody
 * needs to move are constants (HISTORY.md R6-7). The 39 keys left are ALL drawn, in combination, by tests/test_gpu_api_fuzz.py
 * (tests/test_abi.py keeps that pool and this list in step). Except for "strict_order" and its thresholds — which select HOW the
 * sums are rounded — results are the same bits for every setting. Every key can be read back with mcl3dl_hip_get_option.
 *
 * -- how the sums are added up
 *   "strict_order"      2 (default) = the likelihood terms are added up as the reference adds them — float, sequentially, in the
 *                       order the CALLER's cloud holds its points (likelihood.cpp:120-134) — for every scan of at most
 *                       "strict_exact_max" points (4096: the reference's operating range and far beyond) and for every scan of
 *                       at least "strict_auto_min" points (28 147: from there on three standard deviations of the reference's own
 *                       rounding leave north_star's 1e-5; host_context.h derives it): likelihoods are then the reference's bit for
 *                       bit; so are match ratios and beam scores always, and the normalised weights up to 1024 particles (pf::measure's
 *                       float sum, pf.h:255-260). In between the same float terms are summed in a fixed-order fp64 tree (within
 *                       the reference's rounding of its own float: 5e-6 at 16 384 points). Wherever one work-group owns a
 *                       particle's whole scan (below ~1000 points, or few particles: up to 12 288 points) the exact sum costs no
 *                       memory and no launch — the terms wait in LDS at their original indices and a wavefront runs the
 *                       recurrence 512 terms a pass (float_chain.h); elsewhere it costs an n_s x n_p float buffer and a replay
 *                       launch. When that buffer would take more than half of the free device memory (or "strict_auto_max_bytes",
 *                       or its allocation fails) the launch sums in fp64 instead — an update never fails over it; read-only
 *                       "strict_auto_skipped" counts such launches.
 *                       0 = fp64 tree sums always (fastest; ~1e-6, at worst n_s x 6e-8, from the reference).
 *                       1 = exact always, weights at every particle count too (single GPU); fails loudly when the buffer cannot be had.
 *                       3 = the float recurrence INSIDE the tiled likelihood kernel, at every scan size, over the scan in the
 *                       ENGINE's order (mcl3dl_hip_scan_order): bit-identical to the reference's measure() on the scan permuted
 *                       that way; no term buffer, no replay pass. Suits callers whose scan order means nothing to them (a
 *                       sampled cloud): the drop-in classes with MCL3DL_HIP_ENGINE_ORDER=1.
 *   "strict_exact_max", "strict_auto_min", "strict_auto_max_bytes"   the thresholds above (0 = no cap for the last)
 *   "strict_chunk"      0 (default) = a replay keeps the terms of the whole scan; a point count >= 1024 = scans of at least two such
 *                       chunks are ordered chunk by chunk of the caller's order and replayed with two term buffers (C5: 4.3 GB
 *                       instead of 17 GB, 27.9 instead of 25.6 ms). Same bits. Read-only "scan_chunk_in_use".
 *                       (Mode 3 launches of one DEVICE are serialised across contexts — an event per device —: the hand-off between
 *                       tile work-groups relies on in-order dispatch inside one launch, and two such kernels side by side could starve
 *                       each other's producers. The poll is bounded all the same: ~1 s, then error -2 for that update.)
 *   "scan_presorted"    1 = the caller holds its likelihood scans in the engine's order already (mcl3dl_hip_scan_order_host): no
 *                       ordering launches, the permutation is the identity, and the exact modes follow the caller's own order.
 *   "chain_ppl"         strict_order 3: scan tiles per work-group (0 = four for few particles on long scans, 1, 4). Same bits.
 * -- which kernel runs (same bits)
 *   "lik_index"         2 (default) = candidate-voxel records (map_compiler.h); 0 = 27-cell scan of the cell-sorted map
 *   "lik_tiled", "lik_tiled_min", "lik_group"   tile-major XCD-aware kernel for scans >= lik_tiled_min (1024) points and >= 4 particles;
 *                       particles per work-group 0 (= 16 / 8 / 4 by launch size), 4, 8, 16, 32
 *   "lik_small"         1 = scans of <= 32 points with >= 256 particles share each wavefront between several particles
 *   "lik_coop"          1 = quad-cooperative record fetch + VALU-trimmed evaluation in the tiled kernel
 *   "lik_defer"         1 = overflow rounds queued per wavefront and run densely (packed 64-byte records); 0 = at once; 2 = only on
 *                       maps where enough voxels overflow. Read-only "lik_defer_active"
 *   "update_small", "update_small_max"   the whole update as ONE launch up to that many particles (512) where the per-particle
 *                       kernels serve the scans; "update_small_conformant" = its arrival tickets as acq_rel RMWs (the memory
 *                       model's form; slower; a mitigation switch, tests/test_gpu_soak.py runs both)
 *   "pf_fused"          1 = pf::measure as ONE work-group up to 1024 particles; 0 = partial + reduce + apply
 *   "overlap_models"    1 = the two models of a large update run side by side: in ONE launch whose work-groups interleave the
 *                       tiled likelihood kernel's and the beam kernel's (fp64-tree sums, >= 64 beam work-groups), else on two
 *                       streams (from 262 144 rays); 0 = behind each other. Same bits either way
 *   "beam_prepare"      1 = launches of >= 32 768 rays take per-(particle, origin) constants from a small kernel
 * -- the likelihood index (same bits)
 *   "cand_voxel_ratio"  voxel edge / match_dist_min; 0 (default) = per map: 0.5, or 0.36 on maps of voxel-filter centroids
 *   "cand_phase"        grid origin phase in voxels, [0, 1) (0.5)
 *   "cand_record_parts" inline candidates per record: 0 (default) / 4 = 64-byte records, 8 = 128-byte records
 *   "cand_packed", "cand_bound"   packed w words / skip bounds in the records when the map allows them
 *   "cand_aniso"        0 = cubes, 1 = boxes that follow the dist_weight (edge of axis a = base edge x min(w_a / w_min,
 *                       "cand_aniso_max" = 8)), 2 (default) = boxes when cubes exceed the budget
 *   "index_budget_bytes"  upper bound of the records: -1 (default) = a quarter of the device's memory, 0 = none; cubes, then boxes,
 *                       then coarser voxels (<= 1.5 r), then a clean error naming the bytes needed. Read-only "index_note" says
 *                       when the index differs from what was asked for; "index_record_bytes", "cand_edge_ratio_x/_y/_z"
 *   "cand_prune_coop"   1 = the compiler's pruning pass runs 16 lanes per voxel; 0 = one thread per voxel (the form it is checked against)
 *   "grid_build_host"   1 = cell grid and DDA grid built by sequential counting sorts on the host (the form the device builders are
 *                       checked against)
 *   "dda_overlay"       1 = a map update rides on the DDA grid as an overlay instead of forcing a rebuild
 * -- host side
 *   "update_stage", "update_zero_copy"   host-buffer updates: scans / poses / weights taken over by ONE launch that also orders the
 *                       scans; read in place from page-locked memory, results written there by the update's last kernel
 *   "poll_sync", "poll_spin_us"   completion by a polled page-locked word (2 = every synchronisation, 1 = host-buffer updates, 0 =
 *                       hipStreamSynchronize); spin that long (2000 us) before napping between looks
 *   "scan_order_device" scans of at least this many points are ordered on the device (4096; 0 = always on the host)
 *   "batch_slice"       particles per slice of mcl3dl_hip_measure_batch_begin when slice_particles is 0
 *   "timing_mask"       bit k set = kernel group k is timed while kernel timing is on
 *   "test_late_structures"  fault injection for the API-sequence fuzz only */
int mcl3dl_hip_set_option(mcl3dl_hip_ctx* ctx, const char* name, double value);
int mcl3dl_hip_get_option(mcl3dl_hip_ctx* ctx, const char* name, double* value);
/* Candidate-voxel index of the current map: [0] bricks, [1] preliminary candidates, [2] candidates kept,
 * [3] device build time in ms, [4] voxels with at least one candidate, [5] voxels with more candidates than their
 * 64-byte record holds (4), [6] overflow records, [7] voxel edge / match_dist_min in use ([4]..[7]: lik_index 2). */
int mcl3dl_hip_index_stats(mcl3dl_hip_ctx* ctx, double* stats8);

#ifdef __cplusplus
}
#endif
#endif /* MCL3DL_HIP_H */
