/* oracle/mcl3dl_oracle.h — TEST INFRASTRUCTURE ONLY (never linked into, imported by or executed from the product).
 *
 * Plain-C restatement of the reference's LiDAR measurement-update hot path (at-wat/mcl_3dl v0.7.0).
 * Same C ABI as oracle/ref_harness.cpp (which wraps the real reference sources) but prefix orc_,
 * so oracle/pyoracle.py drives both with one wrapper and tests can diff them call by call.
 *
 * Parity status: PINNED for everything that lives under /root/reference — checked against the
 * reference's own golden vectors (test/src/test_raycast_dda.cpp:185-286, test_chunked_kdtree.cpp:38-88,
 * test_quat.cpp:234-273, test_pf.cpp:330-391) and, call by call, against oracle/_ref (the reference
 * sources themselves, compiled here).  UNPINNED only at the PCL/FLANN boundary
 * (pcl::KdTreeFLANN::radiusSearch is third-party, absent from /root/reference, version unpinned:
 * CMakeLists.txt:26-31): both oracles define it as the exact nearest neighbour (eps = 0) in the
 * dist_weight-rescaled metric, see oracle/shims/pcl/kdtree/kdtree_flann.h.
 */
#ifndef MCL3DL_ORACLE_H
#define MCL3DL_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

void* orc_create(float chunk_length, float max_search_radius);
void orc_destroy(void* h);
int orc_max_threads(void);
void orc_set_map(void* h, const float* xyz, const uint32_t* label, size_t n, uint64_t stamp, const float* dist_weight,
                 float epsilon);
void orc_set_likelihood_params(void* h, float match_dist_min, float match_dist_flat, float match_weight,
                               uint32_t num_points, uint32_t num_points_global, float clip_near, float clip_far,
                               float clip_z_min, float clip_z_max);
void orc_set_beam_params(void* h, float map_grid_x, float map_grid_y, float map_grid_z, float dda_grid_size,
                         float ray_angle_half, float hit_range, float beam_likelihood_min, uint32_t num_points,
                         uint32_t num_points_global, float ang_total_ref, uint32_t filter_label_max,
                         int add_penalty_short_only_mode, int use_raycast_using_dda, float clip_near, float clip_far,
                         float clip_z_min, float clip_z_max);
void orc_radius_search(void* h, const float* q_xyz, size_t n, float radius, int* found, int* id, float* sqdist);
void orc_transform(const float* pose7, const float* xyz_in, size_t n, float* xyz_out);
void orc_quat_rotate(const float* q4, const float* v3, float* out3);
double orc_likelihood_measure(void* h, const float* poses, size_t n_p, const float* scan_xyz,
                              const uint32_t* scan_label, size_t n_s, float* out_lik, float* out_quality, int threads);
double orc_beam_measure(void* h, const float* poses, size_t n_p, const float* scan_xyz, const uint32_t* scan_label,
                        size_t n_b, const float* origins_xyz, size_t n_o, float* out_lik, float* out_quality,
                        int threads);
void orc_beam_status(void* h, const float* begin_xyz, const float* end_xyz, size_t n, int* status, int* hit_index);
int orc_dda_waypoints(void* h, double map_grid_x, double map_grid_y, double map_grid_z, double dda_grid_size,
                      double ray_angle_half, double hit_tolerance, const float* begin3, const float* end3,
                      float* out_xyz, int max_out, int* collided, int* hit_index, int stop_at_collision);
int orc_pf_measure(float* weight_inout, const float* likelihood, size_t n, float* entropy);
double orc_measure_update(void* h, const float* poses, const float* odom_err_integ_lin, float* weight_inout,
                          size_t n_p, const float* scan_lik_xyz, size_t n_s, const float* scan_beam_xyz,
                          const uint32_t* scan_beam_label, size_t n_b, const float* origins_xyz, size_t n_o,
                          float odom_err_integ_lin_sigma, float* out_lik, float* out_beam, float* out_quality,
                          float* entropy, float* match_ratio_min_out, float* match_ratio_max_out, int* restored_out);

void orc_expectation(const float* poses, const float* weights, const float* bias, size_t n, float* out_mean7,
                     int* out_max_index, int* out_max_biased_index);
void orc_covariance(const float* poses, const float* weights, size_t n, float* out_cov36, float* out_mean7);

float orc_resample_pstep(const float* weight, size_t n, size_t n_out);
void orc_resample_plan(const float* weight, size_t n, size_t n_out, int mode, float initial_p, uint32_t* out_source,
                       uint8_t* out_dup);
void orc_resample_apply(const float* state13_in, const uint32_t* source, const uint8_t* dup, const float* noise13,
                        size_t n_out, float* state13_out);

/* Extra, oracle-only: exact workload statistics used by bench.py / DESIGN.md for the algorithmic-bytes
 * accounting: K = map points in the 3x3x3 cell neighbourhood (cell edge = match_dist_min
 * in the weighted metric) of each transformed scan point, summed over a batch of poses. */
void orc_count_neighbourhood(void* h, const float* poses, size_t n_p, const float* scan_xyz, size_t n_s,
                             double* sum_k, double* sum_found);
/* DDA step statistics for a batch (voxel steps S, occupied voxels visited O, points tested T), SURVEY.md §8d. */
void orc_count_dda(void* h, const float* poses, size_t n_p, const float* scan_xyz, const uint32_t* scan_label,
                   size_t n_b, const float* origins_xyz, size_t n_o, double* steps, double* occupied, double* tested);

#ifdef __cplusplus
}
#endif
#endif
