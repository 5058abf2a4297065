/* tests/cpp/group_all_devices.c — TEST INFRASTRUCTURE, plain C99 against include/mcl3dl_hip.h only: one device group over
 * EVERY GPU this process sees (mcl3dl_hip_device_count), the particles sharded over them, the update's single collective
 * through RCCL (ncclCommInitAll + ncclAllReduce with one rank per GPU) — so the moment `pytest -m gpu` runs on a multi-GPU
 * box this is an N-rank RCCL test of the in-process path the reference's single process would use (src/mcl_3dl.cpp:1466).
 * With one GPU it still takes the sharded path (option direct_single = 0): RCCL with one rank.
 * Check: the group's update against ONE context over all particles — likelihood / match ratio / beam score equal bit for
 * bit, weights within 2e-7 relative (the fp64 sum is associated per shard), entropy within 1e-6. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mcl3dl_hip.h"

static unsigned long long rng_state = 88172645463325252ull;
static float uniform01(void)
{
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return (float)((rng_state >> 11) * (1.0 / 9007199254740992.0));
}

int main(int argc, char** argv)
{
  const int n_dev = mcl3dl_hip_device_count();
  if (n_dev < 1)
  {
    fprintf(stderr, "no HIP device\n");
    return 2;
  }
  const size_t n_p = argc > 1 ? (size_t)atol(argv[1]) : 3001, n_s = 700, n_b = 24, side = 60;
  /* map: a hollow box of points, 0.1 m spacing (the scene family of test/src/test_expansion_resetting.cpp:78-91) */
  const size_t n_m = 6 * side * side;
  float* map = (float*)malloc(sizeof(float) * 3 * n_m);
  size_t k = 0;
  const float half = 0.05f * side;
  for (int f = 0; f < 6; ++f)
    for (size_t i = 0; i < side; ++i)
      for (size_t j = 0; j < side; ++j)
      {
        const float u = (i + 0.5f) * 0.1f - half, v = (j + 0.5f) * 0.1f - half, w = (f & 1) ? half : -half;
        float* p = map + 3 * k++;
        p[f / 2] = w;
        p[(f / 2 + 1) % 3] = u;
        p[(f / 2 + 2) % 3] = v;
      }
  /* scan: points of the walls seen from the origin, with noise; beam points on the +x wall; particles around the origin */
  float* scan = (float*)malloc(sizeof(float) * 3 * n_s);
  for (size_t i = 0; i < n_s; ++i)
  {
    const float* m = map + 3 * (size_t)(uniform01() * (n_m - 1));
    for (int a = 0; a < 3; ++a)
      scan[3 * i + a] = m[a] + 0.01f * (uniform01() - 0.5f);
  }
  float* beam = (float*)malloc(sizeof(float) * 3 * n_b);
  uint32_t* beam_org = (uint32_t*)calloc(n_b, sizeof(uint32_t));
  for (size_t i = 0; i < n_b; ++i)
  {
    beam[3 * i] = half - 0.3f * uniform01();
    beam[3 * i + 1] = 2.0f * (uniform01() - 0.5f);
    beam[3 * i + 2] = 1.0f * (uniform01() - 0.5f);
  }
  const float origins[3] = { 0.f, 0.f, 0.2f };
  float* pose = (float*)malloc(sizeof(float) * 7 * n_p);
  float* w0 = (float*)malloc(sizeof(float) * n_p);
  for (size_t i = 0; i < n_p; ++i)
  {
    float* q = pose + 7 * i;
    q[0] = 0.3f * (uniform01() - 0.5f);
    q[1] = 0.3f * (uniform01() - 0.5f);
    q[2] = 0.1f * (uniform01() - 0.5f);
    const float yaw = 0.2f * (uniform01() - 0.5f);
    q[3] = 0.f;
    q[4] = 0.f;
    q[5] = sinf(0.5f * yaw);
    q[6] = cosf(0.5f * yaw);
    w0[i] = 1.0f / (float)n_p;
  }
  int* ids = (int*)malloc(sizeof(int) * n_dev);
  for (int d = 0; d < n_dev; ++d)
    ids[d] = d;

  float *lik[2], *ratio[2], *bm[2], *w[2], ent[2], rmin[2], rmax[2];
  int restored[2];
  for (int v = 0; v < 2; ++v)
  {
    lik[v] = (float*)malloc(sizeof(float) * n_p);
    ratio[v] = (float*)malloc(sizeof(float) * n_p);
    bm[v] = (float*)malloc(sizeof(float) * n_p);
    w[v] = (float*)malloc(sizeof(float) * n_p);
    memcpy(w[v], w0, sizeof(float) * n_p);
  }
  /* v = 0: one context, all particles */
  mcl3dl_hip_ctx* ctx = NULL;
  if (mcl3dl_hip_create(&ctx, 0) != 0)
    return 3;
  int rc = mcl3dl_hip_set_map(ctx, map, NULL, n_m, 1, NULL);
  rc = rc ? rc : mcl3dl_hip_set_likelihood_params(ctx, 0.2f, 0.05f, 5.0f);
  rc = rc ? rc : mcl3dl_hip_set_beam_params(ctx, 0.1f, 0.1f, 0.1f, 0.2, 0.25 * M_PI / 180.0, 0.3, 0.2f, (uint32_t)n_b, (float)(M_PI / 6.0),
                                            0xffffffffu, 1);
  rc = rc ? rc : mcl3dl_hip_measure_update(ctx, pose, NULL, w[0], n_p, scan, n_s, beam, beam_org, n_b, origins, 1, lik[0], ratio[0],
                                           bm[0], &ent[0], &rmin[0], &rmax[0], &restored[0]);
  if (rc)
  {
    fprintf(stderr, "single context: %s\n", mcl3dl_hip_last_error(ctx));
    return 4;
  }
  mcl3dl_hip_destroy(ctx);
  /* v = 1: the group over every device, RCCL collective */
  mcl3dl_hip_group* g = NULL;
  if (mcl3dl_hip_group_create(&g, ids, n_dev) != 0)
    return 5;
  rc = mcl3dl_hip_group_set_option(g, "direct_single", 0);
  rc = rc ? rc : mcl3dl_hip_group_set_map(g, map, NULL, n_m, 1, NULL);
  rc = rc ? rc : mcl3dl_hip_group_set_likelihood_params(g, 0.2f, 0.05f, 5.0f);
  rc = rc ? rc : mcl3dl_hip_group_set_beam_params(g, 0.1f, 0.1f, 0.1f, 0.2, 0.25 * M_PI / 180.0, 0.3, 0.2f, (uint32_t)n_b,
                                                  (float)(M_PI / 6.0), 0xffffffffu, 1);
  for (int rep = 0; rep < 3 && !rc; ++rep)
  {
    memcpy(w[1], w0, sizeof(float) * n_p);
    rc = mcl3dl_hip_group_measure_update(g, pose, NULL, w[1], n_p, scan, n_s, beam, beam_org, n_b, origins, 1, lik[1], ratio[1], bm[1],
                                         &ent[1], &rmin[1], &rmax[1], &restored[1]);
  }
  if (rc)
  {
    fprintf(stderr, "group of %d: %s\n", n_dev, mcl3dl_hip_group_last_error(g));
    return 6;
  }
  uint64_t n_rccl = 0, n_host = 0;
  mcl3dl_hip_group_collective_stats(g, &n_rccl, &n_host);
  /* ---- a whole filter iteration with the particles RESIDENT on the devices: upload once, then pf::measure (all-reduce),
   * expectation + covariance (host combine of one record per shard), resample (all-gather of the 13-float states) — and the
   * second update reads the NEW generation without any pose upload. Checked against the first update above and against the
   * single-context functions on the same inputs. */
  float* st13 = (float*)calloc(13 * n_p, sizeof(float));
  for (size_t i = 0; i < n_p; ++i)
    memcpy(st13 + 13 * i, pose + 7 * i, sizeof(float) * 7);
  float ent_r = 0, rmin_r = 0, rmax_r = 0, mean7[7], total = 0, cov[36], pstep = 0;
  int restored_r = 0;
  int64_t imax = -1, imax_b = -1;
  size_t n_dup = 0, bad_r = 0;
  float* w_r = (float*)malloc(sizeof(float) * n_p);
  float* lik_r = (float*)malloc(sizeof(float) * n_p);
  rc = mcl3dl_hip_group_upload_state(g, st13, w0, n_p);
  rc = rc ? rc : mcl3dl_hip_group_update_resident(g, NULL, scan, n_s, beam, beam_org, n_b, origins, 1, w_r, lik_r, NULL, NULL, &ent_r,
                                                  &rmin_r, &rmax_r, &restored_r);
  rc = rc ? rc : mcl3dl_hip_group_expectation(g, NULL, mean7, &total, &imax, &imax_b);
  rc = rc ? rc : mcl3dl_hip_group_covariance(g, mean7, cov);
  rc = rc ? rc : mcl3dl_hip_group_resample_begin(g, 0, &pstep);
  rc = rc ? rc : mcl3dl_hip_group_resample_plan(g, 0, 0.37f * pstep, NULL, NULL, &n_dup);
  float* noise = (float*)calloc(13 * (n_dup + 1), sizeof(float));
  for (size_t i = 0; i < n_dup; ++i)
  {
    noise[13 * i + 0] = 0.01f * (uniform01() - 0.5f);
    noise[13 * i + 6] = 1.0f; /* identity rotation */
  }
  rc = rc ? rc : mcl3dl_hip_group_resample_apply(g, noise, n_dup);
  float* st_new = (float*)malloc(sizeof(float) * 13 * n_p);
  float* w_new = (float*)malloc(sizeof(float) * n_p);
  rc = rc ? rc : mcl3dl_hip_group_download_state(g, st_new, w_new, n_p);
  float ent_2 = 0;
  rc = rc ? rc : mcl3dl_hip_group_update_resident(g, NULL, scan, n_s, beam, beam_org, n_b, origins, 1, NULL, NULL, NULL, NULL, &ent_2,
                                                  NULL, NULL, NULL);
  if (rc)
  {
    fprintf(stderr, "resident iteration over %d device(s): %s\n", n_dev, mcl3dl_hip_group_last_error(g));
    return 7;
  }
  uint64_t n_rccl2 = 0;
  mcl3dl_hip_group_collective_stats(g, &n_rccl2, &n_host);
  mcl3dl_hip_group_destroy(g);
  /* the resident update = the host-array update of the same particles */
  double worst_r = 0;
  for (size_t i = 0; i < n_p; ++i)
  {
    bad_r += memcmp(&lik_r[i], &lik[1][i], 4) != 0;
    const double rel = fabs((double)w_r[i] - (double)w[1][i]) / fmax(fabs((double)w[1][i]), 1e-30);
    worst_r = rel > worst_r ? rel : worst_r;
  }
  /* the new generation against one context resampling the same weights */
  mcl3dl_hip_ctx* c2 = NULL;
  size_t n_dup1 = 0, bad_st = 0;
  float pstep1 = 0;
  float* st_ref = (float*)malloc(sizeof(float) * 13 * n_p);
  rc = mcl3dl_hip_create(&c2, 0);
  rc = rc ? rc : mcl3dl_hip_resample_begin(c2, w_r, n_p, n_p, &pstep1);
  rc = rc ? rc : mcl3dl_hip_resample_plan(c2, 0, 0.37f * pstep1, NULL, NULL, &n_dup1);
  rc = rc ? rc : mcl3dl_hip_resample_apply(c2, st13, noise, n_dup, st_ref);
  if (rc)
    return 8;
  mcl3dl_hip_destroy(c2);
  bad_st = (size_t)(memcmp(st_ref, st_new, sizeof(float) * 13 * n_p) != 0);
  printf("group_all_devices (resident): update %zu mismatching likelihoods, worst weight rel err %.3g, mean (%.4f %.4f %.4f), sum w %.6f, "
         "max at %lld, cov[0] %.3g, %zu duplicates (%zu on one context), new generation %s, weights %.3g, entropy after %.4f, "
         "collectives %llu\n",
         bad_r, worst_r, mean7[0], mean7[1], mean7[2], total, (long long)imax, cov[0], n_dup, n_dup1, bad_st ? "DIFFERS" : "identical",
         w_new[0], ent_2, (unsigned long long)n_rccl2);
  if (bad_r || worst_r > 2e-7 || bad_st || n_dup != n_dup1 || pstep != pstep1 || fabs(total - 1.0f) > 1e-5f || imax < 0 ||
      imax >= (int64_t)n_p || !(cov[0] > 0.f) || fabs(w_new[0] * (float)n_p - 1.0f) > 1e-6f || !(ent_2 > 0.f) ||
      n_rccl2 != n_rccl + 3 /* two all-reduces + one all-gather */)
    return 9;
  size_t bad = 0, matched = 0;
  double worst = 0;
  for (size_t i = 0; i < n_p; ++i)
  {
    if (memcmp(&lik[0][i], &lik[1][i], 4) || memcmp(&ratio[0][i], &ratio[1][i], 4) || memcmp(&bm[0][i], &bm[1][i], 4))
      ++bad;
    matched += lik[0][i] > 0.f;
    const double rel = fabs((double)w[0][i] - (double)w[1][i]) / fmax(fabs((double)w[0][i]), 1e-30);
    worst = rel > worst ? rel : worst;
  }
  printf("group_all_devices: %d device(s), %zu particles, %zu with matches, rccl all-reduces %llu, host combines %llu, "
         "mismatching likelihood/ratio/beam %zu, worst weight rel err %.3g, entropy %.6f vs %.6f\n",
         n_dev, n_p, matched, (unsigned long long)n_rccl, (unsigned long long)n_host, bad, worst, ent[1], ent[0]);
  if (bad || worst > 2e-7 || fabs(ent[0] - ent[1]) > 1e-6 * fabs(ent[0]) || n_rccl != 3 || n_host != 0 || restored[0] != restored[1] ||
      rmin[0] != rmin[1] || rmax[0] != rmax[1] || matched < n_p / 2)
    return 1;
  return 0;
}
