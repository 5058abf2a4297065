/* tools/benchloop.c — MEASUREMENT INFRASTRUCTURE, plain C99 against include/mcl3dl_hip.h only (never linked into the
 * product library): the timed loop of bench.py's `update_8d` figure run from C, the language of the reference's caller
 * (src/mcl_3dl.cpp:398-426 calls pf_->measure from C++), so that the figure is the cost of the C ABI and not of ctypes
 * argument conversion (~30 us per call from Python at 19 arguments). One step = what the node does per scan: prior weights
 * restored (resampling leaves them uniform, include/mcl_3dl/pf.h:203,207), then ONE mcl3dl_hip_measure_update on host
 * arrays: scan upload + ordering, pose / weight H2D, both models, pf::measure, results D2H, synchronised. */
#define _POSIX_C_SOURCE 199309L
#include <string.h>
#include <time.h>

#include "mcl3dl_hip.h"

static double now_ms(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* Runs the update for warm_ms of wall time (clock ramp), then `steps` timed steps. Returns the mean ms per step, or the
 * (negative) error code of a failing call. per_step_ms (optional): each timed step's own duration. */
double mcl3dl_benchloop_measure_update(mcl3dl_hip_ctx* ctx, const float* pose, const float* extra, const float* w0, float* w,
                                       size_t n_p, const float* scan_lik_xyz, size_t n_s, const float* scan_beam_xyz,
                                       const uint32_t* scan_beam_origin, size_t n_b, const float* origins, size_t n_o,
                                       float* out_lik, float* out_ratio, float* out_beam, int steps, double warm_ms,
                                       double* per_step_ms)
{
  float ent, rmin, rmax;
  int restored;
  const double t_warm = now_ms();
  do
  {
    memcpy(w, w0, sizeof(float) * n_p);
    const int rc = mcl3dl_hip_measure_update(ctx, pose, extra, w, n_p, scan_lik_xyz, n_s, scan_beam_xyz, scan_beam_origin, n_b,
                                             origins, n_o, out_lik, out_ratio, out_beam, &ent, &rmin, &rmax, &restored);
    if (rc != 0)
      return rc;
  } while (now_ms() - t_warm < warm_ms);
  const double t0 = now_ms();
  double prev = t0;
  for (int s = 0; s < steps; ++s)
  {
    memcpy(w, w0, sizeof(float) * n_p);
    const int rc = mcl3dl_hip_measure_update(ctx, pose, extra, w, n_p, scan_lik_xyz, n_s, scan_beam_xyz, scan_beam_origin, n_b,
                                             origins, n_o, out_lik, out_ratio, out_beam, &ent, &rmin, &rmax, &restored);
    if (rc != 0)
      return rc;
    if (per_step_ms)
    {
      const double t = now_ms();
      per_step_ms[s] = t - prev;
      prev = t;
    }
  }
  return (now_ms() - t0) / (steps > 0 ? steps : 1);
}
