"""Soak: a long mixed stream of the engine's calls (maps of changing size, replacing map updates, updates of changing shape in
every summation mode, progressive batches begun and abandoned, scan preparation, matched / unmatched, resampling, engines and
device groups created and destroyed) with the device's free memory and the process's resident set sampled along the way. A leak
shows as a steady decline of free device memory / growth of RSS across rounds that repeat the same work.
    PYTHONPATH=. python scripts/r05_soak.py [rounds]"""
import resource
import sys
import time

import numpy as np
import torch

from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_scene

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(7)
scenes = [make_scene(n=61, n_p=600, n_s=3000, n_b=64, seed=1), make_scene(n=121, n_p=2500, n_s=9000, n_b=256, seed=2),
          make_scene(n=91, n_p=1200, n_s=40000, n_b=128, seed=3, map_jitter=0.045)]


def rss_mb():
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0


def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0] / 2**20


def one_round(eng, grp, r):
    for si, sc in enumerate(scenes):
        dw = ((1.0, 1.0, 1.0), (1.0, 1.0, 5.0), (1.0, 1.0, 2.0))[(r + si) % 3]
        eng.set_map(sc.map_xyz, sc.map_label, stamp=1000 * r + si, dist_weight=dw)
        eng.set_likelihood_params()
        eng.set_beam_params(num_points=len(sc.scan_beam))
        n_p = len(sc.poses)
        w0 = np.full(n_p, 1.0 / n_p, np.float32)
        for mode in (2, 3, 1, 0):
            eng.set_option("strict_order", mode)
            eng.measure_update(sc.poses, w0, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
        eng.set_option("strict_order", 2)
        # a replacing map update (a patch of points moved a little), measurement behind it, matched / unmatched
        k = rng.integers(0, len(sc.map_xyz) - 400)
        upd = sc.map_xyz[k:k + 400] + rng.normal(0, 0.02, (400, 3)).astype(np.float32)
        eng.map_update(upd, None, leaf=(0.05, 0.05, 0.05), stamp=1000 * r + si + 500)
        eng.measure_batch(sc.poses, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
        eng.match_split(sc.poses[0], sc.scan_lik)
        # progressive batch, abandoned half way in every other round
        eng.measure_batch_begin(sc.poses, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins, slice_particles=max(256, n_p // 4))
        eng.measure_batch_wait(0)
        if r % 2 == 0:
            eng.measure_batch_end()
        # scan preparation
        raw = np.repeat(sc.scan_lik, 3, axis=0) + rng.normal(0, 0.01, (3 * len(sc.scan_lik), 3)).astype(np.float32)
        n_full, n_l, n_b = eng.scan_begin(raw, None, leaf=(0.05, 0.05, 0.05))
        if n_l and n_b:
            eng.scan_finish(rng.integers(0, n_l, 2048).astype(np.uint32), rng.integers(0, n_b, 64).astype(np.uint32), sc.origins[:1])
        # resampling
        w = rng.uniform(0.1, 1.0, n_p).astype(np.float32)
        w /= w.sum()
        eng.resample_begin(w)
        _src, _dup, n_dup = eng.resample_plan(0, float(rng.uniform(0, 1.0 / n_p)))
        state = np.zeros((n_p, 13), np.float32)
        state[:, :7] = sc.poses
        eng.resample_apply(state, rng.normal(0, 0.01, (n_dup, 13)).astype(np.float32) if n_dup else None)
        if grp is not None:
            grp.set_map(sc.map_xyz, sc.map_label, stamp=1000 * r + si, dist_weight=dw)
            grp.set_likelihood_params()
            grp.set_beam_params(num_points=len(sc.scan_beam))
            grp.measure_update(sc.poses, w0, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)


t0 = time.time()
eng = capi.Engine(0)
grp = capi.Group((0, 0), collective="host")
hist = []
for r in range(rounds):
    one_round(eng, grp, r)
    if r % 3 == 2:
        # engines and groups come and go
        grp.close()
        grp = capi.Group((0, 0), collective="host")
        e2 = capi.Engine(0)
        one_round(e2, None, r)
        e2.close()
    eng.synchronize()
    hist.append((r, free_mb(), rss_mb()))
    print("round %2d  free device memory %9.1f MiB  max RSS %8.1f MiB  (%.0f s)" % (r, hist[-1][1], hist[-1][2], time.time() - t0), flush=True)
half = len(hist) // 2
d_free = hist[-1][1] - hist[half][1]
d_rss = hist[-1][2] - hist[half][2]
print("second half of the run: free device memory %+.1f MiB, max RSS %+.1f MiB over %d rounds" % (d_free, d_rss, len(hist) - 1 - half))
print("LEAK SUSPECTED" if d_free < -64 or d_rss > 256 else "no leak seen")
