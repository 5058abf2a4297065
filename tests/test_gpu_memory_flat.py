"""A short soak in the suite (the long one: scripts/r05_soak.py, profiles/r05l_soak_600_rounds.txt): rounds of the same mixed
work — maps of changing size and dist_weight, replacing map updates, updates in every summation mode, a progressive batch
abandoned every other round, scan preparation, resampling, engines and device groups created and destroyed — must leave the
device's free memory where it was. A leak of a staging block, an index or a context per round shows within thirty rounds."""
import numpy as np
import pytest
import torch

from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_scene

pytestmark = pytest.mark.gpu


def free_mib():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0] / 2**20


def one_round(eng, grp, scenes, rng, r):
    for si, sc in enumerate(scenes):
        dw = ((1.0, 1.0, 1.0), (1.0, 1.0, 5.0), (1.0, 1.0, 2.0))[(r + si) % 3]
        eng.set_map(sc.map_xyz, sc.map_label, stamp=1000 * r + si + 1, dist_weight=dw)
        eng.set_likelihood_params()
        eng.set_beam_params(num_points=len(sc.scan_beam))
        n_p = len(sc.poses)
        w0 = np.full(n_p, 1.0 / n_p, np.float32)
        args = (sc.poses, w0, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
        for mode in (2, 3, 1, 0):
            eng.set_option("strict_order", mode)
            eng.measure_update(*args)
        eng.set_option("strict_order", 2)
        k = int(rng.integers(0, len(sc.map_xyz) - 300))
        upd = sc.map_xyz[k:k + 300] + rng.normal(0, 0.02, (300, 3)).astype(np.float32)
        eng.map_update(upd, None, leaf=(0.05, 0.05, 0.05), stamp=1000 * r + si + 500)
        eng.measure_batch(*args[:1], *args[2:])
        eng.match_split(sc.poses[0], sc.scan_lik)
        eng.measure_batch_begin(sc.poses, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins, slice_particles=256)
        eng.measure_batch_wait(0)
        if r % 2 == 0:
            eng.measure_batch_end()
        raw = np.repeat(sc.scan_lik, 2, axis=0) + rng.normal(0, 0.01, (2 * len(sc.scan_lik), 3)).astype(np.float32)
        _, n_l, n_b = eng.scan_begin(raw, None, leaf=(0.05, 0.05, 0.05))
        if n_l and n_b:
            eng.scan_finish(rng.integers(0, n_l, 1024).astype(np.uint32), rng.integers(0, n_b, 32).astype(np.uint32), sc.origins[:1])
        w = rng.uniform(0.1, 1.0, n_p).astype(np.float32)
        eng.resample_begin(w / w.sum())
        _, _, n_dup = eng.resample_plan(0, float(rng.uniform(0, 1.0 / n_p)))
        state = np.zeros((n_p, 13), np.float32)
        state[:, :7] = sc.poses
        eng.resample_apply(state, rng.normal(0, 0.01, (n_dup, 13)).astype(np.float32) if n_dup else None)
        if grp is not None:
            grp.set_map(sc.map_xyz, sc.map_label, stamp=1000 * r + si + 1, dist_weight=dw)
            grp.set_likelihood_params()
            grp.set_beam_params(num_points=len(sc.scan_beam))
            grp.measure_update(*args)


def test_thirty_rounds_of_mixed_calls_leave_the_device_memory_where_it_was():
    rng = np.random.default_rng(7)
    scenes = [make_scene(n=61, n_p=600, n_s=3000, n_b=64, seed=1), make_scene(n=91, n_p=1200, n_s=33000, n_b=128, seed=3, map_jitter=0.045)]
    eng = capi.Engine(0)
    grp = capi.Group((0, 0), collective="host")
    free = []
    try:
        for r in range(30):
            one_round(eng, grp, scenes, rng, r)
            if r % 3 == 2:
                grp.close()
                grp = capi.Group((0, 0), collective="host")
                e2 = capi.Engine(0)
                one_round(e2, None, scenes, rng, r)
                e2.close()
            eng.synchronize()
            free.append(free_mib())
    finally:
        grp.close()
        eng.close()
    # rounds repeat with period 6 (dist_weight x engine churn): compare like with like, after the pools have warmed up
    assert free[29] > free[11] - 64.0 and free[28] > free[10] - 64.0, free
