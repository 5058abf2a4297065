"""The polled completion word written by the update's last kernel itself (option update_fold_done) against a kernel of its own
behind it: the host-buffer update of SURVEY.md 8d, timed from C (tools/benchloop.c), where the update ends in a one-work-group
kernel — the fused pf::measure (up to 1024 particles) and the one-block apply behind a float-order replay.
    PYTHONPATH=. python scripts/r05_time_fold.py [steps]"""
import sys

import numpy as np

from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_config

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for workload, over, strict in (("C1", {}, 0), ("C2", dict(n_p=256), 0), ("C2", dict(n_p=1024), 0), ("C2", dict(n_p=1024, n_s=2048), 0),
                               ("C2", dict(n_p=4096), 1)):
    sc = make_config(workload, seed=12345, **over)
    n_p, n_s, n_b = len(sc.poses), len(sc.scan_lik), len(sc.scan_beam)
    eng = capi.Engine(0)
    eng.set_map(sc.map_xyz, sc.map_label, stamp=1, dist_weight=(1.0, 1.0, 1.0))
    eng.set_likelihood_params()
    eng.set_beam_params(num_points=max(n_b, 1), dda_grid_size=0.2)
    eng.set_option("strict_order", strict)
    poses = np.ascontiguousarray(sc.poses, np.float32)
    w0 = np.full(n_p, 1.0 / n_p, np.float32)
    scan = np.ascontiguousarray(sc.scan_lik, np.float32)
    beam = np.ascontiguousarray(sc.scan_beam, np.float32) if n_b else None
    lab = np.ascontiguousarray(sc.scan_beam_label, np.uint32) if n_b else None
    org = np.ascontiguousarray(sc.origins, np.float32)
    o_lik, o_q, o_b = (np.zeros(n_p, np.float32) for _ in range(3))
    res = {0: [], 1: []}
    liks = {}
    for rep in range(4):
        for fold in (1, 0):
            eng.set_option("update_fold_done", fold)
            ms, _ = eng.time_measure_update(poses, w0, w0.copy(), scan, beam, lab, org, o_lik, o_q, o_b, steps, warm_ms=100.0)
            res[fold].append(ms)
            liks[fold] = o_lik.copy()
    print("%s %5d particles x %5d points + %3d rays, strict_order %d: folded %.4f ms (%s)  own kernel %.4f ms (%s)  same bits %s" % (
        workload, n_p, n_s, n_b, strict, min(res[1]), " ".join("%.4f" % v for v in res[1]), min(res[0]),
        " ".join("%.4f" % v for v in res[0]), bool(np.array_equal(liks[0], liks[1]))), flush=True)
    del eng
