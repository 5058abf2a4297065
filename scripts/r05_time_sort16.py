"""The one-launch sort of 16-bit keys (option sort_one_launch) against the launches per 8-bit pass in the host-buffer update of
SURVEY.md 8d (timed from C, tools/benchloop.c).   PYTHONPATH=. python scripts/r05_time_sort16.py"""
import numpy as np

from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_config

for workload, over in (("C2", {}), ("C3", {}), ("C2", dict(n_s=32768)), ("C2", dict(n_s=4096))):
    sc = make_config(workload, seed=12345, **over)
    n_p, n_s, n_b = len(sc.poses), len(sc.scan_lik), len(sc.scan_beam)
    eng = capi.Engine(0)
    eng.set_map(sc.map_xyz, sc.map_label, stamp=1, dist_weight=(1.0, 1.0, 1.0))
    eng.set_likelihood_params()
    eng.set_beam_params(num_points=max(n_b, 1), dda_grid_size=0.2)
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
        for one in (1, 0):
            eng.set_option("sort_one_launch", one)
            ms, _ = eng.time_measure_update(poses, w0, w0.copy(), scan, beam, lab, org, o_lik, o_q, o_b, 200, warm_ms=100.0)
            res[one].append(ms)
            liks[one] = o_lik.copy()
    print("%s %d particles x %d points + %d rays: host-buffer update  one-launch sort %.4f ms (%s)   passes %.4f ms (%s)  same bits %s" % (
        workload, n_p, n_s, n_b, min(res[1]), " ".join("%.4f" % v for v in res[1]), min(res[0]), " ".join("%.4f" % v for v in res[0]),
        bool(np.array_equal(liks[0], liks[1]))), flush=True)
    del eng
