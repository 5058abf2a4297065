"""What a caller that holds its scan in the engine's order gains (option scan_presorted) in the host-buffer update of SURVEY.md 8d
— mcl3dl_hip_measure_update timed from C (tools/benchloop.c) — with the float sums in the fp64 tree (strict_order 0), inside the
kernel (3: bit-identical to the reference on the caller's own array) and replayed behind it (1).
    PYTHONPATH=. python scripts/r05_time_presorted.py [C2|C3] [steps]"""
import sys

import numpy as np

from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_config

workload = sys.argv[1] if len(sys.argv) > 1 else "C2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sc = make_config(workload, seed=12345)
n_p, n_s, n_b = len(sc.poses), len(sc.scan_lik), len(sc.scan_beam)
eng = capi.Engine(0)
eng.set_map(sc.map_xyz, sc.map_label, stamp=1, dist_weight=(1.0, 1.0, 1.0))
eng.set_likelihood_params()
eng.set_beam_params(num_points=max(n_b, 1), dda_grid_size=0.2)
poses = np.ascontiguousarray(sc.poses, np.float32)
w0 = np.full(n_p, 1.0 / n_p, np.float32)
raw = np.ascontiguousarray(sc.scan_lik, np.float32)
held = np.ascontiguousarray(raw[capi.scan_order_host(raw)])
beam = np.ascontiguousarray(sc.scan_beam, np.float32) if n_b else None
lab = np.ascontiguousarray(sc.scan_beam_label, np.uint32) if n_b else None
org = np.ascontiguousarray(sc.origins, np.float32)
o_lik, o_q, o_b = (np.zeros(n_p, np.float32) for _ in range(3))


def run(tag, scan, strict, presorted):
    eng.set_option("strict_order", strict)
    eng.set_option("scan_presorted", presorted)
    best = []
    for _ in range(3):
        ms, per = eng.time_measure_update(poses, w0, w0.copy(), scan, beam, lab, org, o_lik, o_q, o_b, steps, warm_ms=150.0)
        best.append(ms)
    print("%-52s %.4f ms (runs %s)" % (tag, min(best), " ".join("%.4f" % b for b in best)), flush=True)
    return min(best), o_lik.copy()


base, l0 = run("caller's order, fp64 sums (default path)", raw, 0, 0)
run("caller's order, float replay behind (strict 1)", raw, 1, 0)
_, l3 = run("caller's order, sums in the kernel in engine order (3)", raw, 3, 0)
run("scan held in engine order, ordered again, fp64", held, 0, 0)
p0, lp0 = run("scan held in engine order, presorted, fp64", held, 0, 1)
p3, lp3 = run("scan held in engine order, presorted, in-kernel sums", held, 3, 1)
print("in-kernel sums: engine-order result identical whether the engine or the caller ordered the scan:", bool(np.array_equal(l3, lp3)))
print("presorted + in-kernel float sums vs the default path: %.3f x" % (p3 / base))
