"""mcl3dl_hip_map_update wall time, repeated (the first call pays for its scratch blocks, later ones recycle them)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from mcl_3dl_amd import capi  # noqa: E402
from mcl_3dl_amd.synthetic import make_config  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "C2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
sc = make_config(workload, seed=12345)
n_b = len(sc.scan_beam)
eng = capi.Engine(0)
eng.set_map(sc.map_xyz, sc.map_label, stamp=1, dist_weight=(1.0, 1.0, 1.0))
eng.set_likelihood_params()
eng.set_beam_params(num_points=max(n_b, 1), dda_grid_size=0.2)


def measure():
    return eng.measure_batch(sc.poses[:64], sc.scan_lik, sc.scan_beam if n_b else None, sc.scan_beam_label if n_b else None,
                             sc.origins)


measure()
tp = sc.true_pose[:3]
rng_to = np.linalg.norm(sc.map_xyz - tp, axis=1)
near = sc.map_xyz[np.argsort(rng_to)[:len(sc.map_xyz) // 100]]
inward = (tp - near) / np.maximum(np.linalg.norm(tp - near, axis=1, keepdims=True), 1e-6)
for rep in range(reps):
    upd = (near + (0.10 + 0.01 * rep) * inward).astype(np.float32)
    t0 = time.perf_counter()
    n_map, st = eng.map_update(upd, None, leaf=(0.1, 0.1, 0.1), stamp=70 + rep)
    t1 = time.perf_counter()
    measure()
    t2 = time.perf_counter()
    measure()
    t3 = time.perf_counter()
    print("update %d: wall %.3f ms (index part %.3f ms device), %d bricks, next measure %.3f ms, the one after %.3f ms" % (
        rep, (t1 - t0) * 1e3, st["device_ms"], st["bricks_recompiled"], (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
