"""Debug driver for the first find of tests/test_gpu_api_fuzz.py: a stream of replacing map updates on a small map, the
updated engine against a fresh engine on the downloaded merged map (match ratios must be equal), per index structure."""
import numpy as np
from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_scene

sc = make_scene(n=41, n_p=700, n_s=2600, n_b=64, seed=500, label_wall=2, lik_clip=(0.5, 4.0, -2.0, 2.0), beam_clip=(0.5, 3.0, -2.0, 2.0))
eng, fresh = capi.Engine(0), capi.Engine(0)
rng = np.random.default_rng(1)
half = 41 * 0.1 / 2
bad = 0
for trial in range(30):
    dw = [(1.0, 1.0, 1.0), (1.0, 1.0, 5.0)][trial % 2]
    eng.set_map(sc.map_xyz, sc.map_label, stamp=1000 + trial, dist_weight=dw)
    for k in range(5):
        n_new = int(rng.integers(0, 250))
        span = half - 0.3
        pts = rng.uniform(-span, span, (n_new, 3)).astype(np.float32)
        pts[:, 2] = rng.uniform(-half + 0.2, -half + 1.2, n_new)
        lab = rng.integers(0, 4, n_new).astype(np.uint32)
        n_map, st = eng.map_update(pts, lab, leaf=(0.2, 0.2, 0.2), stamp=5000 + 10 * trial + k)
        mx, ml = eng.map_download()
        fresh.set_map(mx, ml, stamp=9000 + 10 * trial + k, dist_weight=dw)
        # queries: scan points seen from poses + the update's own neighbourhood
        poses = sc.poses[:64]
        scan = sc.scan_lik[:777]
        res = {}
        for idx in (2, 0):
            eng.set_option("lik_index", idx)
            res[idx] = eng.measure_batch(poses, scan)
        eng.set_option("lik_index", 2)
        f = fresh.measure_batch(poses, scan)
        for idx in (2, 0):
            if not np.array_equal(res[idx][1], f[1]):
                bad += 1
                d = np.nonzero(res[idx][1] != f[1])[0]
                print("trial %d update %d (n_new %d, outcome %s, n_map %d): lik_index %d differs from a fresh engine for %d particles, e.g. %s vs %s"
                      % (trial, k, n_new, st, n_map, idx, len(d), res[idx][1][d[:3]], f[1][d[:3]]))
print("mismatches:", bad)
