"""Where did C2's scan preparation go from 0.35 to 4-5 ms?  Replays the stages bench.py runs
before `cloud_path_extras` one by one and times scan_begin / scan_finish after each of them, with and without beam points."""
import gc
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402,F401
from mcl_3dl_amd import capi  # noqa: E402
from mcl_3dl_amd.synthetic import make_config  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
sc = make_config(wl, seed=12345)
n_s, n_b = len(sc.scan_lik), len(sc.scan_beam)
rng = np.random.default_rng(4)
raw = np.concatenate([sc.scan_lik + rng.normal(0, 0.02, sc.scan_lik.shape).astype(np.float32) for _ in range(4)], 0)[:65536]
raw = np.ascontiguousarray(raw, np.float32)
leaf, cl, cb = (0.1, 0.1, 0.05), (0.5, 10.0, -2.0, 2.0), (0.5, 4.0, -2.0, 2.0)
org = np.array([[0, 0, 0.5]], np.float32)
eng = capi.Engine(0)

# evidence for the cause: every collection of Python's cyclic garbage collector with its generation and duration
_gc_t0 = [0.0]
gc_log = []


def _gc_cb(phase, info):
    if phase == "start":
        _gc_t0[0] = time.perf_counter()
    else:
        gc_log.append((info["generation"], (time.perf_counter() - _gc_t0[0]) * 1e3))


gc.callbacks.append(_gc_cb)
if os.environ.get("NOGC"):
    gc.disable()


def prep(tag, nb, n=10):
    tb = tf = 0.0
    worst = 0.0
    for it in range(n + 3):
        t0 = time.perf_counter()
        f, l, b = eng.scan_begin(raw, None, leaf=leaf, clip_lik=cl, clip_beam=cb)
        t1 = time.perf_counter()
        il = rng.integers(0, l, 16384).astype(np.uint32)
        ib = rng.integers(0, b, nb).astype(np.uint32) if nb else None
        t2 = time.perf_counter()
        eng.scan_finish(il, ib, origins=org)
        t3 = time.perf_counter()
        if it >= 3:
            tb += t1 - t0
            tf += t3 - t2
            worst = max(worst, (t1 - t0) + (t3 - t2))
    slow_gc = [(g, round(ms, 2)) for g, ms in gc_log if ms > 0.3]
    del gc_log[:]
    print("%-44s nb=%-4d begin %.3f ms finish %.3f ms worst call %.3f ms  gc passes > 0.3 ms (generation, ms): %s"
          % (tag, nb, tb / n * 1e3, tf / n * 1e3, worst * 1e3, slow_gc), flush=True)


prep("fresh engine", 0)
prep("fresh engine", 512)
eng.set_map(sc.map_xyz, sc.map_label, stamp=1, dist_weight=(1, 1, 1))
eng.set_likelihood_params()
eng.set_beam_params(num_points=max(n_b, 1), dda_grid_size=0.2)
eng.upload_scan(sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
d_pose = torch.from_numpy(sc.poses).cuda().contiguous()
d_w = torch.full((len(sc.poses),), 1.0 / len(sc.poses), device="cuda")
d_lik, d_ratio, d_beam = (torch.empty(len(sc.poses), device="cuda") for _ in range(3))
d_stats = torch.zeros(4, device="cuda")
eng.update_device(d_pose, len(sc.poses), d_w, d_stats, d_lik=d_lik, d_ratio=d_ratio, d_beam=d_beam if n_b else None)
eng.synchronize()
prep("after map + one update", 0)
prep("after map + one update", 512)
eng.measure_update(sc.poses, sc.weights, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
prep("after host-buffer update", 0)
st13 = np.zeros((len(sc.poses), 13), np.float32)
st13[:, :7] = sc.poses
w_host = d_w.cpu().numpy()
pstep = eng.resample_begin(w_host)
_s, _d, nd = eng.resample_plan(0, 0.37 * pstep)
ident = np.zeros((nd, 13), np.float32)
ident[:, 6] = 1
eng.resample_apply(st13, ident)
prep("after resample", 0)
for g in (0, 1):
    eng.set_option("use_graph", g)
    for _ in range(4):
        eng.update_device(d_pose, len(sc.poses), d_w, d_stats, d_lik=d_lik, d_ratio=d_ratio, d_beam=d_beam if n_b else None)
    eng.synchronize()
eng.set_option("use_graph", 0)
prep("after graph replay", 0)
tp = sc.true_pose[:3]
near = sc.map_xyz[np.argsort(np.linalg.norm(sc.map_xyz - tp, axis=1))[:len(sc.map_xyz) // 100]]
inward = (tp - near) / np.maximum(np.linalg.norm(tp - near, axis=1, keepdims=True), 1e-6)
eng.map_update((near + 0.12 * inward).astype(np.float32), None, leaf=(0.1, 0.1, 0.1), stamp=77)
eng.map_update(None, None, stamp=78)
prep("after map_update + withdraw", 0)
scj = make_config(wl, seed=12345, map_jitter=0.045)
eng.set_map(scj.map_xyz, scj.map_label, stamp=2, dist_weight=(1, 1, 1))
eng.upload_scan(scj.scan_lik, scj.scan_beam, scj.scan_beam_label, scj.origins)
eng.measure_device(d_pose, len(sc.poses), d_lik, d_ratio, None)
eng.synchronize()
print("index after jitter:", eng.index_stats(), flush=True)
prep("after jittered map", 0)
prep("after jittered map", 512)
eng.set_kernel_timing(True)
eng.measure_device(d_pose, len(sc.poses), d_lik, d_ratio, None)
print(eng.kernel_time(capi.KERNEL_LIKELIHOOD))
eng.set_kernel_timing(False)
prep("after kernel timing", 0)
