"""Where the host-buffer update (SURVEY.md 8d's timed region) spends its time at C2: scan upload + ordering, pose / weight
H2D, kernels, D2H. Run on the GPU box."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from mcl_3dl_amd import capi  # noqa: E402
from mcl_3dl_amd.synthetic import make_config  # noqa: E402

sc = make_config(sys.argv[1] if len(sys.argv) > 1 else "C2", seed=12345)
eng = capi.Engine(0)
eng.set_map(sc.map_xyz, sc.map_label, stamp=1, dist_weight=(1.0, 1.0, 1.0))
eng.set_likelihood_params()
nb = len(sc.scan_beam)
if nb:
    eng.set_beam_params(num_points=nb)


def timeit(f, n=50):
    """Median of n individually synchronised calls (a one-off stall of tens of milliseconds — a lazily loaded code object, a
    staging chunk being pinned — otherwise lands in whichever loop happens to be running and reads as +0.7 ms per call)."""
    for _ in range(5):
        f()
    eng.synchronize()
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        f()
        eng.synchronize()
        ts.append(time.perf_counter() - t)
    return float(np.median(ts)) * 1e3


# clock ramp: a GPU coming out of idle needs a few hundred milliseconds of load before its timings mean anything
eng.upload_scan(sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
_p = torch.from_numpy(sc.poses).cuda()
_w = torch.from_numpy(sc.weights).cuda()
_s = torch.zeros(4, device="cuda")
_t = time.perf_counter()
while time.perf_counter() - _t < 0.5:
    for _ in range(20):
        eng.update_device(_p, len(sc.poses), _w, _s)
    eng.synchronize()

for thr in (4096, 0):
    eng.set_option("scan_order_device", thr)
    # (synchronise per call: staged copies are recycled at a synchronisation, a free-running loop would only grow the
    # pinned staging area)
    print("scan_order_device=%d: upload_scan %.4f ms" % (thr, timeit(lambda: (eng.upload_scan(sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins), eng.synchronize()))))
    print("   measure_update (host buffers) %.4f ms" % timeit(lambda: eng.measure_update(sc.poses, sc.weights, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)))
eng.set_option("scan_order_device", 4096)
print("upload_poses %.4f ms" % timeit(lambda: eng.upload_poses(sc.poses)))
d_pose = torch.from_numpy(sc.poses).cuda()
d_w = torch.from_numpy(sc.weights).cuda()
d_st = torch.zeros(4, device="cuda")
eng.upload_scan(sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
print("update_device (async enqueue + one sync per call) %.4f ms" % timeit(lambda: (eng.update_device(d_pose, len(sc.poses), d_w, d_st), eng.synchronize())))
