"""Where the CPU time of a long completion wait goes: the caller's thread (RUSAGE_THREAD) against the whole process (RUSAGE_SELF:
the HIP runtime's helper threads included), for the polled word with and without the hipStreamQuery health checks, and for
hipStreamSynchronize.   PYTHONPATH=. python scripts/r05_dbg_cpushare.py"""
import resource
import time

import numpy as np
import torch

from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_scene

sc = make_scene(n=91, n_p=64, n_s=8192, seed=77)
n_p = 400000
rng = np.random.default_rng(1)
poses = np.repeat(sc.poses, (n_p + 63) // 64, axis=0)[:n_p].copy()
poses[:, :3] += rng.normal(0, 0.05, (n_p, 3)).astype(np.float32)
eng = capi.Engine(0)
eng.set_map(sc.map_xyz, sc.map_label, stamp=1, dist_weight=(1.0, 1.0, 1.0))
eng.set_likelihood_params()
eng.upload_scan(sc.scan_lik)
dev = torch.device("cuda", 0)
d_pose = torch.from_numpy(poses).to(dev)
d_lik, d_q = torch.zeros(n_p, device=dev), torch.zeros(n_p, device=dev)
torch.cuda.synchronize()


def cpu(who):
    r = resource.getrusage(who)
    return r.ru_utime + r.ru_stime


def run(tag, reps=8, **opts):
    saved = {k: eng.get_option(k) for k in opts}
    for k, v in opts.items():
        eng.set_option(k, v)
    try:
        for _ in range(2):
            eng.measure_device(d_pose, n_p, d_lik, d_q, None)
            eng.synchronize()
        p0, t0, w0 = cpu(resource.RUSAGE_SELF), cpu(resource.RUSAGE_THREAD), time.perf_counter()
        launch = 0.0
        for _ in range(reps):
            a = time.perf_counter()
            eng.measure_device(d_pose, n_p, d_lik, d_q, None)
            launch += time.perf_counter() - a
            eng.synchronize()
        p1, t1, w1 = cpu(resource.RUSAGE_SELF), cpu(resource.RUSAGE_THREAD), time.perf_counter()
    finally:
        for k, v in saved.items():
            eng.set_option(k, v)
    wall = w1 - w0
    print("%-44s %.2f ms/update (launch call %.2f ms)  process CPU %3.0f %%  caller thread %3.0f %%" % (
        tag, wall / reps * 1e3, launch / reps * 1e3, 100 * (p1 - p0) / wall, 100 * (t1 - t0) / wall), flush=True)


for rep in range(2):
    run("polled word, query every 5 ms (default)")
    run("polled word, no spin phase", poll_spin_us=0)
    run("polled word, pure spin", poll_spin_us=1e6)
    run("hipStreamSynchronize", poll_sync=0)
