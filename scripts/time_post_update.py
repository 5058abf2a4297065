"""The rows after the update in a loop — expectationBiased + max + covariance, then resampling of
device-resident states — for rocprofv3 --kernel-trace --stats: which launches they are made of."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from mcl_3dl_amd import capi  # noqa: E402
from mcl_3dl_amd.synthetic import make_config  # noqa: E402

n_p = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sc = make_config("C2", n_p=n_p, seed=12345)
eng = capi.Engine(0)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
w = rng.uniform(0.1, 1.0, n_p).astype(np.float32)
w /= w.sum()
d_pose = torch.from_numpy(sc.poses).to(dev)
d_w = torch.from_numpy(w).to(dev)
st13 = np.zeros((n_p, 13), np.float32)
st13[:, :7] = sc.poses
d_in = torch.from_numpy(st13).to(dev)
d_out = torch.empty_like(d_in)
ident = np.zeros((n_p, 13), np.float32)
ident[:, 6] = 1.0
N = 30
t_red = t_res = 0.0
for it in range(N + 3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mean7, _tot, _im, _ib = eng.expectation_device(d_pose, d_w, None, n_p)
    eng.covariance_device(d_pose, d_w, n_p, mean7)
    t1 = time.perf_counter()
    pstep = eng.resample_begin_device(d_w, n_p)
    _s, _d, nd = eng.resample_plan(0, 0.37 * pstep, want_plan=False)
    eng.resample_apply_device(d_in, ident[:nd], d_out)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if it >= 3:
        t_red += t1 - t0
        t_res += t2 - t1
print("reductions %.4f ms  resample %.4f ms  (%d particles, %d duplicates)" % (t_red / N * 1e3, t_res / N * 1e3, n_p, nd))
