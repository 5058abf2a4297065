"""Scan preparation in a loop, with the host time of each call: for rocprofv3 --kernel-trace --stats."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from mcl_3dl_amd import capi  # noqa: E402
from mcl_3dl_amd.synthetic import make_config  # noqa: E402

sc = make_config("C3", seed=12345)
rng = np.random.default_rng(4)
raw = np.concatenate([sc.scan_lik + rng.normal(0, 0.02, sc.scan_lik.shape).astype(np.float32) for _ in range(4)], 0)[:65536]
raw = np.ascontiguousarray(raw, np.float32)
eng = capi.Engine(0)
leaf, cl, cb = (0.1, 0.1, 0.05), (0.5, 10.0, -2.0, 2.0), (0.5, 4.0, -2.0, 2.0)
org = np.array([[0, 0, 0.5]], np.float32)
tb = tf = 0.0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for it in range(N + 3):
    t0 = time.perf_counter()
    f, l, b = eng.scan_begin(raw, None, leaf=leaf, clip_lik=cl, clip_beam=cb)
    t1 = time.perf_counter()
    il = rng.integers(0, l, 16384).astype(np.uint32)
    ib = rng.integers(0, b, 512).astype(np.uint32)
    t2 = time.perf_counter()
    eng.scan_finish(il, ib, origins=org)
    t3 = time.perf_counter()
    if it >= 3:
        tb += t1 - t0
        tf += t3 - t2
print("scan_begin %.4f ms  scan_finish %.4f ms  (%d points -> %d -> %d / %d)" % (tb / N * 1e3, tf / N * 1e3, len(raw), f, l, b))
