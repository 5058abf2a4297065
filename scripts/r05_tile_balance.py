"""How evenly the tiled likelihood kernel's work is shared out over the eight XCDs: XCD x evaluates the scan tiles x, x + 8, ...
(likelihood_kernels.h), and tiles differ in cost (points in empty bricks are cheap, points along walls are not). Every 256-point
tile of the headline scan (in engine order) is timed on its own — a scan of 64 copies of that tile, i.e. the real launch shape
with every work-group doing that tile's work — and the per-XCD sums are compared with their mean.
    PYTHONPATH=. python scripts/r05_tile_balance.py [C2|C3] [map_jitter]"""
import sys

import numpy as np
import torch

from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_config

workload = sys.argv[1] if len(sys.argv) > 1 else "C2"
jitter = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
sc = make_config(workload, seed=12345, **({"map_jitter": jitter} if jitter else {}))
n_p = len(sc.poses)
eng = capi.Engine(0)
eng.set_map(sc.map_xyz, sc.map_label, stamp=1, dist_weight=(1.0, 1.0, 1.0))
eng.set_likelihood_params()
dev = torch.device("cuda", 0)
d_pose = torch.from_numpy(np.ascontiguousarray(sc.poses, np.float32)).to(dev)
d_lik, d_q = torch.zeros(n_p, device=dev), torch.zeros(n_p, device=dev)
held = np.ascontiguousarray(sc.scan_lik[capi.scan_order_host(sc.scan_lik)], np.float32)
n_tiles = (len(held) + 255) // 256
eng.set_option("scan_presorted", 1)
eng.set_option("strict_order", 0)
eng.set_kernel_timing(True)


def timed(scan, reps=30):
    eng.upload_scan(scan)
    for _ in range(5):
        eng.measure_device(d_pose, n_p, d_lik, d_q, None)
    eng.synchronize()
    eng.reset_kernel_time()
    for _ in range(reps):
        eng.measure_device(d_pose, n_p, d_lik, d_q, None)
    eng.synchronize()
    ms, n = eng.kernel_time(0)
    return ms / max(n, 1)


whole = timed(held, 100)
cost = np.zeros(n_tiles)
for j in range(n_tiles):
    tile = held[256 * j:256 * (j + 1)]
    if len(tile) < 256:
        tile = np.concatenate([tile, np.repeat(tile[-1:], 256 - len(tile), axis=0)])
    cost[j] = timed(np.ascontiguousarray(np.tile(tile, (n_tiles, 1))))
whole2 = timed(held, 100)
per_xcd = np.array([cost[x::8].sum() for x in range(8)]) / n_tiles
print("%s%s: whole scan %.4f / %.4f ms per launch; mean of the per-tile launches %.4f ms (min %.4f, max %.4f)" % (
    workload, " jitter %.3f" % jitter if jitter else "", whole, whole2, cost.mean(), cost.min(), cost.max()))
print("tile costs (ms when all %d tiles are that tile): %s" % (n_tiles, " ".join("%.3f" % c for c in cost)))
print("predicted share per XCD (ms): %s" % " ".join("%.4f" % v for v in per_xcd))
print("slowest XCD / mean: %.3f   (sum of shares %.4f ms against the whole launch %.4f ms)" % (per_xcd.max() / per_xcd.mean(), per_xcd.sum(), whole))
# what a balanced assignment of whole tiles to XCDs could reach (longest-processing-time greedy, 8 tiles per XCD)
order = np.argsort(-cost)
load, cnt = np.zeros(8), np.zeros(8, int)
for j in order:
    free = np.where(cnt < (n_tiles + 7) // 8)[0]
    x = free[np.argmin(load[free])]
    load[x] += cost[j]
    cnt[x] += 1
print("LPT assignment of the same tiles: slowest XCD / mean %.3f" % (load.max() / load.mean()))
