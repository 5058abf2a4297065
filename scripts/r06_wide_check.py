"""Round 6: the per-particle likelihood kernel with caller-order rows now serves every default-mode scan of up to 4096 points below
2048 particles — which work-group size (256 or 1024 threads per particle) should it take there?  PYTHONPATH=. python scripts/r06_wide_check.py"""
import numpy as np
import torch

from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_config

dev = torch.device("cuda", 0)
for n_p, n_s in ((64, 1000), (64, 4096), (128, 2048), (256, 1024), (256, 2048), (300, 3000), (512, 1024), (512, 4096), (1024, 2048), (1024, 4096), (2000, 4096), (3, 12288)):
    sc = make_config("C2", seed=12345, n_p=n_p, n_s=n_s)
    eng = capi.Engine(0)
    eng.set_map(sc.map_xyz, sc.map_label, stamp=1, dist_weight=(1.0, 1.0, 1.0))
    eng.set_likelihood_params()
    eng.upload_scan(sc.scan_lik)
    d_pose = torch.from_numpy(np.ascontiguousarray(sc.poses, np.float32)).to(dev)
    d_lik, d_q = torch.zeros(n_p, device=dev), torch.zeros(n_p, device=dev)
    eng.set_kernel_timing(True)
    res, liks = {}, {}
    for tag, wide in (("256 threads", 0), ("1024 threads", 1 << 20)):
        eng.set_option("lik_wide_max_particles", wide)
        best = 1e9
        for rep in range(3):
            for _ in range(5):
                eng.measure_device(d_pose, n_p, d_lik, d_q, None)
            eng.synchronize()
            eng.reset_kernel_time()
            for _ in range(30):
                eng.measure_device(d_pose, n_p, d_lik, d_q, None)
            eng.synchronize()
            ms, n = eng.kernel_time(0)
            best = min(best, ms / max(n, 1))
        res[tag] = best
        liks[tag] = d_lik.cpu().numpy().copy()
    print("%5d x %5d (default mode, rows): 256 threads %.4f ms | 1024 threads %.4f ms (x %.2f) | same bits %s" % (
        n_p, n_s, res["256 threads"], res["1024 threads"], res["1024 threads"] / res["256 threads"],
        bool(np.array_equal(liks["256 threads"], liks["1024 threads"]))), flush=True)
    del eng
