"""strict_order = 3 with four tiles per work-group (likelihood_chain_multi.h, option chain_ppl) against the one-tile form and
the fp64 sums: the likelihood kernel (hipEvents around the launches) for few particles on a 16 384-point scan.
    PYTHONPATH=. python scripts/r05_time_chain_multi.py"""
import numpy as np
import torch

from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_config

dev = torch.device("cuda", 0)
for n_p, n_s in ((64, 16384), (256, 16384), (512, 16384), (1024, 16384), (2048, 16384), (3000, 16384), (4096, 16384), (1024, 4096), (1024, 65536)):
    sc = make_config("C2", seed=12345, n_p=n_p, n_s=n_s)
    eng = capi.Engine(0)
    eng.set_map(sc.map_xyz, sc.map_label, stamp=1, dist_weight=(1.0, 1.0, 1.0))
    eng.set_likelihood_params()
    eng.upload_scan(sc.scan_lik)
    d_pose = torch.from_numpy(np.ascontiguousarray(sc.poses, np.float32)).to(dev)
    d_lik, d_q = torch.zeros(n_p, device=dev), torch.zeros(n_p, device=dev)
    eng.set_kernel_timing(True)
    res, liks = {}, {}
    for tag, strict, ppl in (("fp64", 0, 0), ("one tile", 3, 1), ("four tiles", 3, 4)):
        eng.set_option("strict_order", strict)
        eng.set_option("chain_ppl", ppl)
        best = 1e9
        for rep in range(3):
            for _ in range(10):
                eng.measure_device(d_pose, n_p, d_lik, d_q, None)
            eng.synchronize()
            eng.reset_kernel_time()
            for _ in range(50):
                eng.measure_device(d_pose, n_p, d_lik, d_q, None)
            eng.synchronize()
            ms, n = eng.kernel_time(0)
            best = min(best, ms / max(n, 1))
        res[tag] = best
        liks[tag] = d_lik.cpu().numpy().copy()
    print("%5d particles x %5d points: fp64 sums %.4f ms | in-kernel float sums, one tile per work-group %.4f ms (x %.2f) | four tiles %.4f ms (x %.2f) | same bits %s" % (
        n_p, n_s, res["fp64"], res["one tile"], res["one tile"] / res["fp64"], res["four tiles"], res["four tiles"] / res["fp64"],
        bool(np.array_equal(liks["one tile"], liks["four tiles"]))), flush=True)
    del eng
