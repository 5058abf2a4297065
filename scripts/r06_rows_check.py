"""Round 6: caller-order float sums inside the per-particle kernels (lik_particle's LDS row, float_chain.h).
(1) correctness against the reference compiled here (oracle/_ref): likelihoods and — through the fused update — weights must
be the reference's bits in the DEFAULT mode; (2) what the per-particle kernel with rows costs against the tiled kernel with
fp64 sums for scans of 768 .. 12 288 points (which one should the default take below the row limit?).
    PYTHONPATH=. python scripts/r06_rows_check.py"""
import sys

import numpy as np
import torch

from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_config
from oracle import pyoracle

dev = torch.device("cuda", 0)
kind = "ref" if pyoracle.available("ref") else "port"
bad = 0
for n_p, n_s in ((64, 96), (64, 1000), (3, 5000), (300, 700), (1000, 8), (5000, 32), (2, 12288), (513, 300), (1500, 100)):
    sc = make_config("C1", seed=7, n_p=n_p, n_s=n_s)
    eng = capi.Engine(0)
    eng.set_map(sc.map_xyz, sc.map_label, stamp=1, dist_weight=(1.0, 1.0, 1.0))
    eng.set_likelihood_params()
    orc = pyoracle.Oracle(kind)
    orc.set_map(sc.map_xyz, sc.map_label, dist_weight=(1.0, 1.0, 1.0))
    orc.set_likelihood_params(pyoracle.LikelihoodParams())
    want, want_q = orc.likelihood_measure(sc.poses, sc.scan_lik)
    lik, q, _ = eng.measure_batch(sc.poses, sc.scan_lik)
    ok = np.array_equal(lik, want) and np.array_equal(q, want_q)
    w0 = np.random.default_rng(1).uniform(0.1, 1.0, n_p).astype(np.float32)
    w0 /= w0.sum()
    wantu = orc.measure_update(sc.poses, w0, sc.scan_lik, np.zeros((0, 3), np.float32), np.zeros(0, np.uint32), sc.origins, odom_err=None, odom_sigma=1.0)
    extra = np.full(n_p, np.float32(1.0 / np.sqrt(2.0 * np.pi)), np.float32)  # the oracle's odometry factor at zero error (nd.h:41-58)
    got = eng.measure_update(sc.poses, w0, sc.scan_lik, extra=extra)
    okw = np.array_equal(got["weights"], wantu["weights"])
    relw = float(np.max(np.abs(got["weights"] - wantu["weights"]) / np.maximum(wantu["weights"], 1e-30)))
    print("%5d x %5d: default-mode likelihoods == reference: %s (max rel %.2e)   weights of the fused update == reference: %s (max rel %.2e)" % (
        n_p, n_s, ok, float(np.max(np.abs(lik - want) / np.maximum(np.abs(want), 1e-30))), okw, relw), flush=True)
    bad += (not ok) + (not okw)
    del eng

print("---- per-particle kernel with caller-order rows (default mode) against the tiled kernel with fp64 sums (strict_order 0)")
for n_p, n_s in ((64, 1000), (4096, 768), (4096, 1024), (4096, 2048), (4096, 4096), (300, 3000), (64, 4096), (1024, 4096), (512, 8192), (4096, 8192), (64, 12288)):
    sc = make_config("C2", seed=12345, n_p=n_p, n_s=n_s)
    eng = capi.Engine(0)
    eng.set_map(sc.map_xyz, sc.map_label, stamp=1, dist_weight=(1.0, 1.0, 1.0))
    eng.set_likelihood_params()
    eng.upload_scan(sc.scan_lik)
    d_pose = torch.from_numpy(np.ascontiguousarray(sc.poses, np.float32)).to(dev)
    d_lik, d_q = torch.zeros(n_p, device=dev), torch.zeros(n_p, device=dev)
    eng.set_kernel_timing(True)
    res, liks = {}, {}
    for tag, strict, tmin in (("tiled fp64", 0, 1024), ("per-particle fp64", 0, 500000000), ("per-particle rows", 2, 500000000), ("tiled replay", 1, 1024)):
        eng.set_option("strict_order", strict)
        eng.set_option("lik_tiled_min", tmin)
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
    same = bool(np.array_equal(liks["per-particle rows"], liks["tiled replay"]))
    print("%5d x %5d: tiled fp64 %.4f ms | per-particle fp64 %.4f | per-particle rows %.4f (x %.2f of tiled fp64) | tiled + replay %.4f (x %.2f) | rows == replay bits: %s" % (
        n_p, n_s, res["tiled fp64"], res["per-particle fp64"], res["per-particle rows"], res["per-particle rows"] / res["tiled fp64"],
        res["tiled replay"], res["tiled replay"] / res["tiled fp64"], same), flush=True)
    bad += not same
    del eng
sys.exit(1 if bad else 0)
