#!/bin/bash
# round 3, session 13: division-free ray set-up — self-test, parity, and what it buys (A/B on one box)
OUT=gpurun_out/r03m
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kats.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_c4c5.py tests/test_gpu_fullsize.py tests/test_gpu_update_small.py -x -q 2>&1 | tail -6 | tee $OUT/pytest.log
python - <<'P' 2>&1 | tee $OUT/beam_ab.log
import sys, numpy as np, torch
sys.path.insert(0, ".")
from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_config
eng = capi.Engine(0)
dev = torch.device("cuda", 0)
for n_b in (512, 16384):
    sc = make_config("C3", n_b=n_b)
    eng.set_map(sc.map_xyz, sc.map_label, stamp=n_b, dist_weight=(1, 1, 1)); eng.set_likelihood_params()
    eng.set_beam_params(num_points=n_b, dda_grid_size=0.2)
    eng.upload_scan(sc.scan_lik[:64], sc.scan_beam, sc.scan_beam_label, sc.origins)
    d_pose = torch.from_numpy(sc.poses).to(dev); n_p = len(sc.poses)
    d_l, d_r, d_b = (torch.empty(n_p, device=dev) for _ in range(3))
    torch.cuda.synchronize()
    for rep in range(2):
        for fast in (0, 1):
            eng.set_option("beam_fast_div", fast)
            for _ in range(5):
                eng.measure_device(d_pose, n_p, None, None, d_b)
            eng.synchronize()
            eng.set_option("timing_mask", 7); eng.set_kernel_timing(True); eng.reset_kernel_time()
            for _ in range(20):
                eng.measure_device(d_pose, n_p, None, None, d_b)
            ms, n = eng.kernel_time(capi.KERNEL_BEAM)
            eng.set_kernel_timing(False)
            print("C3 beam, %5d rays/particle, fast_div=%d: %.4f ms per launch group" % (n_b, fast, ms / max(n, 1)), flush=True)
P
