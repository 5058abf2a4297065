#!/bin/bash
# round 3, session 9: ticket tree + adapter (one launch for both models, voxel-centre pos_) + route A numbers
OUT=gpurun_out/r03i
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_update_small.py tests/test_gpu_adapter.py tests/test_gpu_group.py -x -q 2>&1 | tail -6 | tee $OUT/pytest.log
python - <<'P' 2>&1 | tee $OUT/shapes.log
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_config, make_scene
eng = capi.Engine(0)
dev = torch.device("cuda", 0)
def run(tag, sc, n_p, n_s, n_b):
    eng.set_beam_params(num_points=max(n_b, 1))
    eng.upload_scan(sc.scan_lik[:n_s], sc.scan_beam[:n_b] if n_b else None, sc.scan_beam_label[:n_b] if n_b else None, sc.origins)
    d_pose = torch.from_numpy(np.ascontiguousarray(sc.poses[:n_p])).to(dev)
    d_w = torch.full((n_p,), 1.0 / n_p, device=dev); d_st = torch.zeros(4, device=dev)
    d_l, d_r, d_b = (torch.empty(n_p, device=dev) for _ in range(3))
    torch.cuda.synchronize()
    out = []
    for one in (0, 1):
        eng.set_option("update_small", one)
        eng.set_option("update_small_max", 8192)
        for _ in range(200):
            eng.update_device(d_pose, n_p, d_w, d_st, d_lik=d_l, d_ratio=d_r, d_beam=d_b)
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(2000):
            eng.update_device(d_pose, n_p, d_w, d_st, d_lik=d_l, d_ratio=d_r, d_beam=d_b)
        eng.synchronize()
        out.append((time.perf_counter() - t0) / 2000 * 1e6)
        lat = []
        for _ in range(200):
            t1 = time.perf_counter(); eng.update_device(d_pose, n_p, d_w, d_st, d_lik=d_l, d_ratio=d_r, d_beam=d_b); eng.synchronize(); lat.append((time.perf_counter() - t1) * 1e6)
        out.append(float(np.median(lat)))
    eng.set_option("update_small", 1)
    print("%-22s %5d x %5d + %3d : separate %6.1f us/update (single %6.1f) | one launch %6.1f (single %6.1f)" % (tag, n_p, n_s, n_b, out[0], out[1], out[2], out[3]), flush=True)
sc = make_config("C1")
eng.set_map(sc.map_xyz, sc.map_label, stamp=1, dist_weight=(1, 1, 1)); eng.set_likelihood_params()
big = make_scene(n=91, n_p=4096, n_s=1000, n_b=96, seed=3)
run("C1", sc, 64, 1000, 0)
eng.set_map(big.map_xyz, big.map_label, stamp=2, dist_weight=(1, 1, 1))
for n_p, n_s, n_b in ((64, 96, 3), (64, 1000, 32), (256, 96, 3), (500, 300, 0), (500, 300, 40), (1024, 96, 3), (2048, 96, 3), (4096, 96, 3), (4096, 512, 0), (4096, 96, 0)):
    run("reference-scale", big, n_p, n_s, n_b)
P
python bench.py --workload C2 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/C2.json
python bench.py --workload C3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/C3.json
python bench.py --workload C1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/C1.json
python - <<P
import json
for n in ("C1","C2","C3"):
    d=json.load(open("$OUT/%s.json"%n)); k=d["kernels_ms_per_step"]
    print("%-4s ms/step %.4f lik %.4f beam %.4f pf %.4f | 8d %.4f | route_a %s | prep %.3f" % (n,d["ms_per_step"],k["likelihood"],k["beam"],k["pf"], d["update_8d"]["ms_per_update"], json.dumps(d.get("route_a",{}).get("ms_per_update")), d["scan_preparation"]["ms"]))
    print("    ", json.dumps(d.get("route_a",{}).get("breakdown")))
P
