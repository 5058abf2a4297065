"""A/B of the host-buffer update (SURVEY.md 8d's timed region: mcl3dl_hip_measure_update on host arrays) timed from C
(tools/benchloop.c): general path vs scan_stage_kernel in front (one H2D copy / zero-copy) vs pf_tail_kernel behind, pageable
vs page-locked caller arrays; next to it the device-resident update with and without the tail. Run on the GPU box:
    python scripts/time_update_8d.py C2 [steps]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from mcl_3dl_amd import capi  # noqa: E402
from mcl_3dl_amd.synthetic import make_config  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "C2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
over = {}
for a in sys.argv[3:]:
    k, v = a.split("=")
    over[k] = int(v)
sc = make_config(workload, seed=12345, **over)
n_p, n_s, n_b = len(sc.poses), len(sc.scan_lik), len(sc.scan_beam)
eng = capi.Engine(0)
eng.set_map(sc.map_xyz, sc.map_label, stamp=1, dist_weight=(1.0, 1.0, 1.0))
eng.set_likelihood_params()
eng.set_beam_params(num_points=max(n_b, 1), dda_grid_size=0.2)

poses = np.ascontiguousarray(sc.poses, np.float32)
w0 = np.full(n_p, 1.0 / n_p, np.float32)
lik_xyz = np.ascontiguousarray(sc.scan_lik, np.float32)
beam_xyz = np.ascontiguousarray(sc.scan_beam, np.float32) if n_b else None
beam_lab = np.ascontiguousarray(sc.scan_beam_label, np.uint32) if n_b else None
org = np.ascontiguousarray(sc.origins, np.float32)
out = {}


def pageable():
    return dict(poses=poses, w0=w0, w=w0.copy(), lik=lik_xyz, beam=beam_xyz, lab=beam_lab, org=org,
                o_lik=np.zeros(n_p, np.float32), o_ratio=np.zeros(n_p, np.float32), o_beam=np.zeros(n_p, np.float32))


def pinned():
    a = dict(poses=eng.host_array((n_p, 7)), w0=w0, w=eng.host_array(n_p), lik=eng.host_array((n_s, 3)),
             beam=eng.host_array((n_b, 3)) if n_b else None, lab=eng.host_array(n_b, np.uint32) if n_b else None,
             org=eng.host_array((len(org), 3)), o_lik=eng.host_array(n_p), o_ratio=eng.host_array(n_p),
             o_beam=eng.host_array(n_p))
    a["poses"][:] = poses
    a["lik"][:] = lik_xyz
    a["org"][:] = org
    if n_b:
        a["beam"][:] = beam_xyz
        a["lab"][:] = beam_lab
    return a


def run(tag, arrays, stage, zero_copy, tail):
    eng.set_option("update_stage", stage)
    eng.set_option("update_zero_copy", zero_copy)
    a = arrays
    eng.set_kernel_timing(False)
    ms, per = eng.time_measure_update(a["poses"], a["w0"], a["w"], a["lik"], a["beam"], a["lab"], a["org"], a["o_lik"], a["o_ratio"],
                                      a["o_beam"], steps, warm_ms=300.0)
    # kernel groups of the same loop (hipEvents: each timed group costs two event records, so not inside the figure above)
    eng.set_option("timing_mask", 31)
    eng.set_kernel_timing(True)
    eng.reset_kernel_time()
    eng.time_measure_update(a["poses"], a["w0"], a["w"], a["lik"], a["beam"], a["lab"], a["org"], a["o_lik"], a["o_ratio"],
                            a["o_beam"], 20, warm_ms=0.0)
    kt = {}
    for name, kid in (("lik", 0), ("beam", 1), ("pf", 2), ("update", 3), ("stage", 4)):
        t, n = eng.kernel_time(kid)
        kt[name] = round(t / 20.0, 5)
    eng.set_kernel_timing(False)
    out[tag] = dict(ms=round(ms, 5), median=round(float(np.median(per)), 5), min=round(float(per.min()), 5),
                    p90=round(float(np.percentile(per, 90)), 5), kernels_ms=kt, checksum=float(a["w"].astype(np.float64).sum()),
                    lik0=float(a["o_lik"][0]))
    print(tag, json.dumps(out[tag]), flush=True)


pg = pageable()
run("general(stage0,tail0)", pg, 0, 0, 0)
run("stage+h2d,tail0", pg, 1, 0, 0)
run("stage+zerocopy,tail0", pg, 1, 1, 0)
run("stage+zerocopy,tail1", pg, 1, 1, 1)
pn = pinned()
run("pinned,stage+zerocopy,tail0", pn, 1, 1, 0)
run("stage+zerocopy,tail0 (again)", pg, 1, 1, 0)

# device-resident update (bench.py's `value`), eager, tail on / off
dev = torch.device("cuda", 0)
eng.upload_scan(sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
d_pose = torch.from_numpy(poses).to(dev)
d_w0 = torch.from_numpy(w0).to(dev)
d_w = d_w0.clone()
d_lik, d_ratio, d_beam = (torch.zeros(n_p, device=dev) for _ in range(3))
d_st = torch.zeros(4, device=dev)
for tail in (0, 1):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(10):
            eng.update_device(d_pose, n_p, d_w, d_st, d_lik=d_lik, d_ratio=d_ratio, d_beam=d_beam if n_b else None)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        d_w.copy_(d_w0)
        eng.update_device(d_pose, n_p, d_w, d_st, d_lik=d_lik, d_ratio=d_ratio, d_beam=d_beam if n_b else None)
    torch.cuda.synchronize()
    out["device_resident,tail%d" % tail] = round((time.perf_counter() - t0) / steps * 1e3, 5)
    print("device_resident tail", tail, out["device_resident,tail%d" % tail], flush=True)
print("RESULT " + json.dumps(dict(workload=workload, n_p=n_p, n_s=n_s, n_b=n_b, steps=steps, results=out)))
