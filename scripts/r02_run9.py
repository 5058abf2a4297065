#!/usr/bin/env python3
"""Round-2 GPU session 9 (final tree of the round: predicated DDA walk, prepared beam origins, hybrid XCD mapping, device grid builders): whole -m gpu suite; cooperative overflow + per-map voxel edge on jittered maps; tiled kernel for
short scans (lik_tiled_min); device-side scan ordering in the host-buffer path; full bench lines + profiles (C2, C3, C5)."""
import json
import os
import subprocess
import time

OUT = "gpurun_out/r02h"
os.makedirs(OUT, exist_ok=True)
T0 = time.time()


def sh(cmd, log, timeout, env=None):
    t = time.time()
    try:
        with open(os.path.join(OUT, log), "w") as f:
            rc = subprocess.run(cmd, shell=True, stdout=f, stderr=subprocess.STDOUT, timeout=timeout,
                                env=dict(os.environ, **(env or {}))).returncode
    except subprocess.TimeoutExpired:
        rc = -9
    print("[%6.0fs] rc=%s %.0fs  %s" % (time.time() - T0, rc, time.time() - t, cmd[:140]), flush=True)
    return rc


def bench(name, args, timeout=600):
    sh("python bench.py %s 2>%s/%s.err | tail -1 > %s/%s.json" % (args, OUT, name, OUT, name), name + ".log", timeout)
    try:
        d = json.load(open("%s/%s.json" % (OUT, name)))
        k = d["kernels_ms_per_step"]
        ix = d["index"]
        extra = "  ovf-vox %.3f ratio %.2f" % (ix.get("voxels_with_overflow", 0) / max(ix.get("voxels_with_candidates", 1), 1),
                                              ix.get("voxel_ratio", 0))
        print("    %-22s value %.4g  ms/step %.4f  lik %.4f  beam %.4f  pf %.4f  8d %s%s" % (
            name, d["value"], d["ms_per_step"], k["likelihood"], k["beam"], k["pf"],
            ("%.4f" % d["update_8d"]["ms_per_update"]) if "update_8d" in d else "-", extra), flush=True)
        return d
    except Exception as e:  # noqa: BLE001
        print("    %s: no JSON (%s)" % (name, e), flush=True)
        return None


sh("python -m pytest tests -m gpu -q 2>&1 | grep -E 'passed|failed|FAILED'", "pytest.log", 1200)
sh("python -m pytest tests/test_gpu_c4c5.py -q -s 2>&1 | grep -E 'worst|passed|failed'", "c4c5.log", 900)
quick = "--steps 20 --warmup 3 --no-extras --no-cpu-baseline"
bench("C2_full", "--workload C2", 600)
bench("C2j045", "--workload C2 --map-jitter 0.045 %s" % quick, 300)
bench("C2j020", "--workload C2 --map-jitter 0.02 %s" % quick, 300)
bench("C3_full", "--workload C3 --no-cpu-baseline", 600)
bench("C3_stress", "--workload C3 --beam-points 16384 --no-cpu-baseline --no-extras", 600)
bench("C1_full", "--workload C1", 400)
bench("C2_strict", "--workload C2 --strict-order 1 --no-cpu-baseline --no-extras", 400)
bench("C2_forcedist", "--workload C2 --force-dist --no-cpu-baseline --no-extras", 400)
bench("C4_shard", "--workload C4 --particles 32768 --no-cpu-baseline --no-extras", 600)
bench("C4_8pt", "--workload C4 --scan-points 8 --no-cpu-baseline --no-extras", 600)
bench("C5_shard", "--workload C5 --particles 8192 --no-extras --cpu-particles 8", 900)
for p, s in ((4096, 96), (4096, 512), (4096, 1000), (4096, 2048), (64, 16384), (100000, 96), (500, 300)):
    bench("shape_%dx%d" % (p, s), "--workload C2 --particles %d --scan-points %d %s" % (p, s, quick), 300)
print("total %.0f s" % (time.time() - T0))
