#!/usr/bin/env python3
"""Round-2 GPU session 4: the whole -m gpu suite, the placement-exact VALU micro-benchmark, full bench lines of every
BASELINE configuration, the voxel-edge sweep on the jittered map, the RCCL path with one rank, profiles (kernel stats +
PMC) of C2 / C3 / C5."""
import json
import os
import subprocess
import time

OUT = "gpurun_out/r02d"
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
        extra = ""
        if "index" in d:
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


sh("python -m pytest tests/test_gpu_map_path.py -m gpu -q -x 2>&1 | tail -40", "pytest_map.log", 600)
sh("python -m pytest tests -m gpu -q 2>&1 | tail -60", "pytest.log", 1800)
sh("./profiles/valu_microbench.bin", "valu_microbench.txt", 300)
quick = "--steps 20 --warmup 3 --no-extras --no-cpu-baseline"
# voxel edge vs overflow on lattice and jittered maps
for r in (0.5, 0.42, 0.36, 0.3):
    bench("C2j_r%02d" % int(r * 100), "--workload C2 --map-jitter 0.045 --cand-voxel-ratio %g %s" % (r, quick))
    bench("C2_r%02d" % int(r * 100), "--workload C2 --cand-voxel-ratio %g %s" % (r, quick))
bench("C2j02_r50", "--workload C2 --map-jitter 0.02 %s" % quick)
# full lines
bench("C2_full", "--workload C2", 900)
bench("C1_full", "--workload C1", 600)
bench("C3_full", "--workload C3", 900)
bench("C3_stress", "--workload C3 --beam-points 16384 --no-cpu-baseline --no-extras", 900)
bench("C4_shard", "--workload C4 --particles 32768 --no-cpu-baseline", 900)
bench("C4_8pt", "--workload C4 --scan-points 8 --no-cpu-baseline", 900)
bench("C5_shard", "--workload C5 --particles 8192", 1200)
bench("C2_strict", "--workload C2 --strict-order 1 --no-cpu-baseline --no-extras", 600)
bench("C2_forcedist", "--workload C2 --force-dist --no-cpu-baseline --no-extras", 600)
bench("C2_w5", "--workload C2 --dist-weight-z 5 --no-extras", 600)
for p, s in ((4096, 512), (4096, 96), (64, 96), (100000, 96), (64, 16384)):
    bench("shape_%dx%d" % (p, s), "--workload C2 --particles %d --scan-points %d %s" % (p, s, quick))
for tag, a in (("r02d_C2", "--workload C2"), ("r02d_C3", "--workload C3"), ("r02d_C5", "--workload C5 --particles 8192")):
    sh("bash profiles/run_profiles.sh %s %s" % (tag, a), "prof_%s.log" % tag, 1500)
sh("bash profiles/run_pmc_extra.sh r02d_C2 --workload C2", "profx_r02d_C2.log", 1500)
print("total %.0f s" % (time.time() - T0))
