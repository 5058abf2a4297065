#!/usr/bin/env python3
"""One GPU session through `gpurun`: the -m gpu suite, the bench line of every workload / shape, optional profiles — the
parameterised form of the seven near-identical round-2 session drivers (r02_run1 .. r02_run9, `git log` has them).

  gpurun --timeout 3000 -- 'python scripts/gpu_session.py r03m --suite --workloads --shapes --profiles C2 C3'

  <tag>          results go to gpurun_out/<tag>/ (copy what should be judged into profiles/)
  --suite        python -m pytest tests -m gpu (+ the C4 / C5 tests with their worst-case print-outs)
  --workloads    bench lines: C2 (full), jittered C2, C3, C3 stress, C1, C2 strict, C2 with the RCCL call, C4 shards, C5 shard
  --shapes       bench lines of the launch-bound shapes (4096 x 96 ... 100 000 x 96)
  --prep         scripts/time_scan_prep.py plain and under rocprofv3 --kernel-trace --stats (launch count of the scan preparation)
  --profiles W.. profiles/run_profiles.sh for the named workloads (kernel stats + separate PMC passes)"""
import argparse
import json
import os
import subprocess
import time

ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("--suite", action="store_true")
ap.add_argument("--workloads", action="store_true")
ap.add_argument("--shapes", action="store_true")
ap.add_argument("--profiles", nargs="*", default=[])
ap.add_argument("--prep", action="store_true")
args = ap.parse_args()
OUT = "gpurun_out/" + args.tag
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


def bench(name, bargs, timeout=600):
    sh("python bench.py %s 2>%s/%s.err | tail -1 > %s/%s.json" % (bargs, OUT, name, OUT, name), name + ".log", timeout)
    try:
        d = json.load(open("%s/%s.json" % (OUT, name)))
        k = d["kernels_ms_per_step"]
        print("    %-22s value %.4g  ms/step %.4f  lik %.4f  beam %.4f  pf %.4f  8d %s" % (
            name, d["value"], d["ms_per_step"], k["likelihood"], k["beam"], k["pf"],
            ("%.4f" % d["update_8d"]["ms_per_update"]) if "update_8d" in d else "-"), flush=True)
        return d
    except Exception as e:  # noqa: BLE001
        print("    %s: no JSON (%s)" % (name, e), flush=True)
        return None


if args.suite:
    sh("python -m pytest tests -m gpu -q 2>&1 | grep -E 'passed|failed|FAILED|Error'", "pytest.log", 2400)
    sh("python -m pytest tests/test_gpu_c4c5.py -q -s 2>&1 | grep -E 'worst|passed|failed'", "c4c5.log", 900)
quick = "--steps 20 --warmup 3 --no-extras --no-cpu-baseline"
if args.workloads:
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
if args.shapes:
    for p, s in ((64, 96), (4096, 96), (4096, 512), (4096, 1000), (4096, 2048), (64, 16384), (100000, 96), (500, 300)):
        bench("shape_%dx%d" % (p, s), "--workload C2 --particles %d --scan-points %d %s" % (p, s, quick), 300)
for w in args.profiles:
    extra = " --particles 8192" if w == "C5" else ""
    wl = "C2 --map-jitter 0.045" if w == "C2j" else w   # C2j = C2 on the map of displaced points (voxel-filter centroids)
    sh("bash profiles/run_profiles.sh %s_%s --workload %s%s" % (args.tag, w, wl, extra), "prof_%s.log" % w, 900)
if args.prep:
    sh("python scripts/time_scan_prep.py 50", "prep_plain.log", 300)
    sh("cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv "
       "-d %s/prep_stats -o prep -- python scripts/time_scan_prep.py 50" % OUT, "prep_under_rocprof.log", 400)
print("total %.0f s" % (time.time() - T0))
