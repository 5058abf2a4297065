#!/usr/bin/env python3
"""[historical: option lik_ilp no longer exists] Round-2 GPU session 1 (run through gpurun from the repo root): the whole -m gpu suite, the VALU micro-benchmark, the
tiled-kernel variant sweep (option lik_ilp) at C2 / C3 / C5, and rocprofv3 stats + PMC passes for the baseline and the
best variant.  Everything lands in gpurun_out/r02a/; every child runs under a timeout."""
import json
import os
import subprocess
import sys
import time

OUT = "gpurun_out/r02a"
os.makedirs(OUT, exist_ok=True)
T0 = time.time()


def sh(cmd, log, timeout):
    t = time.time()
    try:
        with open(os.path.join(OUT, log), "w") as f:
            rc = subprocess.run(cmd, shell=True, stdout=f, stderr=subprocess.STDOUT, timeout=timeout).returncode
    except subprocess.TimeoutExpired:
        rc = -9
    print("[%6.0fs] rc=%s %.0fs  %s" % (time.time() - T0, rc, time.time() - t, cmd[:150]), flush=True)
    return rc


def bench(name, args, timeout=400):
    rc = sh("python bench.py %s 2>%s/%s.err | tail -1 > %s/%s.json" % (args, OUT, name, OUT, name), name + ".log", timeout)
    try:
        d = json.load(open("%s/%s.json" % (OUT, name)))
        print("    %-22s value %.4g  ms/step %.4f  lik %.4f ms  beam %.4f  pf %.4f" % (
            name, d["value"], d["ms_per_step"], d["kernels_ms_per_step"]["likelihood"], d["kernels_ms_per_step"]["beam"],
            d["kernels_ms_per_step"]["pf"]), flush=True)
        return d
    except Exception as e:  # noqa: BLE001
        print("    %s: no JSON (%s)" % (name, e), flush=True)
        return None


sh("python -m pytest tests -m gpu -x -q 2>&1 | tail -15", "pytest.log", 1500)
sh("./profiles/valu_microbench.bin", "valu_microbench.txt", 300)

quick = "--steps 20 --warmup 3 --no-extras --no-cpu-baseline"
res = {}
for ilp in (0, 1, 2, 3):
    res[ilp] = bench("C2_ilp%d" % ilp, "--workload C2 --lik-ilp %d %s" % (ilp, quick))
ok = {k: v for k, v in res.items() if v}
best = min(ok, key=lambda k: ok[k]["kernels_ms_per_step"]["likelihood"]) if ok else 0
print("best lik_ilp at C2:", best, flush=True)
for ilp in sorted(set([0, best])):
    bench("C3_ilp%d" % ilp, "--workload C3 --lik-ilp %d %s" % (ilp, quick))
    bench("C5_ilp%d" % ilp, "--workload C5 --particles 8192 --lik-ilp %d %s" % (ilp, quick), 600)
    if ilp == best:
        bench("C4_ilp%d" % ilp, "--workload C4 --particles 32768 --lik-ilp %d %s" % (ilp, quick), 600)
    bench("C2j_ilp%d" % ilp, "--workload C2 --map-jitter 0.045 --lik-ilp %d %s" % (ilp, quick))
for g in (8, 32):
    bench("C2_ilp%d_g%d" % (best, g), "--workload C2 --lik-ilp %d --lik-group %d %s" % (best, g, quick))
# full line (extras, CPU baseline, route A) with the best variant
bench("C2_full_ilp%d" % best, "--workload C2 --lik-ilp %d" % best, 900)

# profiles: kernel stats + PMC passes (separate rocprofv3 runs)
for ilp in sorted(set([0, best])):
    tag = "r02a_C2_ilp%d" % ilp
    sh("bash profiles/run_profiles.sh %s --workload C2 --lik-ilp %d --no-extras" % (tag, ilp), "prof_%s.log" % tag, 1500)
    if ilp == best:
        sh("bash profiles/run_pmc_extra.sh %s --workload C2 --lik-ilp %d --no-extras" % (tag, ilp), "profx_%s.log" % tag, 1500)
# what SQ_ACTIVE_INST_VALU counts per instruction: the micro-benchmark under the same counters
sh("cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU "
   "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d %s/pmc_microbench -o mb -- "
   "./profiles/valu_microbench.bin" % OUT, "pmc_microbench.log", 300)
print("total %.0f s" % (time.time() - T0))
