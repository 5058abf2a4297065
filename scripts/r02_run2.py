#!/usr/bin/env python3
"""[historical: the lik_ilp / lik_trim options it sweeps were folded into lik_coop afterwards] Round-2 GPU session 2: whole -m gpu suite, extended VALU micro-benchmark, A/B of {SLP vectoriser on/off} x {lik_ilp} x
{lik_trim} at C2, the winner at C3 / C4 / C5 / jittered map, profiles (kernel stats + PMC incl. VALU class counters)."""
import json
import os
import subprocess
import time

OUT = "gpurun_out/r02b"
os.makedirs(OUT, exist_ok=True)
T0 = time.time()
SLP_LIB = os.path.abspath("mcl_3dl_amd/variants/libmcl3dl_hip_slp.so")


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


def bench(name, args, timeout=400, env=None):
    sh("python bench.py %s 2>%s/%s.err | tail -1 > %s/%s.json" % (args, OUT, name, OUT, name), name + ".log", timeout, env)
    try:
        d = json.load(open("%s/%s.json" % (OUT, name)))
        k = d["kernels_ms_per_step"]
        print("    %-26s value %.4g  ms/step %.4f  lik %.4f  beam %.4f  pf %.4f  8d %s" % (
            name, d["value"], d["ms_per_step"], k["likelihood"], k["beam"], k["pf"],
            ("%.4f" % d["update_8d"]["ms_per_update"]) if "update_8d" in d else "-"), flush=True)
        return d
    except Exception as e:  # noqa: BLE001
        print("    %s: no JSON (%s)" % (name, e), flush=True)
        return None


sh("python -m pytest tests -m gpu -q 2>&1 | tail -40", "pytest.log", 1800)
sh("./profiles/valu_microbench.bin", "valu_microbench.txt", 300)

quick = "--steps 20 --warmup 3 --no-extras --no-cpu-baseline"
res = {}
for lib, env in (("noslp", None), ("slp", {"MCL3DL_HIP_LIB": SLP_LIB})):
    for ilp in (0, 1):
        for trim in (0, 1):
            d = bench("C2_%s_ilp%d_trim%d" % (lib, ilp, trim), "--workload C2 --lik-ilp %d --lik-trim %d %s" % (ilp, trim, quick),
                      env=env)
            if d and lib == "noslp":
                res[(ilp, trim)] = d["kernels_ms_per_step"]["likelihood"]
best = min(res, key=res.get) if res else (0, 0)
print("best (lik_ilp, lik_trim) with the shipped build:", best, flush=True)
v = "--lik-ilp %d --lik-trim %d" % best
for cfgname, a in (("C3", "--workload C3"), ("C5", "--workload C5 --particles 8192"), ("C4", "--workload C4 --particles 32768"),
                   ("C2j", "--workload C2 --map-jitter 0.045"), ("C2w5", "--workload C2 --dist-weight-z 5"),
                   ("C2_g8", "--workload C2 --lik-group 8"), ("C2_64p", "--workload C2 --particles 64"),
                   ("C2_2048pts", "--workload C2 --scan-points 2048")):
    bench("%s_best" % cfgname, "%s %s %s" % (a, v, quick), 600)
    if cfgname in ("C3", "C5", "C2j"):
        bench("%s_base" % cfgname, "%s --lik-ilp 0 --lik-trim 0 %s" % (a, quick), 600)
# full lines
bench("C2_full", "--workload C2 %s" % v, 900)
bench("C1_full", "--workload C1 %s" % v, 600)
bench("C3_full", "--workload C3 %s --no-cpu-baseline" % v, 900)

tag = "r02b_C2"
sh("bash profiles/run_profiles.sh %s --workload C2 %s" % (tag, v), "prof_%s.log" % tag, 1500)
sh("bash profiles/run_pmc_extra.sh %s --workload C2 %s" % (tag, v), "profx_%s.log" % tag, 1500)
print("total %.0f s" % (time.time() - T0))
