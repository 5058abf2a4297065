#!/usr/bin/env python3
"""Round-2 GPU session 3: whole -m gpu suite (new: scan preparation, map path, cooperative fetch), A/B of the
quad-cooperative record fetch (lik_coop 0 / 1) at every shape, profiles of the winner."""
import json
import os
import subprocess
import time

OUT = "gpurun_out/r02c"
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


def bench(name, args, timeout=400):
    sh("python bench.py %s 2>%s/%s.err | tail -1 > %s/%s.json" % (args, OUT, name, OUT, name), name + ".log", timeout)
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


sh("python -m pytest tests -m gpu -q -x 2>&1 | tail -60", "pytest.log", 1800)

quick = "--steps 20 --warmup 3 --no-extras --no-cpu-baseline"
shapes = [("C2", "--workload C2"), ("C3", "--workload C3"), ("C5", "--workload C5 --particles 8192"),
          ("C4", "--workload C4 --particles 32768"), ("C4_8pt", "--workload C4 --scan-points 8"),
          ("C2j", "--workload C2 --map-jitter 0.045"), ("C2w5", "--workload C2 --dist-weight-z 5"),
          ("C2_64p", "--workload C2 --particles 64"), ("C2_512pts", "--workload C2 --scan-points 512"),
          ("C2_96pts", "--workload C2 --scan-points 96"), ("C1", "--workload C1")]
for name, a in shapes:
    for coop in (0, 1):
        bench("%s_coop%d" % (name, coop), "%s --lik-coop %d %s" % (a, coop, quick), 600)
bench("C2_full", "--workload C2", 900)
bench("C3_full", "--workload C3 --no-cpu-baseline", 900)
tag = "r02c_C2"
sh("bash profiles/run_profiles.sh %s --workload C2" % tag, "prof_%s.log" % tag, 1500)
sh("bash profiles/run_pmc_extra.sh %s --workload C2" % tag, "profx_%s.log" % tag, 1500)
print("total %.0f s" % (time.time() - T0))
