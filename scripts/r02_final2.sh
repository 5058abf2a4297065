#!/bin/bash
# One box: the default bench line; the same workload without the extras under rocprofv3 --kernel-trace --stats and plain
# (the extras re-run the likelihood kernel on a jittered map under the same kernel name, which would mix into rocprofv3's
# average); PMC passes of C2 on the jittered map with the current (128-byte record) index.
OUT=gpurun_out/r02i
mkdir -p $OUT
python bench.py 2>$OUT/C2_default.err | tail -1 > $OUT/C2_default.json
python bench.py --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/C2_noextras.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -o r02i_C2 -- python bench.py --no-extras --no-cpu-baseline > $OUT/C2_noextras_under_rocprof.log 2>&1
bash profiles/run_profiles.sh r02i_C2j --workload C2 --map-jitter 0.045 > /dev/null 2>&1
python - <<P
import json,csv
for n in ("C2_default","C2_noextras"):
    d=json.load(open("$OUT/%s.json"%n)); k=d["kernels_ms_per_step"]; r=d["roofline"]
    print("%-14s value %.4g ms/step %.4f lik %.4f pf %.4f | %s frac %.3f src %s | %s" % (n,d["value"],d["ms_per_step"],k["likelihood"],k["pf"], r["bound"], r["frac"], r["counters_source"], {a:round(b["frac"],3) for a,b in r["resources"].items()}))
for r in csv.DictReader(open("$OUT/stats2/r02i_C2_kernel_stats.csv")):
    if "likelihood_tiled" in r["Name"]: print("rocprofv3 (same command, same box): likelihood_tiled avg %.4f ms over %s launches, min %.4f" % (float(r["AverageNs"])/1e6, r["Calls"], float(r["MinNs"])/1e6))
P
