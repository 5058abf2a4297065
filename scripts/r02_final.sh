#!/bin/bash
# Final check of the round on ONE box: the whole -m gpu suite, smoke, the default bench line, and the SAME command under
# rocprofv3 --kernel-trace --stats (so that the hipEvent average of the roofline kernel and rocprofv3's can be compared on
# the same GPU), then the C3 / C5 lines.
OUT=gpurun_out/r02i
mkdir -p $OUT
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -3 | tee $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py 2>$OUT/C2_default.err | tail -1 > $OUT/C2_default.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r02i_C2_default -- python bench.py > $OUT/C2_default_under_rocprof.log 2>&1
tail -1 $OUT/C2_default_under_rocprof.log > $OUT/C2_default_under_rocprof.json
python bench.py --workload C3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/C3_full.json
python bench.py --workload C5 --particles 8192 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $OUT/C5_shard.json
python bench.py --workload C2 --map-jitter 0.045 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $OUT/C2j045.json
python bench.py --workload C1 2>/dev/null | tail -1 > $OUT/C1_full.json
python - <<P
import json,csv
for n in ("C2_default","C2_default_under_rocprof","C3_full","C5_shard","C2j045","C1_full"):
    try:
        d=json.load(open("$OUT/%s.json"%n)); k=d["kernels_ms_per_step"]; r=d["roofline"]
        print("%-26s value %.4g ms/step %.4f lik %.4f beam %.4f pf %.4f | %s frac %s" % (n,d["value"],d["ms_per_step"],k["likelihood"],k["beam"],k["pf"], r["bound"], r["frac"]))
    except Exception as e: print(n,"failed",e)
for r in csv.DictReader(open("$OUT/stats/r02i_C2_default_kernel_stats.csv")):
    if "likelihood_tiled" in r["Name"]: print("rocprofv3: likelihood_tiled avg %.4f ms over %s launches" % (float(r["AverageNs"])/1e6, r["Calls"]))
P
