#!/bin/bash
# r03 session 16: the full GPU suite on the current tree, the default bench line, the C3/C5/C1/jitter lines, scan-prep kernel stats
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03q; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python bench.py 2>$OUT/C2_default.err | tail -1 > $OUT/C2_default.json
python bench.py --workload C2 --map-jitter 0.045 --no-extras --no-cpu-baseline 2>$OUT/C2j045.err | tail -1 > $OUT/C2j045.json
python bench.py --workload C3 --no-cpu-baseline 2>$OUT/C3_full.err | tail -1 > $OUT/C3_full.json
python bench.py --workload C1 --no-cpu-baseline 2>$OUT/C1_full.err | tail -1 > $OUT/C1_full.json
python bench.py --workload C5 --particles 8192 --no-extras --no-cpu-baseline 2>$OUT/C5_shard.err | tail -1 > $OUT/C5_shard.json
python bench.py --workload C4 --particles 32768 --no-extras --no-cpu-baseline 2>$OUT/C4_shard.err | tail -1 > $OUT/C4_shard.json
python bench.py --workload C2 --strict-order 1 --no-extras --no-cpu-baseline 2>$OUT/C2_strict.err | tail -1 > $OUT/C2_strict.json
for sh in "64 96" "4096 96" "4096 512" "4096 2048" "500 300"; do set -- $sh
  python bench.py --workload C2 --particles $1 --scan-points $2 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/shape_$1x$2.json; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prep_stats -o prep -- python scripts/time_scan_prep.py 50 > $OUT/prep_under_rocprof.log 2>&1
python scripts/time_scan_prep.py 50 > $OUT/prep_plain.log 2>&1; tail -3 $OUT/prep_plain.log
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03q/*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], "%.4g"%d["value"], "%.4f"%d["ms_per_step"], d["kernels_ms_per_step"], d.get("value_8d"))
    except Exception as e: print(f, "failed", e)
P
