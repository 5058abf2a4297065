#!/bin/bash
# r03 session 15: timing events without the system-scope fence (the default event's L2 writeback + invalidate made the
# bracketed kernel start cold): bench lines + a kernel trace of the same command to compare event and rocprof durations
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03p; mkdir -p $OUT
python bench.py --workload C2 --no-cpu-baseline 2>$OUT/C2.err | tail -1 > $OUT/C2.json
python bench.py --workload C2 --map-jitter 0.045 --no-extras --no-cpu-baseline 2>$OUT/C2j.err | tail -1 > $OUT/C2j.json
python bench.py --workload C3 --no-extras --no-cpu-baseline 2>$OUT/C3.err | tail -1 > $OUT/C3.json
python bench.py --workload C1 --no-extras --no-cpu-baseline 2>$OUT/C1.err | tail -1 > $OUT/C1.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o C2 -- python bench.py --workload C2 --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $OUT/trace.log 2>&1
python - <<'P'
import json
for n in ("C2","C2j","C3","C1"):
    try:
        d=json.load(open("gpurun_out/r03p/%s.json"%n)); print(n, d["value"], d["ms_per_step"], d["kernels_ms_per_step"], d.get("value_8d"))
    except Exception as e: print(n, "failed", e)
P
