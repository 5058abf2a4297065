#!/bin/bash
# r03 session 25: does the slow host-buffer update of the C1 / C3 lines of session r03z reproduce?
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03z2; mkdir -p $OUT
for rep in 1 2; do
  python bench.py --workload C1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/C1_$rep.json
  python bench.py --workload C3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/C3_$rep.json
  python bench.py --workload C1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $OUT/C1ne_$rep.json
  python bench.py --workload C3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $OUT/C3ne_$rep.json
done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03z2/*.json")):
    d=json.load(open(f)); print("%-10s"%f.split('/')[-1][:-5], "step %.4f"%d["ms_per_step"], "8d %.4f"%d["update_8d"]["ms_per_update"], "red", (d.get("post_update_reductions") or {}).get("ms"), "resample", (d.get("resample") or {}).get("ms"))
P
