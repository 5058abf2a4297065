#!/bin/bash
# r03 session 21: 16-bit Morton key (two radix passes), results of the host-buffer update in one D2H copy: suite, C2 / C3 lines, scan prep
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03x; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
python bench.py --workload C2 --no-cpu-baseline 2>$OUT/C2.err | tail -1 > $OUT/C2.json
python bench.py --workload C3 --no-cpu-baseline 2>$OUT/C3.err | tail -1 > $OUT/C3.json
python bench.py --workload C5 --particles 8192 --no-extras --no-cpu-baseline 2>$OUT/C5.err | tail -1 > $OUT/C5.json
python scripts/time_scan_prep.py 50 2>&1 | tail -2
python scripts/time_host_path.py 2>&1 | tail -12
python - <<'P'
import json
for n in ("C2","C3","C5"):
    d=json.load(open("gpurun_out/r03x/%s.json"%n)); print(n, "%.4g"%d["value"], "%.4f"%d["ms_per_step"], d["kernels_ms_per_step"], d.get("update_8d",{}).get("ms_per_update"), (d.get("scan_preparation") or {}).get("ms"), (d.get("route_a") or {}).get("ms_per_update"))
P
