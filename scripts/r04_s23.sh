#!/bin/bash
O=gpurun_out/r04u
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_map_path.py tests/test_gpu_adapter.py tests/test_gpu_scan_prep.py -x -q 2>&1 | tail -4 | tee $O/tests.txt
timeout 900 python bench.py --workload C2 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/bench_C2.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04u/bench_C2.json"))
print("match_split", {k:v for k,v in d["match_split"].items() if k!="what"})
print("scan_prep", d["scan_preparation"]["ms"], "route_a", d["route_a"]["ms_per_update"])
PY
