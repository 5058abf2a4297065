#!/bin/bash
O=gpurun_out/r04j
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest_all.log; head -4 $O/pytest_all.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench.err | tail -1 > $O/bench_C2.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04j/bench_C2.json"))
print("value %.4g ms/step %.4f  8d %.4f pinned %.4f" % (d["value"], d["ms_per_step"], d["update_8d"]["ms_per_update"], d["update_8d"]["ms_per_update_page_locked_arrays"]))
print("map_update", json.dumps(d.get("map_update")))
print("map_jitter", d["map_jitter"]["likelihood_ms"], d["map_jitter"]["vs_lattice"])
print("scan_prep", d["scan_preparation"]["ms"], "filter_iteration", d.get("filter_iteration"), "resample", d["resample"]["ms"])
PY
