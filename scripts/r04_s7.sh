#!/bin/bash
O=gpurun_out/r04g
mkdir -p $O
timeout 900 python -X faulthandler bench.py --steps 20 --warmup 5 > $O/bench_C2.json 2> $O/bench_C2.err; echo "rc=$?"; tail -c 1500 $O/bench_C2.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r04g/bench_C2.json"))
print("value %.4g ms/step %.4f 8d %s" % (d["value"], d["ms_per_step"], json.dumps(d.get("update_8d"))[:900]))
print("kernels", d["kernels_ms_per_step"], "roofline frac", d["roofline"]["frac"], d["roofline"].get("counters_note"))
print("route_a", json.dumps(d.get("route_a"))[:400])
print("map_jitter", json.dumps(d.get("map_jitter"))[:300])
PY
