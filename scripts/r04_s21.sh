#!/bin/bash
O=gpurun_out/r04s
mkdir -p $O
timeout 900 python bench.py --workload C2 > $O/bench_C2.out 2> $O/bench_C2.err; tail -1 $O/bench_C2.out > $O/bench_C2.json
timeout 900 python bench.py --workload C5 --steps 5 --warmup 2 > $O/bench_C5.out 2> $O/bench_C5.err; tail -1 $O/bench_C5.out > $O/bench_C5.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04s/bench_C2.json"))
print(json.dumps(d.get("dist_weight_shipped"),indent=1))
print("route_a",d["route_a"]["ms_per_update"],"8d",d["update_8d"]["ms_per_update"],"ms",d["ms_per_step"])
d=json.load(open("gpurun_out/r04s/bench_C5.json"))
r=d["roofline"]; print("C5", d["ms_per_step"], r["bound"], r["frac"], r.get("frac_vs_measured_peaks"), r["traffic"], r.get("counters_note"))
PY
