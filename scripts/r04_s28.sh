#!/bin/bash
# last session of round 4: GPU suite, smoke, every bench line (counters: profiles/r04z_*_pmc_summary.csv)
O=gpurun_out/r04z_final
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for w in C2 C3 C1 C4 C5; do
  S=""; if [ $w = C4 ] || [ $w = C5 ]; then S="--steps 5 --warmup 2"; fi
  timeout 900 python bench.py --workload $w $S > $O/bench_$w.out 2> $O/bench_$w.err; tail -1 $O/bench_$w.out > $O/bench_$w.json
done
python - <<'PY'
import json
for w in ("C2","C3","C1","C4","C5"):
    try:
        d=json.load(open("gpurun_out/r04z_final/bench_%s.json"%w)); r=d["roofline"]
        print(w,"ms %.4f"%d["ms_per_step"],"8d %.4f"%d["update_8d"]["ms_per_update"],"roof",r["bound"],r["frac"],"jit",(d.get("map_jitter") or {}).get("vs_lattice"),"mu",(d.get("map_update") or {}).get("wall_ms"),"ra",(d.get("route_a") or {}).get("ms_per_update"),"split",(d.get("match_split") or {}).get("ms"),"ipg",(d.get("in_process_group") or {}).get("ms_per_update"))
    except Exception as e: print(w,"failed",e)
PY
