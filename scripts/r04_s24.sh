#!/bin/bash
# final tree of round 4 (commit e447685ecda5): PMC sessions, GPU suite, smoke, every bench line
export GIT_HEAD=e447685ecda5
O=gpurun_out/r04v
mkdir -p $O
for t in "C2 --workload C2" "C2j --workload C2 --map-jitter 0.045" "C3 --workload C3" "C5 --workload C5"; do
  set -- $t; tag=$1; shift
  bash profiles/run_profiles.sh r04z_$tag "$@" 2>&1 | tail -2
done
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for w in C2 C3 C1 C4 C5; do
  S=""; if [ $w = C4 ] || [ $w = C5 ]; then S="--steps 5 --warmup 2"; fi
  timeout 900 python bench.py --workload $w $S > $O/bench_$w.out 2> $O/bench_$w.err; tail -1 $O/bench_$w.out > $O/bench_$w.json
done
python - <<'PY'
import json
for w in ("C2","C3","C1","C4","C5"):
    try:
        d=json.load(open("gpurun_out/r04v/bench_%s.json"%w)); r=d["roofline"]
        print(w,"ms %.4f"%d["ms_per_step"],"8d %.4f"%d["update_8d"]["ms_per_update"],"roof",r["bound"],r["frac"],r.get("counters_source"),"jit",(d.get("map_jitter") or {}).get("vs_lattice"),"mu",(d.get("map_update") or {}).get("wall_ms"),"ra",(d.get("route_a") or {}).get("ms_per_update"),"ms",(d.get("match_split") or {}).get("ms"))
    except Exception as e: print(w,"failed",e)
PY
