#!/bin/bash
# full GPU suite + the bench lines of every workload + the C5 PMC session (commit c4cf6593570d)
export GIT_HEAD=c4cf6593570d
O=gpurun_out/r04q
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
for w in C2 C3 C1; do
  timeout 900 python bench.py --workload $w > $O/bench_$w.out 2> $O/bench_$w.err; tail -1 $O/bench_$w.out > $O/bench_$w.json; echo "bench $w rc=$?"
done
timeout 900 python bench.py --workload C5 --steps 5 --warmup 2 > $O/bench_C5.out 2> $O/bench_C5.err; tail -1 $O/bench_C5.out > $O/bench_C5.json
timeout 900 python bench.py --workload C4 --steps 5 --warmup 2 > $O/bench_C4.out 2> $O/bench_C4.err; tail -1 $O/bench_C4.out > $O/bench_C4.json
bash profiles/run_profiles.sh r04z_C5 --workload C5 2>&1 | tail -3
python - <<'PY'
import json
for w in ("C2","C3","C1","C5","C4"):
    try:
        d=json.load(open("gpurun_out/r04q/bench_%s.json"%w))
        r=d["roofline"]
        print(w,"ms/step %.4f"%d["ms_per_step"],"value %.4g"%d["value"],"8d",d.get("update_8d",{}).get("ms_per_update"),"roofline",r["bound"],r["frac"],"jit",(d.get("map_jitter") or {}).get("vs_lattice"),"mu",(d.get("map_update") or {}).get("wall_ms"),(d.get("map_update") or {}).get("first_measure_after_update_ms"),"ra",(d.get("route_a") or {}).get("ms_per_update"))
    except Exception as e:
        print(w,"failed",e)
PY
