#!/bin/bash
# round 6, the final sessions (one gpurun call per part: profiles are long). usage: r05_final.sh <part>
#   prof_a : kernel stats + PMC passes of C1, C2, C2 with in-kernel sums (C2c), C2 on the map of centroids (C2j), C3
#   prof_b : ... of C4, C5, C5 with in-kernel sums (C5c)
#   bench  : the bench line of every BASELINE configuration (+ C2 / C5 with strict_order 3, C2 with strict_order 1) and the suite
part=${1:-bench}
O=gpurun_out/r06z; mkdir -p $O
prof() { # tag, bench args
  bash profiles/run_profiles.sh r06z_$1 $2 > $O/prof_$1.log 2>&1
  cp gpurun_out/prof_r06z_$1/r06z_$1_kernel_stats.csv gpurun_out/prof_r06z_$1/r06z_$1_pmc_summary.csv $O/ 2>/dev/null
  echo "profile $1: $(tail -1 $O/prof_$1.log)"
}
line() { # name, bench args
  timeout 1500 python bench.py $2 2>$O/bench_$1.err | tail -1 > $O/bench_$1.json
  python - "$O/bench_$1.json" "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels_ms_per_step"]; r=d["roofline"]
    print("%-10s value %.4g ms/step %.4f lik %.4f beam %.4f pf %.4f 8d %s frac %s (%s) err %s" % (
        sys.argv[2], d["value"], d["ms_per_step"], k["likelihood"], k["beam"], k["pf"],
        ("%.4f" % d["update_8d"]["ms_per_update"]) if "update_8d" in d else "-", r["frac"], r["bound"],
        d["result_check"].get("max_rel_err_vs_cpu")), flush=True)
except Exception as e:
    print(sys.argv[2], "no json", e)
PY
}
case $part in
  prof_a)
    prof C1 "--workload C1"
    prof C2 "--workload C2"
    prof C2c "--workload C2 --strict-order 3"
    prof C2j "--workload C2 --map-jitter 0.045"
    prof C3 "--workload C3" ;;
  prof_b)
    prof C4 "--workload C4"
    prof C5 "--workload C5"
    prof C5c "--workload C5 --strict-order 3" ;;
  bench)
    line C2 ""
    line C1 "--workload C1"
    line C3 "--workload C3"
    line C4 "--workload C4"
    line C5 "--workload C5"
    line C5c "--workload C5 --strict-order 3"
    line C2c "--workload C2 --strict-order 3"
    line C2s1 "--workload C2 --strict-order 1 --no-extras"
    timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log ;;
esac
