#!/bin/bash
# round 6 session 3: caller-order rows in the per-particle kernels — correctness + cost (r06_rows_check.py), the bench lines of
# C1 / C2 / small shapes, then the whole GPU suite
O=gpurun_out/r06d; mkdir -p $O
PYTHONPATH=. timeout 600 python scripts/r06_rows_check.py > $O/rows_check.txt 2>&1; tail -22 $O/rows_check.txt
Q="--steps 20 --warmup 3 --no-extras --no-cpu-baseline"
run() { # name, options, bench args
  MCL3DL_HIP_OPTIONS="$2" timeout 600 python bench.py $3 2>$O/$1.err | tail -1 > $O/$1.json
  python - "$O/$1.json" "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels_ms_per_step"]
    print("%-28s ms/step %.4f lik %.4f beam %.4f pf %.4f one-launch %.4f" % (sys.argv[2], d["ms_per_step"], k["likelihood"], k["beam"], k["pf"], k.get("update_one_launch", 0)), flush=True)
except Exception as e:
    print(sys.argv[2], "no json", e)
PY
}
run C1_default "" "--workload C1 $Q"
run C1_strict0 "strict_order=0" "--workload C1 $Q"
run C2_default "" "--workload C2 $Q"
run s64x96 "" "--workload C2 --particles 64 --scan-points 96 $Q"
run s64x96_s0 "strict_order=0" "--workload C2 --particles 64 --scan-points 96 $Q"
run s4096x512 "" "--workload C2 --scan-points 512 $Q"
run s4096x512_s0 "strict_order=0" "--workload C2 --scan-points 512 $Q"
run s4096x96 "" "--workload C2 --scan-points 96 $Q"
run s4096x96_s0 "strict_order=0" "--workload C2 --scan-points 96 $Q"
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.log; tail -15 $O/pytest_gpu.log
