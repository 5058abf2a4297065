#!/bin/bash
# round 6 session 9: instruction trimming of the cooperative evaluation (own_word as selects, std::max as one v_max, the term as a
# select) — parity first, then the bench lines
O=gpurun_out/r06k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_defer.py tests/test_gpu_bound.py tests/test_gpu_chain.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -3
Q="--steps 20 --warmup 3 --no-extras --no-cpu-baseline"
run() { # name, options, bench args
  MCL3DL_HIP_OPTIONS="$2" timeout 900 python bench.py $3 2>$O/$1.err | tail -1 > $O/$1.json
  python - "$O/$1.json" "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels_ms_per_step"]
    print("%-28s ms/step %.4f lik %.4f beam %.4f pf %.4f one-launch %.4f" % (sys.argv[2], d["ms_per_step"], k["likelihood"], k["beam"], k["pf"], k.get("update_one_launch", 0)), flush=True)
except Exception as e:
    print(sys.argv[2], "no json", e)
PY
}
for rep in 1 2; do
run C2_$rep "" "--workload C2 $Q"
run C2j_$rep "" "--workload C2 --map-jitter 0.045 $Q"
done
run C2_s3 "" "--workload C2 --strict-order 3 $Q"
run C3 "" "--workload C3 $Q"
run C4s "" "--workload C4 --particles 32768 $Q"
run C5s "strict_order=0" "--workload C5 --particles 8192 $Q"
run C1 "" "--workload C1 $Q"
run s4096x512 "" "--workload C2 --scan-points 512 $Q"
