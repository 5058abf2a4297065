#!/bin/bash
# round 5 session 2: the in-kernel float-order chain (strict_order = 3): its tests, the whole GPU suite, A/B bench lines
O=gpurun_out/r05b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -15 > $O/chain_tests.log; tail -4 $O/chain_tests.log
Q="--steps 20 --warmup 3 --no-extras --no-cpu-baseline"
run() { # name, options, bench args
  MCL3DL_HIP_OPTIONS="$2" timeout 600 python bench.py $3 2>$O/$1.err | tail -1 > $O/$1.json
  python - "$O/$1.json" "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels_ms_per_step"]
    print("%-28s ms/step %.4f lik %.4f beam %.4f pf %.4f" % (sys.argv[2], d["ms_per_step"], k["likelihood"], k["beam"], k["pf"]), flush=True)
except Exception as e:
    print(sys.argv[2], "no json", e)
PY
}
run C2_s0 "" "--workload C2 --strict-order 0 $Q"
run C2_s3 "" "--workload C2 --strict-order 3 $Q"
run C2_s0b "" "--workload C2 --strict-order 0 $Q"
run C2_s3b "" "--workload C2 --strict-order 3 $Q"
run C2_s3_g8 "" "--workload C2 --strict-order 3 --lik-group 8 $Q"
run C2_s3_g4 "" "--workload C2 --strict-order 3 --lik-group 4 $Q"
run C2j_s0 "" "--workload C2 --map-jitter 0.045 --strict-order 0 $Q"
run C2j_s3 "" "--workload C2 --map-jitter 0.045 --strict-order 3 $Q"
run C3_s3 "" "--workload C3 --strict-order 3 $Q"
run P1k_s0 "" "--workload C2 --particles 1024 --strict-order 0 $Q"
run P1k_s3 "" "--workload C2 --particles 1024 --strict-order 3 $Q"
run P16k_s0 "" "--workload C2 --particles 16384 --strict-order 0 $Q"
run P16k_s3 "" "--workload C2 --particles 16384 --strict-order 3 $Q"
run C4s_s0 "" "--workload C4 --particles 32768 --strict-order 0 $Q"
run C4s_s3 "" "--workload C4 --particles 32768 --strict-order 3 $Q"
run C5s_s0 "" "--workload C5 --particles 8192 --strict-order 0 $Q"
run C5s_s2 "" "--workload C5 --particles 8192 --strict-order 2 $Q"
run C5s_s3 "" "--workload C5 --particles 8192 --strict-order 3 $Q"
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
