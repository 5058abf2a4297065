#!/bin/bash
# round 5 session 7: the caller-order replay in chunks (tests, then C5 shard / C5 / C2-strict timings against one piece)
O=gpurun_out/r05g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_strict_chunks.py tests/test_gpu_c4c5.py tests/test_gpu_parity.py tests/test_gpu_scan_prep.py -x -q 2>&1 | tail -25 > $O/tests.log; tail -8 $O/tests.log
Q="--steps 10 --warmup 2 --no-extras --no-cpu-baseline"
run() { # name, options, bench args
  MCL3DL_HIP_OPTIONS="$2" timeout 900 python bench.py $3 2>$O/$1.err | tail -1 > $O/$1.json
  python - "$O/$1.json" "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels_ms_per_step"]
    print("%-28s ms/step %.4f lik %.4f beam %.4f pf %.4f" % (sys.argv[2], d["ms_per_step"], k["likelihood"], k["beam"], k["pf"]), flush=True)
except Exception as e:
    print(sys.argv[2], "no json", e)
PY
}
run C5s_one "strict_chunk=0" "--workload C5 --particles 8192 $Q"
run C5s_chunk16k "" "--workload C5 --particles 8192 $Q"
run C5s_chunk8k "strict_chunk=8192" "--workload C5 --particles 8192 $Q"
run C5s_chunk32k "strict_chunk=32768" "--workload C5 --particles 8192 $Q"
run C5s_fp64 "" "--workload C5 --particles 8192 --strict-order 0 $Q"
run C5s_chain "" "--workload C5 --particles 8192 --strict-order 3 $Q"
run C5_one "strict_chunk=0" "--workload C5 $Q"
run C5_chunk16k "" "--workload C5 $Q"
run C5_chain "" "--workload C5 --strict-order 3 $Q"
run C5_fp64 "" "--workload C5 --strict-order 0 $Q"
