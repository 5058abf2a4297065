#!/bin/bash
# (the option strict_pipeline it sets existed only in that session: the pipelined replay was measured and reverted, profiles/r06i_replay_pipeline_ab.txt)
# round 6 session 7: the caller-order replay pipelined over particle batches, the replay of batch b - 1 by
# the light (256-thread, 16 KB) form of the replay kernel beside the evaluation of batch b — against one replay behind everything
O=gpurun_out/r06i; mkdir -p $O
Q="--steps 20 --warmup 3 --no-extras --no-cpu-baseline"
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
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_strict_chunks.py tests/test_gpu_c4c5.py -m gpu -q -x 2>&1 | tail -3
run C2_fp64 "strict_order=0" "--workload C2 $Q"
for p in 0 1 2 4 8 16; do run C2_s1_pipe$p "strict_pipeline=$p" "--workload C2 --strict-order 1 $Q"; done
run C5s_fp64 "strict_order=0" "--workload C5 --particles 8192 $Q"
for p in 0 1 4 8; do run C5s_pipe$p "strict_pipeline=$p" "--workload C5 --particles 8192 $Q"; done
run s4096x4096_fp64 "strict_order=0" "--workload C2 --scan-points 4096 $Q"
for p in 0 1 2 4; do run s4096x4096_pipe$p "strict_pipeline=$p" "--workload C2 --scan-points 4096 $Q"; done
run C5_fp64 "strict_order=0" "--workload C5 $Q"
for p in 0 1; do run C5_pipe$p "strict_pipeline=$p" "--workload C5 $Q"; done
