#!/bin/bash
# round 6 session 10: the split pf::measure as two launches (lik_finalize folded into pf_partial_kernel, pf_reduce into every
# work-group of pf_apply_kernel) — the suite, then the bench lines
O=gpurun_out/r06o; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl" | tail -8
Q="--steps 20 --warmup 3 --no-extras --no-cpu-baseline"
run() { # name, options, bench args
  MCL3DL_HIP_OPTIONS="$2" timeout 900 python bench.py $3 2>$O/$1.err | tail -1 > $O/$1.json
  python - "$O/$1.json" "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels_ms_per_step"]
    print("%-20s ms/step %.4f 8d %s lik %.4f beam %.4f pf %.4f" % (sys.argv[2], d["ms_per_step"], d.get("ms_per_step_8d"), k["likelihood"], k["beam"], k["pf"]), flush=True)
except Exception as e:
    print(sys.argv[2], "no json", e)
PY
}
Q8="--steps 20 --warmup 3 --no-cpu-baseline"
run C2_a "" "--workload C2 $Q8"
run C2_b "" "--workload C2 $Q8"
run C3 "" "--workload C3 $Q"
run C4s "" "--workload C4 --particles 32768 $Q"
run s4096x96 "" "--workload C2 --scan-points 96 $Q"
run s4096x2048_s0 "strict_order=0" "--workload C2 --scan-points 2048 $Q"
