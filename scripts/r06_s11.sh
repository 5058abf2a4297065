#!/bin/bash
# round 6 session 11: lik_finalize_kernel folded into the first pf::measure launch WITH lik_finalize's parallelism (lik_pf_partial_kernel):
# the suite, then an A/B on one box (MCL3DL_TAIL_AB=0: the five launches of 1461162; 1: four; 2: three — pf_reduce inside pf_apply too)
O=gpurun_out/r06p; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl" | tail -8
MCL3DL_TAIL_AB=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_update_staged.py tests/test_gpu_group.py tests/test_gpu_c4c5.py -m gpu -q -x 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl" | tail -4
run() { # name, AB, bench args
  MCL3DL_TAIL_AB="$2" timeout 900 python bench.py $3 2>$O/$1.err | tail -1 > $O/$1.json
  python - "$O/$1.json" "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels_ms_per_step"]
    print("%-20s ms/step %.4f 8d %s lik %.4f beam %.4f pf %.4f" % (sys.argv[2], d["ms_per_step"], d.get("ms_per_step_8d"), k["likelihood"], k["beam"], k["pf"]), flush=True)
except Exception as e:
    print(sys.argv[2], "no json", e)
PY
}
Q8="--steps 40 --warmup 5 --no-cpu-baseline"
for r in 1 2 3; do
  run C2_ab0_$r 0 "--workload C2 $Q8"
  run C2_ab1_$r 1 "--workload C2 $Q8"
  run C2_ab2_$r 2 "--workload C2 $Q8"
done
for v in 0 1 2; do
  run C3_ab$v $v "--workload C3 $Q8"
done
for v in 0 1 2; do
  run C4s_ab$v $v "--workload C4 --particles 32768 --steps 20 --warmup 3 --no-extras --no-cpu-baseline"
done
