#!/bin/bash
# round 6 session 14: both models in one launch (lik_beam_kernel: the tiled kernel's and the beam kernel's work-groups interleaved):
# suite, then A/B on one box (MCL3DL_MERGE_AB=0: two launches)
O=gpurun_out/r06s; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl" | tail -5
run() { # name, switch, bench args
  MCL3DL_MERGE_AB="$2" timeout 900 python bench.py $3 2>$O/$1.err | tail -1 > $O/$1.json
  python - "$O/$1.json" "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels_ms_per_step"]
    print("%-20s ms/step %.4f 8d %s lik %.4f beam %.4f pf %.4f err %s" % (sys.argv[2], d["ms_per_step"], d.get("ms_per_step_8d"), k["likelihood"], k["beam"], k["pf"], d["result_check"].get("max_rel_err_vs_cpu")), flush=True)
except Exception as e:
    print(sys.argv[2], "no json", e)
PY
}
Q="--steps 40 --warmup 5 --no-extras --no-cpu-baseline"
for r in 1 2 3; do
  run C3_two_$r 0 "--workload C3 $Q"
  run C3_one_$r 1 "--workload C3 $Q"
done
for v in 0 1; do
  run C3b48_$v $v "--workload C3 --beam-points 48 $Q"
  run C3b4096_$v $v "--workload C3 --beam-points 4096 $Q"
  run C5s_$v $v "--workload C5 --particles 8192 --steps 10 --warmup 2 --no-extras --no-cpu-baseline"
  run C5sc_$v $v "--workload C5 --particles 8192 --strict-order 3 --steps 10 --warmup 2 --no-extras --no-cpu-baseline"
  run C3j_$v $v "--workload C3 --map-jitter 0.045 $Q"
done
