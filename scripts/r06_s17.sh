#!/bin/bash
# round 6 session 17: the caller-order replay with the recurrence run by a whole wavefront per particle (lik_strict_sum_wave_kernel):
# the tests of the exact modes, then A/B on one box (MCL3DL_REPLAY_AB=0: lik_strict_sum_rows_kernel)
O=gpurun_out/r06v; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_c4c5.py tests/test_gpu_strict_chunks.py tests/test_gpu_fuzz.py -m gpu -q 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl" | tail -3
run() { # name, switch, bench args
  MCL3DL_REPLAY_AB="$2" timeout 900 python bench.py $3 2>$O/$1.err | tail -1 > $O/$1.json
  python - "$O/$1.json" "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels_ms_per_step"]
    print("%-20s ms/step %.4f 8d %s lik %.4f beam %.4f pf %.4f err %s" % (sys.argv[2], d["ms_per_step"], d.get("ms_per_step_8d"), k["likelihood"], k["beam"], k["pf"], d["result_check"].get("max_rel_err_vs_cpu")), flush=True)
except Exception as e:
    print(sys.argv[2], "no json", e)
PY
}
Q="--no-extras --no-cpu-baseline"
for r in 1 2; do
for v in 0 1; do
  run C2s1_$v$r $v "--workload C2 --strict-order 1 --steps 40 --warmup 5 $Q"
  run C5s_$v$r $v "--workload C5 --particles 8192 --steps 10 --warmup 2 $Q"
done
done
for v in 0 1; do
  run C5q_$v $v "--workload C5 --particles 32768 --steps 6 --warmup 2 $Q"
  run p4096x4096_$v $v "--workload C2 --scan-points 4096 --steps 40 --warmup 5 $Q"
  run p1024x16384s1_$v $v "--workload C2 --particles 1024 --strict-order 1 --steps 40 --warmup 5 $Q"
done
