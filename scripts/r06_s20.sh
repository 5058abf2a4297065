#!/bin/bash
# round 6 session 20: the merged launch in the in-kernel-chain mode (strict_order 3)
O=gpurun_out/r06zz; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_chain.py tests/test_gpu_launch_paths.py tests/test_gpu_c4c5.py tests/test_gpu_fullsize.py tests/test_gpu_adapter.py tests/test_gpu_api_fuzz.py -m gpu -q 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl" | tail -5
run() { # name, overlap, bench args
  timeout 900 python bench.py $3 --overlap-models $2 2>$O/$1.err | tail -1 > $O/$1.json
  python - "$O/$1.json" "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels_ms_per_step"]
    print("%-24s ms/step %.4f 8d %s lik %.4f beam %.4f pf %.4f err %s" % (sys.argv[2], d["ms_per_step"], d.get("ms_per_step_8d"), k["likelihood"], k["beam"], k["pf"], d["result_check"].get("max_rel_err_vs_cpu")), flush=True)
except Exception as e:
    print(sys.argv[2], "no json", e)
PY
}
Q="--no-extras --no-cpu-baseline"
for r in 1 2; do
  run C3c_serial_$r 0 "--workload C3 --strict-order 3 --steps 40 --warmup 5 $Q"
  run C3c_merged_$r 1 "--workload C3 --strict-order 3 --steps 40 --warmup 5 $Q"
  run C5sc_serial_$r 0 "--workload C5 --particles 8192 --strict-order 3 --steps 10 --warmup 2 $Q"
  run C5sc_merged_$r 1 "--workload C5 --particles 8192 --strict-order 3 --steps 10 --warmup 2 $Q"
done
run C5c_merged 1 "--workload C5 --strict-order 3 --steps 6 --warmup 2 $Q"
