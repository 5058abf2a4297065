#!/bin/bash
# round 6 session 19: the per-particle likelihood kernel and the beam kernel in one launch (lik_particle_beam_kernel)
O=gpurun_out/r06y; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_launch_paths.py tests/test_gpu_pf_fused.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_rows.py tests/test_gpu_api_fuzz.py tests/test_gpu_update_staged.py tests/test_gpu_group.py -m gpu -q 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl" | tail -6
run() { # name, overlap, bench args
  timeout 900 python bench.py $3 --overlap-models $2 2>$O/$1.err | tail -1 > $O/$1.json
  python - "$O/$1.json" "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels_ms_per_step"]
    print("%-24s ms/step %.4f 8d %s lik %.4f beam %.4f pf %.4f" % (sys.argv[2], d["ms_per_step"], d.get("ms_per_step_8d"), k["likelihood"], k["beam"], k["pf"]), flush=True)
except Exception as e:
    print(sys.argv[2], "no json", e)
PY
}
Q="--steps 40 --warmup 5 --no-extras --no-cpu-baseline"
for r in 1 2; do
for shape in "4096 96 3" "2000 96 3" "1100 64 16" "8192 128 8" "1000 1000 16"; do
  set -- $shape
  run p$1x$2+$3_serial_$r 0 "--workload C3 --particles $1 --scan-points $2 --beam-points $3 $Q"
  run p$1x$2+$3_merged_$r 1 "--workload C3 --particles $1 --scan-points $2 --beam-points $3 $Q"
done
done
