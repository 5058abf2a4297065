#!/bin/bash
# round 6 session 21: with the caller-order replay behind the tiled kernel, the beam kernel forked BEHIND the tiled kernel (beside the
# replay) instead of beside the tiled kernel (MCL3DL_BEAM_LATE=0)
O=gpurun_out/r06zy; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_c4c5.py tests/test_gpu_fullsize.py tests/test_gpu_launch_paths.py tests/test_gpu_strict_chunks.py -m gpu -q 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl" | tail -4
run() { # name, switch, bench args
  MCL3DL_BEAM_LATE="$2" timeout 900 python bench.py $3 2>$O/$1.err | tail -1 > $O/$1.json
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
  run C5s_early_$r 0 "--workload C5 --particles 8192 --steps 10 --warmup 2 $Q"
  run C5s_late_$r 1 "--workload C5 --particles 8192 --steps 10 --warmup 2 $Q"
  run p4096x4096+128_early_$r 0 "--workload C3 --scan-points 4096 --beam-points 128 --steps 30 --warmup 4 $Q"
  run p4096x4096+128_late_$r 1 "--workload C3 --scan-points 4096 --beam-points 128 --steps 30 --warmup 4 $Q"
done
run C5_early 0 "--workload C5 --steps 6 --warmup 2 $Q"
run C5_late 1 "--workload C5 --steps 6 --warmup 2 $Q"
