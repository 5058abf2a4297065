#!/bin/bash
# round 6 session 15: did the refactor of the tiled kernel into a body + wrapper cost anything (this tree against the previous commit's
# library, same box, C2), the merged launch at G = 8 / 4 shapes, then the whole suite
O=gpurun_out/r06t; mkdir -p $O
run() { # name, lib, env, bench args
  MCL3DL_HIP_LIB="$2" MCL3DL_HIP_OPTIONS="$3" timeout 900 python bench.py $4 2>$O/$1.err | tail -1 > $O/$1.json
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
PREV=$PWD/mcl_3dl_amd/variants/libmcl3dl_hip_prev.so
for r in 1 2 3; do
  run C2_prev_$r "$PREV" "" "--workload C2 $Q"
  run C2_new_$r "" "" "--workload C2 $Q"
done
for shape in "2048 8192 256" "4096 16384 512"; do
  set -- $shape
  run p$1x$2+$3_two "" "overlap_models=0" "--workload C3 --particles $1 --scan-points $2 --beam-points $3 --overlap-models 0 $Q"
  run p$1x$2+$3_one "" "" "--workload C3 --particles $1 --scan-points $2 --beam-points $3 $Q"
done
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_gpu_bench_contract.py::test_headline_workload_gates 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl" | tail -8
