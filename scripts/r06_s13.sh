#!/bin/bash
# round 6 session 13: the beam counts taken by pf_partial_kernel / pf_fused_kernel too (no beam_finalize, no fill launch in the reference's
# own beam configuration: 3 rays per particle): suite, then shapes against the previous commit's library (variants/libmcl3dl_hip_prev.so)
O=gpurun_out/r06r; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl" | tail -5
run() { # name, lib, bench args
  MCL3DL_HIP_LIB="$2" timeout 900 python bench.py $3 2>$O/$1.err | tail -1 > $O/$1.json
  python - "$O/$1.json" "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels_ms_per_step"]
    print("%-24s ms/step %.4f 8d %s lik %.4f beam %.4f pf %.4f one %.4f" % (sys.argv[2], d["ms_per_step"], d.get("ms_per_step_8d"), k["likelihood"], k["beam"], k["pf"], k["update_one_launch"]), flush=True)
except Exception as e:
    print(sys.argv[2], "no json", e)
PY
}
Q="--steps 60 --warmup 5 --no-extras --no-cpu-baseline"
PREV=$PWD/mcl_3dl_amd/variants/libmcl3dl_hip_prev.so
for r in 1 2; do
for shape in "4096 96 3" "1000 96 3" "4096 512 16" "2048 2048 48"; do
  set -- $shape
  run p$1x$2+$3_prev_$r "$PREV" "--workload C3 --particles $1 --scan-points $2 --beam-points $3 $Q"
  run p$1x$2+$3_new_$r "" "--workload C3 --particles $1 --scan-points $2 --beam-points $3 $Q"
done
done
