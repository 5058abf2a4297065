#!/bin/bash
# round 6 session 22: the overflow test of the tiled kernel's first round in MASK form (scalar ANDs of compare masks, inverse_ballot)
# instead of ballots of combined booleans — four VALU instructions per evaluation less; this tree against the previous commit's library
O=gpurun_out/r06zx; mkdir -p $O
run() { # name, lib, bench args
  MCL3DL_HIP_LIB="$2" timeout 900 python bench.py $3 2>$O/$1.err | tail -1 > $O/$1.json
  python - "$O/$1.json" "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels_ms_per_step"]
    print("%-20s ms/step %.4f 8d %s lik %.4f beam %.4f pf %.4f" % (sys.argv[2], d["ms_per_step"], d.get("ms_per_step_8d"), k["likelihood"], k["beam"], k["pf"]), flush=True)
except Exception as e:
    print(sys.argv[2], "no json", e)
PY
}
Q="--steps 40 --warmup 5 --no-extras --no-cpu-baseline"
PREV=$PWD/mcl_3dl_amd/variants/libmcl3dl_hip_prev.so
for r in 1 2 3; do
  run C2_prev_$r "$PREV" "--workload C2 $Q"
  run C2_new_$r "" "--workload C2 $Q"
done
for r in 1 2; do
  run C2j_prev_$r "$PREV" "--workload C2 --map-jitter 0.045 $Q"
  run C2j_new_$r "" "--workload C2 --map-jitter 0.045 $Q"
done
run C4s_prev "$PREV" "--workload C4 --particles 32768 --steps 20 --warmup 3 --no-extras --no-cpu-baseline"
run C4s_new "" "--workload C4 --particles 32768 --steps 20 --warmup 3 --no-extras --no-cpu-baseline"
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_gpu_bench_contract.py::test_headline_workload_gates 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl" | tail -4
