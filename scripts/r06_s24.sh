#!/bin/bash
# round 6 session 24: the tiled kernel's XCD balance — the second half of a row's particle groups takes the tile of the XCD four places
# on (sixteen half-tiles per XCD instead of eight tiles); this tree against the previous commit's library
O=gpurun_out/r06zv; mkdir -p $O
run() { # name, lib, bench args
  MCL3DL_HIP_LIB="$2" timeout 900 python bench.py $3 2>$O/$1.err | tail -1 > $O/$1.json
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
PREV=$PWD/mcl_3dl_amd/variants/libmcl3dl_hip_prev.so
for r in 1 2 3; do
  run C2_prev_$r "$PREV" "--workload C2 $Q"
  run C2_new_$r "" "--workload C2 $Q"
done
for r in 1 2; do
  run C2j_prev_$r "$PREV" "--workload C2 --map-jitter 0.045 $Q"
  run C2j_new_$r "" "--workload C2 --map-jitter 0.045 $Q"
  run C3_prev_$r "$PREV" "--workload C3 $Q"
  run C3_new_$r "" "--workload C3 $Q"
done
run C4s_prev "$PREV" "--workload C4 --particles 32768 --steps 20 --warmup 3 --no-extras --no-cpu-baseline"
run C4s_new "" "--workload C4 --particles 32768 --steps 20 --warmup 3 --no-extras --no-cpu-baseline"
run C5s_prev "$PREV" "--workload C5 --particles 8192 --steps 10 --warmup 2 --no-extras --no-cpu-baseline"
run C5s_new "" "--workload C5 --particles 8192 --steps 10 --warmup 2 --no-extras --no-cpu-baseline"
