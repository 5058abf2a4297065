#!/bin/bash
# round 5 session 1: XCD-by-tile (lik_map=0) against XCD-by-group (lik_map=1) mapping of the tiled kernel; strict baselines
O=gpurun_out/r05a; mkdir -p $O
Q="--steps 20 --warmup 3 --no-extras --no-cpu-baseline"
run() { # name, options, bench args
  MCL3DL_HIP_OPTIONS="$2" timeout 600 python bench.py $3 2>$O/$1.err | tail -1 > $O/$1.json
  python - "$O/$1.json" "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels_ms_per_step"]
    print("%-28s ms/step %.4f lik %.4f beam %.4f pf %.4f" % (sys.argv[2], d["ms_per_step"], k["likelihood"], k["beam"], k["pf"]), flush=True)
except Exception as e:
    print(sys.argv[2], "no json", e)
PY
}
run C2_map0 "lik_map=0" "--workload C2 $Q"
run C2_map1 "lik_map=1" "--workload C2 $Q"
run C2_map0b "lik_map=0" "--workload C2 $Q"
run C2_map1b "lik_map=1" "--workload C2 $Q"
run C2_map1_g8 "lik_map=1" "--workload C2 --lik-group 8 $Q"
run C2_map0_g8 "lik_map=0" "--workload C2 --lik-group 8 $Q"
run C2j_map0 "lik_map=0" "--workload C2 --map-jitter 0.045 $Q"
run C2j_map1 "lik_map=1" "--workload C2 --map-jitter 0.045 $Q"
run C2_strict1 "" "--workload C2 --strict-order 1 $Q"
run C2_strict2 "strict_auto_min=1" "--workload C2 --strict-order 2 $Q"
run C4s_map0 "lik_map=0" "--workload C4 --particles 32768 $Q"
run C4s_map1 "lik_map=1" "--workload C4 --particles 32768 $Q"
run C5s_map0_s0 "lik_map=0" "--workload C5 --particles 8192 --strict-order 0 $Q"
run C5s_map1_s0 "lik_map=1" "--workload C5 --particles 8192 --strict-order 0 $Q"
run C5s_map0_s2 "lik_map=0" "--workload C5 --particles 8192 $Q"
