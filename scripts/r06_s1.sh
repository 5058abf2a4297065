#!/bin/bash
# round 6 session 1: calibration of this box + how the tiled kernel takes fewer wavefronts per SIMD and smaller particle groups
# (inputs for the design of the owner kernel: a work-group owning G particles' whole scan, terms in LDS in caller order)
O=gpurun_out/r06a; mkdir -p $O
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
run C2_default "" "--workload C2 $Q"
run C2_g4 "" "--workload C2 --lik-group 4 $Q"
run C2_g8 "" "--workload C2 --lik-group 8 $Q"
run C2_g32 "" "--workload C2 --lik-group 32 $Q"
run C2_map1 "lik_map=1" "--workload C2 $Q"
run C2_map1_g4 "lik_map=1" "--workload C2 --lik-group 4 $Q"
run C2_strict1 "" "--workload C2 --strict-order 1 $Q"
run C2_strict3 "" "--workload C2 --strict-order 3 $Q"
run C2j_default "" "--workload C2 --map-jitter 0.045 $Q"
run C2j_g4 "" "--workload C2 --map-jitter 0.045 --lik-group 4 $Q"
run C1_default "" "--workload C1 $Q"
run s1024x16384_s0 "" "--workload C2 --particles 1024 --strict-order 0 $Q"
run s1024x16384_s3 "" "--workload C2 --particles 1024 --strict-order 3 $Q"
run s4096x512 "" "--workload C2 --scan-points 512 $Q"
run s4096x96 "" "--workload C2 --scan-points 96 $Q"
run s64x96 "" "--workload C2 --particles 64 --scan-points 96 $Q"
