#!/bin/bash
# round 5 session 6: chain mode (strict_order = 3) at mid-size particle counts with the group size picked for the chain
O=gpurun_out/r05f; mkdir -p $O
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
for P in 256 1024 2048 3000; do
  run P${P}_s0 "" "--workload C2 --particles $P --strict-order 0 $Q"
  run P${P}_s3 "" "--workload C2 --particles $P --strict-order 3 $Q"
  run P${P}_s1 "" "--workload C2 --particles $P --strict-order 1 $Q"
done
run P1024_s3_g16 "" "--workload C2 --particles 1024 --strict-order 3 --lik-group 16 $Q"
run P1024x4k_s0 "" "--workload C2 --particles 1024 --scan-points 4096 --strict-order 0 $Q"
run P1024x4k_s3 "" "--workload C2 --particles 1024 --scan-points 4096 --strict-order 3 $Q"
run C1_s0 "" "--workload C1 --strict-order 0 $Q"
run C1_s3 "" "--workload C1 --strict-order 3 $Q"
run C1_s1 "" "--workload C1 --strict-order 1 $Q"
