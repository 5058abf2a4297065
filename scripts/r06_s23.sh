#!/bin/bash
# round 6 session 23: in front of the caller-order replay, up to how many rays is the merged launch better than two streams?
O=gpurun_out/r06zw; mkdir -p $O
run() { # name, max rays, bench args
  MCL3DL_MERGE_STRICT_MAX_RAYS="$2" timeout 900 python bench.py $3 2>$O/$1.err | tail -1 > $O/$1.json
  python - "$O/$1.json" "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels_ms_per_step"]
    print("%-28s ms/step %.4f 8d %s lik %.4f beam %.4f pf %.4f" % (sys.argv[2], d["ms_per_step"], d.get("ms_per_step_8d"), k["likelihood"], k["beam"], k["pf"]), flush=True)
except Exception as e:
    print(sys.argv[2], "no json", e)
PY
}
Q="--steps 30 --warmup 4 --no-extras --no-cpu-baseline"
for r in 1 2; do
for shape in "4096 4096 128" "4096 4096 512" "8192 4096 512" "16384 4096 512" "8192 32768 512"; do
  set -- $shape
  run p$1x$2+$3_streams_$r 262144 "--workload C3 --particles $1 --scan-points $2 --beam-points $3 $Q"
  run p$1x$2+$3_merged_$r 1000000000 "--workload C3 --particles $1 --scan-points $2 --beam-points $3 $Q"
done
done
