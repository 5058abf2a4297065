#!/bin/bash
# round 6 session 8: the map of centroids — where the tiled kernel stands across voxel edges and record
# sizes today (the choice of round 3 re-checked with the queue, the bounds and the packed words in place)
O=gpurun_out/r06j; mkdir -p $O
Q="--steps 20 --warmup 3 --no-extras --no-cpu-baseline --workload C2 --map-jitter 0.045"
run() { # name, options, bench args
  MCL3DL_HIP_OPTIONS="$2" timeout 900 python bench.py $3 2>$O/$1.err | tail -1 > $O/$1.json
  python - "$O/$1.json" "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels_ms_per_step"]; i=d["index"]
    print("%-26s lik %.4f | voxels %.2fM over4 %.1f %% over8 %.2f %% cand/voxel %.2f records %.0f MB parts %d defer %d" % (sys.argv[2], k["likelihood"], i["voxels_with_candidates"]/1e6, 100.0*i["voxels_with_overflow"]/max(i["voxels_with_candidates"],1), 100.0*i["voxels_over8"]/max(i["voxels_with_candidates"],1), i["candidates"]/max(i["voxels_with_candidates"],1), i["footprint_bytes"]["cand_start"]/1e6, i["record_parts"], i["deferred_overflow"]), flush=True)
except Exception as e:
    print(sys.argv[2], "no json", e)
PY
}
run lattice "" "--steps 20 --warmup 3 --no-extras --no-cpu-baseline --workload C2"
for r in 0.25 0.30 0.36 0.42 0.50; do
  run j_r${r}_p4 "" "$Q --cand-voxel-ratio $r --cand-record-parts 4"
  run j_r${r}_p8 "lik_defer=0" "$Q --cand-voxel-ratio $r --cand-record-parts 8"
done
run j_r0.36_p4_nobound "cand_bound=0" "$Q --cand-voxel-ratio 0.36 --cand-record-parts 4"
run j_r0.36_p4_nodefer "lik_defer=0" "$Q --cand-voxel-ratio 0.36 --cand-record-parts 4"
