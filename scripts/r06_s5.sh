#!/bin/bash
# round 6 session 5: the tiled kernel's plateau — A/B of (a) the dense record grid (no brick table),
# (b) particle groups ordered by pose, (c) G / MINW on top of them. Timing first; counters for the interesting ones (s6).
O=gpurun_out/r06g; mkdir -p $O
Q="--steps 20 --warmup 3 --no-extras --no-cpu-baseline"
run() { # name, options, bench args
  MCL3DL_HIP_OPTIONS="$2" timeout 900 python bench.py $3 2>$O/$1.err | tail -1 > $O/$1.json
  python - "$O/$1.json" "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels_ms_per_step"]
    print("%-28s ms/step %.4f lik %.4f pf %.4f index %.0f MB build %.1f ms" % (sys.argv[2], d["ms_per_step"], k["likelihood"], k["pf"], d["index"]["footprint_bytes"]["cand_start"]/1e6, d["index"]["build_ms"]), flush=True)
except Exception as e:
    print(sys.argv[2], "no json", e)
PY
}
for rep in 1 2; do
run C2_base_$rep "" "--workload C2 $Q"
run C2_dense_$rep "cand_dense=1,index_budget_bytes=0" "--workload C2 $Q"
run C2_sort_yaw_$rep "" "--workload C2 --sort-poses yaw $Q"
run C2_sort_xy_$rep "" "--workload C2 --sort-poses xy $Q"
run C2_sort_cluster_$rep "" "--workload C2 --sort-poses cluster $Q"
run C2_dense_cluster_$rep "cand_dense=1,index_budget_bytes=0" "--workload C2 --sort-poses cluster $Q"
done
run C2_dense_cluster_g8 "cand_dense=1,index_budget_bytes=0" "--workload C2 --sort-poses cluster --lik-group 8 $Q"
run C2_dense_cluster_g32 "cand_dense=1,index_budget_bytes=0" "--workload C2 --sort-poses cluster --lik-group 32 $Q"
run C2j_base "" "--workload C2 --map-jitter 0.045 $Q"
run C2j_dense "cand_dense=1,index_budget_bytes=0" "--workload C2 --map-jitter 0.045 $Q"
run C2j_cluster "" "--workload C2 --map-jitter 0.045 --sort-poses cluster $Q"
run C4s_base "" "--workload C4 --particles 32768 $Q"
run C4s_dense "cand_dense=1,index_budget_bytes=0" "--workload C4 --particles 32768 $Q"
