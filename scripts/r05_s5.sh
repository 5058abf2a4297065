#!/bin/bash
# round 5 session 5: per-axis voxel edges (cand_aniso) — parity tests, then A/B at the shipped dist_weight (z x 5)
O=gpurun_out/r05e; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bound.py tests/test_gpu_index_random.py tests/test_gpu_map_path.py tests/test_gpu_defer.py tests/test_gpu_dist_weight_fullsize.py tests/test_gpu_api_fuzz.py tests/test_gpu_fuzz.py tests/test_gpu_errors.py -x -q 2>&1 | tail -25 > $O/tests.log; tail -6 $O/tests.log
Q="--steps 20 --warmup 3 --no-extras --no-cpu-baseline"
run() { # name, options, bench args
  MCL3DL_HIP_OPTIONS="$2" timeout 900 python bench.py $3 2>$O/$1.err | tail -1 > $O/$1.json
  python - "$O/$1.json" "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels_ms_per_step"]; f=d["index"]["footprint_bytes"]
    print("%-26s ms/step %.4f lik %.4f beam %.4f | records %.3f GB bricks %d over %d build %.1f ms" % (sys.argv[2], d["ms_per_step"], k["likelihood"], k["beam"], f["cand_start"]/1e9, d["index"]["bricks"], d["index"]["voxels_with_overflow"], d["index"]["build_ms"]), flush=True)
except Exception as e:
    print(sys.argv[2], "no json", e)
PY
}
run C2_w1 "" "--workload C2 $Q"
run C2_w1b "" "--workload C2 $Q"
run C2_w5_iso "cand_aniso=0" "--workload C2 --dist-weight-z 5 $Q"
run C2_w5_aniso "cand_aniso=1" "--workload C2 --dist-weight-z 5 $Q"
run C2_w5_aniso3 "cand_aniso=1,cand_aniso_max=3" "--workload C2 --dist-weight-z 5 $Q"
run C2_w2_iso "cand_aniso=0" "--workload C2 --dist-weight-z 2 $Q"
run C2_w2_aniso "cand_aniso=1" "--workload C2 --dist-weight-z 2 $Q"
run C2j_w5_iso "cand_aniso=0" "--workload C2 --map-jitter 0.045 --dist-weight-z 5 $Q"
run C2j_w5_aniso "cand_aniso=1" "--workload C2 --map-jitter 0.045 --dist-weight-z 5 $Q"
run C3_w5_iso "cand_aniso=0" "--workload C3 --dist-weight-z 5 $Q"
run C3_w5_aniso "cand_aniso=1" "--workload C3 --dist-weight-z 5 $Q"
run C5s_w5_iso "cand_aniso=0" "--workload C5 --particles 8192 --dist-weight-z 5 --strict-order 0 $Q"
run C5s_w5_aniso "cand_aniso=1" "--workload C5 --particles 8192 --dist-weight-z 5 --strict-order 0 $Q"
run C5s_w5_budget "cand_aniso=0,index_budget_bytes=10e9" "--workload C5 --particles 8192 --dist-weight-z 5 --strict-order 0 $Q"
run C2_w1_budget "index_budget_bytes=200e6" "--workload C2 $Q"
