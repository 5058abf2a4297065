#!/bin/bash
# r03 session 18: deferred overflow rounds — parity tests, then C2 / jittered C2 with the queue off / on and 64- / 128-byte records
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03u; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_defer.py -x -q > $OUT/pytest_defer.log 2>&1; tail -15 $OUT/pytest_defer.log
Q="--no-extras --no-cpu-baseline"
for cfg in "j8_d0 --map-jitter 0.045 --cand-record-parts 8 --lik-defer 0" "j4_d0 --map-jitter 0.045 --cand-record-parts 4 --lik-defer 0" \
           "j4_d1 --map-jitter 0.045 --cand-record-parts 4 --lik-defer 1" "j4_d1_r30 --map-jitter 0.045 --cand-record-parts 4 --lik-defer 1 --cand-voxel-ratio 0.30" \
           "j4_d1_r42 --map-jitter 0.045 --cand-record-parts 4 --lik-defer 1 --cand-voxel-ratio 0.42" "j4_d1_r50 --map-jitter 0.045 --cand-record-parts 4 --lik-defer 1 --cand-voxel-ratio 0.50" \
           "jauto --map-jitter 0.045" "j020_auto --map-jitter 0.02" "j020_8_d0 --map-jitter 0.02 --cand-record-parts 8 --lik-defer 0" \
           "l4_d0 --lik-defer 0" "l4_d1 --lik-defer 1" "lauto"; do
  set -- $cfg; name=$1; shift
  python bench.py --workload C2 $* $Q 2>$OUT/$name.err | tail -1 > $OUT/$name.json
done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03u/*.json")):
    try:
        d=json.load(open(f)); ix=d["index"]
        print("%-12s"%f.split('/')[-1][:-5], "lik %.4f"%d["kernels_ms_per_step"]["likelihood"], "ratio", round(ix.get("voxel_ratio",0),3), "parts", ix["record_parts"], "packed", ix.get("packed_words"), "defer", ix.get("deferred_overflow"), "relerr", d.get("result_check"))
    except Exception as e: print(f, "failed", e)
P
