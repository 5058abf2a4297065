#!/bin/bash
# r03 session 19: deferred overflow rounds — parity tests; voxel edge of the candidate index with the queue on
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03v; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_defer.py -x -q > $OUT/pytest_defer.log 2>&1; tail -5 $OUT/pytest_defer.log
Q="--no-extras --no-cpu-baseline --cand-record-parts 4 --lik-defer 1"
for j in 0.045 0.02; do for r in 0.25 0.28 0.30 0.33 0.36; do
  python bench.py --workload C2 --map-jitter $j --cand-voxel-ratio $r $Q 2>/dev/null | tail -1 > $OUT/j${j}_r$r.json
done; done
for r in 0.36 0.42 0.50 0.55; do python bench.py --workload C2 --cand-voxel-ratio $r $Q 2>/dev/null | tail -1 > $OUT/lat_r$r.json; done
python bench.py --workload C2 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/lat_auto.json
python bench.py --workload C2 --map-jitter 0.045 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/j0.045_auto.json
python bench.py --workload C2 --map-jitter 0.02 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/j0.02_auto.json
python bench.py --workload C5 --particles 8192 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/C5_auto.json
python bench.py --workload C5 --particles 8192 --lik-defer 0 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/C5_d0.json
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03v/*.json")):
    try:
        d=json.load(open(f)); ix=d["index"]
        print("%-14s"%f.split('/')[-1][:-5], "lik %.4f"%d["kernels_ms_per_step"]["likelihood"], "parts", ix["record_parts"], "packed", ix.get("packed_words"), "defer", ix.get("deferred_overflow"), "ovf %d of %d"%(ix["voxels_with_overflow"], ix["voxels_with_candidates"]), "build %.1f"%ix["build_ms"], "entropy", d["result_check"]["entropy"])
    except Exception as e: print(f, "failed", e)
P
