#!/bin/bash
# r03 session 17: jittered map (C2j) against the voxel edge of the candidate index and the record width, warm clocks
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03t; mkdir -p $OUT
for ratio in 0.0 0.30 0.36 0.42 0.50; do for parts in 0 4 8; do
  python bench.py --workload C2 --map-jitter 0.045 --cand-voxel-ratio $ratio --cand-record-parts $parts --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/j_${ratio}_${parts}.json
done; done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03t/j_*.json")):
    try:
        d=json.load(open(f)); ix=d["index"]
        print(f.split('/')[-1], "lik %.4f"%d["kernels_ms_per_step"]["likelihood"], "ratio", round(ix.get("voxel_ratio",0),3), "parts", ix["record_parts"], "build %.1f"%ix["build_ms"], "ovf", ix["voxels_with_overflow"], "of", ix["voxels_with_candidates"], "bytes", ix["footprint_bytes"]["cand_start"])
    except Exception as e: print(f, "failed", e)
P
