#!/bin/bash
# r03 session 20: packed w words on the per-particle kernels (A/B on one box), parity file
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03w; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_defer.py -x -q 2>&1 | tail -3
Q="--no-extras --no-cpu-baseline"
for rep in 1 2; do for pk in 0 1; do for sh in "4096 96" "4096 512" "64 96" "500 300" "64 1000"; do set -- $sh
  python bench.py --workload C2 --particles $1 --scan-points $2 --cand-packed $pk $Q 2>/dev/null | tail -1 > $OUT/s_$1x$2_p${pk}_$rep.json
done; done; done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03w/*.json")):
    try:
        d=json.load(open(f)); print("%-26s"%f.split('/')[-1][:-5], "step %.4f"%d["ms_per_step"], "lik %.4f"%d["kernels_ms_per_step"]["likelihood"], "pf %.4f"%d["kernels_ms_per_step"]["pf"], "packed", d["index"].get("packed_words"))
    except Exception as e: print(f, "failed", e)
P
