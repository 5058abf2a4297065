#!/bin/bash
# r03 session 26: particles per work-group of the tiled kernel (lik_group) with the queue in place, lattice / centroids / C5
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03z3; mkdir -p $OUT
Q="--no-extras --no-cpu-baseline"
for g in 0 8 16 32; do
  python bench.py --workload C2 --lik-group $g $Q 2>/dev/null | tail -1 > $OUT/C2_g$g.json
  python bench.py --workload C2 --map-jitter 0.045 --lik-group $g $Q 2>/dev/null | tail -1 > $OUT/C2j_g$g.json
  python bench.py --workload C5 --particles 8192 --strict-order 0 --lik-group $g $Q 2>/dev/null | tail -1 > $OUT/C5_g$g.json
  python bench.py --workload C2 --particles 4096 --scan-points 2048 --lik-group $g $Q 2>/dev/null | tail -1 > $OUT/s2048_g$g.json
done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03z3/*.json")):
    d=json.load(open(f)); print("%-12s"%f.split('/')[-1][:-5], "lik %.4f"%d["kernels_ms_per_step"]["likelihood"], "step %.4f"%d["ms_per_step"], d["roofline"]["kernel"])
P
