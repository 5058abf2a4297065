#!/bin/bash
# r03 session 22: how many Morton key bits the scan ordering needs (16 / 22 / 30: two / three / four radix passes), one box, twice
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03y; mkdir -p $OUT
Q="--no-extras --no-cpu-baseline"
for rep in 1 2; do for b in 16 22 30; do   # variants: make -C mcl_3dl_amd/csrc with -DMCL3DL_MORTON_BITS=16 / 30 into mcl_3dl_amd/variants/; the tree builds 22
  if [ $b = 16 ]; then unset MCL3DL_HIP_LIB; else export MCL3DL_HIP_LIB=$GRAFT_REPO_ROOT/mcl_3dl_amd/variants/libmcl3dl_hip_m$b.so; fi
  python bench.py --workload C2 $Q 2>/dev/null | tail -1 > $OUT/C2_m${b}_$rep.json
  python bench.py --workload C5 --particles 8192 $Q 2>/dev/null | tail -1 > $OUT/C5_m${b}_$rep.json
  python bench.py --workload C2 --map-jitter 0.045 $Q 2>/dev/null | tail -1 > $OUT/C2j_m${b}_$rep.json
done; done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03y/*.json")):
    d=json.load(open(f)); print("%-14s"%f.split('/')[-1][:-5], "lik %.4f"%d["kernels_ms_per_step"]["likelihood"], "step %.4f"%d["ms_per_step"], "8d %.4f"%d["update_8d"]["ms_per_update"], "entropy", d["result_check"]["entropy"])
P
