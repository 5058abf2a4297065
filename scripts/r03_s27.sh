#!/bin/bash
# r03 session 27: chunk size of the strict-order sum (64 / 32 / 16 KB of terms per LDS buffer: one / two / four work-groups per CU), one box
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03z4; mkdir -p $OUT
Q="--no-extras --no-cpu-baseline"
for rep in 1 2; do for c in 65536 32768 16384; do   # variants built with -DMCL3DL_STRICT_CHUNK=... while the chunk was a macro; the tree now picks 32768 above n_cus groups
  if [ $c = 65536 ]; then unset MCL3DL_HIP_LIB; else export MCL3DL_HIP_LIB=$GRAFT_REPO_ROOT/mcl_3dl_amd/variants/libmcl3dl_hip_sc$c.so; fi
  python bench.py --workload C5 --particles 8192 $Q 2>/dev/null | tail -1 > $OUT/C5_c${c}_$rep.json
  python bench.py --workload C2 --strict-order 1 $Q 2>/dev/null | tail -1 > $OUT/C2s_c${c}_$rep.json
done; done
unset MCL3DL_HIP_LIB
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03z4/*.json")):
    d=json.load(open(f)); print("%-16s"%f.split('/')[-1][:-5], "lik %.4f"%d["kernels_ms_per_step"]["likelihood"], "step %.4f"%d["ms_per_step"], d["result_check"]["entropy"])
P
