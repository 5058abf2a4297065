#!/bin/bash
# r03 session 24: pf::measure as one launch between 1024 and 16 384 particles — bit-identity, then what it buys per step
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03z1; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_pf_ticket.py tests/test_gpu_pf_fused.py tests/test_gpu_defer.py -q -x 2>&1 | tail -4
Q="--no-extras --no-cpu-baseline"
for rep in 1 2; do for t in 0 1; do
  MCL3DL_HIP_OPTIONS="pf_ticket=$t" python bench.py --workload C2 $Q 2>/dev/null | tail -1 > $OUT/C2_t${t}_$rep.json
  MCL3DL_HIP_OPTIONS="pf_ticket=$t" python bench.py --workload C2 --particles 4096 --scan-points 96 $Q 2>/dev/null | tail -1 > $OUT/s4096x96_t${t}_$rep.json
  MCL3DL_HIP_OPTIONS="pf_ticket=$t" python bench.py --workload C2 --particles 4096 --scan-points 512 $Q 2>/dev/null | tail -1 > $OUT/s4096x512_t${t}_$rep.json
  MCL3DL_HIP_OPTIONS="pf_ticket=$t" python bench.py --workload C2 --particles 16384 --scan-points 2048 $Q 2>/dev/null | tail -1 > $OUT/s16384x2048_t${t}_$rep.json
done; done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03z1/*.json")):
    d=json.load(open(f)); print("%-22s"%f.split('/')[-1][:-5], "step %.4f"%d["ms_per_step"], "lik %.4f"%d["kernels_ms_per_step"]["likelihood"], "pf %.4f"%d["kernels_ms_per_step"]["pf"], "8d %.4f"%d["update_8d"]["ms_per_update"], d["result_check"]["entropy"])
P
