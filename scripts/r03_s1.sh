#!/bin/bash
# round 3, session 1: scan-preparation bisect (C2 and C3 sequences), L2 request calibration (times + PMC), default bench line
OUT=gpurun_out/r03a
mkdir -p $OUT
python scripts/r03_scanprep_bisect.py C2 > $OUT/bisect_C2.log 2>&1
python scripts/r03_scanprep_bisect.py C3 > $OUT/bisect_C3.log 2>&1
profiles/l2_calib.bin 5 > $OUT/l2_calib.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for SET in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/l2pmc_$i -o l2calib -- profiles/l2_calib.bin 1 > $OUT/l2pmc_$i.log 2>&1
  echo "pmc [$SET] rc=$?"
done
python bench.py 2>$OUT/C2_default.err | tail -1 > $OUT/C2_default.json
tail -3 $OUT/C2_default.err
