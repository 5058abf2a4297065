#!/bin/bash
# mid-size shapes (the reference's operating range; 4096 x 512 is VERDICT's target shape): the tiled kernel with the
# contiguous-range XCD mapping from 256 / 512 / 1024 points up, against the one-work-group-per-particle kernel
OUT=gpurun_out/${1:-shapes}
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -3
Q="--steps 30 --warmup 5 --no-extras --no-cpu-baseline --workload C2"
for sh in "4096 256" "4096 512" "4096 1000" "4096 1500" "4096 2048" "1024 300" "512 1000" "64 1000" "2000 500" "4096 16384"; do
  set -- $sh
  for tm in 256 512 1024 100000; do
    python bench.py $Q --particles $1 --scan-points $2 --lik-tiled-min $tm 2>/dev/null | tail -1 > $OUT/s_$1x$2_tm$tm.json
  done
done
python - <<P
import json,glob
for f in sorted(glob.glob("$OUT/s_*.json")):
    try:
        d=json.load(open(f)); k=d["kernels_ms_per_step"]
        print("%-32s lik %.4f ms/step %.4f" % (f.split("/")[-1], k["likelihood"], d["ms_per_step"]))
    except Exception as e: print(f,"failed",e)
P
