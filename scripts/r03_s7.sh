#!/bin/bash
# round 3, session 7: strict sum with two groups per work-group (A/B against the HEAD build on one box)
OUT=gpurun_out/r03g
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_c4c5.py -x -q 2>&1 | tail -5 | tee $OUT/pytest.log
OLD=$PWD/mcl_3dl_amd/variants/libmcl3dl_hip_oldstrict.so
for V in new old new old; do
  if [ $V = old ]; then export MCL3DL_HIP_LIB=$OLD; else unset MCL3DL_HIP_LIB; fi
  python bench.py --workload C5 --particles 8192 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $OUT/C5_$V.json
  python - <<P
import json
d=json.load(open("$OUT/C5_$V.json")); k=d["kernels_ms_per_step"]
print("C5_%-4s ms/step %.4f lik %.4f beam %.4f pf %.4f" % ("$V",d["ms_per_step"],k["likelihood"],k["beam"],k["pf"]))
P
done
