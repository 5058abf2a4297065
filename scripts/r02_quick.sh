#!/bin/bash
# quick GPU check used while iterating on the likelihood kernel: parity tests that pin the kernels bit for bit, then short
# bench lines (lattice C2, jittered C2, C5 shard with its > 4 GB record array, C3).   usage: scripts/r02_quick.sh <tag>
TAG=${1:-q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_c4c5.py tests/test_gpu_fuzz.py tests/test_gpu_index_random.py tests/test_gpu_map_path.py -q -x 2>&1 | tail -5 > $OUT/pytest.log
tail -3 $OUT/pytest.log
Q="--steps 20 --warmup 3 --no-extras --no-cpu-baseline"
python bench.py --workload C2 $Q 2>/dev/null | tail -1 > $OUT/C2.json
python bench.py --workload C2 --map-jitter 0.045 $Q 2>/dev/null | tail -1 > $OUT/C2j.json
python bench.py --workload C3 $Q 2>/dev/null | tail -1 > $OUT/C3.json
python bench.py --workload C5 --particles 8192 $Q 2>/dev/null | tail -1 > $OUT/C5.json
python bench.py --workload C1 $Q 2>/dev/null | tail -1 > $OUT/C1.json
python - <<P
import json
for n in ("C2","C2j","C3","C5","C1"):
    try:
        d=json.load(open("$OUT/%s.json"%n)); k=d["kernels_ms_per_step"]
        print("%-4s value %.4g ms/step %.4f lik %.4f beam %.4f pf %.4f" % (n,d["value"],d["ms_per_step"],k["likelihood"],k["beam"],k["pf"]))
    except Exception as e: print(n,"failed",e)
P
