#!/bin/bash
# quick GPU check while iterating on the beam kernel: every test that pins rays bit for bit, then C3 / C3 stress / C5 lines
OUT=gpurun_out/${1:-qb}
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_kats.py tests/test_gpu_fullsize.py tests/test_gpu_c4c5.py tests/test_gpu_fuzz.py tests/test_gpu_adapter.py tests/test_gpu_map_path.py -q -x 2>&1 | grep -E "passed|failed|Error" | tail -3
Q="--steps 20 --warmup 3 --no-extras --no-cpu-baseline"
python bench.py --workload C3 $Q 2>/dev/null | tail -1 > $OUT/C3.json
python bench.py --workload C3 --beam-points 16384 $Q 2>/dev/null | tail -1 > $OUT/C3s.json
python bench.py --workload C5 --particles 8192 $Q 2>/dev/null | tail -1 > $OUT/C5.json
python - <<P
import json
for n in ("C3","C3s","C5"):
    try:
        d=json.load(open("$OUT/%s.json"%n)); k=d["kernels_ms_per_step"]
        print("%-4s ms/step %.4f lik %.4f beam %.4f | rays/s %.4g steps/s %.4g" % (n,d["ms_per_step"],k["likelihood"],k["beam"],d["beam"]["rays_per_s"],d["beam"]["dda_steps_per_s"]))
    except Exception as e: print(n,"failed",e)
P
