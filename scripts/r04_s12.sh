#!/bin/bash
O=gpurun_out/r04k
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bound.py tests/test_gpu_defer.py tests/test_gpu_map_path.py tests/test_gpu_parity.py tests/test_gpu_index_random.py -x -q -s 2>&1 | grep -E "crowded map|passed|failed|Error|error|assert" | head -20 | cut -c1-250
for b in 0 1; do
  for j in 0.045 0.02 0; do
    timeout 600 python bench.py --map-jitter $j --cand-bound $b --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_j${j}_b$b.json
    python - <<PY
import json
d=json.load(open("$O/bench_j${j}_b$b.json"))
print("jitter $j cand_bound $b: lik %.4f ms, step %.4f, index ovf %s" % (d["kernels_ms_per_step"]["likelihood"], d["ms_per_step"], d["index"]["voxels_with_overflow"]))
PY
  done
done
