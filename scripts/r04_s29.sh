#!/bin/bash
O=gpurun_out/r04aa
mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_bound.py tests/test_gpu_map_path.py tests/test_gpu_index_random.py tests/test_gpu_defer.py tests/test_gpu_parity.py tests/test_gpu_update_small.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -4
for j in 0.045 0.02 0; do
  for g in 0 1; do
    MCL3DL_HIP_OPTIONS=cand_bound_groups=$g timeout 600 python bench.py --map-jitter $j --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_j${j}_g$g.json
    python - <<PY
import json
d=json.load(open("$O/bench_j${j}_g$g.json"))
print("jitter $j groups $g: lik %.4f ms" % d["kernels_ms_per_step"]["likelihood"])
PY
  done
done
