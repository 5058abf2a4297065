#!/bin/bash
O=gpurun_out/r04l
mkdir -p $O
for r in 0.28 0.30 0.33 0.36 0.40 0.45 0.5; do
  for j in 0.045 0.02; do
    timeout 600 python bench.py --map-jitter $j --cand-voxel-ratio $r --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_j${j}_r$r.json
    python - <<PY
import json
d=json.load(open("$O/bench_j${j}_r$r.json"))
i=d["index"]
print("jitter $j ratio $r: lik %.4f ms, voxels %d, overflow %d (%.1f%%), records %.2f GB, build %.1f ms" % (d["kernels_ms_per_step"]["likelihood"], i["voxels_with_candidates"], i["voxels_with_overflow"], 100.0*i["voxels_with_overflow"]/max(i["voxels_with_candidates"],1), i["footprint_bytes"]["cand_start"]/1e9, i["build_ms"]))
PY
  done
done
