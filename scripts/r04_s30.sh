#!/bin/bash
O=gpurun_out/r04ab
mkdir -p $O
for r in 0 0.30 0.33 0.40 0.45; do
  timeout 600 python bench.py --map-jitter 0.045 --cand-voxel-ratio $r --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_j045_r$r.json
  python - <<PY
import json
d=json.load(open("$O/bench_j045_r$r.json")); i=d["index"]
print("jitter 0.045 ratio $r: lik %.4f ms, overflow voxels %d of %d" % (d["kernels_ms_per_step"]["likelihood"], i["voxels_with_overflow"], i["voxels_with_candidates"]))
PY
done
timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lattice lik %.4f' % d['kernels_ms_per_step']['likelihood'])"
