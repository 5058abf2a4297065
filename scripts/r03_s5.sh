#!/bin/bash
# round 3, session 5: strict-sum kernel with two chunks in flight, group vote tests, ADVICE fixes
OUT=gpurun_out/r03e
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_group.py tests/test_gpu_c4c5.py tests/test_gpu_map_path.py tests/test_gpu_moments.py -x -q 2>&1 | tail -8 | tee $OUT/pytest.log
python bench.py --workload C5 --particles 8192 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $OUT/C5_shard_auto.json
python bench.py --workload C2 --no-cpu-baseline --no-extras --strict-order 1 2>/dev/null | tail -1 > $OUT/C2_strict.json
python bench.py --workload C2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $OUT/C2.json
python - <<P
import json
for n in ("C5_shard_auto","C2_strict","C2"):
    d=json.load(open("$OUT/%s.json"%n)); k=d["kernels_ms_per_step"]
    print("%-16s ms/step %.4f lik %.4f beam %.4f pf %.4f" % (n,d["ms_per_step"],k["likelihood"],k["beam"],k["pf"]))
P
