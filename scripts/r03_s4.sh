#!/bin/bash
# round 3, session 4: sort with separate histogram launches, LDS-staged centroids, merged index upload; strict_order = 2 (C5
# tolerance); bench contract
OUT=gpurun_out/r03d
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_sort.py -x -q 2>&1 | tail -5 | tee $OUT/pytest_sort.log
timeout 900 python -m pytest tests/test_gpu_scan_prep.py tests/test_gpu_map_path.py tests/test_gpu_parity.py tests/test_gpu_resample.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -5 | tee $OUT/pytest_users.log
timeout 900 python -m pytest tests/test_gpu_c4c5.py tests/test_gpu_bench_contract.py -x -q -s 2>&1 | grep -E "passed|failed|error|C5|Error" | tail -12 | tee $OUT/pytest_c5.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prep_stats -o prep -- python scripts/time_scan_prep.py 50 > $OUT/prep_under_rocprof.log 2>&1
python scripts/time_scan_prep.py 50 > $OUT/prep_plain.log 2>&1
tail -1 $OUT/prep_plain.log
python bench.py --workload C5 --particles 8192 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $OUT/C5_shard_auto.json
python bench.py --workload C5 --particles 8192 --no-cpu-baseline --no-extras --strict-order 0 2>/dev/null | tail -1 > $OUT/C5_shard_fp64.json
python - <<P
import json
for n in ("C5_shard_auto","C5_shard_fp64"):
    d=json.load(open("$OUT/%s.json"%n)); k=d["kernels_ms_per_step"]
    print("%-16s ms/step %.4f lik %.4f beam %.4f pf %.4f" % (n,d["ms_per_step"],k["likelihood"],k["beam"],k["pf"]))
P
