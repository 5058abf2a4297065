#!/bin/bash
# round 3, session 2: the new sort / fused cloud kernels — their own tests, the tests of their users, launch count and time
# of the scan preparation under rocprofv3, and the GC evidence for the round-2 "regression"
OUT=gpurun_out/r03b
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_sort.py -x -q 2>&1 | tail -15 | tee $OUT/pytest_sort.log
timeout 900 python -m pytest tests/test_gpu_scan_prep.py tests/test_gpu_map_path.py tests/test_gpu_parity.py tests/test_gpu_resample.py -x -q 2>&1 | tail -15 | tee $OUT/pytest_users.log
python scripts/r03_scanprep_bisect.py C2 > $OUT/bisect_C2_gc.log 2>&1
NOGC=1 python scripts/r03_scanprep_bisect.py C2 > $OUT/bisect_C2_nogc.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prep_stats -o prep -- python scripts/time_scan_prep.py 50 > $OUT/prep_under_rocprof.log 2>&1
python scripts/time_scan_prep.py 50 > $OUT/prep_plain.log 2>&1
tail -2 $OUT/prep_plain.log
