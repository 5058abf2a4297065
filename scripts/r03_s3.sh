#!/bin/bash
# round 3, session 3: the sort spread over the chip — tests, then launch count and time of the scan preparation
OUT=gpurun_out/r03c
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_sort.py -x -q 2>&1 | tail -5 | tee $OUT/pytest_sort.log
timeout 900 python -m pytest tests/test_gpu_scan_prep.py tests/test_gpu_map_path.py tests/test_gpu_parity.py tests/test_gpu_resample.py -x -q 2>&1 | tail -5 | tee $OUT/pytest_users.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prep_stats -o prep -- python scripts/time_scan_prep.py 50 > $OUT/prep_under_rocprof.log 2>&1
python scripts/time_scan_prep.py 50 > $OUT/prep_plain.log 2>&1
tail -1 $OUT/prep_plain.log
python scripts/time_host_path.py > $OUT/host_path.log 2>&1
tail -12 $OUT/host_path.log
