#!/bin/bash
# round 5, session 21: one-launch 16-bit sort, second form (batched loads, bank-swizzled counters, lane-owned prefix): tests, A/B, kernel time
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05s21; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_sort16.py tests/test_gpu_sort.py -q -x -rf 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl" | tail -5 > $O/pytest.log
tail -2 $O/pytest.log
PYTHONPATH=. timeout 300 python scripts/r05_time_sort16.py 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids" | tee $O/sort16_8d_ab.txt
PYTHONPATH=. timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o tl -- python scripts/r05_timeline_8d.py C2 > $O/run.log 2>&1
F=$(find $O/trace -name "*kernel_stats.csv" | head -1); grep -E "rs_sort16|stage_pack|rs_pass|rs_keygen|rs_hist" "$F" | cut -c1-60,100-200 | head; 
T=$(find $O/trace -name "*kernel_trace.csv" | head -1); python scripts/r05_timeline_8d.py --table "$T" | head -14 | tee $O/timeline.txt; rm -rf $O/trace
