#!/bin/bash
# round 5, session 23: stage_pack with contiguous 16-byte host reads — tests of the staged update, A/B is box-to-box only
# (no option): timing of the 8d update + kernel trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05s23; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_update_staged.py tests/test_gpu_parity.py tests/test_gpu_chain.py tests/test_gpu_fuzz.py tests/test_gpu_sort16.py tests/test_gpu_api_fuzz.py -q -x -rf 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl" | tail -8 > $O/pytest.log
tail -2 $O/pytest.log
for i in 1 2; do PYTHONPATH=. timeout 300 python scripts/r05_time_presorted.py C2 100 2>&1 | grep -E "default path|presorted, fp64|presorted, in-kernel"; done | tee $O/time8d.txt
PYTHONPATH=. timeout 300 python scripts/r05_time_presorted.py C3 100 2>&1 | grep -E "default path" | tee -a $O/time8d.txt
PYTHONPATH=. timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o tl -- python scripts/r05_timeline_8d.py C2 > $O/run.log 2>&1
T=$(find $O/trace -name "*kernel_trace.csv" | head -1); python scripts/r05_timeline_8d.py --table "$T" | head -14 | tee $O/timeline.txt; rm -rf $O/trace
