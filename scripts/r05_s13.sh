#!/bin/bash
# round 5, session 13: the API-sequence fuzz far beyond the default suite's 300 x 30 (new seeds, longer sequences)
O=gpurun_out/r05s13; mkdir -p $O
export MCL3DL_FUZZ_SEQUENCES=${1:-3000} MCL3DL_FUZZ_CALLS=${2:-40}
( time timeout 1100 python -m pytest tests/test_gpu_api_fuzz.py -q -rf -k random_call 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl" | tail -60 ) > $O/fuzz_long.log 2>&1
grep -E "passed|failed|real" $O/fuzz_long.log | tail -3
