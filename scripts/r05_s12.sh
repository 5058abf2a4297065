#!/bin/bash
# round 5, session 12: completion word folded into the update's last kernel, A/B on one box; then the suite on this tree
O=gpurun_out/r05s12; mkdir -p $O
PYTHONPATH=. timeout 600 python scripts/r05_time_fold.py 200 2>&1 | grep -v -E "RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids" | tee $O/fold_ab.txt
timeout 2400 python -m pytest tests -m gpu -q -rf 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl" | tail -150 > $O/pytest_gpu.log
grep -E "passed|failed" $O/pytest_gpu.log | tail -2
