#!/bin/bash
# round 5, session 10: the completion word folded into the last kernel of the host-buffer update — suite, smoke, 8d timing
O=gpurun_out/r05s10; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -rf 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl" | tail -40 > $O/pytest_gpu.log
grep -E "passed|failed" $O/pytest_gpu.log | tail -2
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $O/smoke.log)"
for i in 1 2; do PYTHONPATH=. timeout 300 python scripts/r05_time_presorted.py C2 100 2>&1 | tail -8; done | tee $O/time8d.log
