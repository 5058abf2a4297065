#!/bin/bash
# round 5, session 25: in-kernel float sums with four tiles per work-group — the chain tests, then the kernel time by particle count
O=gpurun_out/r05s25; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_chain.py -q -x -rf 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl" | tail -12 > $O/pytest.log
tail -3 $O/pytest.log
PYTHONPATH=. timeout 400 python scripts/r05_time_chain_multi.py 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids" | tee $O/chain_multi.txt
