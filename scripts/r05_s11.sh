#!/bin/bash
# round 5, session 11: the two failures of session 10 with their whole output + where a long wait's CPU time goes
O=gpurun_out/r05s11; mkdir -p $O
PYTHONPATH=. timeout 300 python scripts/r05_dbg_cpushare.py 2>&1 | grep -v -E "RCCL|HIP version|ROCm version|Hostname|Librccl" | tee $O/cpushare.log
timeout 900 python -m pytest tests/test_gpu_bench_contract.py::test_headline_workload_gates tests/test_gpu_errors.py -q -rf 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl" > $O/pytest.log; tail -5 $O/pytest.log
