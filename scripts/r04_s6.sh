#!/bin/bash
O=gpurun_out/r04f
mkdir -p $O
./profiles/launch_cost.bin > $O/launch_cost.txt 2>&1; tail -3 $O/launch_cost.txt
timeout 900 python -m pytest tests/test_gpu_update_staged.py tests/test_gpu_scan_prep.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
timeout 300 python scripts/time_update_8d.py C2 200 > $O/time8d_C2.log 2>&1; grep -v RESULT $O/time8d_C2.log | tail -8
timeout 300 python scripts/time_update_8d.py C3 200 > $O/time8d_C3.log 2>&1; grep -v RESULT $O/time8d_C3.log | tail -8
timeout 300 python scripts/time_update_8d.py C1 500 > $O/time8d_C1.log 2>&1; grep -v RESULT $O/time8d_C1.log | tail -8
timeout 300 python scripts/time_update_8d.py C1 500 n_s=96 n_b=3 > $O/time8d_64x96.log 2>&1; grep -v RESULT $O/time8d_64x96.log | tail -8
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest_all.log; head -3 $O/pytest_all.log
