#!/bin/bash
# round 4, session 1: the staged host-buffer update — parity tests first, then the A/B timings, then the whole GPU suite
mkdir -p gpurun_out/r04a
O=gpurun_out/r04a
timeout 600 python -m pytest tests/test_gpu_update_staged.py -x -q 2>&1 | tail -25 > $O/staged_tests.log
cat $O/staged_tests.log | tail -8
timeout 300 python scripts/time_update_8d.py C2 200 > $O/time8d_C2.log 2>&1; tail -14 $O/time8d_C2.log
timeout 300 python scripts/time_update_8d.py C3 200 > $O/time8d_C3.log 2>&1; grep RESULT $O/time8d_C3.log
timeout 300 python scripts/time_update_8d.py C1 500 n_b=3 n_s=96 > $O/time8d_64x96.log 2>&1; grep RESULT $O/time8d_64x96.log
timeout 300 python scripts/time_update_8d.py C2 300 n_s=96 n_b=3 > $O/time8d_4096x96.log 2>&1; grep RESULT $O/time8d_4096x96.log
timeout 300 python scripts/time_update_8d.py C2 300 n_s=512 > $O/time8d_4096x512.log 2>&1; grep RESULT $O/time8d_4096x512.log
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_all.log; tail -6 $O/pytest_all.log
