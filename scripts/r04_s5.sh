#!/bin/bash
O=gpurun_out/r04e
mkdir -p $O
./profiles/launch_cost.bin > $O/launch_cost.txt 2>&1; cat $O/launch_cost.txt
timeout 600 python -m pytest tests/test_gpu_update_staged.py tests/test_gpu_pf_fused.py -x -q 2>&1 | tail -3
timeout 300 python scripts/time_update_8d.py C2 200 > $O/time8d_C2.log 2>&1; grep -v RESULT $O/time8d_C2.log | tail -8
timeout 300 python scripts/time_update_8d.py C2 300 n_s=96 n_b=3 > $O/time8d_4096x96.log 2>&1; grep -v RESULT $O/time8d_4096x96.log | tail -8
