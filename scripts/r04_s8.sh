#!/bin/bash
O=gpurun_out/r04h
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_group_state.py tests/test_gpu_group.py -x -q 2>&1 | tail -30 | cut -c1-250 > $O/group_tests.log; tail -30 $O/group_tests.log
./tests/cpp/group_all_devices.bin 2>&1 | tail -4
