#!/bin/bash
O=gpurun_out/r04c
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_update_staged.py -x -q 2>&1 | tail -5
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest_all.log; tail -5 $O/pytest_all.log
