#!/bin/bash
O=gpurun_out/r04r
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch_progressive.py tests/test_gpu_adapter.py -x -q 2>&1 | tail -4 | tee $O/tests.txt
timeout 600 python scripts/time_route_a.py C3 40 0 512 100000 2>&1 | tee $O/route_a_C3.txt | grep -v "^{"
timeout 600 python scripts/time_route_a.py C2 40 0 2>&1 | tee $O/route_a_C2.txt | grep -v "^{"
