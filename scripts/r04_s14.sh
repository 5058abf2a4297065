#!/bin/bash
# progressive batch (measure_batch_begin/_wait/_end): tests, then Route A at C2/C3/C1 for several slice sizes
O=gpurun_out/r04m
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch_progressive.py tests/test_gpu_adapter.py tests/test_gpu_update_staged.py -x -q 2>&1 | tail -15 > $O/tests.txt
cat $O/tests.txt
timeout 600 python scripts/time_route_a.py C2 40 0 256 512 1024 2048 100000 2>&1 | tee $O/route_a_C2.txt | grep -v "^{"
timeout 600 python scripts/time_route_a.py C3 40 0 512 1024 100000 2>&1 | tee $O/route_a_C3.txt | grep -v "^{"
timeout 600 python scripts/time_route_a.py C1 40 0 100000 2>&1 | tee $O/route_a_C1.txt | grep -v "^{"
