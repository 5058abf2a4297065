#!/bin/bash
O=gpurun_out/r04o
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bound.py tests/test_gpu_map_path.py tests/test_gpu_index_random.py tests/test_gpu_parity.py tests/test_gpu_defer.py -x -q 2>&1 | tail -8 | tee $O/tests.txt
timeout 300 python scripts/time_map_update.py C2 8 2>&1 | tee $O/map_update_C2.txt
bash scripts/r04_s15.sh C2 14
