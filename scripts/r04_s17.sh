#!/bin/bash
O=gpurun_out/r04p
mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_map_path.py tests/test_gpu_parity.py tests/test_gpu_kats.py tests/test_gpu_fuzz.py tests/test_gpu_update_small.py tests/test_gpu_graph.py tests/test_gpu_adapter.py tests/test_gpu_group.py -x -q 2>&1 | tail -12 | tee $O/tests.txt
