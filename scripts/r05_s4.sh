#!/bin/bash
# round 5 session 4: the stateful API-sequence fuzz and the engine-order adapter test
O=gpurun_out/r05d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_adapter.py -x -q 2>&1 | tail -30 > $O/adapter.log; tail -3 $O/adapter.log
timeout 1500 python -m pytest tests/test_gpu_api_fuzz.py -q -x --durations=5 2>&1 | tail -120 > $O/api_fuzz.log; tail -60 $O/api_fuzz.log
