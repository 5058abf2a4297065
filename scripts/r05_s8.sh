#!/bin/bash
# round 5 session 8: scan_presorted (tests + what skipping the ordering launches is worth in the host-buffer update)
O=gpurun_out/r05h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_adapter.py tests/test_gpu_update_staged.py -x -q 2>&1 | tail -12 > $O/tests.log; tail -4 $O/tests.log
PYTHONPATH=. timeout 600 python scripts/r05_time_presorted.py C2 100 2>&1 | tail -12 > $O/presorted_C2.txt; cat $O/presorted_C2.txt
PYTHONPATH=. timeout 600 python scripts/r05_time_presorted.py C3 100 2>&1 | tail -12 > $O/presorted_C3.txt; cat $O/presorted_C3.txt
