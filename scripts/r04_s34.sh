#!/bin/bash
MCL3DL_HIP_OPTIONS=poll_sync=2 MCL3DL_HIP_BATCH_SLICE=0 timeout 300 python scripts/time_route_a.py C2 40 0 2>&1 | grep -E "slice [0-9]+ :|repetition" | head -2 | cut -c1-160
MCL3DL_HIP_OPTIONS=poll_sync=2 timeout 300 python scripts/time_route_a.py C3 40 0 2>&1 | grep -E "slice [0-9]+ :|repetition" | head -2 | cut -c1-160
MCL3DL_HIP_OPTIONS=poll_sync=2 timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^FAILED" | tail -3
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^FAILED" | tail -3
