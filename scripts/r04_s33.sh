#!/bin/bash
for opt in "poll_sync=2" "poll_sync=1"; do
  for sl in 0 100000; do
    for rep in 1 2 3; do
      MCL3DL_HIP_OPTIONS=$opt MCL3DL_HIP_BATCH_SLICE=$sl timeout 300 python scripts/time_route_a.py C2 40 $sl 2>&1 | grep -E "slice|repetition|error" | head -2 | cut -c1-200
    done
  done
done
