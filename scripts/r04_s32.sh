#!/bin/bash
# every synchronisation polled (poll_sync = 2) against the default: the rows either side of the update
for p in 1 2 1 2; do
  MCL3DL_HIP_OPTIONS=poll_sync=$p timeout 600 python bench.py --workload C2 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('poll_sync=$p: scan_prep %.4f split %.4f resample %.4f/%.4f reductions %.4f iteration %.4f map_update %.3f 8d %.4f' % (d['scan_preparation']['ms'], d['match_split']['ms'], d['resample']['ms'], d['resample']['ms_device_resident'], d['post_update_reductions']['ms'], d['filter_iteration']['ms'], d['map_update']['wall_ms'], d['update_8d']['ms_per_update']))"
done
