#!/bin/bash
# same box: the library before the quarter bounds (variants/libmcl3dl_hip_prev.so, built from commit 7c086d3) against the tree's
# (the box's copy of the tree is scratch: the library file itself is swapped, so that tools/libmcl3dl_benchloop.so sees the same one)
O=gpurun_out/r04ac
mkdir -p $O
cp mcl_3dl_amd/libmcl3dl_hip.so /tmp/cur.so
for rep in 1 2; do
for lib in prev cur; do
  if [ $lib = prev ]; then cp mcl_3dl_amd/variants/libmcl3dl_hip_prev.so mcl_3dl_amd/libmcl3dl_hip.so; else cp /tmp/cur.so mcl_3dl_amd/libmcl3dl_hip.so; fi
  for j in 0 0.045; do
    timeout 600 python bench.py --map-jitter $j --no-extras --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib jitter $j: lik %.4f ms, update %.4f' % (d['kernels_ms_per_step']['likelihood'], d['ms_per_step']))"
  done
done
done
cp /tmp/cur.so mcl_3dl_amd/libmcl3dl_hip.so
