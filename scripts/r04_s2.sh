#!/bin/bash
# round 4, session 2: staged update with the chip-wide sort behind stage_pack_kernel + results emitted by pf_apply; Morton-bits A/B
O=gpurun_out/r04b
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_update_staged.py tests/test_gpu_update_small.py tests/test_gpu_pf_fused.py -x -q 2>&1 | tail -25 > $O/staged_tests.log
tail -4 $O/staged_tests.log
timeout 300 python scripts/time_update_8d.py C2 200 > $O/time8d_C2.log 2>&1; grep -v RESULT $O/time8d_C2.log | tail -9
timeout 300 python scripts/time_update_8d.py C3 200 > $O/time8d_C3.log 2>&1; grep -v RESULT $O/time8d_C3.log | tail -9
timeout 300 python scripts/time_update_8d.py C1 500 n_b=3 n_s=96 > $O/time8d_64x96.log 2>&1; grep -v RESULT $O/time8d_64x96.log | tail -9
timeout 300 python scripts/time_update_8d.py C1 500 > $O/time8d_C1.log 2>&1; grep -v RESULT $O/time8d_C1.log | tail -9
timeout 300 python scripts/time_update_8d.py C2 300 n_s=96 n_b=3 > $O/time8d_4096x96.log 2>&1; grep -v RESULT $O/time8d_4096x96.log | tail -9
cp mcl_3dl_amd/libmcl3dl_hip.so /tmp/keep.so
for m in 16 8; do
  cp mcl_3dl_amd/variants/libmcl3dl_hip_m$m.so mcl_3dl_amd/libmcl3dl_hip.so
  timeout 300 python scripts/time_update_8d.py C2 200 > $O/time8d_C2_m$m.log 2>&1; echo "== morton bits $m"; grep -v RESULT $O/time8d_C2_m$m.log | tail -9
done
cp /tmp/keep.so mcl_3dl_amd/libmcl3dl_hip.so
