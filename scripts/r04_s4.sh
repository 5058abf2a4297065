#!/bin/bash
# round 4, session 4: two-launch tail (pf_norm), per-particle update form, one-launch-per-pass sort
O=gpurun_out/r04d
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_update_staged.py tests/test_gpu_sort.py tests/test_gpu_update_small.py tests/test_gpu_pf_fused.py tests/test_gpu_scan_prep.py -x -q 2>&1 | tail -15 | cut -c1-200 > $O/tests.log; tail -6 $O/tests.log
timeout 300 python scripts/time_update_8d.py C2 200 > $O/time8d_C2.log 2>&1; grep -v RESULT $O/time8d_C2.log | tail -8
timeout 300 python scripts/time_update_8d.py C3 200 > $O/time8d_C3.log 2>&1; grep -v RESULT $O/time8d_C3.log | tail -8
timeout 300 python scripts/time_update_8d.py C2 300 n_s=96 n_b=3 > $O/time8d_4096x96.log 2>&1; grep -v RESULT $O/time8d_4096x96.log | tail -8
timeout 300 python scripts/time_update_8d.py C2 300 n_s=512 > $O/time8d_4096x512.log 2>&1; grep -v RESULT $O/time8d_4096x512.log | tail -8
timeout 300 python scripts/time_update_8d.py C1 500 > $O/time8d_C1.log 2>&1; grep -v RESULT $O/time8d_C1.log | tail -8
cp mcl_3dl_amd/libmcl3dl_hip.so /tmp/keep.so
cp mcl_3dl_amd/variants/libmcl3dl_hip_m16.so mcl_3dl_amd/libmcl3dl_hip.so
timeout 300 python scripts/time_update_8d.py C2 200 > $O/time8d_C2_m16.log 2>&1; echo "== morton bits 16"; grep -v RESULT $O/time8d_C2_m16.log | tail -8
cp /tmp/keep.so mcl_3dl_amd/libmcl3dl_hip.so
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest_all.log; tail -5 $O/pytest_all.log
