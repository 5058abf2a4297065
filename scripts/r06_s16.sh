#!/bin/bash
# round 6 session 16: the tail fusions in the device-group path (per rank: lik_finalize's sum + the beam model's last step inside the
# first pf::measure kernel): the group / distributed / fuzz tests, then the in-process group against the previous library
O=gpurun_out/r06u; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_group.py tests/test_gpu_group_state.py tests/test_gpu_distributed.py tests/test_gpu_pf_fused.py tests/test_gpu_api_fuzz.py tests/test_gpu_adapter.py tests/test_gpu_c4c5.py -m gpu -q 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl" | tail -5
PREV=$PWD/mcl_3dl_amd/variants/libmcl3dl_hip_prev.so
for r in 1 2 3; do
for v in prev new; do
  LIB=""; [ $v = prev ] && LIB=$PREV
  for w in C2 C3; do
  MCL3DL_HIP_LIB="$LIB" timeout 900 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>$O/${w}_$v$r.err | tail -1 > $O/${w}_$v$r.json
  python - "$O/${w}_$v$r.json" "${w}_$v$r" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); ip=d["in_process_group"]
    print("%-12s ms/step %.4f 8d %.4f in-process group %.4f ms (%s)" % (sys.argv[2], d["ms_per_step"], d["ms_per_step_8d"], ip["ms_per_update"], ip["collective"]), flush=True)
except Exception as e:
    print(sys.argv[2], "no json", e)
PY
  done
done
done
