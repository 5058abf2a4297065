#!/bin/bash
# ONE command for the first multi-GPU lease: everything N > 1 that has never run on hardware, in the
# order of increasing cost, each step under its own timeout, results under gpurun_out/multi_gpu/ and a summary on stdout.
#   usage: bash scripts/multi_gpu.sh [N]        (N = GPUs to use, default: every GPU rocminfo / torch sees)
# 1. tests/cpp/group_all_devices.bin   plain C over the C ABI: ONE device group over every GPU, the update's single collective
#                                      through RCCL (ncclCommInitAll, one rank per GPU), checked against one context
# 2. pytest -m gpu tests/test_gpu_group*.py tests/test_gpu_distributed.py   the suite's multi-device cases
# 3. bench.py --gpus k, k = 1, 2, 4, ... N   one process per GPU (torch.distributed / RCCL), weak scaling (the contract's line),
#                                      then strong scaling at N; and the in-process group (one process, N worker threads)
# 4. prints, per k: value, ms_per_step, the collective's share, RCCL's rank count and how many updates went through RCCL
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
N=${1:-$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)}
O=gpurun_out/multi_gpu; mkdir -p $O
echo "== $N GPU(s)"
echo "== 1. device group over every GPU through the C ABI (RCCL in-process)"
timeout 600 tests/cpp/group_all_devices.bin 20000 2>&1 | tee $O/group_all_devices.txt | tail -5
echo "== 2. the suite's multi-device cases"
timeout 1500 python -m pytest tests/test_gpu_group.py tests/test_gpu_group_state.py tests/test_gpu_distributed.py tests/test_gpu_batch_progressive.py -m gpu -q 2>&1 | tail -4 | tee $O/pytest_multi.txt
line() { # tag, n, extra bench args
  local port=$((29500 + RANDOM % 2000))
  if [ "$2" -eq 1 ] || [[ "$1" == inproc_* ]]; then   # (the in-process group is ONE process with a worker thread per GPU)
    timeout 1200 python bench.py --gpus $2 --steps 20 --warmup 3 $3 2>$O/$1.err | tail -1 > $O/$1.json
  else
    timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $2 --master-addr 127.0.0.1 --master-port $port \
      bench.py --gpus $2 --steps 20 --warmup 3 $3 2>$O/$1.err | tail -1 > $O/$1.json
  fi
  python - "$O/$1.json" "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = d.get("kernels_ms_per_step", {})
    g = d.get("in_process_group") or {}
    print("%-18s n_gpus %d scaling %-6s value %.4g evals/s ms/step %.4f collective %.4f ms | in-process group: %s" % (
        sys.argv[2], d["n_gpus"], d["scaling"], d["value"], d["ms_per_step"], k.get("collective", 0.0),
        ("%.4f ms/update, collective %s, %s" % (g.get("ms_per_update", float("nan")), g.get("collective"), g.get("collectives")))
        if g and "error" not in g else g.get("error", "-")), flush=True)
except Exception as e:
    print(sys.argv[2], "no bench line:", e, flush=True)
PY
}
echo "== 3. bench.py, one process per GPU (weak scaling), then strong scaling and the in-process group at N"
k=1
while [ $k -le $N ]; do
  line weak_$k $k "--no-extras"
  k=$((k * 2))
done
[ $N -gt 1 ] && line strong_$N $N "--scaling strong --no-extras"
line c4_$N $N "--workload C4 --scaling strong --no-extras"
line c5_$N $N "--workload C5 --scaling strong --no-extras"
line inproc_$N $N "--in-process"
echo "== RCCL: $(python - <<'PY'
import torch
print("torch", torch.__version__, "devices", torch.cuda.device_count(), "nccl", ".".join(map(str, torch.cuda.nccl.version())))
PY
)"
echo "== done: $O/"
