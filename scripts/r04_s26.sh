#!/bin/bash
O=gpurun_out/r04x
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_group.py tests/test_gpu_group_state.py tests/test_gpu_distributed.py tests/test_gpu_adapter.py -x -q 2>&1 | tail -4 | tee $O/tests.txt
timeout 600 python bench.py --workload C2 --in-process --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/inproc_C2.json
timeout 600 python bench.py --workload C3 --in-process --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/inproc_C3.json
python - <<'PY'
import json
for w in ("C2","C3"):
    d=json.load(open("gpurun_out/r04x/inproc_%s.json"%w))
    print(w,"in-process ms/step",d["ms_per_step"],"value",d["value"], d["in_process_group"].get("ms_per_update"), d["in_process_group"].get("collective"))
PY
