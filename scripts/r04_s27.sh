#!/bin/bash
O=gpurun_out/r04y
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_group.py tests/test_gpu_group_state.py tests/test_gpu_adapter.py tests/test_gpu_batch_progressive.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
timeout 300 ./tests/cpp/group_all_devices.bin 2>&1 | tail -3
for w in C2 C3 C1; do
timeout 600 python bench.py --workload $w --in-process --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w in-process ms/step', d['ms_per_step'])"
done
