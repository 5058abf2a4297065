#!/bin/bash
# round 5, session 24: the GPU suite, smoke and the default bench line on the last tree of the round (bench first: a box that
# has just run the suite for three minutes clocks ~10 % lower)
O=gpurun_out/r05s24; mkdir -p $O
timeout 600 python bench.py 2>$O/bench.err | tail -1 > $O/bench_C2.json; python -c "
import json; d=json.load(open('$O/bench_C2.json')); print(d['value'], d['ms_per_step'], d['headline']['ms_per_update_8d'], d['roofline']['frac'], d['result_check']['max_rel_err_vs_cpu'])"
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $O/smoke.log)"
timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl" | tail -60 > $O/pytest_gpu.log
grep -E "passed|failed" $O/pytest_gpu.log | tail -1
