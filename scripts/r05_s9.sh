#!/bin/bash
# round 5 session 9: whole GPU suite + the default bench lines of C2 and C1 on the current tree
O=gpurun_out/r05i; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 900 python bench.py 2>$O/bench_C2.err | tail -1 > $O/bench_C2.json; python -c "
import json; d=json.load(open('$O/bench_C2.json')); print('C2 value %.4g ms %.4f 8d %.4f' % (d['value'], d['ms_per_step'], d['update_8d']['ms_per_update'])); print(json.dumps(d.get('engine_order'), indent=1)); print(json.dumps(d.get('headline'), indent=1)); print(json.dumps({k: d['map_update'][k] for k in ('wall_ms','first_measure_after_update_ms','same_measure_steady_ms','first_match_split_after_update_ms','same_match_split_steady_ms','cell_grid_merges','cell_grid_rebuilds')}, indent=1)); print(d['roofline']['frac'], d['roofline'].get('counters_note'), d['roofline'].get('frac_of_update'))"
timeout 600 python bench.py --workload C1 2>$O/bench_C1.err | tail -1 > $O/bench_C1.json; python -c "
import json; d=json.load(open('$O/bench_C1.json')); print('C1 value %.4g ms %.4f 8d %.4f' % (d['value'], d['ms_per_step'], d['update_8d']['ms_per_update'])); print(d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('counters_note'))"
tail -3 $O/bench_C2.err
