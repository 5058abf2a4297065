#!/bin/bash
# The commands behind DESIGN.md section 6 (run from the repo root on a box with one MI355X; through gpurun here:
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash scripts/reproduce_round1.sh').
# Output: gpurun_out/r01g_bench_*.json (one bench.py line each); profiles via profiles/run_profiles.sh.
set -u
mkdir -p gpurun_out
run() { name=$1; shift; timeout 600 python bench.py "$@" 2>/dev/null | tail -1 > gpurun_out/r01g_bench_$name.json; echo "$name done"; }
run C2                                   # headline: 4096 particles x 16 384 points, 1 M-pt map (includes the CPU baseline)
run C1 --workload C1
run C3 --workload C3
run C3_stress --workload C3 --beam-points 16384 --no-cpu-baseline
run C4 --workload C4 --particles 32768 --no-cpu-baseline
run C4_8pt --workload C4 --scan-points 8 --no-cpu-baseline
run C5 --workload C5 --particles 8192 --no-cpu-baseline
run C2_strict --strict-order 1 --no-cpu-baseline
# per-kernel statistics + PMC passes (separate rocprofv3 runs, each under `timeout`):
#   bash profiles/run_profiles.sh r01g_C2 --workload C2 && python profiles/summarize_pmc.py r01g_C2
#   bash profiles/run_profiles.sh r01g_C3 --workload C3 && python profiles/summarize_pmc.py r01g_C3
#   bash profiles/run_pmc_extra.sh r01h_C2 --workload C2
