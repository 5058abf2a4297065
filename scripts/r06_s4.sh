#!/bin/bash
# round 6 session 4: kernel trace of the C1 update, default (caller-order rows) against strict_order = 0 (fp64 tree)
O=gpurun_out/r06e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
Q="--workload C1 --steps 200 --warmup 20 --no-extras --no-cpu-baseline --timing-mask 0"
for M in 2 0; do
  MCL3DL_HIP_OPTIONS="strict_order=$M" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$M -o t$M -- python bench.py $Q > $O/run$M.log 2>&1
  f=$(find $O/t$M -name "*kernel_stats.csv" | head -1); echo "== strict_order $M"; head -4 $f | cut -c1-90,200-330
  tail -1 $O/run$M.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
done
