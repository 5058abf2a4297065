#!/bin/bash
# round 6 session 2: where the caller-order replay's time goes at C2 (kernel trace of strict_order = 1)
O=gpurun_out/r06b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
Q="--steps 20 --warmup 3 --no-extras --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $R/$O/s1 -o s1 -- python $R/bench.py --workload C2 --strict-order 1 $Q > $R/$O/s1.log 2>&1
find $R/$O/s1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/$O/C2s1_kernel_stats.csv
head -8 $R/$O/C2s1_kernel_stats.csv | cut -c1-200
