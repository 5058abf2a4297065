#!/bin/bash
O=gpurun_out/r04i
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_c4c5.py -x -q 2>&1 | grep -E "Error|error|assert|FAILED|passed|failed" | head -12 | cut -c1-300
for k in 0 1; do
  MCL3DL_HIP_OPTIONS="strict_skew=$k" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof3_skew$k -o c5 -- python bench.py --workload C5 --particles 8192 --no-extras --no-cpu-baseline --steps 10 --warmup 3 --prewarm-ms 100 --overlap-models 0 > $O/prof3_skew$k.log 2>&1
  find $O/prof3_skew$k -name "*_kernel_trace.csv" -delete
done
