#!/bin/bash
# round 5, session 19: kernel-trace timeline of the host-buffer update (C2, C3)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05s19; mkdir -p $O
for W in C2 C3; do
  PYTHONPATH=. timeout 280 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$W -o tl -- python scripts/r05_timeline_8d.py $W > $O/run_$W.log 2>&1
  F=$(find $O/trace_$W -name "*kernel_trace.csv" | head -1)
  ( grep "update_8d under" $O/run_$W.log; python scripts/r05_timeline_8d.py --table "$F" ) > $O/timeline_8d_$W.txt 2>&1
  rm -rf $O/trace_$W
  head -20 $O/timeline_8d_$W.txt
done
