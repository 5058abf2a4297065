#!/bin/bash
# kernel + HIP API statistics of a stream of map updates (scripts/time_map_update.py)
O=gpurun_out/r04n
W=${1:-C3}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d $O/prof -o mu -- python scripts/time_map_update.py $W 20 > $O/prof.log 2>&1
for f in $(find $O/prof -name "*kernel_stats.csv"); do echo == $f; head -${2:-30} $f | cut -c1-150,300-420; done
find $O/prof -type f ! -name "*stats*" -delete
