#!/bin/bash
# r03 session 28: the per-update scan arrays in one device block / one H2D copy, A/B on one box at the reference's own sizes (C1):
# mcl_3dl_amd/variants/libmcl3dl_hip_prev.so = the tree before the change. Measured: upload_scan 0.034 -> 0.024 ms; the whole
# host-buffer update stays at 0.058 ms — at this size it is bound by the host's half-dozen API calls and the one synchronisation,
# not by what the GPU does.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do for v in prev new; do
  if [ $v = new ]; then unset MCL3DL_HIP_LIB; else export MCL3DL_HIP_LIB=$GRAFT_REPO_ROOT/mcl_3dl_amd/variants/libmcl3dl_hip_prev.so; fi
  python scripts/time_host_path.py C1 2>&1 | grep -E "4096: upload_scan|measure_update" | head -2 | tr '\n' ' '; echo " [$v]"
done; done
