#!/bin/bash
# Extra PMC passes (pipe activity, TLB, TCP stalls) for bench.py's workload; one counter family per pass, each under timeout.
# usage: profiles/run_pmc_extra.sh <tag> [bench args...]
set -u
TAG=${1:-x}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
BENCH="python bench.py --steps 5 --warmup 1 --prewarm-ms 60 --no-cpu-baseline --no-extras $*"
i=0
for SET in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/pmc_x$i" -o "$TAG" -- $BENCH > "$OUT/pmc_x$i.log" 2>&1
  echo "pmc [$SET] rc=$?"
done
