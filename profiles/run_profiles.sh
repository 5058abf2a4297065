#!/bin/bash
# Runs on the GPU box (through gpurun): per-kernel stats + PMC passes for bench.py's workload.
# Every rocprofv3 invocation is wrapped in `timeout` (a failed counter config can hang the tool's finalisation).
# usage: profiles/run_profiles.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
BENCH="python bench.py --steps 5 --warmup 1 --prewarm-ms 60 --no-cpu-baseline --no-extras $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o "$TAG" -- $BENCH > "$OUT/stats.log" 2>&1
echo "stats rc=$?"
for SET in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"; do
  NAME=$(echo "$SET" | tr ' ' '_' | cut -c1-40)
  timeout 240 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/pmc_$NAME" -o "$TAG" -- $BENCH > "$OUT/pmc_$NAME.log" 2>&1
  echo "pmc [$SET] rc=$?"
done
# summaries next to the raw data, then drop the per-dispatch CSVs (gpurun merges at most 64 MiB of gpurun_out/ back)
python profiles/summarize_pmc.py "$TAG" --out "$OUT"
find "$OUT" -name "*_counter_collection.csv" -delete
find "$OUT"/pmc_* -name "*_kernel_trace.csv" -delete
find "$OUT" -name "*_kernel_trace.csv" -size +4M -delete
du -sh "$OUT"
