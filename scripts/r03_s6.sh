#!/bin/bash
# round 3, session 6: strict-sum kernel A/B on ONE box (old = HEAD build in variants/), with the per-kernel split
OUT=gpurun_out/r03f
mkdir -p $OUT
OLD=$PWD/mcl_3dl_amd/variants/libmcl3dl_hip_oldstrict.so
for rep in 1 2; do
for V in new old; do
  if [ $V = old ]; then export MCL3DL_HIP_LIB=$OLD; else unset MCL3DL_HIP_LIB; fi
  python bench.py --workload C2 --no-cpu-baseline --no-extras --strict-order 1 2>/dev/null | tail -1 > $OUT/C2_strict_$V.json
  python bench.py --workload C5 --particles 8192 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $OUT/C5_$V.json
  python - <<P
import json
for n in ("C2_strict_$V","C5_$V"):
    d=json.load(open("$OUT/%s.json"%n)); k=d["kernels_ms_per_step"]
    print("%-16s ms/step %.4f lik %.4f beam %.4f pf %.4f" % (n,d["ms_per_step"],k["likelihood"],k["beam"],k["pf"]))
P
done
done
unset MCL3DL_HIP_LIB
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_new -o c5 -- python bench.py --workload C5 --particles 8192 --no-cpu-baseline --no-extras --steps 5 --warmup 1 --prewarm-ms 60 > $OUT/c5_new_rocprof.log 2>&1
export MCL3DL_HIP_LIB=$OLD
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_old -o c5 -- python bench.py --workload C5 --particles 8192 --no-cpu-baseline --no-extras --steps 5 --warmup 1 --prewarm-ms 60 > $OUT/c5_old_rocprof.log 2>&1
python - <<P
import csv
for v in ("new","old"):
    for r in csv.DictReader(open("$OUT/stats_%s/c5_kernel_stats.csv"%v)):
        if "strict" in r["Name"] or "tiled" in r["Name"] or "finalize" in r["Name"]:
            print(v, r["Name"][:60], r["Calls"], "%.1f us"%(float(r["AverageNs"])/1e3))
P
