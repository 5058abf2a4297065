#!/bin/bash
O=gpurun_out/r04t
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_c4c5.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -4 | tee $O/tests.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in 0 1; do
  MCL3DL_HIP_OPTIONS=strict_rows=$r timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rows$r -o p -- python bench.py --workload C5 --particles 8192 --no-extras --no-cpu-baseline --steps 10 --warmup 3 --overlap-models 0 > $O/shard_rows$r.out 2>&1
  echo "== strict_rows=$r (C5 shard 8192 x 65536)"; python - <<PY
import csv,glob
for f in glob.glob("$O/prof_rows$r/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if any(k in row["Name"] for k in ("likelihood_tiled","lik_strict","lik_finalize")):
            print("  %-70s calls %4s avg %9.1f us min %9.1f" % (row["Name"].split("(")[0][:70], row["Calls"], float(row["AverageNs"])/1e3, float(row["MinNs"])/1e3))
PY
  tail -1 $O/shard_rows$r.out | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ms/step',d['ms_per_step'],d['kernels_ms_per_step'])"
  find $O/prof_rows$r -type f ! -name "*stats*" -delete
done
for r in 0 1; do
  MCL3DL_HIP_OPTIONS=strict_rows=$r timeout 600 python bench.py --workload C5 --no-extras --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('full C5 strict_rows=$r ms/step',d['ms_per_step'],d['kernels_ms_per_step'])"
done
