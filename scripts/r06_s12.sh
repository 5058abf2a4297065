#!/bin/bash
# round 6 session 12: the beam model's last step (penalty count -> score) folded into the update's tail kernel: suite, then A/B on one box
O=gpurun_out/r06q; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl" | tail -5
run() { # name, switch, bench args
  MCL3DL_BEAM_TAIL="$2" timeout 900 python bench.py $3 2>$O/$1.err | tail -1 > $O/$1.json
  python - "$O/$1.json" "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels_ms_per_step"]
    print("%-20s ms/step %.4f 8d %s lik %.4f beam %.4f pf %.4f" % (sys.argv[2], d["ms_per_step"], d.get("ms_per_step_8d"), k["likelihood"], k["beam"], k["pf"]), flush=True)
except Exception as e:
    print(sys.argv[2], "no json", e)
PY
}
Q8="--steps 40 --warmup 5 --no-cpu-baseline"
for r in 1 2 3; do
  run C3_off_$r 0 "--workload C3 $Q8"
  run C3_on_$r 1 "--workload C3 $Q8"
done
run C3s_off 0 "--workload C3 --beam-points 48 --steps 40 --warmup 5 --no-extras --no-cpu-baseline"
run C5s_off 0 "--workload C5 --particles 8192 --steps 10 --warmup 2 --no-extras --no-cpu-baseline"
run C5s_on 1 "--workload C5 --particles 8192 --steps 10 --warmup 2 --no-extras --no-cpu-baseline"
run C3s_on 1 "--workload C3 --beam-points 48 --steps 40 --warmup 5 --no-extras --no-cpu-baseline"
