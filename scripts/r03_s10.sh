#!/bin/bash
# round 3, session 10: the whole -m gpu suite + smoke + default bench on one box
OUT=gpurun_out/r03j
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 | tee $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.log
python bench.py 2>$OUT/C2_default.err | tail -1 > $OUT/C2_default.json
python - <<P
import json
d=json.load(open("$OUT/C2_default.json")); k=d["kernels_ms_per_step"]; r=d["roofline"]
print("C2 value %.4g value_8d %.4g ms/step %.4f lik %.4f pf %.4f | bound %s frac %.3f vs measured %s | hbm_target %s" % (d["value"], d["value_8d"], d["ms_per_step"], k["likelihood"], k["pf"], r["bound"], r["frac"], r["frac_vs_measured_peaks"], r["hbm_target"]["frac"]))
print("prep", d["scan_preparation"]["ms"], d["scan_preparation"]["ms_max"], "cpu", d["scan_preparation"].get("cpu_reference_ms"), "split", d["match_split"]["ms"], "jitter", d["map_jitter"]["vs_lattice"], "route_a", d["route_a"]["ms_per_update"])
print("resample", json.dumps(d["resample"]))
P
