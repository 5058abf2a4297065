#!/bin/bash
# Round-2 GPU session 8: whole -m gpu suite on the tree with the device grid builders, then the bench lines that carry the new
# keys (grid_build, scan_preparation, match_split) with the r02f counters behind the roofline block.
OUT=gpurun_out/r02g
mkdir -p $OUT
python -m pytest tests -m gpu -q 2>&1 | tail -15 > $OUT/pytest.log; tail -3 $OUT/pytest.log
python bench.py --workload C2 2>$OUT/C2_full.err | tail -1 > $OUT/C2_full.json
python bench.py --workload C3 --no-cpu-baseline 2>$OUT/C3_full.err | tail -1 > $OUT/C3_full.json
python bench.py --workload C1 2>$OUT/C1_full.err | tail -1 > $OUT/C1_full.json
python bench.py --workload C5 --particles 8192 --no-cpu-baseline 2>$OUT/C5_shard.err | tail -1 > $OUT/C5_shard.json
python - <<P
import json
for n in ("C2_full","C3_full","C1_full","C5_shard"):
    try:
        d=json.load(open("$OUT/%s.json"%n)); k=d["kernels_ms_per_step"]; r=d["roofline"]
        print("%-8s value %.4g ms/step %.4f lik %.4f beam %.4f pf %.4f | %s frac %s src %s" % (n,d["value"],d["ms_per_step"],k["likelihood"],k["beam"],k["pf"], r["bound"], r["frac"], r.get("counters_source")))
        print("    grid_build", json.dumps({a:b for a,b in d.get("grid_build",{}).items() if a!="what"}))
        print("    scan_prep", d.get("scan_preparation",{}).get("ms"), "match_split", d.get("match_split",{}).get("ms"))
    except Exception as e: print(n,"failed",e)
P
