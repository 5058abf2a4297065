"""Route A (the node's pf_->measure(measure_func) through the drop-in C++ classes, tests/cpp/adapter_demo.bin) for several
slice sizes of the progressive batch (MCL3DL_HIP_BATCH_SLICE; 0 = automatic, a value above the particle count = one slice).
Run on the GPU box:   python scripts/time_route_a.py C2 [reps] [slice ...]"""
import json
import os
import sys

sys.path.insert(0, ".")
import bench  # noqa: E402
from mcl_3dl_amd.synthetic import make_config  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "C2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
slices = [int(a) for a in sys.argv[3:]] or [0]
sc = make_config(workload, seed=12345)
out = {}
for s in slices:
    os.environ["MCL3DL_HIP_BATCH_SLICE"] = str(s)
    r = bench.route_a(sc, (1.0, 1.0, 1.0), len(sc.scan_beam), reps)
    out[str(s)] = r
    bd = (r or {}).get("breakdown") or {}
    print(workload, "slice", s, ":", (r or {}).get("ms_per_update", r), "ms; begin %.1f us, waiting %.1f us, loop %.1f us" % (
        bd.get("measure_batch_begin_us", -1), bd.get("waiting_for_slices_us", -1), bd.get("reference_pf_loop_us", -1)), flush=True)
print(json.dumps({"workload": workload, "reps": reps, "route_a_by_slice": out}))
