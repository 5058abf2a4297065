"""The host-buffer update of SURVEY.md 8d under `rocprofv3 --kernel-trace`: a few updates through mcl3dl_hip_measure_update
(pageable arrays, default options), so that the kernel trace shows every launch of one update with its start, end and the gap
to the launch before it.   (driver: scripts/r05_s19.sh; the table is made by --table <kernel_trace.csv>)"""
import csv
import sys

import numpy as np

if len(sys.argv) > 2 and sys.argv[1] == "--table":
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the LAST update: from the last stage_pack_kernel on
    starts = [i for i, r in enumerate(rows) if "stage_pack_kernel" in r["Kernel_Name"]]
    for which, first in (("last update", starts[-1]), ("the one before", starts[-2])):
        t0 = int(rows[first]["Start_Timestamp"])
        prev_end = None
        print("%s (times in us from the start of its first kernel)" % which)
        print("%-58s %9s %9s %9s %9s" % ("kernel", "start", "end", "duration", "gap"))
        for r in rows[first:]:
            if r is not rows[first] and "stage_pack_kernel" in r["Kernel_Name"]:
                break
            s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
            name = r["Kernel_Name"].replace("mcl3dl::", "").replace("void ", "").split("(")[0][:58]
            print("%-58s %9.2f %9.2f %9.2f %9s" % (name, s / 1e3, e / 1e3, (e - s) / 1e3,
                                                 "" if prev_end is None else "%.2f" % ((s - prev_end) / 1e3)))
            prev_end = e
        print()
    sys.exit(0)

from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_config

sc = make_config(sys.argv[1] if len(sys.argv) > 1 else "C2", seed=12345)
n_p, n_b = len(sc.poses), len(sc.scan_beam)
eng = capi.Engine(0)
eng.set_map(sc.map_xyz, sc.map_label, stamp=1, dist_weight=(1.0, 1.0, 1.0))
eng.set_likelihood_params()
eng.set_beam_params(num_points=max(n_b, 1), dda_grid_size=0.2)
poses = np.ascontiguousarray(sc.poses, np.float32)
w0 = np.full(n_p, 1.0 / n_p, np.float32)
scan = np.ascontiguousarray(sc.scan_lik, np.float32)
beam = np.ascontiguousarray(sc.scan_beam, np.float32) if n_b else None
lab = np.ascontiguousarray(sc.scan_beam_label, np.uint32) if n_b else None
org = np.ascontiguousarray(sc.origins, np.float32)
o_lik, o_q, o_b = (np.zeros(n_p, np.float32) for _ in range(3))
ms, _ = eng.time_measure_update(poses, w0, w0.copy(), scan, beam, lab, org, o_lik, o_q, o_b, 200, warm_ms=150.0)
print("update_8d under the tracer: %.4f ms" % ms)
