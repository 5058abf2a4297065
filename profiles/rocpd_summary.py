#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (<out>_results.db) into the per-kernel summary CSV kept under profiles/.
usage: rocpd_summary.py results.db > rNN_kernel_stats.csv"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage", "MinUs", "MaxUs", "VGPR", "SGPR", "LDS", "Grid", "Workgroup"])
rows = db.execute(
    "select name, count(*), sum(duration)/1000.0, avg(duration)/1000.0, min(duration)/1000.0, max(duration)/1000.0, "
    "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name "
    "order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows) or 1.0
for r in rows:
    w.writerow([r[0], r[1], "%.3f" % r[2], "%.3f" % r[3], "%.3f" % (100.0 * r[2] / total), "%.3f" % r[4], "%.3f" % r[5],
                r[6], r[7], r[8], r[9], r[10]])
