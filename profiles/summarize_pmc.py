#!/usr/bin/env python3
"""Collapse gpurun_out/prof_<tag>/pmc_*/<tag>_counter_collection.csv into profiles/<tag>_pmc_summary.csv (mean per launch
per kernel) and copy the --stats kernel summary next to it.   usage: summarize_pmc.py <tag> [kernel-substring] [--out DIR]
profiles/run_profiles.sh runs it ON the GPU box with --out gpurun_out/prof_<tag> and then deletes the raw per-dispatch CSVs
(tens of MB per session: gpurun merges at most 64 MiB back); copy the two summaries from there into profiles/."""
import collections
import csv
import glob
import hashlib
import os
import shutil
import sys

# what the counters belong to: sha256 of the kernel sources (and the build flags) they were collected from — bench.py
# refuses counters whose sources differ from the tree it runs in — and the commit (GIT_HEAD=... in the environment of the
# profiling command: the GPU box has no .git)
SOURCES = ["likelihood_kernels.h", "beam_kernels.h", "device_math.h", "map_structs.h", "map_compiler.h", "Makefile"]


def source_rows(root="."):
    rows = [["__source__", "git_head", os.environ.get("GIT_HEAD", "unknown"), 0]]
    for name in SOURCES:
        path = os.path.join(root, "mcl_3dl_amd", "csrc", name)
        rows.append(["__source__", name, hashlib.sha256(open(path, "rb").read()).hexdigest()[:16], 0])
    return rows


tag = sys.argv[1]
out_dir = 'profiles'
if '--out' in sys.argv:
    i = sys.argv.index('--out')
    out_dir = sys.argv[i + 1]
    del sys.argv[i:i + 2]
out = {}
for f in sorted(glob.glob('gpurun_out/prof_%s/pmc_*/%s_counter_collection.csv' % (tag, tag))):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'mcl3dl' not in k:
            continue
        a = agg[k][r['Counter_Name']]
        a[0] += float(r['Counter_Value'])
        a[1] += 1
    for k, cs in agg.items():
        for c, (s, n) in cs.items():
            out.setdefault(k, {})[c] = (s / n, n)
w = csv.writer(open('%s/%s_pmc_summary.csv' % (out_dir, tag), 'w'))
w.writerow(["kernel", "counter", "mean_per_launch", "launches"])
for row in source_rows():
    w.writerow(row)
for k in sorted(out):
    for c in sorted(out[k]):
        w.writerow([k, c, "%.6g" % out[k][c][0], out[k][c][1]])
shutil.copy('gpurun_out/prof_%s/stats/%s_kernel_stats.csv' % (tag, tag), '%s/%s_kernel_stats.csv' % (out_dir, tag))
for k in out:
    if len(sys.argv) > 2 and sys.argv[2] in k:
        print(k, {c: "%.4g" % v[0] for c, v in out[k].items()})
