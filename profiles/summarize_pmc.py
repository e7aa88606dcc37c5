#!/usr/bin/env python3
"""Reduces the three rocprofv3 --pmc passes of scratch/pmc_conv.sh / pmc_any.sh (gpurun_out/pmc_<tag>/pass{1,2,3}) to one small CSV:
pass, counter, average value per dispatch of the kernels whose name contains <pattern>, dispatches, description.

usage: summarize_pmc.py <gpurun_out/pmc_tag dir> <kernel name pattern> <description> <out.csv>"""
import collections
import csv
import glob
import sys

root, pat, what, out = sys.argv[1:5]
rows = []
for i in (1, 2, 3):
    fs = glob.glob('%s/pass%d/*/*counter_collection.csv' % (root, i))
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if pat in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()):
        rows.append(("pmc%d" % i, k, "%.6e" % (sum(v) / len(v)), len(v), what))
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["pass", "counter", "avg_value_per_dispatch", "dispatches", "what"])
    w.writerows(rows)
print(out, len(rows), "rows")
