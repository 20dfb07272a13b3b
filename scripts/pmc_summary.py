#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel: python scripts/pmc_summary.py <counter_collection.csv>... [-o out.csv]
One row per (kernel, counter): dispatches, mean value per dispatch."""
import collections
import csv
import sys

args = sys.argv[1:]
out = None
if "-o" in args:
    i = args.index("-o")
    out = args[i + 1]
    del args[i:i + 2]
acc = collections.defaultdict(lambda: [0, 0.0])
for path in args:
    for r in csv.DictReader(open(path)):
        k = (r["Kernel_Name"], r["Counter_Name"])
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
rows = sorted(((k[0], k[1], v[0], v[1] / v[0]) for k, v in acc.items()), key=lambda t: (t[0], t[1]))
w = csv.writer(open(out, "w", newline="") if out else sys.stdout)
w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
for r in rows:
    w.writerow([r[0], r[1], r[2], f"{r[3]:.1f}"])
