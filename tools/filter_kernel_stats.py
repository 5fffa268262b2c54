#!/usr/bin/env python3
"""Keep the library's kernels (dvo::) of a rocprofv3 --stats kernel_stats.csv and recompute the percentages: the synthetic-frame
generator of bench.py (torch elementwise kernels) otherwise fills the table.  python tools/filter_kernel_stats.py in.csv out.csv"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if "dvo::" in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in keep) or 1.0
with open(sys.argv[2], "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in keep:
        r["Percentage"] = "%.4f" % (100 * float(r["TotalDurationNs"]) / tot)
        w.writerow(r)
