#!/usr/bin/env python3
"""Print a window of consecutive kernels (duration, gap to the previous one) from a rocprofv3 --kernel-trace CSV."""
import csv, glob, sys
rows = []
for fn in glob.glob(sys.argv[1]):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("dvo::", "").split("(")[0][:44],
                     r.get("Grid_Size") or r.get("Grid_Size_X") or ""))
rows.sort()
start = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 6
n = int(sys.argv[3]) if len(sys.argv) > 3 else 70
prev = None
for s, e, name, g in rows[start:start + n]:
    print("%-46s grid %9s dur %7.1f us gap %7.1f us" % (name, g, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0)); prev = e
