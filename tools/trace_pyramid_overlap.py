#!/usr/bin/env python3
"""For every k_pyramid in a rocprofv3 kernel trace: its duration and how much of it other kernels ran beside it."""
import csv, glob, sys
rows = []
for fn in glob.glob(sys.argv[1]):
    for r in csv.DictReader(open(fn)):
        if "dvo::" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r.get("Queue_Id", "?")))
rows.sort()
pyr = [r for r in rows if "k_pyramid" in r[2]]
for s, e, n, q in pyr[-6:]:
    ov = 0; names = {}
    for s2, e2, n2, q2 in rows:
        if n2 is n and s2 == s: continue
        a, b = max(s, s2), min(e, e2)
        if b > a and "k_pyramid" not in n2:
            ov += b - a; names[n2[-24:]] = names.get(n2[-24:], 0) + (b - a)
    print("pyramid %8.1f us on queue %s, other kernels beside it for %8.1f us: %s" % ((e - s) / 1e3, q, ov / 1e3, {k: round(v / 1e3) for k, v in names.items()}))
