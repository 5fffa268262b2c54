#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per (kernel, grid size)."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for pat in sys.argv[1:]:
    for fn in glob.glob(pat):
        with open(fn) as f:
            seen = set()
            for r in csv.DictReader(f):
                key = (r["Kernel_Name"].replace("void ", "").split("(")[0], int(r["Grid_Size"]), r.get("VGPR_Count", ""), r.get("Scratch_Size", ""))
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                did = (fn, r["Dispatch_Id"])
                if did not in seen:
                    seen.add(did)
                    dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
LAST = int(__import__("os").environ.get("PMC_LAST", "0"))  # only the last N dispatches of each kernel/grid
for key in sorted(acc):
    if LAST:
        dur[key] = dur[key][-LAST:]
        for c in acc[key]:
            acc[key][c] = acc[key][c][-LAST:]
    n = len(dur[key])
    print("%s grid=%d vgpr=%s scratch=%s dispatches=%d avg_ns=%.0f" % (key[0], key[1], key[2], key[3], n, sum(dur[key]) / max(n, 1)))
    for c in sorted(acc[key]):
        v = acc[key][c]
        print("    %-28s %14.1f" % (c, sum(v) / len(v)))
