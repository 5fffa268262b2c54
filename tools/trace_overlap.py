#!/usr/bin/env python3
"""How much do kernels overlap in a rocprofv3 --kernel-trace CSV?  usage: trace_overlap.py 'glob'"""
import csv, glob, sys
rows = []
for fn in glob.glob(sys.argv[1]):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:30], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows = [r for r in rows if "dvo::" in r[2]]
rows.sort()
rows = rows[len(rows) // 2:]
ev = []
for s, e, n, q, st in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, hist = 0, ev[0][0], {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + (t - last)
    depth += d; last = t
tot = sum(hist.values())
print("time by number of kernels in flight:", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
print("queues:", sorted(set(r[3] for r in rows)), "streams:", sorted(set(r[4] for r in rows)))
for r in rows[200:330]:
    print(r[0] - rows[200][0], r[1] - r[0], r[2], "q", r[3], "s", r[4])
