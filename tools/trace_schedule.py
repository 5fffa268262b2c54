#!/usr/bin/env python3
"""Per-position timing of the tracker's launch schedule from a rocprofv3 --kernel-trace CSV.

  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 bench.py --no-cpu-baseline --pcie-steps 0 --no-roofline --steps 10
  python3 tools/trace_schedule.py /tmp/tr/*/*kernel_trace.csv 60

Prints, for each position in the per-step schedule (level-major, `launches_per_step` GN launches), the mean duration
of k_track_gn, of the k_gn_solve that follows, and the idle gaps before each of them.
"""
import csv, glob, sys
from collections import defaultdict

def main():
    files = glob.glob(sys.argv[1])
    per_step = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    rows = []
    for fn in files:
        for r in csv.DictReader(open(fn)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # keep whole steps only: a step starts at k_pyramid
    steps, cur = [], None
    for s, e, n in rows:
        if "k_pyramid" in n and ("raw4" in n or "k_pyramid(" in n):
            if cur: steps.append(cur)
            cur = []
        if cur is not None:
            cur.append((s, e, n))
    steps = [st for st in steps if sum("k_track_gn" in n for _, _, n in st) == per_step]
    steps = steps[len(steps) // 2:]  # the later half: warm
    print("steps analysed:", len(steps))
    gn_d, gn_gap, so_d, so_gap = defaultdict(float), defaultdict(float), defaultdict(float), defaultdict(float)
    other = defaultdict(float)
    wall = 0.0
    for st in steps:
        wall += (st[-1][1] - st[0][0]) / 1e3
        pos, prev_end = -1, None
        for s, e, n in st:
            gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
            if "k_track_gn" in n:
                pos += 1
                gn_d[pos] += (e - s) / 1e3; gn_gap[pos] += gap
            elif "k_gn_solve" in n:
                so_d[pos] += (e - s) / 1e3; so_gap[pos] += gap
            else:
                other[n.split("(")[0][:40]] += (e - s) / 1e3 + gap
            prev_end = e
    n = len(steps)
    print("wall per step %.1f us" % (wall / n))
    tot = [0, 0, 0, 0]
    for p in range(per_step):
        v = (gn_d[p] / n, gn_gap[p] / n, so_d[p] / n, so_gap[p] / n)
        for i in range(4): tot[i] += v[i]
        print("pos %2d  gn %7.1f us (gap %4.1f)   solve %5.1f us (gap %4.1f)" % ((p,) + v))
    print("totals: gn %.1f  gn_gap %.1f  solve %.1f  solve_gap %.1f us" % tuple(tot))
    for k, v in sorted(other.items(), key=lambda kv: -kv[1]):
        print("other %-40s %.1f us (incl. gap before)" % (k, v / n))

main()
