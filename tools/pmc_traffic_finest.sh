#!/bin/bash
# HBM traffic of the FINEST-level k_track_gn launches (largest grid) of a bench command: two --pmc passes, counters only.
#   bash tools/pmc_traffic_finest.sh [bench.py args] -> gpurun_out/traffic_finest.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=/tmp/pmctf_$$
ARGS="--no-cpu-baseline --pcie-steps 0 --no-roofline --no-secondary $*"
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex k_track_gn --output-format csv -d $OUT/f -- python3 $R/bench.py $ARGS > $OUT.f.json 2> $OUT.f.err || tail -3 $OUT.f.err
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex k_track_gn --output-format csv -d $OUT/w -- python3 $R/bench.py $ARGS > $OUT.w.json 2> $OUT.w.err || tail -3 $OUT.w.err
python3 - "$OUT" > $R/gpurun_out/traffic_finest.json <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
def rows(pat, name):
    r = []
    for fn in glob.glob(pat):
        for x in csv.DictReader(open(fn)):
            if x["Counter_Name"] == name:
                r.append((int(x["Grid_Size"]), float(x["Counter_Value"]), int(x["End_Timestamp"]) - int(x["Start_Timestamp"])))
    return r
f, w = rows(out + "/f/*/*counter_collection.csv", "FETCH_SIZE"), rows(out + "/w/*/*counter_collection.csv", "WRITE_SIZE")
g = max(x[0] for x in f)
# launches of the finest level with every sequence active = the largest grid; keep the slowest half (full launches, not the tail ones)
ff = sorted([x for x in f if x[0] == g], key=lambda x: -x[1]); ff = ff[:max(1, len(ff) // 2)]
ww = sorted([x for x in w if x[0] == g], key=lambda x: -x[1]); ww = ww[:max(1, len(ww) // 2)]
fm, wm = sum(x[1] for x in ff) / len(ff), sum(x[1] for x in ww) / len(ww)
print(json.dumps({"kernel": "k_track_gn", "level": "finest (largest grid %d), fullest half of its launches" % g, "dispatches": len(ff),
                  "FETCH_SIZE_KiB_per_launch_raw": fm, "WRITE_SIZE_KiB_per_launch": wm,
                  "fetch_bytes_per_launch_corrected": 2 * fm * 1024, "write_bytes_per_launch": wm * 1024,
                  "traffic_bytes_per_launch": 2 * fm * 1024 + wm * 1024, "avg_launch_us": sum(x[2] for x in ff) / len(ff) / 1e3,
                  "note": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM)"}))
PY
cat $R/gpurun_out/traffic_finest.json
