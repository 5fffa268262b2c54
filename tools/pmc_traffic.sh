#!/bin/bash
# HBM traffic of k_track_gn for the default bench command: two separate --pmc passes (FETCH_SIZE costs 3 TCC slots,
# WRITE_SIZE 2), counters only.  Writes gpurun_out/traffic.json (copy it to profiles/ to have bench.py report it).
#   bash tools/pmc_traffic.sh [extra bench.py args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=/tmp/pmct_$$
ARGS="--no-cpu-baseline --pcie-steps 0 --no-roofline --no-secondary $*"
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex k_track_gn --output-format csv -d $OUT/f -- python3 $R/bench.py $ARGS > $OUT.f.json 2> $OUT.f.err || tail -3 $OUT.f.err
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex k_track_gn --output-format csv -d $OUT/w -- python3 $R/bench.py $ARGS > $OUT.w.json 2> $OUT.w.err || tail -3 $OUT.w.err
python3 - "$OUT" "$OUT.f.json" > $R/gpurun_out/traffic.json <<'PY'
import csv, glob, json, sys
out, bench_json = sys.argv[1], sys.argv[2]
def mean(pat, name):
    v = []
    for fn in glob.glob(pat):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == name:
                v.append(float(r["Counter_Value"]))
    return (sum(v) / len(v), len(v)) if v else (0.0, 0)
f, nf = mean(out + "/f/*/*counter_collection.csv", "FETCH_SIZE")
w, nw = mean(out + "/w/*/*counter_collection.csv", "WRITE_SIZE")
cfg = json.load(open(bench_json))["config"]
# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM):
# calibrated here on the probe launch (TCC_EA0_RDREQ x 128 B = 105 MB vs 98 MB compulsory) -> x2.
res = {"kernel": "k_track_gn", "dispatches": nf, "sequences_per_gpu": cfg["sequences_per_gpu"],
       "FETCH_SIZE_KiB_per_launch_raw": f, "WRITE_SIZE_KiB_per_launch": w,
       "fetch_bytes_per_launch_corrected": 2 * f * 1024, "write_bytes_per_launch": w * 1024,
       "traffic_bytes_per_launch": 2 * f * 1024 + w * 1024}
print(json.dumps(res))
PY
cat $R/gpurun_out/traffic.json
