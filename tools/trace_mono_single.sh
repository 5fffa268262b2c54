# Kernel + copy timeline of the mono dvo_vo loop (vo.odometrize per frame): bash tools/trace_mono_single.sh   (on the GPU box)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/trm && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/trm -- python3 $GRAFT_REPO_ROOT/tools/trace_mono_single.py > /tmp/trm.log 2>&1
python3 - <<'PY'
import csv, glob
ev=[]
for fn in glob.glob('/tmp/trm/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(fn)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'KERNEL '+r['Kernel_Name'].split('(')[0][-44:]))
for fn in glob.glob('/tmp/trm/*/*memory_copy_trace.csv'):
    for r in csv.DictReader(open(fn)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY '+r.get('Direction','')))
ev.sort()
idx=[i for i,e in enumerate(ev) if 'k_track_persist' in e[2]]
i0=idx[-7]; t0=ev[i0][0]
for s,e,n in ev[i0-3:idx[-2]+6]:
    print("%9.1f us  dur %7.1f  %s" % ((s-t0)/1e3, (e-s)/1e3, n))
PY
