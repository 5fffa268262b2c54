R=$GRAFT_REPO_ROOT
cd $R && python3 -c "
import sys
sys.path.insert(0, 'direct-visual-odometry_amd')
import numpy as np
from dvo_amd import synth
g,d,s,_=synth.sequence(6,seed=42,sigma_value=0.1)
np.stack([g.numpy(),d.numpy(),s.numpy()],axis=1).astype(np.float32).tofile('/tmp/frames.f32')
"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/trh && rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d /tmp/trh -- $R/direct-visual-odometry_amd/lib/single_stream_bench /tmp/frames.f32 6 640 480 525.0 525.0 319.5 239.5 12 > /tmp/trh.log 2>&1
ls /tmp/trh/*/ | head
python3 - <<'PY'
import csv, glob
ev=[]
for fn in glob.glob('/tmp/trh/*/*hip_api_trace.csv'):
    for r in csv.DictReader(open(fn)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'API '+r['Function']))
for fn in glob.glob('/tmp/trh/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(fn)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'KERNEL '+r['Kernel_Name'].split('(')[0][-40:]))
for fn in glob.glob('/tmp/trh/*/*memory_copy_trace.csv'):
    for r in csv.DictReader(open(fn)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY '+r.get('Direction','')+' '+r.get('Bytes', r.get('Size',''))))
ev.sort()
# print a window in the middle: around the 30th frame start (first mode)
idx=[i for i,e in enumerate(ev) if 'k_track_persist' in e[2]]
i0=idx[8]
t0=ev[i0][1]
for s,e,n in ev[i0:i0+60]:
    print("%9.1f us  dur %8.1f  %s" % ((s-t0)/1e3, (e-s)/1e3, n))
PY
