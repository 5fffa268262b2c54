#!/bin/bash
# kernel trace of the C++ single-stream loop (lib/single_stream_bench): durations and gaps of one window of launches
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R && python3 -c "
import sys
sys.path.insert(0, 'direct-visual-odometry_amd')
import numpy as np
from dvo_amd import synth
g,d,s,_=synth.sequence(6,seed=42,sigma_value=0.1)
np.stack([g.numpy(),d.numpy(),s.numpy()],axis=1).astype(np.float32).tofile('/tmp/frames.f32')
"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/trc && rocprofv3 --kernel-trace --output-format csv -d /tmp/trc -- $R/direct-visual-odometry_amd/lib/single_stream_bench /tmp/frames.f32 6 640 480 525.0 525.0 319.5 239.5 30 > /tmp/trc.log 2>&1
python3 $R/tools/trace_single.py "/tmp/trc/*/*kernel_trace.csv" ${1:-40} ${2:-40}
