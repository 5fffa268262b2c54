cd $GRAFT_REPO_ROOT
python3 -c "
import sys,os
sys.path.insert(0, 'direct-visual-odometry_amd')
import numpy as np
from dvo_amd import synth
g,d,s,_=synth.sequence(6,seed=42,sigma_value=0.1)
np.stack([g.numpy(),d.numpy(),s.numpy()],axis=1).astype(np.float32).tofile('/tmp/frames.f32')
"
which rocgdb gdb
timeout -k 5 120 /opt/rocm/bin/rocgdb -batch -ex run -ex bt --args direct-visual-odometry_amd/lib/facade_demo /tmp/frames.f32 6 640 480 525.0 525.0 319.5 239.5 depth /tmp/p.f32 0 2>&1 | tail -40
