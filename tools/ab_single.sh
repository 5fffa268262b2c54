cd $GRAFT_REPO_ROOT
python3 -c "
import sys,os
sys.path.insert(0, 'direct-visual-odometry_amd')
import numpy as np
from dvo_amd import synth
g,d,s,_=synth.sequence(6,seed=42,sigma_value=0.1)
np.stack([g.numpy(),d.numpy(),s.numpy()],axis=1).astype(np.float32).tofile('/tmp/frames.f32')
"
for ppt in 1 2 4; do for sl in 0 1 -1; do
  echo "== ppt $ppt single_launch $sl"
  DVO_PPT=$ppt DVO_SINGLE_LAUNCH=$sl timeout -k 5 60 direct-visual-odometry_amd/lib/single_stream_bench /tmp/frames.f32 6 640 480 525.0 525.0 319.5 239.5 300 | grep pinned
done; done
