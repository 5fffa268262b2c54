import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "direct-visual-odometry_amd"))
import numpy as np, torch
import dvo_amd as dvo
from dvo_amd import synth
g, d, s, _ = synth.sequence(16, seed=42, sigma_value=0.1)
g, d = g.numpy(), d.numpy()
idx = [i if i < 16 else 30 - i for i in range(31)]
vo = dvo.VisualOdometry(synth.K_640, 640, 480, cfg=dvo.default_config(rng_seed=1))
d0 = d[0][::4, ::4].copy()
vo.setInitialDepth(d0, np.full_like(d0, 0.5))
for k in range(40):
    vo.odometrize(g[idx[k % 30]])
vo.close()
