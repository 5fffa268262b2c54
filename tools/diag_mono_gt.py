#!/usr/bin/env python3
"""Diagnostic: mono batch (1 sequence) vs the oracle on a synthetic sequence with a ground-truth-initialised depth map."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "direct-visual-odometry_amd"), os.path.join(ROOT, "oracle")]
import numpy as np, torch
import dvo_amd as dvo, orc
from dvo_amd import synth
K = synth.K_640
F = 6
poses = synth.trajectory(F, seed=42)
g, d = synth.render_batch(np.stack(poses), K, 640, 480, device="cuda", newton_iters=6)
g8 = torch.clamp(torch.round(g * 255.0), 0, 255).to(torch.uint8)
gf = (g8.float() * (1.0 / 255.0)).cpu().numpy()
rng = np.random.RandomState(1)
d0 = (d[0, ::4, ::4].cpu().numpy() + rng.normal(0, 0.1, (120, 160))).astype(np.float32)
def ring(k): 
    p = 2 * (F - 1); r = k % p
    return r if r < F else p - r
mb = dvo.MonoBatch(1, K, 640, 480, cfg=dvo.default_config(rng_seed=1))
mb.setInitialDepth(d0, np.full_like(d0, 0.5))
ovo = orc.OVO(K, 640, 480, seed=1); ovo.set_initial_depth(d0, np.full_like(d0, 0.5))
for k in range(24):
    f = ring(k)
    mb.odometrize_raw_device(g8[f:f + 1].contiguous().data_ptr(), 1)
    xi, T, key = mb.world_poses()
    To, keyo = ovo.odometrize(gf[f])
    kf = mb.keyframe(0); okf = ovo.keyframe(ovo.keyframe_count() - 1)
    lg = mb.last_track_log(0) if k > 0 else {"n_iter": []}
    print("frame %2d key %d/%d |T-To| %.2e iters %s valid %d/%d age mismatch %.4f depth>1e-3 %.4f" % (
        k, key[0], keyo, np.abs(T[0] - To).max(), lg["n_iter"], kf["valid_updates"], ovo.last_valid_updates(),
        (kf["age"] != okf.age()).mean(), (np.abs(kf["depth"] - okf.depth(2)) > 1e-3).mean()))
mb.close()
