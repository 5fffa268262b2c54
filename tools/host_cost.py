#!/usr/bin/env python3
"""Host enqueue cost vs GPU time of one dvo_batch_push_device (is the launch schedule CPU bound?).

    python tools/host_cost.py --batch 512 --streams 2
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "direct-visual-odometry_amd"))
import torch
import dvo_amd as dvo
from dvo_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--streams", type=int, default=0)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda", 0)
W, H, K = 640, 480, synth.K_640
gs, ds = [], []
for b in range(4):
    poses = synth.trajectory(3, seed=42 + b)
    fr = [synth.render(p, K, W, H, device=dev) for p in poses]
    gs.append(torch.stack([f[0] for f in fr])); ds.append(torch.stack([f[1] for f in fr]))
rep = (a.batch + 3) // 4
gray = torch.stack(gs, 1).repeat(1, rep, 1, 1)[:, :a.batch].contiguous()
depth = torch.stack(ds, 1).repeat(1, rep, 1, 1)[:, :a.batch].contiguous()
sigma = torch.full_like(gray, 0.1)
torch.cuda.synchronize()
cfg = dvo.default_config(stream=torch.cuda.current_stream().cuda_stream, track_streams=a.streams)
bt = dvo.Batch(a.batch, K, W, H, 4, 1, cfg=cfg)
def push(k):
    f = k % 3
    bt.push_device(gray[f].data_ptr(), depth[f].data_ptr(), sigma[f].data_ptr())
for k in range(4): push(k)
bt.synchronize()
host = 0.0
t0 = time.perf_counter()
for k in range(a.steps):
    t1 = time.perf_counter(); push(4 + k); host += time.perf_counter() - t1
t_enq = time.perf_counter() - t0
bt.synchronize()
t_all = time.perf_counter() - t0
print("streams %d: host enqueue %.3f ms/step, wall %.3f ms/step (enqueue loop finished after %.3f ms/step)" %
      (a.streams, host / a.steps * 1e3, t_all / a.steps * 1e3, t_enq / a.steps * 1e3))
bt.close()
