#!/usr/bin/env python3
"""Experiment: do two tracking chains on different HIP streams overlap (one in its latency-bound coarse levels, the other in its
VALU-bound fine level)?  One 4096-sequence batch vs two 2048-sequence batches staggered by half a step, adaptive schedule off."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "direct-visual-odometry_amd"))
import numpy as np, torch
import dvo_amd as dvo
from dvo_amd import synth
dev = torch.device("cuda", 0)
B, F, W, H, K = 4096, 4, 640, 480, synth.K_640
g8 = torch.empty((F, B, H, W), dtype=torch.uint8, device=dev); d16 = torch.empty((F, B, H, W), dtype=torch.int16, device=dev)
for b0 in range(0, B, 24):
    b1 = min(B, b0 + 24)
    Ts = np.stack([synth.trajectory(F, seed=42 + b)[f] for b in range(b0, b1) for f in range(F)])
    g, d = synth.render_batch(Ts, K, W, H, device=dev)
    g = g.reshape(b1 - b0, F, H, W).permute(1, 0, 2, 3); d = d.reshape(b1 - b0, F, H, W).permute(1, 0, 2, 3)
    g8[:, b0:b1] = torch.clamp(torch.round(g * 255), 0, 255).to(torch.uint8); d16[:, b0:b1] = torch.round(d * 5000).to(torch.int32).to(torch.int16)
torch.cuda.synchronize()
def ring(k): 
    p = 2 * (F - 1); r = k % p
    return r if r < F else p - r
def run(nb, adaptive, steps=20, stagger=True):
    per = B // nb
    streams = [torch.cuda.Stream() for _ in range(nb)]
    bts = [dvo.Batch(per, K, W, H, 4, 1, cfg=dvo.default_config(stream=streams[i].cuda_stream, track_adaptive=adaptive)) for i in range(nb)]
    def push(i, k):
        f = ring(k)
        bts[i].push_raw_device(g8[f, i * per:(i + 1) * per].contiguous().data_ptr() if nb > 1 and False else g8[f].data_ptr() + i * per * H * W,
                               1, d16[f].data_ptr() + 2 * i * per * H * W)
    for i in range(nb):
        push(i, 0); push(i, 1)
    if stagger and nb > 1:
        push(0, 2)      # batch 0 is one step ahead... the streams then drift to a steady relative phase
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        for i in range(nb):
            push(i, 3 + k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for b in bts: b.close()
    return B * steps / dt
for nb, ad in ((1, 0), (1, -1), (2, -1), (2, -1), (4, -1)):
    print("batches %d adaptive %d: %.0f frames/s" % (nb, ad, run(nb, ad)))
