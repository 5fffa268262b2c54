#!/usr/bin/env python3
"""Launch k_track_gn back to back on one pyramid level with every sequence active (roofline probe / PMC target).

    python tools/probe_gn.py --batch 64 --level 3 --launches 20 [--ppt 8] [--sigma 0.1]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "direct-visual-odometry_amd"))
import torch

import dvo_amd as dvo
from dvo_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--level", type=int, default=3)
ap.add_argument("--launches", type=int, default=20)
ap.add_argument("--ppt", type=int, default=0)
ap.add_argument("--sigma", type=float, default=0.1)
ap.add_argument("--group", type=int, default=0)
ap.add_argument("--lds", type=int, default=-1, help="-1 auto, 0 global gathers, N>0 LDS patch margin")
ap.add_argument("--distinct", type=int, default=4, help="distinct rendered sequences (tiled to --batch)")
ap.add_argument("--raw", action="store_true", help="feed raw u8 gray + u16 depth frames (sigma 0.1: constant weight, no wgt map) instead of float maps")
a = ap.parse_args()
dev = torch.device("cuda", 0)
W, H, K = 640, 480, synth.K_640
gs, ds = [], []
for b in range(a.distinct):
    poses = synth.trajectory(2, seed=42 + b)
    fr = [synth.render(p, K, W, H, device=dev) for p in poses]
    gs.append(torch.stack([f[0] for f in fr]))
    ds.append(torch.stack([f[1] for f in fr]))
rep = (a.batch + a.distinct - 1) // a.distinct
gray = torch.stack(gs, 1).repeat(1, rep, 1, 1)[:, :a.batch].contiguous()   # [2][B][H][W]
depth = torch.stack(ds, 1).repeat(1, rep, 1, 1)[:, :a.batch].contiguous()
sigma = torch.full_like(gray, a.sigma)
torch.cuda.synchronize()
cfg = dvo.default_config(stream=torch.cuda.current_stream().cuda_stream, gn_pixels_per_thread=a.ppt, gn_gather_group=a.group, gn_use_lds_patch=a.lds)
bt = dvo.Batch(a.batch, K, W, H, 4, 1, cfg=cfg)
if a.raw:
    g8 = (gray.clamp(0, 1) * 255).round().to(torch.uint8).contiguous()
    d16 = (depth.clamp(0, 13) * 5000).round().to(torch.int32).to(torch.int16).contiguous()   # (bit pattern of the u16 value)
    torch.cuda.synchronize()
for f in range(2):
    if a.raw:
        bt.push_raw_device(g8[f].data_ptr(), 1, d16[f].data_ptr())
    else:
        bt.push_device(gray[f].data_ptr(), depth[f].data_ptr(), sigma[f].data_ptr())
bt.synchronize()
for lvl in ([a.level] if a.level >= 0 else range(4)):
    ms, px = bt.probe_gn(lvl, a.launches)
    print("level %d: %.2f us/launch, %d px/launch, %.1f GB/s algorithmic (16 B/px), %.2f Gpx/s" %
          (lvl, ms * 1e3, px, 16 * px / (ms * 1e-3) / 1e9, px / (ms * 1e-3) / 1e9))
bt.close()
