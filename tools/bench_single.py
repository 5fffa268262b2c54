#!/usr/bin/env python3
"""Single-stream (latency) rates of the drop-in entry points: one sequence, host frames in, pose out per call."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "direct-visual-odometry_amd"))
import numpy as np
import torch  # noqa: F401

import dvo_amd as dvo
from dvo_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
g, d, s, _ = synth.sequence(16, seed=42, sigma_value=0.1)
g, d, s = g.numpy(), d.numpy(), s.numpy()
idx = [i if i < 16 else 30 - i for i in range(31)]
K = synth.K_640
FUSED = int(os.environ.get("DVO_FUSED", "0"))  # track_fused_tiles
for sigma, name in ((0.1, "sensor sigma 0.1 (over-relaxed, many iterations)"), (0.5, "sigma 0.5 (1-2 iterations per level)")):
    vo = dvo.VisualOdometry(K, 640, 480, cfg=dvo.default_config(track_fused_tiles=FUSED))
    ss = np.full_like(s[0], sigma)
    for k in range(3):
        vo.odometrizeUsingDepth(g[idx[k % 30]], d[idx[k % 30]], ss)
    t0 = time.perf_counter(); its = 0
    for k in range(n):
        j = idx[(3 + k) % 30]
        vo.odometrizeUsingDepth(g[j], d[j], ss)
        its += sum(vo.lastTrackLog()["n_iter"])
    dt = time.perf_counter() - t0
    print("odometrizeUsingDepth  %-50s %8.1f frames/s  (%.2f ms/frame, %.1f GN iterations/frame)" % (name, n / dt, dt / n * 1e3, its / n))
    vo.close()
# the same sensor-depth loop fed with raw u8 gray + u16 depth (0.9 instead of 3.7 MB per frame over PCIe)
g8 = [np.clip(np.rint(x * 255), 0, 255).astype(np.uint8) for x in g]
d16 = [np.clip(np.rint(x * 5000), 0, 65535).astype(np.uint16) for x in d]
vo = dvo.VisualOdometry(K, 640, 480)
for k in range(3):
    vo.odometrizeUsingDepthRaw(g8[idx[k % 30]], d16[idx[k % 30]])
t0 = time.perf_counter(); its = 0
for k in range(n):
    j = idx[(3 + k) % 30]
    vo.odometrizeUsingDepthRaw(g8[j], d16[j])
    its += sum(vo.lastTrackLog()["n_iter"])
dt = time.perf_counter() - t0
print("odometrizeUsingDepthRaw (u8 gray + u16 depth, sigma 0.1)                 %8.1f frames/s  (%.2f ms/frame, %.1f GN iterations/frame)" % (n / dt, dt / n * 1e3, its / n))
vo.close()
# mono tracking + mapping (odometrize)
import ctypes
vo = dvo.VisualOdometry(K, 640, 480, cfg=dvo.default_config(rng_seed=1, track_fused_tiles=FUSED))
d0 = d[0][::4, ::4].copy()
vo.setInitialDepth(d0, np.full_like(d0, 0.5))
for k in range(3):
    vo.odometrize(g[idx[k]])
t0 = time.perf_counter(); keys = 0
for k in range(n):
    T, key = vo.odometrize(g[idx[(3 + k) % 30]])
    keys += key
dt = time.perf_counter() - t0
print("odometrize (mono, track + map)  %8.1f frames/s  (%.2f ms/frame, %d keyframes in %d frames)" % (n / dt, dt / n * 1e3, keys, n))
vo.close()
