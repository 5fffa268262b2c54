#!/usr/bin/env python3
"""Aggregate rate of T independent single-sequence handles driven from T host threads (each its own HIP stream):
the reference's per-sequence keyframe logic intact, concurrency across sequences.   python tools/bench_multi.py 8 [mono|depth]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "direct-visual-odometry_amd"))
import numpy as np
import torch  # noqa: F401
import dvo_amd as dvo
from dvo_amd import synth

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mode = sys.argv[2] if len(sys.argv) > 2 else "mono"
n = 60
K = synth.K_640
data = []
for t in range(T):
    g, d, s, _ = synth.sequence(12, seed=100 + t, sigma_value=0.1)
    data.append((g.numpy(), d.numpy(), s.numpy()))
idx = [i if i < 12 else 22 - i for i in range(23)]

def work(t, out):
    g, d, s = data[t]
    vo = dvo.VisualOdometry(K, 640, 480, cfg=dvo.default_config(rng_seed=1 + t))
    if mode == "mono":
        d0 = d[0][::4, ::4].copy()
        vo.setInitialDepth(d0, np.full_like(d0, 0.5))
    step = (lambda j: vo.odometrize(g[j])) if mode == "mono" else (lambda j: vo.odometrizeUsingDepth(g[j], d[j], s[j]))
    for k in range(3):
        step(idx[k])
    barrier.wait()
    t0 = time.perf_counter()
    for k in range(n):
        step(idx[(3 + k) % 22])
    out[t] = time.perf_counter() - t0
    vo.close()

barrier = threading.Barrier(T)
out = [0.0] * T
th = [threading.Thread(target=work, args=(t, out)) for t in range(T)]
t0 = time.perf_counter()
for x in th: x.start()
for x in th: x.join()
print("%s: %d handles x %d frames: %.1f frames/s aggregate (slowest thread %.3f s)" % (mode, T, n, T * n / max(out), max(out)))
