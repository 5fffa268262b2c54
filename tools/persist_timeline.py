#!/usr/bin/env python3
"""Where one k_track_persist iteration goes: wall-clock stamps (100 MHz) the solver workgroup and worker 0 leave per step when
DVO_PERSIST_TIMELINE=<worker index>.  python tools/persist_timeline.py [worker index]"""
import ctypes as C
import os
import sys

os.environ["DVO_PERSIST_TIMELINE"] = sys.argv[1] if len(sys.argv) > 1 else "0"   # index of the tile worker that leaves the stamps (0: a corner tile)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "direct-visual-odometry_amd"))
import numpy as np
import torch  # noqa: F401
import dvo_amd as dvo
from dvo_amd import synth

g, d, s, _ = synth.sequence(6, seed=42, sigma_value=0.1)
g, d, s = g.numpy(), d.numpy(), s.numpy()
vo = dvo.VisualOdometry(synth.K_640, 640, 480)
for k in range(5):
    vo.odometrizeUsingDepth(g[k], d[k], s[k])
buf = np.zeros((2, 64, 8), np.int64)
L = dvo.lib()
assert L.dvo_debug_persist_timeline(vo._p, buf.ctypes.data_as(C.c_void_p)) == 0
lg = vo.lastTrackLog()
print("iterations per level", lg["n_iter"])
sol, wk = buf[0], buf[1]
n = int((sol[:, 5] > 0).sum())
us = lambda a, b: (b - a) / 100.0
print("(solve = 6x6 solve + pose update + log / state stores: the two parts in brackets)")
print("step level | solver: wait->slots  fence  sum  solve  publish | worker0: wake->args  gn_tile  drain  announce | step total (publish to publish)")
for k in range(n):
    tot = us(sol[k - 1, 5], sol[k, 5]) if k > 0 else float("nan")
    print("%3d   %d    | %8.2f %6.2f %5.2f %6.2f [%.2f %.2f] %7.2f | %8.2f %8.2f %6.2f %8.2f | %6.2f   (worker woke %.2f us after the publish)" % (
        k, sol[k, 6], us(sol[k, 0], sol[k, 1]), us(sol[k, 1], sol[k, 2]), us(sol[k, 2], sol[k, 3]), us(sol[k, 3], sol[k, 4]), us(sol[k, 3], wk[k, 5]), us(wk[k, 5], wk[k, 6]), us(sol[k, 4], sol[k, 5]),
        us(wk[k, 0], wk[k, 3]), us(wk[k, 3], wk[k, 4]), us(wk[k, 4], wk[k, 1]), us(wk[k, 1], wk[k, 2]), tot,
        us(sol[k - 1, 5], wk[k, 0]) if k > 0 else float("nan")))
vo.close()
