#!/usr/bin/env python3
"""Writes a few synthetic frames and runs lib/single_stream_bench on them (the reference's sensor-depth loop from C++, no Python in
the timed region).  python tools/bench_single_cpp.py [frames_to_time]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "direct-visual-odometry_amd"))
import numpy as np
from dvo_amd import synth

n = 6
g, d, s, _ = synth.sequence(n, seed=42, sigma_value=0.1)
fr = np.stack([g.numpy(), d.numpy(), s.numpy()], axis=1).astype(np.float32)
K = np.asarray(synth.K_640, np.float32).reshape(3, 3)
with tempfile.TemporaryDirectory() as td:
    fn = os.path.join(td, "frames.f32")
    fr.tofile(fn)
    exe = os.path.join(ROOT, "direct-visual-odometry_amd", "lib", "single_stream_bench")
    sys.exit(subprocess.call([exe, fn, str(n), "640", "480", repr(float(K[0, 0])), repr(float(K[1, 1])), repr(float(K[0, 2])), repr(float(K[1, 2])),
                              sys.argv[1] if len(sys.argv) > 1 else "400"]))
