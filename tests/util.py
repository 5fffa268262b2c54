"""Shared helpers of the test-suite: seeded synthetic frames (rendered once per process) and tolerances."""
import functools

import numpy as np

from dvo_amd import synth

K640 = synth.K_640

# Stated float tolerances of the GPU-vs-oracle parity tests.  Index/mask work is bit-exact.
TOL_H_REL = 3e-5        # |H_gpu - H_oracle| <= TOL * max|H|   (fp32 per-thread partial sums vs double raster sum)
TOL_UPD_REL = 2e-4      # |xi_update diff| <= TOL * |xi_update| + TOL_UPD_ABS
TOL_UPD_ABS = 2e-7
TOL_POSE = 2e-5         # metres / radians on a full Tracker::track call with contracting gain


@functools.lru_cache(maxsize=None)
def frames(n=4, seed=42, w=640, h=480, sigma=0.1):
    g, d, s, poses = synth.sequence(n, width=w, height_px=h, seed=seed, sigma_value=sigma)
    return g.numpy(), d.numpy(), s.numpy(), poses


def level_maps(of, level):
    """(gray, depth, sigma, K) of an oracle frame at a level."""
    return of.gray(level), of.depth(level), of.sigma(level), of.K(level)
