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


TOL_BACKWARD = 2e-6     # backward error of an xi_update against the oracle's normal equations (conditioning independent)


def backward_error(H21, g, x):
    """max|H x - g| / max(|H||x| + |g|) of a 6-vector x against the normal equations (H as the 21 upper-triangle sums, g):
    independent of the conditioning of H, unlike a forward comparison of two solutions."""
    import orc
    H = orc.upper_to_full(np.asarray(H21, np.float64)); g = np.asarray(g, np.float64); x = np.asarray(x, np.float64)
    return float(np.abs(H @ x - g).max() / max((np.abs(H) @ np.abs(x) + np.abs(g)).max(), 1e-300))


def assert_composed(xi_in, upd, xi_after, tag=""):
    """xi_after == log(exp(xi_in) exp(upd)) (tracker.cpp:46) up to the rounding of the result to float: both sides evaluate the
    composition in double and round once, so the allowance is a few units in the last place of the pose, nothing pose-scaled."""
    import orc
    nxt = orc.se3_concatenate(xi_in, upd)
    if not np.all(np.isfinite(nxt)):
        return   # testXi (tracker.cpp:47-51): the pose is left unchanged on both sides; checked by the caller through xi_after
    np.testing.assert_allclose(xi_after, nxt, rtol=0, atol=4 * float(np.spacing(np.float32(max(1.0, np.abs(nxt).max())))), err_msg=str(tag))
