"""CPU tests of the oracle's primitives (no GPU).  The reference pins none of these results
(every test/*.cpp is an interactive viewer), so the oracle is cross-checked against
independent restatements written here in numpy / pure Python and against analytic properties.
"""
import itertools
import math

import numpy as np
import pytest

import orc

INV = np.float32(-2.0)


# ---------------------------------------------------------------- SE(3) (src/math/se3.cpp)
def _exp_np(xi):
    from dvo_amd.synth import se3_exp_np
    return se3_exp_np(xi)


@pytest.mark.parametrize("seed", range(5))
def test_se3_exp_matches_closed_form(seed):
    rng = np.random.RandomState(seed)
    xi = np.concatenate([rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3)]).astype(np.float32)
    T = orc.se3_exp(xi)
    np.testing.assert_allclose(T, _exp_np(xi.astype(np.float64)), atol=2e-7)
    R = T[:3, :3].astype(np.float64)
    np.testing.assert_allclose(R @ R.T, np.eye(3), atol=5e-7)


@pytest.mark.parametrize("scale", [1e-7, 1e-4, 1e-2, 0.3, 2.0])
def test_se3_exp_log_round_trip(scale):
    # test/se3.cpp:17-45 does this round trip by eye
    rng = np.random.RandomState(3)
    for _ in range(20):
        xi = (rng.uniform(-1, 1, 6) * scale).astype(np.float32)
        back = orc.se3_log(orc.se3_exp(xi))
        np.testing.assert_allclose(back, xi, rtol=2e-5, atol=3e-7 * max(1.0, scale))


def test_se3_concatenate_is_matrix_product():
    rng = np.random.RandomState(7)
    for _ in range(20):
        a = (rng.uniform(-0.2, 0.2, 6)).astype(np.float32)
        b = (rng.uniform(-0.05, 0.05, 6)).astype(np.float32)
        c = orc.se3_concatenate(a, b)
        np.testing.assert_allclose(_exp_np(c.astype(np.float64)), _exp_np(a.astype(np.float64)) @ _exp_np(b.astype(np.float64)), atol=3e-7)
    z = np.zeros(6, np.float32)
    np.testing.assert_array_equal(orc.se3_concatenate(z, z), z)


def test_se3_float_literal_restatement_agrees_within_float_noise():
    # D2: the float-literal se3.cpp and the double evaluation differ by float cancellation noise only
    rng = np.random.RandomState(11)
    worst = 0.0
    for _ in range(50):
        a = np.concatenate([rng.normal(0, 0.02, 3), rng.normal(0, 0.01, 3)]).astype(np.float32)
        b = np.concatenate([rng.normal(0, 0.005, 3), rng.normal(0, 0.005, 3)]).astype(np.float32)
        worst = max(worst, np.abs(orc.se3_concatenate(a, b) - orc.se3_concatenate(a, b, lit=True)).max())
        np.testing.assert_allclose(orc.se3_exp(a), orc.se3_exp(a, lit=True), atol=1e-6)
    assert worst < 2e-6


def test_pose_from_xi_sign():
    xi = np.array([0.01, -0.02, 0.03, 0.02, 0.01, -0.03], np.float32)
    Rt = orc.pose_from_xi(xi, -1.0)
    T = orc.se3_exp(-xi)
    np.testing.assert_array_equal(Rt[:9].reshape(3, 3), T[:3, :3])
    np.testing.assert_array_equal(Rt[9:], T[:3, 3])


# ---------------------------------------------------------------- image primitives (src/core/convert.cpp)
def test_cull_image_sizes_and_invalid():
    img = np.arange(7 * 10, dtype=np.float32).reshape(7, 10)
    img[2, 4] = INV
    img[4, 0] = -3.0
    out = orc.cull_image(img, 1)
    assert out.shape == (3, 5)  # cv::Size / 2 truncates (convert.cpp:12)
    exp = img[0:6:2, 0:10:2].copy()
    exp[exp <= -2] = INV
    np.testing.assert_array_equal(out, exp)
    assert orc.cull_image(img, 2).shape == (1, 2)
    np.testing.assert_array_equal(orc.cull_image(img, 0), img)


def test_cull_intrinsic():
    K = np.array([[525, 0.5, 319.5], [0, 524, 239.5], [0, 0, 1]], np.float32)
    K2 = orc.cull_intrinsic(K, 2)
    np.testing.assert_array_equal(K2, np.array([[131.25, 0.125, 79.875], [0, 131, 59.875], [0, 0, 1]], np.float32))


def test_gradiate_borders_and_invalid_taps():
    rng = np.random.RandomState(0)
    img = rng.uniform(0, 1, (6, 8)).astype(np.float32)
    img[3, 3] = INV
    gx, gy = orc.gradiate(img, True), orc.gradiate(img, False)
    assert (gx[:, 0] == INV).all() and (gx[:, -1] == INV).all()
    assert (gy[0, :] == INV).all() and (gy[-1, :] == INV).all()
    assert gx[3, 2] == INV and gx[3, 4] == INV and gy[2, 3] == INV and gy[4, 3] == INV
    assert gx[3, 3] == img[3, 4] - img[3, 2]  # the centre being invalid does not matter
    assert gx[1, 1] == img[1, 2] - img[1, 0]  # no 1/2 (convert.cpp:58)
    assert gy[2, 5] == img[3, 5] - img[1, 5]


def _subpixel_literal(g, hx, vy):
    """Pure-Python literal transcription of the fill loop semantics (convert.cpp:155-173)."""
    g = [np.float32(v) for v in g]
    valid, idx, last = 0, 0, np.float32(-1)
    while True:
        if g[idx] > INV:
            valid += 1
            last = g[idx]
        elif last > 0:
            g[idx] = last
            valid += 1
        if valid == 4:
            break
        if idx == 3 and valid == 0:
            return INV
        idx = (idx + 1) % 4
    f = np.float32
    omh, omv = f(1) - f(hx), f(1) - f(vy)
    top = f(np.float64(g[1]) * np.float64(hx) + np.float64(g[0] * omh))
    bot = f(np.float64(g[3]) * np.float64(hx) + np.float64(g[2] * omh))
    return f(np.float64(bot) * np.float64(vy) + np.float64(top * omv))


def test_get_subpixel_fill_quirk_all_patterns():
    # every validity pattern x {positive, zero, negative-but-valid} values
    vals = [np.float32(0.75), np.float32(0.0), np.float32(-0.5), INV, np.float32(-2.5)]
    hx, vy = np.float32(0.3), np.float32(0.6)
    for combo in itertools.product(vals, repeat=4):
        img = np.array([[combo[0], combo[1]], [combo[2], combo[3]]], np.float32)
        got = orc.get_subpixel(img, float(hx), float(vy))
        exp = _subpixel_literal(combo, hx, vy)
        assert got == exp, (combo, got, exp)


def test_get_subpixel_edges_and_truncation():
    img = np.array([[0.1, 0.2, 0.3], [0.4, 0.5, 0.6]], np.float32)
    # x0 out of range -> INVALID
    assert orc.get_subpixel(img, 3.0, 0.0) == INV and orc.get_subpixel(img, -1.0, 0.0) == INV
    # truncation toward zero: pt in (-1,0) uses index 0 with a negative weight (convert.cpp:82-83)
    got = orc.get_subpixel(img, -0.5, 0.0, dense=True)
    h = np.float32(-0.5)
    exp = np.float32(np.float32(0.2) * h + np.float32(0.1) * (np.float32(1) - h))
    assert abs(got - exp) < 1e-7
    # last column: x1 out of range -> clamped to g00
    assert orc.get_subpixel(img, 2.5, 0.0, dense=True) == np.float32(0.3)
    # last row
    assert abs(orc.get_subpixel(img, 1.0, 1.5, dense=True) - np.float32(0.5)) < 1e-7
    # non finite / huge -> INVALID (D4)
    for bad in (float("nan"), float("inf"), -float("inf"), 3e9):
        assert orc.get_subpixel(img, bad, 0.0) == INV
        assert orc.get_subpixel(img, 0.0, bad, dense=True) == INV


def test_dense_blends_invalid_as_numbers():
    img = np.array([[0.5, -2.0], [0.5, 0.5]], np.float32)
    got = orc.get_subpixel(img, 0.5, 0.0, dense=True)
    assert got == np.float32(-0.75)  # (0.5 + -2)/2, convert.cpp:103-104
    # the quirky version fills the invalid tap with the last valid value instead
    assert orc.get_subpixel(img, 0.5, 0.0) == np.float32(0.5)


# ---------------------------------------------------------------- geometry (src/core/transform.cpp)
def test_warp_identity_and_translation():
    K = np.array([[500, 0, 320], [0, 510, 240], [0, 0, 1]], np.float32)
    Rt = orc.pose_from_xi(np.zeros(6, np.float32))
    p = orc.warp(Rt, 100, 50, 1.7, K)
    np.testing.assert_allclose(p, [100, 50], atol=1e-4)
    # pure x translation by t moves the projection by fx * t / z
    Rt = orc.pose_from_xi(np.array([0.1, 0, 0, 0, 0, 0], np.float32))
    p = orc.warp(Rt, 100, 50, 2.0, K)
    np.testing.assert_allclose(p, [100 + 500 * 0.1 / 2.0, 50], atol=1e-4)


def test_warp_image_identity():
    rng = np.random.RandomState(1)
    g = rng.uniform(0.1, 0.9, (12, 16)).astype(np.float32)
    d = rng.uniform(1, 2, (12, 16)).astype(np.float32)
    d[3, 4] = 0.0  # no depth -> INVALID (transform.cpp:43-44)
    K = np.array([[20, 0, 8], [0, 20, 6], [0, 0, 1]], np.float32)
    out = orc.warp_image(np.zeros(6, np.float32), g, d, K)
    assert out[3, 4] == INV
    m = np.ones_like(g, bool); m[3, 4] = False
    np.testing.assert_allclose(out[m], g[m], atol=2e-5)


# ---------------------------------------------------------------- Gaussian (src/math/gaussian.cpp)
def test_gaussian_fusion_values():
    d, s, ok = orc.gaussian_fuse(1.0, 0.5, 1.1, 0.5)
    assert ok and abs(d - 1.05) < 1e-6 and abs(s - math.sqrt(0.125)) < 1e-6
    # gate: diff > gain * max(sigma, s) -> rejected, state unchanged (gaussian.cpp:43-44)
    d, s, ok = orc.gaussian_fuse(1.0, 0.1, 2.0, 0.1)
    assert not ok and d == np.float32(1.0) and s == np.float32(0.1)
    # update(): reject resets to the supplied random depth and sigma 0.5 (gaussian.cpp:21-25)
    d, s, ok = orc.gaussian_update(1.0, 0.1, 2.0, 0.1, 1.234)
    assert not ok and d == np.float32(1.234) and s == np.float32(0.5)
    # gain ramp: min(d, diff) < 0.8 -> 0.5 + m/0.8*0.5
    d, s, ok = orc.gaussian_fuse(1.0, 0.3, 1.2, 0.3)  # diff .2 gain .625 -> .1875 < .2 -> reject
    assert not ok
    d, s, ok = orc.gaussian_fuse(1.0, 0.3, 1.18, 0.3)  # diff .18 gain .6125 -> .18375 > .18 -> fuse
    assert ok


def test_rng_depth_distribution():
    v = np.array([orc.rng_depth(5, 3, i) for i in range(4000)])
    assert v.min() > 0.5 - 1e-6 and v.max() <= 2.0
    assert abs(v.mean() - 1.25) < 0.03
    assert orc.rng_depth(5, 3, 17) == orc.rng_depth(5, 3, 17)
    assert orc.rng_depth(5, 3, 17) != orc.rng_depth(5, 4, 17)


# ---------------------------------------------------------------- solvers
def test_solve6_matches_numpy_and_pinv():
    rng = np.random.RandomState(2)
    A = rng.normal(size=(200, 6)) * np.array([30, 30, 10, 40, 40, 20])
    b = rng.normal(size=200)
    H = A.T @ A; g = A.T @ b
    H21 = H[np.triu_indices(6)]
    np.testing.assert_allclose(orc.solve6(H21, g), np.linalg.solve(H, g), rtol=1e-5)
    # rank deficient: last column duplicates the first -> min-norm solution
    A[:, 5] = A[:, 0]
    H = A.T @ A; g = A.T @ b
    x = orc.solve6(H[np.triu_indices(6)], g)
    np.testing.assert_allclose(x, np.linalg.pinv(H, rcond=1e-10) @ g, rtol=1e-4, atol=1e-7)
    np.testing.assert_array_equal(orc.solve6(np.zeros(21), np.zeros(6)), np.zeros(6, np.float32))


def test_lsq_svd_matches_normal_equations():
    rng = np.random.RandomState(4)
    A = (rng.normal(size=(3000, 6)) * np.array([30, 30, 10, 40, 40, 20])).astype(np.float32)
    A[rng.uniform(size=3000) < 0.4] = 0  # zero rows, as in optimize.cpp:17
    B = rng.normal(size=3000).astype(np.float32)
    x = orc.lsq_svd(A, B)
    ref = np.linalg.lstsq(A.astype(np.float64), B.astype(np.float64), rcond=None)[0]
    np.testing.assert_allclose(x, ref, rtol=2e-4, atol=1e-7)
