"""CPU tests of the oracle's tracking and mapping functions against analytic properties (no GPU).

Nothing in the reference pins these numbers, so the checks are the ones SURVEY.md §8c lists:
identity pose -> zero residual, finite differences of the residual -> Jacobian, stacked least squares (the
reference's cv::solve path) vs 6x6 normal equations, variant agreement, propagate / regularize invariants and
a synthetic stereo case with a known answer for the depth update.
"""
import numpy as np
import pytest

import orc
from util import K640, frames

INV = np.float32(-2.0)


def small_pair(level=1, sigma=0.5):
    g, d, s, poses = frames()
    sg = np.full_like(d[0], sigma)
    ref = orc.OFrame(g[0], d[0], sg, K640, 4, 1)
    obj = orc.OFrame(g[1], d[1], sg, K640, 4, 1)
    return ref, obj, poses


def test_identity_pose_gives_zero_residual_and_zero_update():
    ref, _, _ = small_pair()
    o = orc.optimize(ref.gray(1), ref.gray(1), ref.depth(1), ref.sigma(1), ref.K(1), np.zeros(6, np.float32), 1)
    assert o["n_valid"] > 3000
    assert o["sum_r2"] < 1e-8 * o["n_valid"]
    assert np.abs(o["xi_update"]).max() < 1e-4


def test_no_valid_pixels_sentinel():
    z = np.zeros((30, 40), np.float32)
    K = np.array([[30, 0, 20], [0, 30, 15], [0, 0, 1]], np.float32)
    o = orc.optimize(z + 0.5, z + 0.5, z, z + 0.5, K, np.zeros(6, np.float32), 0)
    assert o["n_valid"] == 0 and o["residual"] == np.float32(-1) and not o["xi_update"].any()  # optimize.cpp:92-93


def test_crop_and_depth_gates():
    ref, obj, _ = small_pair()
    rg, rd, rs, K = ref.gray(2), ref.depth(2), ref.sigma(2), ref.K(2)
    o = orc.optimize(obj.gray(2), rg, rd, rs, K, np.zeros(6, np.float32), 2, crop=True, want_mask=True)
    m = o["mask"].astype(bool)
    assert not m[:20].any() and not m[101:].any() and not m[:, :20].any() and not m[:, 141:].any()  # optimize.cpp:33-36
    assert m[20, 20] and m[100, 140]
    o2 = orc.optimize(obj.gray(2), rg, rd, rs, K, np.zeros(6, np.float32), 2, crop=False, want_mask=True)
    assert o2["n_valid"] > o["n_valid"]
    rd2 = rd.copy(); rd2[30:40, 50:60] = 0.19
    o3 = orc.optimize(obj.gray(2), rg, rd2, rs, K, np.zeros(6, np.float32), 2, want_mask=True)
    assert not o3["mask"][30:40, 50:60].any()            # depth < 0.20 gate, optimize.cpp:39


def test_jacobian_matches_finite_differences_of_the_warped_image():
    # single-pixel trick: only one pixel has depth >= 0.2, so H = J^T J and g = J^T (w r) of that pixel
    ref, obj, _ = small_pair()
    level = 3
    rg, rd, rs, K = ref.gray(level), ref.depth(level), ref.sigma(level), ref.K(level)
    og = obj.gray(level)
    eps = 2e-3
    for (py, px) in [(100, 150), (60, 200), (170, 80)]:
        d1 = np.zeros_like(rd); d1[py, px] = rd[py, px]
        o = orc.optimize(og, rg, d1, rs, K, np.zeros(6, np.float32), level, crop=False)
        assert o["n_valid"] == 1
        H = orc.upper_to_full(o["H"])
        w = 2.0 / 0.5                                   # step / clamp(sigma)
        r0 = float(rg[py, px]) - float(og[py, px])      # warped(xi=0) = ref itself
        J = o["g"] / (w * r0)
        np.testing.assert_allclose(np.outer(J, J), H, rtol=2e-3, atol=1e-3 * np.abs(H).max())
        # d warped / d xi_k by central differences; the reference's J is built for +xi with un-halved gradients,
        # the warp uses -xi, so d warped / d xi = -J / 2 up to the bilinear interpolation of a smooth texture
        for k in range(6):
            e = np.zeros(6, np.float32); e[k] = eps
            wp = orc.warp_image(e, rg, d1, K)[py, px]
            wm = orc.warp_image(-e, rg, d1, K)[py, px]
            fd = (float(wp) - float(wm)) / (2 * eps)
            assert abs(fd - (-J[k] / 2)) < 0.06 * max(abs(J).max() / 2, 1e-3), (k, fd, -J[k] / 2)


def test_stacked_least_squares_equals_normal_equations():
    ref, obj, _ = small_pair()
    for level in (0, 1, 2):
        a = orc.optimize(obj.gray(level), ref.gray(level), ref.depth(level), ref.sigma(level), ref.K(level),
                         np.zeros(6, np.float32), level, variant=0)
        b = orc.optimize(obj.gray(level), ref.gray(level), ref.depth(level), ref.sigma(level), ref.K(level),
                         np.zeros(6, np.float32), level, variant=1)
        assert a["n_valid"] == b["n_valid"]
        np.testing.assert_array_equal(a["H"], b["H"])            # same per-pixel values in both variants
        np.testing.assert_allclose(b["xi_update"], a["xi_update"], rtol=2e-3, atol=2e-6)
        np.testing.assert_allclose(a["xi_update"], np.linalg.solve(orc.upper_to_full(a["H"]), a["g"]), rtol=1e-4)


def test_gauss_newton_step_with_unit_gain_reduces_the_residual():
    # gain = w / 2 (un-halved gradients): with step/sigma = 2 the update IS the Gauss-Newton step
    g, d, s, _ = frames()
    sg = np.full_like(d[0], 0.5)
    ref = orc.OFrame(g[0], d[0], sg, K640, 4, 1)
    obj = orc.OFrame(g[1], d[1], sg, K640, 4, 1)
    level = 2                                            # step 1.0 / sigma 0.5 -> w = 2
    args = (obj.gray(level), ref.gray(level), ref.depth(level), ref.sigma(level), ref.K(level))
    xi = np.zeros(6, np.float32)
    r = []
    for _ in range(4):
        o = orc.optimize(*args, xi, level)
        r.append(float(o["residual"]))
        xi = orc.se3_concatenate(xi, o["xi_update"])
    assert r[-1] < 0.5 * r[0]


def test_track_variants_agree_and_log_is_consistent():
    ref, obj, _ = small_pair(sigma=0.5)
    xa, la = orc.track(obj, ref, variant=0)
    xb, lb = orc.track(obj, ref, variant=1)
    assert la["n_iter"] == lb["n_iter"]
    np.testing.assert_allclose(xb, xa, rtol=0, atol=3e-5)
    for l in range(4):
        assert 1 <= la["n_iter"][l] <= 15                # tracker.cpp:19
        last = la["n_iter"][l] - 1
        stopped = la["upd_norm"][l][last] < 5e-4 or la["residual"][l][last] < 5e-3 or la["n_iter"][l] == 15
        assert stopped                                    # tracker.cpp:68-73
    np.testing.assert_array_equal(xa, la["xi_after"][3][-1])
    xf, lf = orc.track(obj, ref, fixed_iters=3)
    assert lf["n_iter"] == [3, 3, 3, 3]


# ---------------------------------------------------------------- mapping
def test_propagate_identity_is_age_plus_one():
    ref, _, _ = small_pair()
    d, s, K = ref.depth(2), ref.sigma(2), ref.K(2)
    age = np.arange(d.size, dtype=np.float32).reshape(d.shape) % 7
    d[5, 5] = 0.0                                        # no depth: not propagated (implement.cpp:237)
    od, os_, oa = orc.propagate(d, s, age, np.zeros(6, np.float32), K)
    m = np.ones(d.shape, bool); m[5, 5] = False
    np.testing.assert_array_equal(od[m], d[m])
    np.testing.assert_array_equal(oa[m], age[m] + 1)
    np.testing.assert_allclose(os_[m], np.sqrt(s[m] ** 2 + np.float32(0.06) ** 2), rtol=1e-6)
    assert od[5, 5] == 1.0 and os_[5, 5] == 1.0 and oa[5, 5] == 0.0   # holes: implement.cpp:229-231


def test_propagate_forward_motion_and_last_writer_wins():
    d = np.full((20, 30), 2.0, np.float32); s = np.full_like(d, 0.3); a = np.zeros_like(d)
    K = np.array([[30, 0, 15], [0, 30, 10], [0, 0, 1]], np.float32)
    xi = np.array([0, 0, -0.5, 0, 0, 0], np.float32)     # points come 0.5 m closer
    od, os_, oa = orc.propagate(d, s, a, xi, K)
    hit = oa > 0
    assert hit.sum() > 100
    np.testing.assert_allclose(od[hit], 1.5, atol=1e-6)   # d1 = d0 + xi[2]
    np.testing.assert_allclose(os_[hit], np.sqrt((1.5 / 2.0) ** 4 * 0.09 + 0.0036), rtol=1e-5)
    # points recede: several sources collapse onto one target; the LAST in raster order must win (D7)
    a2 = np.arange(d.size, dtype=np.float32).reshape(d.shape)
    od2, os2, oa2 = orc.propagate(d, s, a2, np.array([0, 0, 2.0, 0, 0, 0], np.float32), K)
    src_y, src_x = np.divmod(np.arange(d.size), d.shape[1])
    tx = np.rint((src_x - 15) * 2.0 / 4.0 + 15).astype(int); ty = np.rint((src_y - 10) * 2.0 / 4.0 + 10).astype(int)
    exp = np.zeros_like(d)
    for i in range(d.size):
        exp[ty[i], tx[i]] = a2.flat[i] + 1
    np.testing.assert_array_equal(oa2, exp)


def test_regularize_constant_map_is_fixed_point_and_spike_is_isolated():
    d = np.full((12, 14), 1.5, np.float32); s = np.full_like(d, 0.3)
    np.testing.assert_allclose(orc.regularize(d, s), d, atol=1e-6)
    d[6, 7] = 3.0; s[6, 7] = 0.1                         # test/regularize.cpp:30-31
    out = orc.regularize(d, s)
    assert out[6, 7] == np.float32(3.0)                  # gated: neighbours too far, nothing fuses
    assert abs(out[6, 6] - 1.5) < 1e-6                   # and the spike does not leak into its neighbours
    d[0, 0] = 9.0
    assert orc.regularize(d, s)[0, 0] == np.float32(6.0)  # cv::min(., 6), implement.cpp:178


def test_depth_update_recovers_depth_on_synthetic_stereo():
    from dvo_amd import synth
    T0 = np.eye(4); T1 = np.eye(4); T1[0, 3] = 0.30       # 30 cm baseline along x (world <- camera)
    K = orc.cull_intrinsic(K640, 2)
    g0, d0 = synth.render(T0, K, 160, 120)
    g1, d1 = synth.render(T1, K, 160, 120)
    g0, d0, g1, d1 = g0.numpy(), d0.numpy(), g1.numpy(), d1.numpy()
    # exp(-r_xi) must map obj (= cam1) points into born (= cam0): p0 = p1 + (0.3, 0, 0)  ->  r_xi = (-0.3, 0, ...)
    r_xi = np.array([-0.30, 0, 0, 0, 0, 0], np.float32)
    err_prior, err_post = [], []
    for (qy, qx) in [(40, 50), (60, 80), (70, 100), (50, 110), (80, 60), (30, 90), (90, 75), (65, 45)]:
        true = float(d1[qy, qx])
        prior = true * 1.08
        nd, ns = orc.implement_update(g1, g0, r_xi, K, qx, qy, prior, 0.25)
        if nd > 0:
            err_prior.append(abs(prior - true)); err_post.append(abs(float(nd) - true))
            assert ns > 0 and np.isfinite(ns)   # often > 0.5 on this low-contrast texture: the mapper gate rejects those
    assert len(err_post) >= 5
    assert np.median(err_post) < 0.5 * np.median(err_prior)


def test_keyframe_policy():
    assert orc.need_new_frame([0.03, 0, 0, 0, 0, 0], 1, 0)           # mapper.cpp:49
    assert not orc.need_new_frame([0.01, 0.01, 0.01, 1, 1, 1], 5, 0)  # translation only, 0.0173 < 0.02
    assert orc.need_new_frame([0, 0, 0, 0, 0, 0], 6, 0)              # mapper.cpp:53: id - ref.id >= 6


def test_visual_odometry_mono_runs_and_keeps_invariants():
    g, d, s, _ = frames(6, seed=7)
    vo = orc.OVO(K640, 640, 480, seed=3)
    d0 = orc.cull_image(d[0], 2)
    vo.set_initial_depth(d0, np.full_like(d0, 0.5))
    keys = []
    for i in range(6):
        T, key = vo.odometrize(g[i])
        keys.append(key)
        assert np.isfinite(T).all()
        R = T[:3, :3].astype(np.float64)
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-5)
    assert keys[0] is True                                # system.hpp:49-54
    kf = vo.keyframe(vo.keyframe_count() - 1)
    assert kf.size(2) == (160, 120) and kf.size(0) == (40, 30)        # Frame(gray,K,3,2), system.hpp:47
    assert kf.depth(2).max() <= 6.0 + 1e-6                            # regularize clamps, implement.cpp:178
    assert 1 <= vo.keyframe_count() <= 6


def test_visual_odometry_depth_mode_relative_pose():
    g, d, s, _ = frames(3, sigma=0.5)
    vo = orc.OVO(K640, 640, 480)
    T0 = vo.odometrize_depth(g[0], d[0], s[0])
    np.testing.assert_array_equal(T0, np.eye(4, dtype=np.float32))     # system.hpp:83-86
    T1 = vo.odometrize_depth(g[1], d[1], s[1])
    ref = orc.OFrame(g[0], d[0], s[0], K640, 4, 1); obj = orc.OFrame(g[1], d[1], s[1], K640, 4, 1)
    xi, _ = orc.track(obj, ref)
    np.testing.assert_array_equal(T1, orc.se3_exp(xi))                 # returns exp(relative_xi), system.hpp:92


def test_row_parallel_oracle_agrees_with_the_sequential_one():
    """orc.set_threads(n): the forEach bodies run row-parallel (what cv::Mat::forEach does, optimize.cpp:28) for bench.py's
    all-core CPU baseline.  Pixel selection is identical; sums differ only by the association order of per-thread partials."""
    from util import frames, K640
    g, d, s, _ = frames()
    ref = orc.OFrame(g[0], d[0], s[0], K640, 4, 1)
    obj = orc.OFrame(g[1], d[1], s[1], K640, 4, 1)
    xi = np.array([0.002, -0.001, 0.003, 0.002, -0.001, 0.001], np.float32)
    res = []
    try:
        for n in (1, 4):
            orc.set_threads(n)
            res.append([orc.optimize(obj.gray(3), ref.gray(3), ref.depth(3), ref.sigma(3), ref.K(3), xi, 3, variant=v, want_mask=True)
                        for v in (0, 1)])
    finally:
        orc.set_threads(1)
    for v in (0, 1):
        a, b = res[0][v], res[1][v]
        np.testing.assert_array_equal(a["mask"], b["mask"])
        assert a["n_valid"] == b["n_valid"]
        np.testing.assert_allclose(b["H"], a["H"], rtol=1e-12)
        np.testing.assert_allclose(b["xi_update"], a["xi_update"], rtol=1e-5, atol=1e-9)
