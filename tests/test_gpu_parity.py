"""GPU parity tests: the HIP path (through the C ABI of libdvo.so) against the CPU oracle on the same seeded
inputs.  Bit-exact for masks / indices / per-pixel maps; stated float tolerances (tests/util.py) for reductions,
solves and poses.  Run on the MI355X box with `pytest -m gpu`.
"""
import numpy as np
import pytest

import dvo_amd as dvo
import orc
from util import K640, TOL_BACKWARD, TOL_H_REL, TOL_POSE, TOL_UPD_ABS, TOL_UPD_REL, assert_composed, backward_error, frames

pytestmark = pytest.mark.gpu
INV = np.float32(-2.0)


def test_library_loads_and_sees_gpu():
    assert dvo.device_count() >= 1
    assert b"gfx950" in dvo.lib().dvo_version()


# ---------------------------------------------------------------- math::se3 on the device
def test_se3_device_matches_oracle():
    rng = np.random.RandomState(0)
    for scale in (1e-7, 1e-3, 0.05, 1.0):
        for _ in range(8):
            a = (rng.uniform(-1, 1, 6) * scale).astype(np.float32)
            b = (rng.uniform(-1, 1, 6) * scale * 0.3).astype(np.float32)
            np.testing.assert_allclose(dvo.se3.exp(a), orc.se3_exp(a), rtol=0, atol=1.2e-7)
            T = orc.se3_exp(a)
            np.testing.assert_allclose(dvo.se3.log(T), orc.se3_log(T), rtol=2e-6, atol=1e-9)
            np.testing.assert_allclose(dvo.se3.concatenate(a, b), orc.se3_concatenate(a, b), rtol=2e-6, atol=1e-9)


# ---------------------------------------------------------------- Convert
def test_cull_image_bit_exact():
    rng = np.random.RandomState(1)
    img = rng.uniform(0, 1, (97, 131)).astype(np.float32)
    img[rng.uniform(size=img.shape) < 0.05] = INV
    img[5, 7] = -7.0
    img[8, 8] = np.nan
    for t in (0, 1, 2, 3):
        np.testing.assert_array_equal(dvo.Convert.cullImage(img, t), orc.cull_image(img, t))


def test_gradiate_bit_exact():
    rng = np.random.RandomState(2)
    img = rng.uniform(0, 1, (60, 83)).astype(np.float32)
    img[rng.uniform(size=img.shape) < 0.05] = INV
    for xdir in (True, False):
        np.testing.assert_array_equal(dvo.Convert.gradiate(img, xdir), orc.gradiate(img, xdir))
    one = np.ones((1, 1), np.float32)  # degenerate sizes
    np.testing.assert_array_equal(dvo.Convert.gradiate(one, True), orc.gradiate(one, True))


def test_pyramid_matches_frame_construction():
    g, d, s, _ = frames()
    gg = g[0].copy(); dd = d[0].copy()
    gg[100:110, 200:230] = INV
    dd[50:60, 10:40] = 0.0
    for levels, culls in ((4, 1), (3, 2), (5, 0)):
        of = orc.OFrame(gg, dd, s[0], K640, levels, culls)
        go, do, so = dvo.pyramid(gg, dd, s[0], levels, culls)
        for l in range(levels):
            np.testing.assert_array_equal(go[l], of.gray(l))
            np.testing.assert_array_equal(do[l], of.depth(l))
            np.testing.assert_array_equal(so[l], of.sigma(l))


# ---------------------------------------------------------------- Transform::warpImage
@pytest.mark.parametrize("level", [0, 2, 3])
def test_warp_image_bit_exact(level):
    g, d, s, _ = frames()
    of = orc.OFrame(g[0], d[0], s[0], K640, 4, 1)
    gray, depth, K = of.gray(level), of.depth(level), of.K(level)
    depth[3:6, 4:9] = 0.0
    gray[10:12, 10:14] = INV   # exercises the getSubpixel fill quirk
    gray[20, 20:24] = 0.0      # black pixels: `last > 0` is false
    xi = np.array([0.01, -0.006, 0.008, 0.004, -0.003, 0.006], np.float32)
    got = dvo.Transform.warpImage(xi, gray, depth, K)
    exp = orc.warp_image(xi, gray, depth, K)
    np.testing.assert_array_equal(got, exp)


# ---------------------------------------------------------------- Track::optimize
def _gn_compare(obj_gray, ref_gray, ref_depth, ref_sigma, K, xi, level, cfg=None, crop=True):
    o = orc.optimize(obj_gray, ref_gray, ref_depth, ref_sigma, K, xi, level, crop=crop, want_mask=True)
    r = dvo.optimize(obj_gray, ref_gray, ref_depth, ref_sigma, K, xi, level, cfg=cfg, want_mask=True)
    np.testing.assert_array_equal(r["mask"], o["mask"])          # pixel selection: bit exact
    assert r["n_valid"] == o["n_valid"]
    if o["n_valid"] == 0:
        assert r["residual"] == np.float32(-1) and not r["xi_update"].any()
        return o, r
    hs = np.abs(o["H"]).max()
    np.testing.assert_allclose(r["H"], o["H"], rtol=0, atol=TOL_H_REL * hs)
    np.testing.assert_allclose(r["g"], o["g"], rtol=0, atol=TOL_H_REL * max(np.abs(o["g"]).max(), 1e-30))
    np.testing.assert_allclose(r["sum_r2"], o["sum_r2"], rtol=2e-5)
    np.testing.assert_allclose(r["residual"], o["residual"], rtol=2e-5)
    un = np.linalg.norm(o["xi_update"])
    np.testing.assert_allclose(r["xi_update"], o["xi_update"], rtol=0, atol=TOL_UPD_REL * un + TOL_UPD_ABS)
    np.testing.assert_allclose(r["xi_next"], orc.se3_concatenate(xi, r["xi_update"]), rtol=2e-6, atol=1e-9)
    return o, r


@pytest.mark.parametrize("level", [0, 1, 2, 3])
def test_gn_step_parity_every_level(level):
    g, d, s, _ = frames()
    ref = orc.OFrame(g[0], d[0], s[0], K640, 4, 1)
    obj = orc.OFrame(g[1], d[1], s[1], K640, 4, 1)
    xi = np.array([0.002, -0.001, 0.003, 0.002, -0.001, 0.001], np.float32)
    o, _ = _gn_compare(obj.gray(level), ref.gray(level), ref.depth(level), ref.sigma(level), ref.K(level), xi, level)
    assert o["n_valid"] > 500


def test_gn_step_invalid_pixels_borders_and_crop_off():
    g, d, s, _ = frames()
    ref = orc.OFrame(g[0], d[0], s[0], K640, 3, 2)
    obj = orc.OFrame(g[2], d[2], s[2], K640, 3, 2)
    rg, rd, rs, K = ref.gray(2), ref.depth(2), ref.sigma(2), ref.K(2)
    og = obj.gray(2)
    rng = np.random.RandomState(5)
    rg[rng.uniform(size=rg.shape) < 0.03] = INV
    og[rng.uniform(size=og.shape) < 0.03] = INV
    rd[rng.uniform(size=rd.shape) < 0.05] = 0.0
    rd[30:40, 50:60] = 0.15                      # below the 0.20 gate
    rs[:, :80] = 0.003; rs[:, 80:] = 0.8          # both clamps of optimize.cpp:83
    rg[60, 60:70] = 0.0
    xi = np.array([0.03, -0.02, 0.01, 0.01, 0.02, -0.015], np.float32)  # large: many warps leave the image
    _gn_compare(og, rg, rd, rs, K, xi, 2)
    cfg = dvo.default_config(crop_enable=0)
    _gn_compare(og, rg, rd, rs, K, xi, 2, cfg=cfg, crop=False)


def test_gn_step_no_valid_pixels():
    z = np.zeros((30, 40), np.float32)
    K = np.array([[30, 0, 20], [0, 30, 15], [0, 0, 1]], np.float32)
    _gn_compare(z + 0.5, z + 0.5, z, z + 0.5, K, np.zeros(6, np.float32), 0)


def test_gn_step_is_bit_reproducible():
    g, d, s, _ = frames()
    ref = orc.OFrame(g[0], d[0], s[0], K640, 4, 1)
    obj = orc.OFrame(g[1], d[1], s[1], K640, 4, 1)
    xi = np.zeros(6, np.float32)
    a = dvo.optimize(obj.gray(3), ref.gray(3), ref.depth(3), ref.sigma(3), ref.K(3), xi, 3)
    b = dvo.optimize(obj.gray(3), ref.gray(3), ref.depth(3), ref.sigma(3), ref.K(3), xi, 3)
    np.testing.assert_array_equal(a["H"], b["H"])
    np.testing.assert_array_equal(a["g"], b["g"])
    np.testing.assert_array_equal(a["xi_update"], b["xi_update"])


# ---------------------------------------------------------------- Tracker::track
def _track_compare(gi, gj, levels, culls, sigma_value, cfg=None):
    g, d, s, _ = frames()
    sg = np.full_like(d[gi], sigma_value)
    ref = orc.OFrame(g[gi], d[gi], sg, K640, levels, culls)
    obj = orc.OFrame(g[gj], d[gj], sg, K640, levels, culls)
    xo, lo = orc.track(obj, ref)
    xg, lg = dvo.track(g[gj], g[gi], d[gi], sg, K640, levels, culls, cfg=cfg)
    return xo, lo, xg, lg


@pytest.mark.parametrize("levels,culls", [(4, 1), (3, 2)])
def test_track_parity_contracting_gain(levels, culls):
    # sigma = 0.5 (what the mono pipeline starts from): gain <= 2, the iteration is stable
    xo, lo, xg, lg = _track_compare(0, 1, levels, culls, 0.5)
    assert lg["n_iter"] == lo["n_iter"]
    for l in range(levels):
        np.testing.assert_array_equal(lg["n_valid"][l], lo["n_valid"][l])
        np.testing.assert_allclose(lg["residual"][l], lo["residual"][l], rtol=1e-4)
        np.testing.assert_allclose(lg["xi_after"][l], lo["xi_after"][l], rtol=0, atol=TOL_POSE)
    np.testing.assert_allclose(xg, xo, rtol=0, atol=TOL_POSE)


def test_track_over_relaxed_gain_per_iteration_parity():
    # sigma = 0.1 (sensor depth, transform.cpp:75): w/2 = 10x over-relaxed, the reference's iteration is chaotic.
    # Whole-call pose parity is meaningless there; what must hold is parity of EVERY iteration given its input pose.
    g, d, s, _ = frames()
    sg = np.full_like(d[0], 0.1)
    ref = orc.OFrame(g[0], d[0], sg, K640, 4, 1)
    obj = orc.OFrame(g[1], d[1], sg, K640, 4, 1)
    xo, lo = orc.track(obj, ref)
    xi = np.zeros(6, np.float32)
    for l in range(4):
        for it in range(lo["n_iter"][l]):
            r = dvo.optimize(obj.gray(l), ref.gray(l), ref.depth(l), ref.sigma(l), ref.K(l), xi, l)
            assert r["n_valid"] == lo["n_valid"][l][it]
            np.testing.assert_allclose(r["residual"], lo["residual"][l][it], rtol=1e-4)
            o = orc.optimize(obj.gray(l), ref.gray(l), ref.depth(l), ref.sigma(l), ref.K(l), xi, l)
            assert backward_error(o["H"], o["g"], r["xi_update"]) <= TOL_BACKWARD, (l, it)
            assert_composed(xi, r["xi_update"], r["xi_next"], tag=(l, it))
            xi = lo["xi_after"][l][it]  # follow the oracle's trajectory


def test_track_fixed_iterations_roofline_preset():
    g, d, s, _ = frames()
    sg = np.full_like(d[0], 0.5)
    cfg = dvo.default_config(fixed_iterations=3, crop_enable=0)
    ref = orc.OFrame(g[0], d[0], sg, K640, 4, 1)
    obj = orc.OFrame(g[1], d[1], sg, K640, 4, 1)
    xo, lo = orc.track(obj, ref, crop=False, fixed_iters=3)
    xg, lg = dvo.track(g[1], g[0], d[0], sg, K640, 4, 1, cfg=cfg)
    assert lg["n_iter"] == [3, 3, 3, 3] == lo["n_iter"]
    np.testing.assert_allclose(xg, xo, rtol=0, atol=5e-5)


# ---------------------------------------------------------------- Map::Implement
def test_propagate_bit_exact():
    g, d, s, _ = frames()
    of = orc.OFrame(g[0], d[0], s[0], K640, 3, 2)
    depth, sigma, K = of.depth(2), of.sigma(2), of.K(2)
    rng = np.random.RandomState(3)
    age = rng.randint(0, 5, depth.shape).astype(np.float32)
    depth[rng.uniform(size=depth.shape) < 0.1] = 0.0
    for xi in ([0.01, -0.01, 0, 0, 0, 0],             # test/propagate.cpp:51
               [0.05, 0.02, -0.3, 0.02, -0.03, 0.05],  # strong forward motion: many collisions
               [0, 0, 0, 0, 0, 0]):
        xi = np.array(xi, np.float32)
        got = dvo.Implement.propagate(depth, sigma, age, xi, K)
        exp = orc.propagate(depth, sigma, age, xi, K)
        for a, b in zip(got, exp):
            np.testing.assert_array_equal(a, b)


def test_regularize_bit_exact():
    rng = np.random.RandomState(4)
    # test/regularize.cpp:26-32: 50x50 randn(1.5,0.5) depth, randn(0.3,0.2) sigma, one spike
    depth = rng.normal(1.5, 0.5, (50, 50)).astype(np.float32)
    sigma = np.abs(rng.normal(0.3, 0.2, (50, 50))).astype(np.float32) + 0.01
    depth[25, 25] = 3.0; sigma[25, 25] = 0.1
    depth[0, 0] = 9.0
    np.testing.assert_array_equal(dvo.Implement.regularize(depth, sigma), orc.regularize(depth, sigma))


def _mapping_scene(n_frames=4, sigma0=0.5):
    g, d, s, poses = frames(6, seed=7)
    return g, d, s, poses


def test_mapper_update_parity():
    g, d, s, poses = frames(6, seed=7)
    # two keyframes + one tracked frame, depth of the reference keyframe perturbed as in test/update.cpp:64-71
    rng = np.random.RandomState(9)
    kf0 = orc.OFrame(g[0], d[0], np.full_like(d[0], 0.5), K640, 3, 2, id=0)
    kf1 = orc.OFrame(g[2], d[2], np.full_like(d[2], 0.5), K640, 3, 2, id=2)
    obj = orc.OFrame(g[3], None, None, K640, 3, 2, id=3)
    T01 = np.linalg.inv(poses[2]) @ poses[0]
    T12 = np.linalg.inv(poses[3]) @ poses[2]
    xi1 = orc.se3_log(T01.astype(np.float32)) * np.float32(20)   # exaggerate the baseline so stereo is informative
    rel = orc.se3_log(T12.astype(np.float32)) * np.float32(20)
    kf1.set_pose(xi1, xi1)
    obj.set_pose(orc.se3_concatenate(xi1, rel), rel)
    top_d = kf1.depth(2) + rng.normal(0, 0.05, kf1.depth(2).shape).astype(np.float32)
    top_s = np.full_like(top_d, 0.3)
    age = (rng.uniform(size=top_d.shape) < 0.5).astype(np.float32)   # half the pixels were born in kf0
    kf1.update_depth_sigma(top_d, top_s)
    kf1.set_age(age)
    K = kf1.K(2)
    got_d, got_s, got_a, got_v = dvo.mapper_update([kf0.gray(2), kf1.gray(2)], [kf0.xi, kf1.xi], obj.gray(2), obj.xi,
                                                   obj.rel_xi, 3, K, top_d, top_s, age,
                                                   cfg=dvo.default_config(rng_seed=11))
    v = orc.mapper_update([kf0, kf1], obj, 11)
    np.testing.assert_array_equal(got_a, kf1.age())
    np.testing.assert_array_equal(got_d, kf1.depth(2))
    np.testing.assert_array_equal(got_s, kf1.sigma(2))
    assert got_v == v
    assert (got_d != top_d).sum() > 100   # the test really exercised the update


# ---------------------------------------------------------------- System::VisualOdometry
def test_odometrize_using_depth_sequence():
    g, d, s, _ = frames(4, sigma=0.5)
    vo = dvo.VisualOdometry(K640, 640, 480)
    ovo = orc.OVO(K640, 640, 480)
    for i in range(4):
        T = vo.odometrizeUsingDepth(g[i], d[i], s[i])
        To = ovo.odometrize_depth(g[i], d[i], s[i])
        np.testing.assert_allclose(T, To, rtol=0, atol=TOL_POSE)
    vo.close()


def test_odometrize_mono_mapping_sequence():
    g, d, s, _ = frames(6, seed=7)
    rng = np.random.RandomState(12)
    d0 = orc.cull_image(d[0], 2)
    init_d = (d0 + rng.normal(0, 0.1, d0.shape)).astype(np.float32)
    init_s = np.full_like(init_d, 0.5)
    vo = dvo.VisualOdometry(K640, 640, 480, cfg=dvo.default_config(rng_seed=3))
    ovo = orc.OVO(K640, 640, 480, seed=3)
    vo.setInitialDepth(init_d, init_s)
    ovo.set_initial_depth(init_d, init_s)
    for i in range(6):
        T, key = vo.odometrize(g[i])
        To, keyo = ovo.odometrize(g[i])
        assert key == keyo
        np.testing.assert_allclose(T, To, rtol=0, atol=1e-4)
        assert vo.keyframeCount() == ovo.keyframe_count()
        kf = vo.keyframe(vo.keyframeCount() - 1)
        okf = ovo.keyframe(ovo.keyframe_count() - 1)
        np.testing.assert_array_equal(kf["age"], okf.age())
        bad = np.abs(kf["depth"] - okf.depth(2)) > 1e-3
        assert bad.mean() < 0.01   # pose differences of 1e-5 may flip a handful of stereo matches
    vo.close()


# ---------------------------------------------------------------- batch
def test_batch_matches_single_and_is_deterministic():
    g, d, s, _ = frames(4, sigma=0.5)
    B = 3
    perm = [[0, 1, 2], [1, 2, 3], [2, 1, 0]]  # sequence b sees frames perm[b]
    xs = []
    for rep in range(2):
        bt = dvo.Batch(B, K640, 640, 480, 4, 1)
        out = []
        for step in range(3):
            gg = np.stack([g[perm[b][step]] for b in range(B)])
            dd = np.stack([d[perm[b][step]] for b in range(B)])
            ss = np.stack([s[perm[b][step]] for b in range(B)])
            bt.push_host(gg, dd, ss)
            if step > 0:
                out.append(bt.last_poses()[0].copy())
        xs.append(np.stack(out))
        bt.close()
    np.testing.assert_array_equal(xs[0], xs[1])          # bit-reproducible
    for b in range(B):
        for step in (1, 2):
            i, j = perm[b][step - 1], perm[b][step]
            x1, _ = dvo.track(g[j], g[i], d[i], s[i], K640, 4, 1)
            np.testing.assert_array_equal(xs[0][step - 1][b], x1)  # batching does not change a sequence's result


def test_bad_arguments_return_status_not_abort():
    L = dvo.lib()
    assert L.dvo_vo_create(None, 640, 480, None, None) == 1
    with pytest.raises(dvo.DvoError):
        dvo.VisualOdometry(K640, 8, 8)
    with pytest.raises(dvo.DvoError):
        dvo.Batch(0, K640, 640, 480)
    vo = dvo.VisualOdometry(K640, 640, 480)
    with pytest.raises(dvo.DvoError):
        vo.keyframeInfo(0)      # FrameHistory::operator[] .at() throws in the reference (frame.hpp:176)
    vo.close()


# ---------------------------------------------------------------- kernel variants (tiling / LDS patch) agree
@pytest.mark.parametrize("lds,ppt,group", [(0, 1, 1), (0, 2, 2), (0, 4, 1), (0, 4, 4), (0, 8, 4), (8, 1, 0), (8, 4, 0), (2, 2, 0), (16, 8, 0)])
def test_gn_kernel_variants_match_oracle(lds, ppt, group):
    g, d, s, _ = frames()
    ref = orc.OFrame(g[0], d[0], s[0], K640, 4, 1)
    obj = orc.OFrame(g[1], d[1], s[1], K640, 4, 1)
    rg = ref.gray(3); rg[100:104, 150:160] = INV           # force the generic sampler inside the image too
    # large motion: many footprints leave a margin-2 patch and some leave the image
    xi = np.array([0.02, -0.015, 0.01, 0.01, 0.012, -0.02], np.float32)
    cfg = dvo.default_config(gn_use_lds_patch=lds, gn_pixels_per_thread=ppt, gn_gather_group=group)
    for level in (1, 3):
        rgl = rg if level == 3 else ref.gray(level)
        _gn_compare(obj.gray(level), rgl, ref.depth(level), ref.sigma(level), ref.K(level), xi, level, cfg=cfg)


# ---------------------------------------------------------------- schedule variants leave the results untouched
def test_batch_schedule_variants_are_bit_identical():
    """Sub-batches on concurrent streams and the one-launch coarse levels (k_track_level) only change WHEN the same
    arithmetic runs: at the same tile size every pose is bit-identical to the default schedule."""
    g, d, s, _ = frames(3, sigma=0.5)
    B = 16
    order = [(b + np.arange(3)) % 3 for b in range(B)]
    res = []
    for kw in ({}, {"track_streams": 2}, {"track_fused_tiles": 8}, {"track_streams": 2, "track_fused_tiles": 8}):
        cfg = dvo.default_config(gn_pixels_per_thread=4, **kw)
        bt = dvo.Batch(B, K640, 640, 480, 4, 1, cfg=cfg)
        out, logs = [], []
        for step in range(3):
            bt.push_host(np.stack([g[order[b][step]] for b in range(B)]), np.stack([d[order[b][step]] for b in range(B)]),
                         np.stack([s[order[b][step]] for b in range(B)]))
            if step > 0:
                out.append(bt.last_poses()[0].copy())
                logs.append([list(bt.last_track_log(b)["n_iter"][:4]) for b in (0, 5, B - 1)])
        res.append((np.stack(out), logs))
        bt.close()
    assert np.isfinite(res[0][0]).all() and np.abs(res[0][0]).max() > 1e-4
    variants = ("default", "2 streams", "k_track_level", "2 streams + k_track_level")
    for name, other in zip(variants[1:], res[1:]):
        bad = np.argwhere((res[0][0] != other[0]).any(axis=2))
        assert bad.size == 0, "%s differs from the default schedule at (step, sequence) %s: %s vs %s" % (
            name, bad.tolist(), res[0][0][tuple(bad[0])], other[0][tuple(bad[0])])
        assert res[0][1] == other[1]


def test_batch_many_iterations_every_sequence_matches_single():
    """sigma = 0.1 (the sensor-depth workload, 10x over-relaxed): sequences run different numbers of iterations, up to
    the cap, so the active-sequence lists are exercised for real.  At the same tile size every sequence of the batch
    must reproduce the single-sequence tracker bit for bit (pose AND per-iteration log)."""
    g, d, s, _ = frames(4, sigma=0.1)
    pairs = [(i, j) for i in range(4) for j in range(4) if i != j]   # 12 (reference, object) pairs
    B = len(pairs)
    cfg = dvo.default_config(gn_pixels_per_thread=4)
    bt = dvo.Batch(B, K640, 640, 480, 4, 1, cfg=cfg)
    bt.push_host(np.stack([g[i] for i, _ in pairs]), np.stack([d[i] for i, _ in pairs]), np.stack([s[i] for i, _ in pairs]))
    bt.push_host(np.stack([g[j] for _, j in pairs]), np.stack([d[j] for _, j in pairs]), np.stack([s[j] for _, j in pairs]))
    xb = bt.last_poses()[0].copy()
    logs = [bt.last_track_log(b) for b in range(B)]
    bt.close()
    iters = set()
    for b, (i, j) in enumerate(pairs):
        x1, lg = dvo.track(g[j], g[i], d[i], s[i], K640, 4, 1, cfg=cfg)
        np.testing.assert_array_equal(xb[b], x1)
        assert list(logs[b]["n_iter"][:4]) == list(lg["n_iter"][:4])
        for lvl in range(4):
            n = lg["n_iter"][lvl]
            np.testing.assert_array_equal(np.asarray(logs[b]["residual"][lvl][:n]), np.asarray(lg["residual"][lvl][:n]))
            iters.add(n)
    assert max(iters) >= 5 and len(iters) >= 3   # the case really has long and differing iteration counts


# ---------------------------------------------------------------- BASELINE config 4 (1920x1080, 5 levels) and ragged sizes
def test_syn1080_five_level_parity():
    """Synthetic 1920x1080 dense alignment, 5-level pyramid, culls 0 (BASELINE.json configs[3]): pyramid bit exact,
    one Gauss-Newton step per level (masks bit exact) and a fixed-iteration track against the oracle."""
    from dvo_amd import synth
    K = synth.K_1080
    g, d, s, _ = synth.sequence(2, width=1920, height_px=1080, seed=7, sigma_value=0.5)
    g, d, s = g.numpy(), d.numpy(), s.numpy()
    ref = orc.OFrame(g[0], d[0], s[0], K, 5, 0)
    obj = orc.OFrame(g[1], d[1], s[1], K, 5, 0)
    pg, pd, _ = dvo.pyramid(g[0], d[0], s[0], 5, 0)
    for l in range(5):
        np.testing.assert_array_equal(pg[l], ref.gray(l))
        np.testing.assert_array_equal(pd[l], ref.depth(l))
    xi = np.array([0.002, -0.001, 0.003, 0.002, -0.001, 0.001], np.float32)
    cfg = dvo.default_config(crop_enable=0)
    for l in (0, 2, 4):   # 120x67, 480x270, 1920x1080
        o, _ = _gn_compare(obj.gray(l), ref.gray(l), ref.depth(l), ref.sigma(l), ref.K(l), xi, l, cfg=cfg, crop=False)
        assert o["n_valid"] > 0.5 * ref.gray(l).size
    cfg = dvo.default_config(fixed_iterations=2, crop_enable=0)
    xo, lo = orc.track(obj, ref, crop=False, fixed_iters=2)
    xg, lg = dvo.track(g[1], g[0], d[0], s[0], K, 5, 0, cfg=cfg)
    assert lg["n_iter"][:5] == [2] * 5 == lo["n_iter"]
    np.testing.assert_allclose(xg, xo, rtol=0, atol=5e-5)


@pytest.mark.parametrize("w,h,levels,culls", [(322, 243, 3, 0), (161, 121, 2, 0), (646, 486, 4, 1), (37, 29, 1, 0)])
def test_ragged_sizes_parity(w, h, levels, culls):
    """Widths that are not multiples of 4 (unaligned rows), tiles that end mid-row, odd pyramid halvings."""
    from dvo_amd import synth
    K = np.array(synth.K_640, np.float32).copy()
    K[0] *= w / 640.0; K[1] *= h / 480.0
    g, d, s, _ = synth.sequence(2, width=w, height_px=h, K=K, seed=11, sigma_value=0.5)
    g, d, s = g.numpy(), d.numpy(), s.numpy()
    ref = orc.OFrame(g[0], d[0], s[0], K, levels, culls)
    obj = orc.OFrame(g[1], d[1], s[1], K, levels, culls)
    xi = np.array([0.004, -0.003, 0.002, 0.003, -0.002, 0.004], np.float32)
    cfg = dvo.default_config(crop_enable=0)
    for l in range(levels):
        _gn_compare(obj.gray(l), ref.gray(l), ref.depth(l), ref.sigma(l), ref.K(l), xi, l, cfg=cfg, crop=False)
    got = dvo.Transform.warpImage(xi, ref.gray(levels - 1), ref.depth(levels - 1), ref.K(levels - 1))
    np.testing.assert_array_equal(got, orc.warp_image(xi, ref.gray(levels - 1), ref.depth(levels - 1), ref.K(levels - 1)))
    xo, lo = orc.track(obj, ref, crop=False)
    xg, lg = dvo.track(g[1], g[0], d[0], s[0], K, levels, culls, cfg=cfg)
    assert lg["n_iter"][:levels] == lo["n_iter"]
    np.testing.assert_allclose(xg, xo, rtol=0, atol=5e-5)


def test_device_reciprocal_is_ieee_for_every_float():
    """The per-pixel 1/Z of project() runs as v_rcp_f32 + two FMA corrections inside [2^-100, 2^100] (IEEE division elsewhere).
    Bit-exact pixel selection rests on it being THE correctly rounded quotient: all 2^32 bit patterns are compared on the device."""
    n_fast, bad, first = dvo.selftest_reciprocal()
    assert bad == 0, "first mismatching bit pattern 0x%08x" % first
    assert n_fast == 2 * (200 * (1 << 23) + 1)   # both signs, exponents 2^-100 .. 2^100 (inclusive end point)


def test_device_trig_kernels_match_the_math_library():
    """The serial end of a Gauss-Newton step (exp / log of SE(3) in double, one lane) uses polynomial sin / cos / atan2 kernels on the
    device.  Against the device math library over their whole fast domain (2^24 arguments: |x| from 1e-12 to 1e5 for sin / cos, the
    (sqrt, trace) plane for atan2): within 1e-15 relative -- a few units in the last place of a double, far below the float rounding
    every pose goes through."""
    es, ec, ea, n = dvo.selftest_trig()
    assert n == 1 << 24
    assert es < 1e-15 and ec < 1e-15 and ea < 1e-15, (es, ec, ea)


def test_device_short_division_and_sqrt_are_ieee():
    """k_regularize(_redecimate) runs its 8 divisions and 4 square roots per pixel as rcp / rsq + FMA corrections when every operand is a
    normal float in [2^-20, 2^20].  Bit-exact maps rest on those being THE correctly rounded results: the square root is compared with
    sqrtf for every float in [2^-100, 2^100]; the division for 2^14 mantissas of b spread over [1, 2) (stride 509: odd, so every
    low-bit pattern occurs) plus the 2^10 on either end, each against ALL 2^23 mantissas of a.  (All 2^46 pairs:
    tools/verify/division_all_pairs.py, profiles/r03_division_all_pairs.txt.)"""
    n, bad, first = dvo.selftest_sqrt()
    assert bad == 0, "first mismatching bit pattern 0x%08x" % first
    assert n == 200 * (1 << 23) + 1
    for (b0, stride, cnt) in ((0, 509, 1 << 14), (0, 1, 1 << 10), ((1 << 23) - (1 << 10), 1, 1 << 10)):
        n, bad, first = dvo.selftest_division(b0, stride, cnt)
        assert n == cnt << 23
        assert bad == 0, "first mismatching pair: mb = 0x%06x, ma = 0x%06x" % (first >> 23, first & 0x7fffff)


def test_adaptive_schedule_changes_no_result():
    """track_adaptive (default): the host stops a level's launches once a launch reported no active sequence.  Skipped launches
    are empty ones, so poses and logs must equal the fixed schedule's (track_adaptive = -1) bit for bit -- single and batch."""
    g, d, s, _ = frames(3, sigma=0.1)
    res = []
    for adaptive in (0, -1):
        cfg = dvo.default_config(gn_pixels_per_thread=4, track_adaptive=adaptive)
        x1, l1 = dvo.track(g[1], g[0], d[0], s[0], K640, 4, 1, cfg=cfg)
        bt = dvo.Batch(12, K640, 640, 480, 4, 1, cfg=cfg)
        bt.push_host(np.stack([g[b % 3] for b in range(12)]), np.stack([d[b % 3] for b in range(12)]), np.stack([s[b % 3] for b in range(12)]))
        bt.push_host(np.stack([g[(b + 1) % 3] for b in range(12)]), np.stack([d[(b + 1) % 3] for b in range(12)]), np.stack([s[(b + 1) % 3] for b in range(12)]))
        xb = bt.last_poses()[0].copy()
        nb = [list(bt.last_track_log(b)["n_iter"][:4]) for b in range(12)]
        bt.close()
        res.append((x1, list(l1["n_iter"][:4]), xb, nb))
    np.testing.assert_array_equal(res[0][0], res[1][0])
    assert res[0][1] == res[1][1]
    np.testing.assert_array_equal(res[0][2], res[1][2])
    assert res[0][3] == res[1][3]


def test_batch_prefetch_changes_no_result():
    """dvo_batch_prefetch_device only moves the pyramid build of the next frame onto a side stream: poses are bit-identical."""
    import torch
    g, d, s, _ = frames(4, sigma=0.1)
    B = 12
    dev = torch.device("cuda", 0)
    tg = [torch.from_numpy(np.stack([g[(b + f) % 4] for b in range(B)])).to(dev) for f in range(4)]
    td = [torch.from_numpy(np.stack([d[(b + f) % 4] for b in range(B)])).to(dev) for f in range(4)]
    ts = [torch.from_numpy(np.stack([s[(b + f) % 4] for b in range(B)])).to(dev) for f in range(4)]
    torch.cuda.synchronize()
    res = []
    for prefetch in (False, True):
        bt = dvo.Batch(B, K640, 640, 480, 4, 1, cfg=dvo.default_config(gn_pixels_per_thread=4))
        out = []
        for k in range(6):
            f = k % 4
            if prefetch and k >= 1 and k < 5:
                fn = (k + 1) % 4
                bt.prefetch_device(tg[fn].data_ptr(), td[fn].data_ptr(), ts[fn].data_ptr())
            bt.push_device(tg[f].data_ptr(), td[f].data_ptr(), ts[f].data_ptr())
            if k >= 1:
                out.append(bt.last_poses()[0].copy())
        res.append(np.stack(out))
        with pytest.raises(dvo.DvoError):   # no more than two frames may wait prefetched
            for f in range(3):
                bt.prefetch_device(tg[f].data_ptr(), td[f].data_ptr(), ts[f].data_ptr())
        bt.close()
    assert np.abs(res[0]).max() > 1e-4
    np.testing.assert_array_equal(res[0], res[1])


@pytest.mark.parametrize("w,h", [(192, 160), (96, 320)])
def test_narrow_2d_tiles_parity(w, h):
    """The 32- and 16-column 2-D tiles of k_track_gn (GnTiling: widths that are multiples of 32 / 16 but not 64) at 4 pixels per
    thread -- until now only covered by the host-compiled geometry check: 192x160 -> levels 96x80 (32-wide tiles) and 192x160 (64);
    96x320 -> 48x160 (16-wide tiles) and 96x320 (32).  Masks bit-exact, H / update within the usual tolerances, fixed-iteration
    track against the oracle."""
    from dvo_amd import synth
    K = np.array(synth.K_640, np.float32).copy()
    K[0] *= w / 640.0; K[1] *= h / 480.0
    g, d, s, _ = synth.sequence(2, width=w, height_px=h, K=K, seed=5, sigma_value=0.5)
    g, d, s = g.numpy(), d.numpy(), s.numpy()
    ref = orc.OFrame(g[0], d[0], s[0], K, 2, 0)
    obj = orc.OFrame(g[1], d[1], s[1], K, 2, 0)
    rg = ref.gray(1); rg[h // 2:h // 2 + 3, w // 3:w // 3 + 9] = INV     # INVALID taps inside the image too
    xi = np.array([0.004, -0.003, 0.002, 0.003, -0.002, 0.004], np.float32)
    cfg = dvo.default_config(crop_enable=0, gn_pixels_per_thread=4)
    for l in range(2):
        _gn_compare(obj.gray(l), rg if l == 1 else ref.gray(l), ref.depth(l), ref.sigma(l), ref.K(l), xi, l, cfg=cfg, crop=False)
    cfg = dvo.default_config(crop_enable=0, gn_pixels_per_thread=4, fixed_iterations=3)
    xo, lo = orc.track(obj, ref, crop=False, fixed_iters=3)
    xg, lg = dvo.track(g[1], g[0], d[0], s[0], K, 2, 0, cfg=cfg)
    assert lg["n_iter"][:2] == [3, 3] == lo["n_iter"]
    np.testing.assert_allclose(xg, xo, rtol=0, atol=5e-5)


def test_levels_smaller_than_4x4_are_refused():
    """k_track_gn gathers a 4 x 4 neighbourhood around (1, 1) for lanes without an interior position, so a pyramid level below
    4 x 4 would read outside its image: make_geometry refuses such a pyramid with a status code (never a fault)."""
    K = np.array(K640, np.float32)
    with pytest.raises(dvo.DvoError):
        dvo.VisualOdometry(K, 24, 12)                      # mono geometry (3 levels, 2 culls): 6 x 3 -> 3 x 1 -> 1 x 0
    with pytest.raises(dvo.DvoError):
        dvo.Batch(2, K, 40, 24, levels=4, culls=1)         # 20 x 12 -> ... -> 2 x 1
    g = np.full((3, 9), 0.5, np.float32)
    with pytest.raises(dvo.DvoError):
        dvo.optimize(g, g, g + 1.0, g, K, np.zeros(6, np.float32), 0)
    bt = dvo.Batch(2, K, 64, 32, levels=4, culls=0)        # coarsest level 8 x 4: the smallest that is accepted
    bt.close()


def test_randomized_gn_step_sweep():
    """Fixed-seed sweep of Track::optimize (optimize.cpp:10-99) over what the targeted tests fix by hand: image sizes from the
    4 x 4 minimum up (so every tile geometry of k_track_gn -- 64 / 32 / 16-column 2-D tiles, raster tiles, partial last tiles),
    level indices (step size, the level-2 crop), pixels per thread / gather groups, sprinkled INVALID gray, zero / sub-gate / NaN
    depth, both sigma clamps, and poses from a fraction of a pixel to far outside the image.  Per case: contributing-pixel masks and
    counts bit-exact, H / g / update within the tolerances of tests/util.py."""
    rng = np.random.RandomState(20260410)
    sizes = [(4, 4), (5, 7), (16, 9), (33, 17), (64, 16), (64, 40), (96, 50), (128, 33), (48, 96), (161, 121), (200, 37), (31, 200)]
    variants = [(1, 1), (2, 2), (4, 1), (4, 2), (4, 4), (8, 2)]
    checked = contributing = 0
    for case in range(36):
        w, h = sizes[case % len(sizes)]
        level = int(rng.randint(0, 4))
        ppt, grp = variants[int(rng.randint(len(variants)))]
        crop = bool(rng.randint(2))
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        ph = rng.uniform(0, 6.28, size=4)
        tex = lambda sx, sy: (0.5 + 0.22 * np.sin(0.35 * (xx + sx) + ph[0]) * np.cos(0.27 * (yy + sy) + ph[1])
                              + 0.15 * np.sin(0.11 * (xx + sx) * 0.9 + 0.13 * (yy + sy) + ph[2])).astype(np.float32)
        rg, og = tex(0.0, 0.0), tex(float(rng.uniform(-0.8, 0.8)), float(rng.uniform(-0.8, 0.8)))
        og = (og + 0.01 * rng.standard_normal(og.shape)).astype(np.float32)
        rd = (1.2 + 0.4 * np.sin(0.05 * xx + ph[3]) + 0.3 * np.cos(0.07 * yy)).astype(np.float32)
        rs = rng.choice(np.array([0.003, 0.05, 0.1, 0.3, 0.8], np.float32), size=rd.shape).astype(np.float32)
        rg[rng.uniform(size=rg.shape) < 0.03] = INV
        og[rng.uniform(size=og.shape) < 0.03] = INV
        rd[rng.uniform(size=rd.shape) < 0.04] = 0.0
        rd[rng.uniform(size=rd.shape) < 0.02] = 0.15
        if case % 5 == 0:
            rd[rng.uniform(size=rd.shape) < 0.01] = np.nan
        if case % 7 == 3:
            rd[rng.uniform(size=rd.shape) < 0.02] = np.float32(3e30)    # beyond 2^100: the IEEE-division branch of recip_gated / recip_rn
        f = 0.9 * max(w, h)
        K = np.array([f, 0, w / 2.0 - 0.3, 0, f, h / 2.0 + 0.2, 0, 0, 1], np.float32)
        scale = [0.002, 0.01, 0.05, 0.4][case % 4]           # from sub-pixel motion to poses that throw most pixels out of the image
        xi = (scale * rng.standard_normal(6)).astype(np.float32)
        cfg = dvo.default_config(crop_enable=1 if crop else 0, gn_pixels_per_thread=ppt, gn_gather_group=grp)
        o = orc.optimize(og, rg, rd, rs, K, xi, level, crop=crop, want_mask=True)
        r = dvo.optimize(og, rg, rd, rs, K, xi, level, cfg=cfg, want_mask=True)
        tag = (case, w, h, level, ppt, grp, crop)
        np.testing.assert_array_equal(r["mask"], o["mask"], err_msg=str(tag))
        assert r["n_valid"] == o["n_valid"], tag
        checked += 1
        if o["n_valid"] == 0:
            assert r["residual"] == np.float32(-1) and not r["xi_update"].any(), tag
            continue
        contributing += 1
        np.testing.assert_allclose(r["H"], o["H"], rtol=0, atol=TOL_H_REL * np.abs(o["H"]).max(), err_msg=str(tag))
        np.testing.assert_allclose(r["g"], o["g"], rtol=0, atol=TOL_H_REL * max(np.abs(o["g"]).max(), 1e-30), err_msg=str(tag))
        np.testing.assert_allclose(r["sum_r2"], o["sum_r2"], rtol=2e-5, err_msg=str(tag))
        # the update against the oracle's normal equations, as a backward error: small images give badly conditioned (or singular:
        # fewer than six pixels) systems, where a forward comparison of the two solutions says nothing
        H = orc.upper_to_full(o["H"]); x = r["xi_update"].astype(np.float64)
        back = np.abs(H @ x - o["g"]).max() / max((np.abs(H) @ np.abs(x) + np.abs(o["g"])).max(), 1e-300)
        assert back <= 2e-5, (tag, back)
    assert checked == 36 and contributing >= 20


def test_randomized_propagate_and_regularize_sweep():
    """Fixed-seed sweep of Implement::propagate (implement.cpp:217-256) and Implement::regularize (implement.cpp:156-180) over
    ragged sizes from 4 x 4 up, with INVALID / zero / huge depth, both fusion outcomes, ages 0..9 and motions from none to strong
    forward motion (scatter collisions): depth, sigma and age bit-exact."""
    rng = np.random.RandomState(777)
    sizes = [(4, 4), (7, 5), (16, 16), (33, 9), (40, 30), (64, 17), (97, 61), (160, 120), (161, 3 * 41), (255, 8)]
    for case in range(30):
        w, h = sizes[case % len(sizes)]
        depth = rng.normal(1.5, 0.5, (h, w)).astype(np.float32)
        sigma = (np.abs(rng.normal(0.3, 0.2, (h, w))) + 0.01).astype(np.float32)
        depth[rng.uniform(size=depth.shape) < 0.05] = INV
        depth[rng.uniform(size=depth.shape) < 0.05] = 0.0
        depth[rng.uniform(size=depth.shape) < 0.02] = 9.0
        sigma[rng.uniform(size=sigma.shape) < 0.03] = INV
        if case % 6 == 0:
            sigma[:] = 0.5                                                  # uniform map: every fusion is accepted or every one rejected
        np.testing.assert_array_equal(dvo.Implement.regularize(depth, sigma), orc.regularize(depth, sigma), err_msg="regularize %d %dx%d" % (case, w, h))
        age = rng.randint(0, 10, depth.shape).astype(np.float32)
        f = 0.9 * max(w, h)
        K = np.array([f, 0, w / 2.0 + 0.1, 0, f, h / 2.0 - 0.2, 0, 0, 1], np.float32)
        xi = ([0.0, 0.0, 0.0, 0.0, 0.0, 0.0] if case % 5 == 0 else
              (np.array([0.02, 0.02, 0.25, 0.03, 0.03, 0.05]) * rng.standard_normal(6)).tolist())
        xi = np.array(xi, np.float32)
        dpos = np.where(depth > 0, depth, 1.0).astype(np.float32) if case % 3 == 0 else depth   # (with and without unusable source pixels)
        got = dvo.Implement.propagate(dpos, sigma, age, xi, K)
        exp = orc.propagate(dpos, sigma, age, xi, K)
        for name, a, b in zip(("depth", "sigma", "age"), got, exp):
            np.testing.assert_array_equal(a, b, err_msg="propagate %s %d %dx%d" % (name, case, w, h))


@pytest.mark.parametrize("seed,w,h,gain,sig,young", [(1, 640, 480, 20.0, 0.3, 0.5), (2, 640, 480, 20.0, 0.2, 0.9), (5, 640, 480, 40.0, 0.5, 0.0),
                                                     (6, 640, 480, 30.0, 0.4, 0.3), (9, 648, 488, 25.0, 0.3, 0.5)])
def test_mapper_update_sweep(seed, w, h, gain, sig, young):
    """Mapper::update / Implement::update (mapper.cpp:76-137, implement.cpp:23-152,182-214) on three keyframes over baselines from
    short (few search steps) to long (the 102-step cap), a map size that is no multiple of the workgroup's pixel count, prior sigmas
    from 0.2 to the 0.5 clamp and age maps from 'all born in the newest keyframe' to 'all born in the oldest': depth, sigma, age and
    the valid-update count bit-exact."""
    K = np.array(K640, np.float32).copy()
    K[0] *= w / 640.0; K[1] *= h / 480.0
    from dvo_amd import synth
    g, d, s, poses = synth.sequence(6, width=w, height_px=h, K=K, seed=30 + seed, sigma_value=0.5)
    g, d = g.numpy(), d.numpy()
    rng = np.random.RandomState(100 + seed)
    kfs = [orc.OFrame(g[i], d[i], np.full_like(d[i], 0.5), K, 3, 2, id=i) for i in (0, 1, 3)]
    obj = orc.OFrame(g[4], None, None, K, 3, 2, id=4)
    def twist(a, b):
        return orc.se3_log((np.linalg.inv(poses[b]) @ poses[a]).astype(np.float32)) * np.float32(gain)
    x1 = twist(0, 1); x3 = orc.se3_concatenate(x1, twist(1, 3)); rel = twist(3, 4)
    kfs[1].set_pose(x1, x1); kfs[2].set_pose(x3, twist(1, 3)); obj.set_pose(orc.se3_concatenate(x3, rel), rel)
    ref = kfs[2]
    top_d = (ref.depth(2) + rng.normal(0, 0.05, ref.depth(2).shape)).astype(np.float32)
    top_d[rng.uniform(size=top_d.shape) < 0.03] = 0.05                      # below every gate
    top_s = np.full_like(top_d, sig)
    u = rng.uniform(size=top_d.shape)
    age = np.where(u < young, 0.0, np.where(u < young + (1 - young) / 2, 1.0, 2.0)).astype(np.float32)
    ref.update_depth_sigma(top_d, top_s)
    ref.set_age(age)
    got_d, got_s, got_a, got_v = dvo.mapper_update([k.gray(2) for k in kfs], [k.xi for k in kfs], obj.gray(2), obj.xi, obj.rel_xi, 4,
                                                   ref.K(2), top_d, top_s, age, cfg=dvo.default_config(rng_seed=seed))
    v = orc.mapper_update(kfs, obj, seed)
    np.testing.assert_array_equal(got_a, ref.age())
    np.testing.assert_array_equal(got_d, ref.depth(2))
    np.testing.assert_array_equal(got_s, ref.sigma(2))
    assert got_v == v
    assert (got_d != top_d).sum() > 100, ((got_d != top_d).sum(), got_v)   # the case really exercised the update (oracle: 222 .. 2376 pixels)


def test_single_handle_schedules_give_the_same_bits(monkeypatch):
    """A dvo_vo handle's sensor-depth tracking on its three launch schedules -- one launch per call (k_track_persist, the default), one
    launch per iteration (k_track_gn_fused) and launch pairs -- and on the persistent kernel's give-up path (polling limit 0: every
    launch gives up at once and the handle re-runs the frame launch by launch): poses and per-iteration logs bit for bit."""
    g, d, s, _ = frames(4, seed=42, sigma=0.1)
    runs = {}
    for name, (sl, limit) in {"persist": (0, None), "per_iteration": (1, None), "pairs": (-1, None), "persist_gives_up": (0, "0")}.items():
        if limit is None:
            monkeypatch.delenv("DVO_PERSIST_SPIN_LIMIT", raising=False)
        else:
            monkeypatch.setenv("DVO_PERSIST_SPIN_LIMIT", limit)
        vo = dvo.VisualOdometry(K640, 640, 480, cfg=dvo.default_config(track_single_launch=sl, gn_pixels_per_thread=4))
        out = []
        for i in range(4):
            T = vo.odometrizeUsingDepth(g[i], d[i], s[i])
            lg = vo.lastTrackLog() if i > 0 else None
            out.append((T, lg))
        vo.close()
        runs[name] = out
    monkeypatch.delenv("DVO_PERSIST_SPIN_LIMIT", raising=False)
    ref = runs["pairs"]
    assert sum(ref[3][1]["n_iter"]) > 8
    for name in ("persist", "per_iteration", "persist_gives_up"):
        for i in range(4):
            np.testing.assert_array_equal(runs[name][i][0].view(np.uint32), ref[i][0].view(np.uint32), err_msg="%s frame %d" % (name, i))
            if i > 0:
                a, b = runs[name][i][1], ref[i][1]
                assert a["n_iter"] == b["n_iter"], (name, i)
                for l in range(4):
                    np.testing.assert_array_equal(a["xi_after"][l], b["xi_after"][l])
                    np.testing.assert_array_equal(a["xi_update"][l], b["xi_update"][l])
                    np.testing.assert_array_equal(a["n_valid"][l], b["n_valid"][l])
                    np.testing.assert_array_equal(a["residual"][l], b["residual"][l])


def test_mono_handle_schedules_give_the_same_bits(monkeypatch):
    """A dvo_vo handle's mono loop (odometrize: track + Mapper::estimate + regularize, system.hpp:44-74) on its schedules: the default
    (k_track_persist with the keyframe decision as its tail, pose and decision through mapped host memory), the persistent kernel's
    give-up path (polling limit 0), one launch per iteration + k_mono_decide, and launch pairs.  World poses, keyframe flags,
    valid-update counts, iteration counts and the newest keyframe's depth / sigma / age maps bit for bit on every frame."""
    g, d, _, _ = frames(12, seed=42, sigma=0.1)
    d0 = d[0][::4, ::4].copy()
    runs = {}
    for name, (sl, limit) in {"persist": (0, None), "persist_gives_up": (0, "0"), "per_iteration": (1, None), "pairs": (-1, None)}.items():
        if limit is None:
            monkeypatch.delenv("DVO_PERSIST_SPIN_LIMIT", raising=False)
        else:
            monkeypatch.setenv("DVO_PERSIST_SPIN_LIMIT", limit)
        vo = dvo.VisualOdometry(K640, 640, 480, cfg=dvo.default_config(track_single_launch=sl, rng_seed=3))
        vo.setInitialDepth(d0, np.full_like(d0, 0.5))
        out = []
        for i in range(12):
            T, key = vo.odometrize(g[i])
            kf = vo.keyframe(vo.keyframeCount() - 1)
            out.append((T.copy(), bool(key), vo.lastValidUpdates(), vo.lastTrackLog()["n_iter"] if i > 0 else None,
                        kf["depth"].copy(), kf["sigma"].copy(), kf["age"].copy(), vo.keyframeCount()))
        vo.close()
        runs[name] = out
    monkeypatch.delenv("DVO_PERSIST_SPIN_LIMIT", raising=False)
    ref = runs["pairs"]
    assert sum(r[1] for r in ref) >= 2 and sum(not r[1] for r in ref) >= 3   # both branches of Mapper::estimate ran (the update counts on real frames: test_real_data.py)
    for name in ("persist", "persist_gives_up", "per_iteration"):
        for i in range(12):
            a, b = runs[name][i], ref[i]
            np.testing.assert_array_equal(a[0].view(np.uint32), b[0].view(np.uint32), err_msg="%s frame %d" % (name, i))
            assert a[1] == b[1] and a[2] == b[2] and a[3] == b[3] and a[7] == b[7], (name, i, a[1:4], b[1:4])
            for k in (4, 5, 6):
                np.testing.assert_array_equal(a[k].view(np.uint32), b[k].view(np.uint32), err_msg="%s frame %d map %d" % (name, i, k))


def test_host_frame_staging_paths_give_the_same_bits(monkeypatch):
    """A mono frame and a raw sensor-depth frame reach the pyramid kernel either through a pinned, device-mapped staging block the
    caller's thread fills (default) or through the runtime's copies (DVO_MONO_STAGE=0 / DVO_RAW_STAGE=0): same poses, same maps."""
    g, d, _, _ = frames(6, seed=42, sigma=0.1)
    g8 = [np.clip(np.rint(x * 255), 0, 255).astype(np.uint8) for x in g]
    d16 = [np.clip(np.rint(x * 5000), 0, 65535).astype(np.uint16) for x in d]
    d0 = d[0][::4, ::4].copy()
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DVO_MONO_STAGE", mode)
        monkeypatch.setenv("DVO_RAW_STAGE", mode)
        out = []
        vo = dvo.VisualOdometry(K640, 640, 480, cfg=dvo.default_config(rng_seed=3))
        vo.setInitialDepth(d0, np.full_like(d0, 0.5))
        for i in range(6):
            T, key = vo.odometrize(g[i]) if i % 2 == 0 else vo.odometrizeRaw(g8[i])
            out.append(T.copy())
        out.append(vo.keyframe(vo.keyframeCount() - 1)["depth"].copy())
        vo.close()
        vo = dvo.VisualOdometry(K640, 640, 480)
        for i in range(4):
            out.append(vo.odometrizeUsingDepthRaw(g8[i], d16[i]).copy())
        vo.close()
        res[mode] = out
    monkeypatch.delenv("DVO_MONO_STAGE", raising=False)
    monkeypatch.delenv("DVO_RAW_STAGE", raising=False)
    for a, b in zip(res["1"], res["0"]):
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))


def _holey(g, d, s):
    """Frames whose border band and interior hold what the generic sampler exists for: INVALID gray blocks, black (0.0) pixels next to
    them (the fill quirk of getSubpixel: `last > 0` is false), depth holes -- along all four image borders and inside."""
    rng = np.random.RandomState(11)
    out = []
    for i in range(4):
        gg, dd = g[i].copy(), d[i].copy()
        for _ in range(60):
            y, x = rng.randint(0, 470), rng.randint(0, 630)
            gg[y:y + rng.randint(1, 9), x:x + rng.randint(1, 9)] = INV
        for _ in range(40):
            y, x = rng.randint(0, 478), rng.randint(0, 638)
            gg[y:y + 2, x:x + 2] = 0.0
        gg[0:3, ::7] = INV; gg[-3:, ::5] = 0.0; gg[::9, 0:3] = INV; gg[::11, -3:] = 0.0
        for _ in range(30):
            y, x = rng.randint(0, 470), rng.randint(0, 630)
            dd[y:y + rng.randint(1, 9), x:x + rng.randint(1, 9)] = 0.0
        out.append((gg, dd, s[i]))
    return out


def test_single_handle_one_launch_schedule_edge_cases():
    """k_track_persist (one launch per odometrizeUsingDepth call) where its control flow is unusual: a fixed iteration count (the
    level never 'stops' by a threshold), a reference without any usable depth (every step has zero contributing pixels: residual -1,
    zero update, optimize.cpp:92-93 -> each level leaves after one iteration and the pose stays the identity), and a reference whose
    depth makes every pixel project outside the image.  Each against the launch-pair schedule, bit for bit; nothing may hang."""
    g, d, s, _ = frames(4, seed=42, sigma=0.1)
    cases = {
        "fixed_iterations": (dict(fixed_iterations=3, crop_enable=0), [(g[i], d[i], s[i]) for i in range(3)]),
        "no_depth": (dict(), [(g[0], np.zeros_like(d[0]), s[0]), (g[1], np.zeros_like(d[1]), s[1]), (g[2], d[2], s[2]), (g[3], d[3], s[3])]),
        "far_outside": (dict(), [(g[0], np.full_like(d[0], 1e-3 + 0.2), s[0]), (g[1], d[1], s[1]), (g[2], d[2], s[2])]),
        "border_and_invalid_taps": (dict(), _holey(g, d, s)),
    }
    for name, (kw, seq) in cases.items():
        res = {}
        for sched in (0, -1):
            vo = dvo.VisualOdometry(K640, 640, 480, cfg=dvo.default_config(track_single_launch=sched, gn_pixels_per_thread=4, **kw))
            out = []
            for (gg, dd, ss) in seq:
                T = vo.odometrizeUsingDepth(gg, dd, ss)
                out.append((T.copy(), vo.lastTrackLog()["n_iter"] if len(out) else None))
            vo.close()
            res[sched] = out
        for i in range(len(seq)):
            np.testing.assert_array_equal(res[0][i][0].view(np.uint32), res[-1][i][0].view(np.uint32), err_msg="%s frame %d" % (name, i))
            assert res[0][i][1] == res[-1][i][1], (name, i)
        if name == "fixed_iterations":
            assert res[0][2][1] == [3, 3, 3, 3]
        if name == "no_depth":
            assert res[0][1][1] == [1, 1, 1, 1] and np.array_equal(res[0][1][0], np.eye(4, dtype=np.float32))
