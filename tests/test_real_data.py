"""Parity on the reference's own images (data/logicool0, data/KINECT_50MM): the only externally authored inputs of this path.
Fixtures: tests/golden/{logicool0,kinect50mm}_excerpt.npz (made by tests/golden/make_real_fixtures.py where /root/reference exists;
nothing here reads /root/reference).  Expected outputs are ORACLE-DERIVED -- parity unpinned, see DESIGN.md §4."""
import os

import numpy as np
import pytest

import orc
from real_data import K_LOGICOOL, frames_from_fixture, ingest_np, undistort_nearest_np, bgr2gray_u8

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _logicool():
    fx = dict(np.load(os.path.join(GOLD, "logicool0_excerpt.npz")))
    return fx, frames_from_fixture(fx)


def test_fixture_frames_have_the_properties_the_test_is_for():
    fx, frames = _logicool()
    assert frames.shape == (20, 480, 640)
    inv = frames <= -2.0
    assert 0.001 < inv.mean() < 0.02                 # the INVALID undistortion border (loader.cpp:39) is present ...
    assert inv[:, 0, 0].all() or inv[:, -1, -1].all() or inv[:, 0, -1].all() or inv[:, -1, 0].all()   # ... in a corner
    g8 = fx["gray_u8"]
    assert (g8 == 0).sum() > 500 and (g8 == 255).sum() > 0   # black webcam pixels (the `last > 0` quirk of convert.cpp:155-173) and saturated ones
    assert fx["key"].sum() >= 5 and fx["age"].max() >= 3 and fx["valid_updates"].max() > 50   # propagate AND update are exercised


def test_oracle_reproduces_the_real_image_fixture():
    """Pins the oracle's mono pipeline (track + propagate / update / regularize) against regressions on real webcam frames."""
    fx, frames = _logicool()
    vo = orc.OVO(K_LOGICOOL, 640, 480, seed=int(fx["seed_vo"]))
    vo.set_initial_depth(fx["init_depth"], np.full_like(fx["init_depth"], 0.5))
    mf = list(fx["map_frames"])
    for i, g in enumerate(frames):
        T, key = vo.odometrize(g)
        assert key == bool(fx["key"][i])
        np.testing.assert_array_equal(T, fx["T_world"][i])
        kf = vo.keyframe(vo.keyframe_count() - 1)
        np.testing.assert_array_equal(kf.age().astype(np.uint8), fx["age"][i])
        assert vo.last_valid_updates() == fx["valid_updates"][i]
        if i in mf:
            np.testing.assert_array_equal(kf.depth(2), fx["depth"][mf.index(i)])
            np.testing.assert_array_equal(kf.sigma(2), fx["sigma"][mf.index(i)])


@pytest.mark.gpu
def test_mono_pipeline_on_logicool0_matches_the_oracle():
    """System::VisualOdometry::odometrize (system.hpp:44-74) over 20 real frames on the GPU: keyframe decisions and age maps
    exact, world poses within 1e-4 (m, rad), depth maps of the newest keyframe within 1e-3 on > 99 % of the pixels."""
    import dvo_amd as dvo
    fx, frames = _logicool()
    vo = dvo.VisualOdometry(K_LOGICOOL, 640, 480, cfg=dvo.default_config(rng_seed=int(fx["seed_vo"])))
    vo.setInitialDepth(fx["init_depth"], np.full_like(fx["init_depth"], 0.5))
    mf = list(fx["map_frames"])
    worst_pose, worst_depth, age_mismatch = 0.0, 0.0, 0
    for i, g in enumerate(frames):
        T, key = vo.odometrize(g)
        assert key == bool(fx["key"][i]), "keyframe decision differs at frame %d" % i
        assert vo.keyframeCount() == fx["n_keyframes"][i]
        worst_pose = max(worst_pose, float(np.abs(T - fx["T_world"][i]).max()))
        kf = vo.keyframe(vo.keyframeCount() - 1)
        age_mismatch += int((kf["age"].astype(np.uint8) != fx["age"][i]).sum())
        if i in mf:
            bad = np.abs(kf["depth"] - fx["depth"][mf.index(i)]) > 1e-3
            worst_depth = max(worst_depth, float(bad.mean()))
    vo.close()
    assert worst_pose <= 1e-4, worst_pose
    assert age_mismatch == 0, age_mismatch
    assert worst_depth < 0.01, worst_depth


@pytest.mark.gpu
def test_kinect_frame_through_k_ingest_is_bit_exact():
    """One data/KINECT_50MM frame pair (u16 depth with real holes + BGRA colour) through the device conversion (loader.cpp:137-147,
    transform.cpp:75): bit-exact against the numpy restatement of that integer arithmetic."""
    import dvo_amd as dvo
    fx = np.load(os.path.join(GOLD, "kinect50mm_excerpt.npz"))
    d16, rgba = fx["depth16"], fx["rgba"]
    assert 0.01 < (d16 == 0).mean() < 0.5 and d16.max() > 5000
    g, d, s = dvo.ingest(rgba, d16)
    eg, ed, es = ingest_np(rgba, d16)
    np.testing.assert_array_equal(g, eg)
    np.testing.assert_array_equal(d, ed)
    np.testing.assert_array_equal(s, es)


@pytest.mark.gpu
def test_undistort_kernel_on_a_real_frame():
    """k_undistort with the loader's constants (loader.cpp:17-25) on a real frame's gray values: the INVALID border is the same set
    of pixels and >= 99.9 % of the remapped values equal the numpy restatement (ties of rint at .5 may differ)."""
    import dvo_amd as dvo
    fx, frames = _logicool()
    src = np.ascontiguousarray(frames[3])
    src[src <= -2.0] = 0.5
    from real_data import D_LOGICOOL
    got = dvo.undistort(src, K_LOGICOOL, D_LOGICOOL)
    exp, inv = undistort_nearest_np(src, K_LOGICOOL, D_LOGICOOL)
    exp = exp.copy(); exp[inv] = np.float32(-2.0)
    assert ((got <= -2.0) != inv).mean() < 1e-3
    assert (got == exp).mean() > 0.999
