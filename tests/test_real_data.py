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
def test_mono_pipeline_on_logicool0_frame_by_frame_given_the_oracle_state(tmp_path):
    """System::VisualOdometry::odometrize (system.hpp:44-74) on 19 real frames, each processed from the ORACLE's FrameHistory
    (loaded through dvo_vo_load), so both sides see identical inputs.  Asserted: keyframe decision and iteration counts exact on
    every frame; world pose within 1e-5 (median) / 5e-4 (worst frame); the newest keyframe's age map differs on < 0.1 % (median) /
    < 2 % (worst) and its depth by > 1e-3 on < 0.5 % (median) / < 5 % (worst) of the pixels.
    Why a worst-frame allowance: every single Gauss-Newton step agrees with the oracle to ~5e-8 on these images
    (test_gn_steps_on_logicool0_match_the_oracle_at_every_iteration), but the reference's 40 x 30 level never converges on them (15
    iterations, every frame) and its iteration EXPANDS differences -- measured 2.8e-7 -> 4.0e-4 over the 15 level-0 iterations of
    frame 17 (tools/diag_real_tf.py, profiles/r02_real_data_divergence.txt); the two finer levels stop after 1-5 iterations
    (residual < 5e-3, tracker.cpp:68-69) and do not contract it away.  A 1e-4 pose difference then flips cvRound / stereo-match
    decisions in the mapping (implement.cpp:230, 106-152).  Free-running, that feedback makes GPU and oracle trajectories part after
    ~5 frames (a different keyframe decision at frame 13): not asserted.  Given the SAME pose, propagate / update / regularize are
    bit-exact on these frames (test_mapping_stages_on_logicool0_are_bit_exact_given_the_oracle_poses)."""
    import dvo_amd as dvo
    from real_data import write_keyframe_store
    fx, frames = _logicool()
    seed = int(fx["seed_vo"])
    ovo = orc.OVO(K_LOGICOOL, 640, 480, seed=seed)
    ovo.set_initial_depth(fx["init_depth"], np.full_like(fx["init_depth"], 0.5))
    ovo.odometrize(frames[0])
    vo = dvo.VisualOdometry(K_LOGICOOL, 640, 480, cfg=dvo.default_config(rng_seed=seed))
    path = str(tmp_path / "oracle_state.dvokf")
    pose, age, depth = [], [], []
    n_key = n_upd = 0
    for i in range(1, len(frames)):
        kfs = [ovo.keyframe(k) for k in range(ovo.keyframe_count())]
        write_keyframe_store(path, K_LOGICOOL, 640, 480, kfs, latest_id=i - 1)
        vo.load(path)
        _, lo = orc.track(orc.OFrame(frames[i], None, None, K_LOGICOOL, 3, 2), kfs[-1])
        T, key = vo.odometrize(frames[i])
        To, keyo = ovo.odometrize(frames[i])
        assert key == keyo, "keyframe decision differs at frame %d" % i
        assert vo.keyframeCount() == ovo.keyframe_count()
        assert vo.lastTrackLog()["n_iter"] == lo["n_iter"], i
        pose.append(float(np.abs(T - To).max()))
        kf = vo.keyframe(vo.keyframeCount() - 1)
        okf = ovo.keyframe(ovo.keyframe_count() - 1)
        age.append(float((kf["age"] != okf.age()).mean()))
        depth.append(float((np.abs(kf["depth"] - okf.depth(2)) > 1e-3).mean()))
        np.testing.assert_array_equal(kf["gray"], okf.gray(2))
        n_key += int(key); n_upd += int(not key and vo.lastValidUpdates() > 0)
    vo.close()
    print("teacher-forced |T - T_oracle|:", ["%.1e" % v for v in pose], "\nage mismatch:", ["%.4f" % v for v in age], "\ndepth mismatch:", ["%.4f" % v for v in depth])
    assert n_key >= 5 and n_upd >= 5          # both branches of Mapper::estimate (mapper.cpp:16-33) ran on real data
    assert np.median(pose) <= 1e-5 and max(pose) <= 5e-4, pose
    assert np.median(age) < 0.001 and max(age) < 0.02, age
    assert np.median(depth) < 0.005 and max(depth) < 0.05, depth


@pytest.mark.gpu
def test_gn_steps_on_logicool0_match_the_oracle_at_every_iteration():
    """Track::optimize (optimize.cpp:10-99) on the real frames, at every input pose the oracle's tracker visits: contributing-pixel
    masks bit-exact, H within 1e-6 max|H|, and the update solves the oracle's normal equations to a backward error <= 2e-6 (a
    conditioning-independent statement; measured 5e-8).  INVALID undistortion border, black pixels (`last > 0` quirk), noisy
    semi-dense depth, all three levels, crop window on the top level."""
    import dvo_amd as dvo
    fx, frames = _logicool()
    seed = int(fx["seed_vo"])
    ovo = orc.OVO(K_LOGICOOL, 640, 480, seed=seed)
    ovo.set_initial_depth(fx["init_depth"], np.full_like(fx["init_depth"], 0.5))
    ovo.odometrize(frames[0])
    n_steps = 0
    for i in range(1, len(frames)):
        ref = ovo.keyframe(ovo.keyframe_count() - 1)
        obj = orc.OFrame(frames[i], None, None, K_LOGICOOL, 3, 2)
        _, lo = orc.track(obj, ref)
        xi = np.zeros(6, np.float32)
        for l in range(3):
            its = list(range(lo["n_iter"][l]))
            for it in (its if i % 4 == 1 else its[:2] + its[-1:]):      # every iteration on every 4th frame, first two + last elsewhere
                xi_in = xi if it == 0 else lo["xi_after"][l][it - 1]
                r = dvo.optimize(obj.gray(l), ref.gray(l), ref.depth(l), ref.sigma(l), ref.K(l), xi_in, l, want_mask=True)
                o = orc.optimize(obj.gray(l), ref.gray(l), ref.depth(l), ref.sigma(l), ref.K(l), xi_in, l, want_mask=True)
                np.testing.assert_array_equal(r["mask"], o["mask"])
                assert r["n_valid"] == o["n_valid"] > 0
                np.testing.assert_allclose(r["H"], o["H"], rtol=0, atol=1e-6 * np.abs(o["H"]).max())
                H = orc.upper_to_full(o["H"]); x = r["xi_update"].astype(np.float64)
                back = np.abs(H @ x - o["g"]).max() / (np.abs(H) @ np.abs(x) + np.abs(o["g"])).max()
                assert back <= 2e-6, (i, l, it, back)
                n_steps += 1
            xi = lo["xi_after"][l][lo["n_iter"][l] - 1]
        ovo.odometrize(frames[i])
    assert n_steps > 150


@pytest.mark.gpu
def test_mapping_stages_on_logicool0_are_bit_exact_given_the_oracle_poses():
    """Implement::propagate, Mapper::update and Implement::regularize on the real frames' maps with the ORACLE's poses as inputs:
    depth, sigma, age and the valid-update count are bit-exact (INVALID undistortion border, black pixels, real keyframe ages)."""
    import dvo_amd as dvo
    fx, frames = _logicool()
    seed = int(fx["seed_vo"])
    ovo = orc.OVO(K_LOGICOOL, 640, 480, seed=seed)
    ovo.set_initial_depth(fx["init_depth"], np.full_like(fx["init_depth"], 0.5))
    ovo.odometrize(frames[0])
    n_prop = n_upd = 0
    for i in range(1, len(frames)):
        n_before = ovo.keyframe_count()
        hist = [ovo.keyframe(k) for k in range(n_before)]
        ref = hist[-1]
        K2 = ref.K(2)
        before = dict(depth=ref.depth(2), sigma=ref.sigma(2), age=ref.age(), grays=[h.gray(2) for h in hist], xis=[h.xi for h in hist])
        To, keyo = ovo.odometrize(frames[i])
        if keyo:      # the new keyframe's maps = regularize(propagate(ref maps, rel_xi)) (mapper.cpp:62-74, 139-144)
            new = ovo.keyframe(ovo.keyframe_count() - 1)
            d, s, a = dvo.Implement.propagate(before["depth"], before["sigma"], before["age"], new.rel_xi, K2)
            np.testing.assert_array_equal(a, new.age())
            np.testing.assert_array_equal(s, new.sigma(2))
            np.testing.assert_array_equal(dvo.Implement.regularize(d, s), new.depth(2))
            n_prop += 1
        else:         # ref maps updated in place by Mapper::update, then regularized
            obj = ovo.last_frame()
            d, s, a, v = dvo.mapper_update(before["grays"], before["xis"], obj.gray(2), obj.xi, obj.rel_xi, i, K2,
                                           before["depth"], before["sigma"], before["age"], cfg=dvo.default_config(rng_seed=seed))
            assert v == ovo.last_valid_updates()
            np.testing.assert_array_equal(a, ref.age())
            np.testing.assert_array_equal(s, ref.sigma(2))
            np.testing.assert_array_equal(dvo.Implement.regularize(d, s), ref.depth(2))
            n_upd += 1
    assert n_prop >= 5 and n_upd >= 5


@pytest.mark.gpu
def test_mono_pipeline_on_logicool0_free_running():
    """The same 20 frames free-running (no teacher forcing): the first tracked frame agrees with the fixture to 1e-6, every pose
    stays finite, the keyframe cadence is comparable; frame-by-frame divergence is reported, not asserted (see the test above)."""
    import dvo_amd as dvo
    fx, frames = _logicool()
    vo = dvo.VisualOdometry(K_LOGICOOL, 640, 480, cfg=dvo.default_config(rng_seed=int(fx["seed_vo"])))
    vo.setInitialDepth(fx["init_depth"], np.full_like(fx["init_depth"], 0.5))
    keys, diffs = [], []
    for i, g in enumerate(frames):
        T, key = vo.odometrize(g)
        assert np.isfinite(T).all()
        keys.append(key); diffs.append(float(np.abs(T - fx["T_world"][i]).max()))
    vo.close()
    assert keys[0] and keys[1] == bool(fx["key"][1]) and diffs[1] <= 1e-6
    assert abs(sum(keys) - int(fx["key"].sum())) <= 3
    print("free-running |T - T_oracle| per frame:", ["%.1e" % d for d in diffs])


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


# ---------------------------------------------------------------- the sensor-depth path on real Kinect frames
K_KINECT_IR = np.array([[365.0, 0, 256.0], [0, 365.0, 212.0], [0, 0, 1]], np.float32)   # nominal Kinect v2 depth / IR intrinsics (512 x 424)


KINECT_SETS = ["kinect50mm_ir_depth.npz", "kinect1deg_ir_depth.npz"]   # ~50 mm translation per frame; ~1 degree rotation per frame


def _kinect_sequence(name="kinect50mm_ir_depth.npz"):
    """Four (IR, depth) pairs of data/KINECT_50MM (or KINECT_1DEG) as raw sensor frames at 512 x 424 (the stored 256 x 212 excerpt, every
    pixel repeated 2 x 2, decimates back to exactly the stored pixels in Frame(g, d, s, K, 4, 1)) and as the float maps the loader makes."""
    fx = np.load(os.path.join(GOLD, name))
    g8 = np.repeat(np.repeat(fx["gray_u8"], 2, axis=1), 2, axis=2)
    d16 = np.repeat(np.repeat(fx["depth16"], 2, axis=1), 2, axis=2)
    fl = [ingest_np(g8[i], d16[i]) for i in range(g8.shape[0])]
    return g8, d16, fl


@pytest.mark.parametrize("name", KINECT_SETS)
def test_kinect_sequence_fixture_has_real_holes(name):
    g8, d16, fl = _kinect_sequence(name)
    assert g8.shape == (4, 424, 512) and 0.03 < (d16 == 0).mean() < 0.2
    g, d, s = fl[0]
    assert (g[d16[0] == 0] == -2.0).all() and (s[d16[0] == 0] == 1.0).all() and (s[d16[0] > 0] == np.float32(0.1)).all()
    ref = orc.OFrame(g, d, s, K_KINECT_IR, 4, 1)
    obj = orc.OFrame(*fl[1], K_KINECT_IR, 4, 1)
    xi, log = orc.track(obj, ref)
    assert np.isfinite(xi).all() and sum(log["n_iter"]) >= 4


@pytest.mark.gpu
@pytest.mark.parametrize("name", KINECT_SETS)
def test_sensor_depth_tracking_on_kinect_frames_matches_the_oracle_at_every_iteration(name):
    """Tracker::track (tracker.cpp:22-85) as odometrizeUsingDepth runs it (Frame(g,d,s,K,4,1), sigma 0.1 / 1.0, INVALID gray in the
    depth holes: transform.cpp:60-76) on real Kinect IR + depth frames with 6 % holes.  Along the GPU's own trajectory every
    iteration is checked against the oracle at the same input pose: contributing-pixel count exact, residual 1e-4, and the
    update solves the oracle's normal equations to a backward error <= 2e-6 (also where the diverged pose leaves a handful of pixels and a singular system); first-iteration masks bit-exact on every level;
    the raw (u8 + u16) entry point equals the float one bit for bit."""
    import dvo_amd as dvo
    g8, d16, fl = _kinect_sequence(name)
    cfg = dvo.default_config(gn_pixels_per_thread=4)
    vo_raw = dvo.VisualOdometry(K_KINECT_IR, 512, 424, cfg=cfg)
    vo_flt = dvo.VisualOdometry(K_KINECT_IR, 512, 424, cfg=cfg)
    n_checked = 0
    for i in range(4):
        Ta = vo_raw.odometrizeUsingDepthRaw(g8[i], d16[i])
        Tb = vo_flt.odometrizeUsingDepth(*fl[i])
        np.testing.assert_array_equal(Ta, Tb)
        if i == 0:
            continue
        lg = vo_flt.lastTrackLog()
        ref = orc.OFrame(*fl[i - 1], K_KINECT_IR, 4, 1)
        obj = orc.OFrame(*fl[i], K_KINECT_IR, 4, 1)
        xi = np.zeros(6, np.float32)
        for l in range(4):
            r = dvo.optimize(obj.gray(l), ref.gray(l), ref.depth(l), ref.sigma(l), ref.K(l), xi, l, cfg=cfg, want_mask=True)
            o = orc.optimize(obj.gray(l), ref.gray(l), ref.depth(l), ref.sigma(l), ref.K(l), xi, l, want_mask=True)
            np.testing.assert_array_equal(r["mask"], o["mask"])
            for it in range(lg["n_iter"][l]):
                o = orc.optimize(obj.gray(l), ref.gray(l), ref.depth(l), ref.sigma(l), ref.K(l), xi, l)
                assert o["n_valid"] == lg["n_valid"][l][it], (i, l, it)
                if o["n_valid"] == 0:     # the over-relaxed iteration (sigma 0.1: gain 10) threw the pose so far that nothing projects into
                    assert lg["residual"][l][it] == np.float32(-1)           # the image: residual -1, zero update (optimize.cpp:92-93)
                    np.testing.assert_array_equal(lg["xi_after"][l][it], xi)
                    n_checked += 1
                    continue
                np.testing.assert_allclose(lg["residual"][l][it], o["residual"], rtol=1e-4)
                r = dvo.optimize(obj.gray(l), ref.gray(l), ref.depth(l), ref.sigma(l), ref.K(l), xi, l, cfg=cfg)   # the same step as an operator call
                assert r["n_valid"] == o["n_valid"]
                upd = r["xi_update"].astype(np.float64)
                H = orc.upper_to_full(o["H"])
                back = np.abs(H @ upd - o["g"]).max() / (np.abs(H) @ np.abs(upd) + np.abs(o["g"])).max()
                assert back <= 2e-6, (i, l, it, back)
                np.testing.assert_allclose(lg["xi_after"][l][it], r["xi_next"], rtol=0, atol=1e-6 * max(1.0, float(np.abs(r["xi_next"]).max())))
                xi = lg["xi_after"][l][it]
                n_checked += 1
    vo_raw.close(); vo_flt.close()
    assert n_checked >= 12
