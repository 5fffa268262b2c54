"""The batched mono pipeline (dvo_batch_create_mono: tracking + inverse-depth filter for n_seq sequences per call, BASELINE
configs[2]) against the single-sequence dvo_vo handle and the CPU oracle."""
import numpy as np
import pytest

import dvo_amd as dvo
import orc
from util import K640, frames

pytestmark = pytest.mark.gpu


def _run_batch_and_singles(B, n_frames, ring, history_limit, order, init_d, seed=3, g=None, K=K640):
    import torch
    if g is None:
        g = frames(6, seed=7)[0]
    dev = torch.device("cuda", 0)
    cfg = dvo.default_config(rng_seed=seed, gn_pixels_per_thread=4)   # one tile size for batch and single
    mb = dvo.MonoBatch(B, K, 640, 480, ring_keyframes=ring, cfg=cfg)
    mb.setInitialDepth(init_d, np.full_like(init_d, 0.5))
    out = []
    for k in range(n_frames):
        gb = torch.from_numpy(np.stack([g[order[b][k]] for b in range(B)])).to(dev)
        mb.odometrize_device(gb.data_ptr())
        xi, T, key = mb.world_poses()
        kfs = [mb.keyframe(b) for b in range(B)]
        out.append((T.copy(), key.copy(), kfs))
    logs = [mb.last_track_log(b) for b in range(B)]
    mb.close()
    singles = []
    for b in range(B):
        vo = dvo.VisualOdometry(K, 640, 480, cfg=cfg)
        vo.setInitialDepth(init_d, np.full_like(init_d, 0.5))
        if history_limit:
            vo.setHistoryLimit(history_limit)
        res = []
        for k in range(n_frames):
            T, key = vo.odometrize(g[order[b][k]])
            kf = vo.keyframe(vo.keyframeCount() - 1)
            res.append((T, key, kf, vo.lastValidUpdates() if (k > 0 and not key) else 0))
        singles.append((res, vo.lastTrackLog()))
        vo.close()
    return out, logs, singles


def _compare(out, logs, singles, B, n_frames):
    n_key = n_upd = 0
    for b in range(B):
        res, slog = singles[b]
        for k in range(n_frames):
            Tb, keyb, kfs = out[k]
            T1, key1, kf1, v1 = res[k]
            assert bool(keyb[b]) == key1, (b, k)
            np.testing.assert_array_equal(Tb[b], T1, err_msg="pose of sequence %d frame %d" % (b, k))
            kb = kfs[b]
            np.testing.assert_array_equal(kb["gray"], kf1["gray"])
            np.testing.assert_array_equal(kb["age"], kf1["age"])
            np.testing.assert_array_equal(kb["sigma"], kf1["sigma"])
            np.testing.assert_array_equal(kb["depth"], kf1["depth"])
            np.testing.assert_array_equal(kb["xi"], kf1["xi"])
            assert kb["id"] == kf1["id"]
            if k > 0 and not key1:
                assert kb["valid_updates"] == v1
                n_upd += 1
            n_key += int(key1 and k > 0)
        assert logs[b]["n_iter"] == slog["n_iter"]
        for l in range(3):
            np.testing.assert_array_equal(logs[b]["residual"][l], slog["residual"][l])
    return n_key, n_upd


def test_mono_batch_every_sequence_matches_single_handle_bit_for_bit():
    """24 sequences that see the 6 frames in different orders (so they take different branches of Mapper::estimate on the same
    step: new keyframe vs stereo update) -- poses, keyframe flags, the newest keyframe's gray / depth / sigma / age maps, ids,
    valid-update counts and the last track log are bit-identical to 24 dvo_vo handles."""
    B, n_frames = 24, 8
    rng = np.random.RandomState(12)
    g, d, s, _ = frames(6, seed=7)
    d0 = orc.cull_image(d[0], 2)
    init_d = (d0 + rng.normal(0, 0.1, d0.shape)).astype(np.float32)
    order = []
    for b in range(B):
        step = 1 + b % 3                      # 1: small motion (mostly updates), 2-3: larger (keyframes by translation)
        start = b % 6
        seq = [(start + step * k) % 6 for k in range(n_frames)]
        if b % 4 == 3:
            seq = [seq[0]] * 3 + seq[3:]      # a static stretch: keyframes by the 6-frame rule only
        order.append(seq)
    out, logs, singles = _run_batch_and_singles(B, n_frames, ring=8, history_limit=0, order=order, init_d=init_d)
    n_key, n_upd = _compare(out, logs, singles, B, n_frames)
    assert n_key >= B and n_upd >= B          # both branches were taken many times
    keys_per_step = [int(out[k][1].sum()) for k in range(1, n_frames)]
    assert any(0 < n < B for n in keys_per_step)   # on some step part of the batch created keyframes while the rest updated


def test_mono_batch_on_real_frames_matches_single_handle_bit_for_bit():
    """The same identity on the reference's own webcam frames (tests/golden/logicool0_excerpt.npz): INVALID undistortion border,
    black pixels, and stereo updates that really change the maps (valid-update counts > 0, ages reset by failed fusions)."""
    import os
    from real_data import K_LOGICOOL, frames_from_fixture
    fx = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "logicool0_excerpt.npz")))
    g = frames_from_fixture(fx)
    B, n_frames = 8, 12
    order = [[(b + k) % 20 for k in range(n_frames)] for b in range(6)] + [[19 - k for k in range(n_frames)], [2 * k % 20 for k in range(n_frames)]]
    out, logs, singles = _run_batch_and_singles(B, n_frames, ring=16, history_limit=0, order=order, init_d=fx["init_depth"],
                                                seed=int(fx["seed_vo"]), g=g, K=K_LOGICOOL)
    n_key, n_upd = _compare(out, logs, singles, B, n_frames)
    assert n_key >= B and n_upd >= B
    assert max(kf["valid_updates"] for k in range(1, n_frames) for kf in out[k][2]) > 20   # the update branch really wrote depth


def test_mono_batch_ring_overflow_equals_bounded_history():
    """More keyframes than ring slots: the ring then equals a dvo_vo handle with dvo_vo_set_history_limit(R) (pixels born in a
    dropped keyframe search the oldest retained one)."""
    B, n_frames = 6, 10
    rng = np.random.RandomState(5)
    g, d, s, _ = frames(6, seed=7)
    d0 = orc.cull_image(d[0], 2)
    init_d = (d0 + rng.normal(0, 0.1, d0.shape)).astype(np.float32)
    order = [[(b + 2 * k + (k // 3)) % 6 for k in range(n_frames)] for b in range(B)]
    out, logs, singles = _run_batch_and_singles(B, n_frames, ring=2, history_limit=2, order=order, init_d=init_d)
    n_key, n_upd = _compare(out, logs, singles, B, n_frames)
    assert max(kf["n_keyframes"] for kf in out[-1][2]) > 2


def test_mono_batch_matches_the_oracle():
    """The batch against the CPU oracle's VisualOdometry on the synthetic sequence: keyframe decisions and age maps equal, world
    poses within 1e-4, depth maps within 1e-3 on > 99 % of the pixels (the tolerances of the single-handle test)."""
    import torch
    g, d, s, _ = frames(6, seed=7)
    rng = np.random.RandomState(12)
    d0 = orc.cull_image(d[0], 2)
    init_d = (d0 + rng.normal(0, 0.1, d0.shape)).astype(np.float32)
    B = 4
    shifts = [0, 1, 2, 3]
    mb = dvo.MonoBatch(B, K640, 640, 480, cfg=dvo.default_config(rng_seed=3))
    mb.setInitialDepth(init_d, np.full_like(init_d, 0.5))
    ovos = []
    for b in range(B):
        o = orc.OVO(K640, 640, 480, seed=3)
        o.set_initial_depth(init_d, np.full_like(init_d, 0.5))
        ovos.append(o)
    dev = torch.device("cuda", 0)
    for k in range(6):
        gb = torch.from_numpy(np.stack([g[(k + shifts[b]) % 6] for b in range(B)])).to(dev)
        mb.odometrize_device(gb.data_ptr())
        xi, T, key = mb.world_poses()
        for b in range(B):
            To, keyo = ovos[b].odometrize(g[(k + shifts[b]) % 6])
            assert bool(key[b]) == keyo
            np.testing.assert_allclose(T[b], To, rtol=0, atol=1e-4)
            kf = mb.keyframe(b)
            okf = ovos[b].keyframe(ovos[b].keyframe_count() - 1)
            np.testing.assert_array_equal(kf["age"], okf.age())
            assert (np.abs(kf["depth"] - okf.depth(2)) > 1e-3).mean() < 0.01
    mb.close()


def test_mono_batch_rejects_sensor_depth_calls_and_vice_versa():
    mb = dvo.MonoBatch(2, K640, 640, 480)
    L = dvo.lib()
    assert L.dvo_batch_push_device(mb._p, None, None, None) != 0
    assert L.dvo_batch_last_poses(mb._p, None, None) == 1
    mb.close()
    bt = dvo.Batch(2, K640, 640, 480)
    assert L.dvo_batch_odometrize_device(bt._p, None) == 1
    bt.close()
    with pytest.raises(dvo.DvoError):
        dvo.MonoBatch(0, K640, 640, 480)


def test_mono_batch_failed_first_frame_leaves_the_batch_not_started():
    """A call that fails (here: a channel count the raw path rejects, a null pointer) must not advance Frame::latest_id: the batch stays
    in the not-started state, and the next good frame IS frame 0 (first keyframe, identity pose) -- frame ids never shift."""
    import torch
    g = frames(6, seed=7)[0]
    dev = torch.device("cuda", 0)
    mb = dvo.MonoBatch(2, K640, 640, 480, cfg=dvo.default_config(rng_seed=3))
    L = dvo.lib()
    g8 = torch.zeros((2, 480, 640, 2), dtype=torch.uint8, device=dev)
    assert L.dvo_batch_odometrize_raw_device(mb._p, dvo.C.c_void_p(g8.data_ptr()), 2) != 0      # 2 channels: rejected
    assert L.dvo_batch_odometrize_device(mb._p, None) != 0
    assert L.dvo_batch_world_poses(mb._p, None, None, None) == dvo.DVO_ERR_NOT_READY            # still not started
    st = dvo.MonoStats()
    assert L.dvo_batch_mono_stats(mb._p, 0, dvo.C.byref(st)) == dvo.DVO_ERR_NOT_READY
    gb = torch.from_numpy(np.stack([g[0], g[1]])).to(dev)
    mb.odometrize_device(gb.data_ptr())
    xi, T, key = mb.world_poses()
    assert key.all() and np.array_equal(T[0], np.eye(4, dtype=np.float32))
    assert mb.stats(0)["frames"] == 1 and mb.stats(1)["keyframes_created"] == 1
    # a failed call in the middle does not consume a frame id either
    assert L.dvo_batch_odometrize_device(mb._p, None) != 0
    gb2 = torch.from_numpy(np.stack([g[1], g[2]])).to(dev)
    mb.odometrize_device(gb2.data_ptr())
    assert mb.stats(0)["frames"] == 2
    mb.close()


def test_mono_batch_ring_clamp_is_counted():
    """ring of 2 keyframes, more than 2 created: pixels older than the ring are searched against the oldest retained keyframe
    (the one deviation from the reference's unbounded FrameHistory) -- and every such pixel is counted (dvo_mono_stats)."""
    import torch
    B, n_frames = 3, 10
    rng = np.random.RandomState(5)
    g, d, s, _ = frames(6, seed=7)
    d0 = orc.cull_image(d[0], 2)
    init_d = (d0 + rng.normal(0, 0.1, d0.shape)).astype(np.float32)
    dev = torch.device("cuda", 0)
    counts = {}
    for ring in (2, 16):
        mb = dvo.MonoBatch(B, K640, 640, 480, ring_keyframes=ring, cfg=dvo.default_config(rng_seed=3))
        mb.setInitialDepth(init_d, np.full_like(init_d, 0.5))
        for k in range(n_frames):
            gb = torch.from_numpy(np.stack([g[(b + 2 * k + (k // 3)) % 6] for b in range(B)])).to(dev)
            mb.odometrize_device(gb.data_ptr())
        counts[ring] = [mb.stats(b) for b in range(B)]
        mb.close()
    assert max(s["keyframes_created"] for s in counts[2]) > 2
    assert sum(s["clamped_pixels"] for s in counts[2]) > 0          # the ring of 2 overflowed and said so
    assert sum(s["clamped_pixels"] for s in counts[16]) == 0        # a ring that holds every keyframe never clamps
    assert counts[2][0]["ring_keyframes"] == 2
