"""keyframe store / checkpoint, bounded history, false-colour debug views."""
import numpy as np
import pytest

import dvo_amd as dvo


def test_ppm_writer(tmp_path):
    rgb = (np.arange(5 * 7 * 3) % 256).astype(np.uint8).reshape(5, 7, 3)
    p = str(tmp_path / "v.ppm")
    dvo.write_ppm(p, rgb)
    raw = open(p, "rb").read()
    assert raw.startswith(b"P6\n7 5\n255\n") and raw[len(b"P6\n7 5\n255\n"):] == rgb.tobytes()


@pytest.mark.gpu
def test_checkpoint_round_trip_continues_identically(tmp_path):
    import orc
    from util import K640, frames
    g, d, s, _ = frames(6, seed=7)
    d0 = orc.cull_image(d[0], 2)
    init_s = np.full_like(d0, 0.5)

    def fresh():
        vo = dvo.VisualOdometry(K640, 640, 480, cfg=dvo.default_config(rng_seed=3))
        vo.setInitialDepth(d0, init_s)
        return vo

    a = fresh()
    for i in range(4):
        a.odometrize(g[i])
    ck = str(tmp_path / "kf.bin")
    a.save(ck)
    n_a = a.keyframeCount()
    b = fresh()
    b.load(ck)
    assert b.keyframeCount() == n_a
    for idx in range(n_a):
        ka, kb = a.keyframe(idx), b.keyframe(idx)
        assert ka["id"] == kb["id"]
        for key in ("gray", "depth", "sigma", "age", "xi", "rel_xi"):
            np.testing.assert_array_equal(ka[key], kb[key])
        np.testing.assert_array_equal(a.keyframe(idx, 0)["depth"], b.keyframe(idx, 0)["depth"])   # re-decimated levels too
    for i in range(4, 6):                                   # both continue: identical poses and keyframes
        Ta, ka_ = a.odometrize(g[i])
        Tb, kb_ = b.odometrize(g[i])
        assert ka_ == kb_
        np.testing.assert_array_equal(Ta, Tb)
    last = a.keyframeCount() - 1
    np.testing.assert_array_equal(a.keyframe(last)["depth"], b.keyframe(last)["depth"])
    other = dvo.VisualOdometry(K640 * 1.01, 640, 480)
    with pytest.raises(dvo.DvoError):
        other.load(ck)                                      # different camera: refused, not silently accepted
    for v in (a, b, other):
        v.close()


@pytest.mark.gpu
def test_history_limit_bounds_the_store():
    import orc
    from util import K640, frames
    g, d, s, _ = frames(6, seed=7)
    d0 = orc.cull_image(d[0], 2)
    # force a keyframe on every frame, keep at most 2
    vo = dvo.VisualOdometry(K640, 640, 480, cfg=dvo.default_config(keyframe_max_frames=1))
    vo.setInitialDepth(d0, np.full_like(d0, 0.5))
    vo.setHistoryLimit(2)
    for i in range(6):
        T, key = vo.odometrize(g[i])
        assert key and np.isfinite(T).all()
        assert vo.keyframeCount() <= 2
    assert vo.keyframe(1)["id"] == 5 and vo.keyframe(0)["id"] == 4
    vo.close()


@pytest.mark.gpu
def test_false_colour_views():
    rng = np.random.RandomState(0)
    gray = rng.uniform(0, 1, (20, 30)).astype(np.float32)
    gray[3, 4] = -2.0
    v = dvo.visualize(dvo.VIS_GRAY, gray)
    assert v.shape == (20, 30, 3) and tuple(v[3, 4]) == (0, 0, 255)                  # INVALID -> blue (draw.cpp:12-17)
    exp = np.clip(np.rint(gray * 255), 0, 255).astype(np.uint8)
    m = np.ones_like(gray, bool); m[3, 4] = False
    assert (v[..., 0][m] == exp[m]).all() and (v[..., 1][m] == exp[m]).all() and (v[..., 2][m] == exp[m]).all()
    depth = rng.uniform(0.8, 3.0, (20, 30)).astype(np.float32); depth[0, 0] = 0.0
    sigma = rng.uniform(0.01, 0.6, (20, 30)).astype(np.float32)
    dv = dvo.visualize(dvo.VIS_DEPTH, depth, sigma)
    assert tuple(dv[0, 0]) == (0, 0, 0)                                               # no depth -> black (draw.cpp:58-61)
    assert dv[sigma > 0.5].max() <= 6                                                 # value = 255 - 500 min(sigma, .5) -> ~5
    near, far = dvo.visualize(dvo.VIS_DEPTH, np.full((2, 2), 0.75, np.float32)), dvo.visualize(dvo.VIS_DEPTH, np.full((2, 2), 2.0, np.float32))
    assert near[0, 0, 0] > 200 and near[0, 0, 2] < 30 and far[0, 0, 1] > 200          # hue 3.5 -> red, hue 91 -> green/cyan
    sv = dvo.visualize(dvo.VIS_SIGMA, sigma)
    assert (sv[..., 0] == np.clip(np.rint(sigma * -500 + 255), 0, 255)).all()
    av = dvo.visualize(dvo.VIS_AGE, np.full((2, 2), 7, np.float32))
    assert (av == 70).all()                                                           # draw.cpp:93-99
    gv = dvo.visualize(dvo.VIS_GRADIENT, np.array([[0.5, -0.25, -2.0]], np.float32))
    assert tuple(gv[0, 0]) == (0, 127, 0) and tuple(gv[0, 1]) == (63, 0, 0) and tuple(gv[0, 2]) == (0, 0, 255)
