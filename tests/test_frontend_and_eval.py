"""Tests of the rows SURVEY.md §8f marks "next": dataset front-end (PNG reader, TUM / list datasets, device ingest,
undistortion) and trajectory evaluation (ATE, RPE, TUM export).  The reference pins nothing here either; the
checkers are PIL (PNG), numpy restatements of the integer/byte arithmetic (ingest: bit exact) and of Umeyama's method."""
import os

import numpy as np
import pytest
from PIL import Image

import dvo_amd as dvo


# ---------------------------------------------------------------- PNG reader (CPU)
@pytest.mark.parametrize("mode,dtype", [("L", np.uint8), ("RGB", np.uint8), ("RGBA", np.uint8), ("I;16", np.uint16)])
def test_png_reader_matches_pil(tmp_path, mode, dtype):
    rng = np.random.RandomState(1)
    h, w = 37, 53
    if mode == "L":
        a = (rng.uniform(size=(h, w)) * 255).astype(np.uint8)
    elif mode == "I;16":
        a = (rng.uniform(size=(h, w)) * 65535).astype(np.uint16)
    else:
        a = (rng.uniform(size=(h, w, len(mode))) * 255).astype(np.uint8)
    # smooth gradients make PIL's encoder pick different scan-line filters (sub / up / average / paeth)
    a[10:20] = np.arange(w, dtype=dtype)[None, :, None] if a.ndim == 3 else np.arange(w, dtype=dtype)[None, :]
    p = str(tmp_path / "t.png")
    Image.fromarray(a, mode=mode).save(p)
    got = dvo.imread(p)
    assert got.dtype == dtype and got.shape == a.shape
    np.testing.assert_array_equal(got, a)


def test_png_reader_errors_are_status_codes(tmp_path):
    with pytest.raises(dvo.DvoError):
        dvo.imread(str(tmp_path / "missing.png"))
    bad = tmp_path / "bad.png"
    bad.write_bytes(b"not a png at all, definitely not one, no header here")
    with pytest.raises(dvo.DvoError):
        dvo.imread(str(bad))


@pytest.mark.skipif(not os.path.exists("/root/reference/data/logicool0/0000.png"), reason="reference data not mounted")
def test_png_reader_on_the_reference_frames():
    for p in ("/root/reference/data/logicool0/0000.png", "/root/reference/data/KINECT_50MM/depth01.png"):
        np.testing.assert_array_equal(dvo.imread(p), np.asarray(Image.open(p)))


# ---------------------------------------------------------------- datasets (CPU)
def test_tum_association_and_groundtruth(tmp_path):
    d = tmp_path / "seq"
    d.mkdir()
    (d / "rgb.txt").write_text("# color images\n# file\n# timestamp filename\n1.00 rgb/1.00.png\n1.10 rgb/1.10.png\n1.20 rgb/1.20.png\n1.50 rgb/1.50.png\n")
    (d / "depth.txt").write_text("# depth\n0.995 depth/0.995.png\n1.105 depth/1.105.png\n1.26 depth/1.26.png\n")
    (d / "groundtruth.txt").write_text("# gt\n1.001 1 2 3 0 0 0 1\n1.099 4 5 6 0 0 0 1\n")
    ds = dvo.Dataset(str(d), tum=True, max_dt=0.02)
    assert len(ds) == 2                                   # 1.20 (dt 0.06) and 1.50 have no depth within 20 ms
    e0, e1 = ds.entry(0), ds.entry(1)
    assert e0["rgb"].endswith("rgb/1.00.png") and e0["depth"].endswith("depth/0.995.png")
    assert e1["rgb"].endswith("rgb/1.10.png") and e1["depth"].endswith("depth/1.105.png")
    np.testing.assert_array_equal(e0["gt"], [1, 2, 3, 0, 0, 0, 1])
    np.testing.assert_array_equal(e1["gt"], [4, 5, 6, 0, 0, 0, 1])
    assert len(dvo.Dataset(str(d), tum=True, max_dt=0.1)) == 3
    with pytest.raises(dvo.DvoError):
        dvo.Dataset(str(tmp_path / "nope"), tum=True)


def test_reference_list_files(tmp_path):
    d = tmp_path / "k"
    d.mkdir()
    (d / "info.txt").write_text("rgb01.png depth01.png\nrgb02.png depth02.png\n")   # data/KINECT_50MM/info.txt format
    ds = dvo.Dataset(str(d))
    assert len(ds) == 2 and ds.entry(1)["depth"].endswith("k/depth02.png")
    (d / "mono.txt").write_text("0000.png\n0001.png\n0002.png\n")                    # data/logicool0/info.txt format
    ds = dvo.Dataset(str(d), list_file=str(d / "mono.txt"))
    assert len(ds) == 3 and ds.entry(2)["rgb"].endswith("0002.png") and ds.entry(2)["depth"] == ""
    with pytest.raises(dvo.DvoError):                    # the reference abort()s (include/core/loader.hpp:33-36)
        dvo.Dataset(str(d), list_file=str(d / "missing.txt"))


# ---------------------------------------------------------------- evaluation (CPU)
def _umeyama_np(est, gt):
    ce, cg = est.mean(0), gt.mean(0)
    H = (est - ce).T @ (gt - cg)
    U, S, Vt = np.linalg.svd(H)
    D = np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))])
    R = Vt.T @ D @ U.T
    t = cg - R @ ce
    return R, t


def test_ate_matches_svd_alignment():
    rng = np.random.RandomState(3)
    gt = np.cumsum(rng.normal(0, 0.05, (200, 3)), 0)
    ang = 0.7
    Rz = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    est = (gt - [1, 2, 3]) @ Rz + rng.normal(0, 0.01, gt.shape)   # rotated, shifted, noisy copy
    rm, R, t, s = dvo.ate(est, gt)
    e32, g32 = est.astype(np.float32).astype(np.float64), gt.astype(np.float32).astype(np.float64)
    Rn, tn = _umeyama_np(e32, g32)
    np.testing.assert_allclose(R, Rn, atol=1e-6)
    np.testing.assert_allclose(t, tn, atol=1e-6)
    exp = np.sqrt(np.mean(np.sum((g32 - (e32 @ Rn.T + tn)) ** 2, 1)))
    assert abs(rm - exp) < 1e-7 and 0.01 < rm < 0.03 and s == 1.0
    rm2, _, _, s2 = dvo.ate(est * 2.0, gt, with_scale=True)
    assert abs(s2 - 0.5) < 5e-3 and abs(rm2 - rm) < 2e-3
    assert dvo.ate(gt, gt)[0] < 1e-6
    with pytest.raises(dvo.DvoError):
        dvo.ate(gt[:2], gt[:2])


def test_rpe_pose_inverse_and_tum_export(tmp_path):
    from dvo_amd.synth import se3_exp_np, trajectory
    gt = np.array(trajectory(30, seed=5))
    assert max(dvo.rpe(gt, gt)) < 1e-3                    # float32 poses in, acos near 1 is noisy: just "small"
    # an estimate that drifts by a constant 1 mm / 0.001 rad per frame
    drift = se3_exp_np([0.001, 0, 0, 0, 0, 0.001])
    est = [np.eye(4)]
    for i in range(1, 30):
        est.append(est[-1] @ (np.linalg.inv(gt[i - 1]) @ gt[i]) @ drift)
    tr, rr = dvo.rpe(np.array(est), gt, delta=1)
    assert abs(tr - 0.001) < 5e-5 and abs(rr - 0.001) < 5e-4
    T = gt[7].astype(np.float32)
    np.testing.assert_allclose(dvo.pose_inverse(T) @ T, np.eye(4), atol=1e-6)   # unlike Convert::inversePose (convert.cpp:31-39)
    p = str(tmp_path / "traj.txt")
    dvo.write_tum_trajectory(p, gt, timestamps=np.arange(30) * 0.1)
    rows = [l.split() for l in open(p) if not l.startswith("#")]
    assert len(rows) == 30 and len(rows[0]) == 8
    q = np.array(rows[12][4:], float)
    xyz = np.array(rows[12][1:4], float)
    np.testing.assert_allclose(xyz, gt[12][:3, 3], atol=1e-6)
    assert abs(np.linalg.norm(q) - 1) < 1e-6
    qx, qy, qz, qw = q
    Rq = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                   [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                   [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
    np.testing.assert_allclose(Rq, gt[12][:3, :3], atol=1e-6)


# ---------------------------------------------------------------- device ingest / undistort (GPU)
def _ingest_np(rgb, d16, scale=1.0 / 5000.0):
    if rgb.ndim == 2:
        g8 = rgb.astype(np.uint32)
    else:
        r, g, b = (rgb[..., i].astype(np.uint32) for i in range(3))
        g8 = (r * 4899 + g * 9617 + b * 1868 + 8192) >> 14
    gray = g8.astype(np.float32) * np.float32(1.0 / 255.0)
    depth = d16.astype(np.float32) * np.float32(scale)
    sigma = np.where(d16 > 0, np.float32(0.1), np.float32(1.0)).astype(np.float32)
    gray = np.where(d16 == 0, np.float32(-2.0), gray).astype(np.float32)
    return gray, depth, sigma


@pytest.mark.gpu
@pytest.mark.parametrize("channels", [1, 3, 4])
def test_ingest_bit_exact(channels):
    rng = np.random.RandomState(7)
    h, w = 61, 97
    rgb = (rng.uniform(size=(h, w) if channels == 1 else (h, w, channels)) * 255).astype(np.uint8)
    d16 = (rng.uniform(size=(h, w)) * 20000).astype(np.uint16)
    d16[rng.uniform(size=(h, w)) < 0.1] = 0
    g, d, s = dvo.ingest(rgb, d16)
    eg, ed, es = _ingest_np(rgb, d16)
    np.testing.assert_array_equal(g, eg)
    np.testing.assert_array_equal(d, ed)
    np.testing.assert_array_equal(s, es)
    g2 = dvo.ingest(rgb)                               # gray only
    assert (g2 >= 0).all() and g2.max() <= 1.0


@pytest.mark.gpu
def test_undistort_nearest_remap():
    rng = np.random.RandomState(8)
    h, w = 120, 160
    img = rng.uniform(0, 1, (h, w)).astype(np.float32)
    K = np.array([[195, 0, 94.5], [0, 199, 55], [0, 0, 1]], np.float32)        # loader.cpp:17 scaled by 1/4
    D = np.array([-0.0462, 0.152, -0.00429, 0.0117, -0.0725], np.float32)      # loader.cpp:18
    got = dvo.undistort(img, K, D)
    v, u = np.mgrid[0:h, 0:w].astype(np.float64)
    x = (u - K[0, 2]) / K[0, 0]
    y = (v - K[1, 2]) / K[1, 1]
    r2 = x * x + y * y
    rad = 1 + r2 * (D[0] + r2 * (D[1] + r2 * D[4]))
    xd = x * rad + 2 * D[2] * x * y + D[3] * (r2 + 2 * x * x)
    yd = y * rad + D[2] * (r2 + 2 * y * y) + 2 * D[3] * x * y
    mx = np.rint((xd * K[0, 0] + K[0, 2]).astype(np.float32)).astype(int)
    my = np.rint((yd * K[1, 1] + K[1, 2]).astype(np.float32)).astype(int)
    ok = (mx >= 0) & (mx < w) & (my >= 0) & (my < h)
    exp = np.full((h, w), -2.0, np.float32)
    exp[ok] = img[my[ok], mx[ok]]
    assert (got == exp).mean() > 0.999                    # ties of rint at .5 may differ in the last double bit
    np.testing.assert_array_equal(dvo.undistort(img, K, np.zeros(5, np.float32)), img)


@pytest.mark.gpu
def test_odometrize_raw_equals_float_path():
    from util import K640, frames
    g, d, s, _ = frames(3, sigma=0.1)
    vo_a = dvo.VisualOdometry(K640, 640, 480)
    vo_b = dvo.VisualOdometry(K640, 640, 480)
    for i in range(3):
        rgb = np.clip(np.rint(g[i] * 255), 0, 255).astype(np.uint8)
        d16 = np.clip(np.rint(d[i] * 5000), 0, 65535).astype(np.uint16)
        d16[5:9, 7:30] = 0                                # holes: gray INVALID, sigma 1 (transform.cpp:60-76)
        fg, fd, fs = dvo.ingest(rgb, d16)
        Ta = vo_a.odometrizeUsingDepth(fg, fd, fs)
        Tb = vo_b.odometrizeUsingDepthRaw(rgb, d16)
        np.testing.assert_array_equal(Ta, Tb)
    vo_a.close()
    vo_b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("channels", [1, 3])
def test_batch_raw_push_equals_float_push(channels):
    """dvo_batch_push_raw_{host,device} (u8 gray / RGB + u16 depth, converted inside k_pyramid on the pixels the pyramid keeps) against
    dvo_op_ingest + dvo_batch_push_host on the same frames: poses and track logs bit-identical, with and without the look-ahead."""
    import torch
    from util import K640, frames
    g, d, s, _ = frames(4, sigma=0.1)
    B = 5
    rng = np.random.RandomState(3)
    raw_rgb, raw_d16, fl = [], [], []
    for k in range(4):
        rr, dd, ff = [], [], []
        for b in range(B):
            g8 = np.clip(np.rint(g[(k + b) % 4] * 255), 0, 255).astype(np.uint8)
            if channels == 3:
                rgb = np.stack([g8, np.roll(g8, 1, axis=1), np.roll(g8, 2, axis=0)], axis=-1)   # three different channels
            else:
                rgb = g8
            d16 = np.clip(np.rint(d[(k + b) % 4] * 5000), 0, 65535).astype(np.uint16)
            d16[rng.uniform(size=d16.shape) < 0.02] = 0
            rr.append(rgb); dd.append(d16); ff.append(dvo.ingest(rgb, d16))
        raw_rgb.append(np.stack(rr)); raw_d16.append(np.stack(dd)); fl.append(ff)
    def run(mode):
        if mode == "host_full":   # whole frames over PCIe (the default transfers only the rows the pyramid keeps: one strided copy)
            os.environ["DVO_UPLOAD_FULL_FRAMES"] = "1"
            os.environ["DVO_WEIGHT_MAPS"] = "1"   # ... and the per-pixel weight maps instead of the constant weight of raw sensor frames
        try:
            bt = dvo.Batch(B, K640, 640, 480, 4, 1, cfg=dvo.default_config(gn_pixels_per_thread=4))
        finally:
            os.environ.pop("DVO_UPLOAD_FULL_FRAMES", None)
            os.environ.pop("DVO_WEIGHT_MAPS", None)
        out = []
        if mode == "device":
            dev = torch.device("cuda", 0)
            tr = [torch.from_numpy(a).to(dev) for a in raw_rgb]; td = [torch.from_numpy(a.view(np.int16)).to(dev) for a in raw_d16]
            torch.cuda.synchronize()
        for k in range(4):
            if mode == "float":
                bt.push_host(np.stack([f[0] for f in fl[k]]), np.stack([f[1] for f in fl[k]]), np.stack([f[2] for f in fl[k]]))
            elif mode in ("host", "host_full"):
                bt.push_raw_host(raw_rgb[k], raw_d16[k])
            else:
                if 1 <= k < 3:
                    bt.prefetch_raw_device(tr[k + 1].data_ptr(), channels, td[k + 1].data_ptr())
                bt.push_raw_device(tr[k].data_ptr(), channels, td[k].data_ptr())
            if k > 0:
                out.append((bt.last_poses()[0].copy(), [bt.last_track_log(b)["residual"] for b in range(B)]))
        bt.close()
        return out
    ref = run("float")
    assert np.abs(ref[0][0]).max() > 1e-5
    for mode in ("host", "host_full", "device"):
        got = run(mode)
        for (xa, la), (xb, lb) in zip(ref, got):
            np.testing.assert_array_equal(xa, xb)
            for b in range(B):
                for l in range(4):
                    np.testing.assert_array_equal(la[b][l], lb[b][l])


@pytest.mark.gpu
def test_mono_batch_raw_equals_float():
    import torch
    from util import K640, frames
    g = frames(4, sigma=0.1)[0]
    B = 3
    dev = torch.device("cuda", 0)
    res = []
    for raw in (False, True):
        mb = dvo.MonoBatch(B, K640, 640, 480, cfg=dvo.default_config(rng_seed=2))
        out = []
        for k in range(4):
            g8 = np.stack([np.clip(np.rint(g[(k + b) % 4] * 255), 0, 255).astype(np.uint8) for b in range(B)])
            if raw:
                t = torch.from_numpy(g8).to(dev); torch.cuda.synchronize()
                mb.odometrize_raw_device(t.data_ptr(), 1)
            else:
                t = torch.from_numpy(np.stack([dvo.ingest(g8[b]) for b in range(B)])).to(dev); torch.cuda.synchronize()
                mb.odometrize_device(t.data_ptr())
            out.append(mb.world_poses()[1].copy())
        out.append(mb.keyframe(1)["depth"])
        mb.close()
        res.append(out)
    for a, b in zip(*res):
        np.testing.assert_array_equal(a, b)
    # ... and so does a single dvo_vo handle fed raw frames (dvo_vo_odometrize_raw) against the float entry point
    va = dvo.VisualOdometry(K640, 640, 480, cfg=dvo.default_config(rng_seed=2)); vb = dvo.VisualOdometry(K640, 640, 480, cfg=dvo.default_config(rng_seed=2))
    for k in range(4):
        g8 = np.clip(np.rint(g[k] * 255), 0, 255).astype(np.uint8)
        Ta, ka = va.odometrizeRaw(g8)
        Tb, kb = vb.odometrize(dvo.ingest(g8))
        np.testing.assert_array_equal(Ta, Tb)
        assert ka == kb
    va.close(); vb.close()
    # the host entry points (copy stream, two staging slots, fixed schedule) give the same bits
    mb = dvo.MonoBatch(B, K640, 640, 480, cfg=dvo.default_config(rng_seed=2))
    for k in range(4):
        g8 = np.stack([np.clip(np.rint(g[(k + b) % 4] * 255), 0, 255).astype(np.uint8) for b in range(B)])
        mb.odometrize_host(g8)
        np.testing.assert_array_equal(mb.world_poses()[1], res[1][k])
    mb.close()


@pytest.mark.gpu
def test_pageable_host_buffers_may_be_reused_when_the_push_returns():
    """include/dvo.h: a pageable buffer handed to dvo_batch_push_host / _push_raw_host / _odometrize_*_host has been copied when the
    call returns (only pinned buffers stay in flight).  The caller here scribbles over its one set of buffers right after every push
    -- what a capture loop that reuses its frame buffer does -- and must get the poses of a run that kept every frame untouched."""
    from util import K640, frames
    g, d, s, _ = frames(4, sigma=0.1)
    B = 24                                                    # 3 x 29 MB per push: the DMA is still running when the call would return
    def stacks(k):
        return (np.stack([g[(k + b) % 4] for b in range(B)]), np.stack([d[(k + b) % 4] for b in range(B)]), np.stack([s[(k + b) % 4] for b in range(B)]))
    def run(reuse):
        bt = dvo.Batch(B, K640, 640, 480, 4, 1, cfg=dvo.default_config(gn_pixels_per_thread=4))
        bufs = [np.empty((B, 480, 640), np.float32) for _ in range(3)]
        keep, out = [], []
        for k in range(4):
            src = stacks(k)
            if reuse:
                for buf, a in zip(bufs, src):
                    buf[...] = a
                bt.push_host(*bufs)
                for buf in bufs:
                    buf[...] = -7.0                           # the frame is gone as far as the caller is concerned
            else:
                keep.append(src)
                bt.push_host(*src)
            if k > 0:
                out.append(bt.last_poses()[0].copy())
        bt.close()
        return np.stack(out)
    ref = run(False)
    assert np.isfinite(ref).all() and np.abs(ref).max() > 1e-5
    np.testing.assert_array_equal(run(True), ref)
    # the mono batch's host entry point, raw u8 frames
    def run_mono(reuse):
        mb = dvo.MonoBatch(B, K640, 640, 480, cfg=dvo.default_config(rng_seed=2))
        buf = np.empty((B, 480, 640), np.uint8)
        keep, out = [], []
        for k in range(4):
            g8 = np.stack([np.clip(np.rint(g[(k + b) % 4] * 255), 0, 255).astype(np.uint8) for b in range(B)])
            if reuse:
                buf[...] = g8
                mb.odometrize_host(buf)
                buf[...] = 0
            else:
                keep.append(g8)
                mb.odometrize_host(g8)
            out.append(mb.world_poses()[1].copy())
        mb.close()
        return np.stack(out)
    np.testing.assert_array_equal(run_mono(True), run_mono(False))


# ---------------------------------------------------------------- whole-trajectory agreement (BASELINE.json: ATE within 1e-3 m of the reference)
@pytest.mark.gpu
def test_trajectory_ate_gpu_vs_oracle_on_synthetic_ground_truth():
    import orc
    from dvo_amd import synth
    from util import K640
    n = 14
    g, d, s, poses = synth.sequence(n, seed=11, sigma_value=0.5)
    g, d, s = g.numpy(), d.numpy(), s.numpy()
    vo = dvo.VisualOdometry(K640, 640, 480)
    ovo = orc.OVO(K640, 640, 480)
    Tg, To = [np.eye(4)], [np.eye(4)]
    for i in range(n):
        rel_g = vo.odometrizeUsingDepth(g[i], d[i], s[i]).astype(np.float64)
        rel_o = ovo.odometrize_depth(g[i], d[i], s[i]).astype(np.float64)
        if i:
            # exp(xi_rel) maps reference-frame points into the new frame: T_world<-new = T_world<-ref * inv(T_rel)
            Tg.append(Tg[-1] @ np.linalg.inv(rel_g))
            To.append(To[-1] @ np.linalg.inv(rel_o))
    vo.close()
    gt = np.array([p[:3, 3] for p in poses])
    xg = np.array([T[:3, 3] for T in Tg]); xo = np.array([T[:3, 3] for T in To])
    ate_g = dvo.ate(xg, gt)[0]
    ate_o = dvo.ate(xo, gt)[0]
    assert abs(ate_g - ate_o) < 1e-3                      # BASELINE.json: within 1e-3 m of the reference path
    assert np.abs(xg - xo).max() < 1e-3                   # and the two trajectories coincide frame by frame
    assert ate_g < 0.05                                    # sanity: it does track (path length ~ 0.1 m)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,levels,culls", [(646, 486, 4, 1), (640, 484, 3, 2), (328, 248, 3, 1), (330, 250, 2, 1), (96, 72, 2, 2)])
def test_host_uploads_of_odd_sizes_equal_device_frames(w, h, levels, culls):
    """The row-decimated host upload (one strided copy per buffer, rows r = 0 mod 2^culls of the caller's [n_seq][h][w] buffer)
    against the same frames already resident on the device, for sizes where the vector pyramid kernel does not apply (widths that are
    no multiple of 4 << culls), where the decimation does not apply (heights that are no multiple of 2^culls) and for both input
    kinds (raw u8 + u16, float maps): poses and per-iteration logs bit-identical."""
    import torch
    from dvo_amd import synth
    K = np.array(synth.K_640, np.float32).copy()
    K[0] *= w / 640.0; K[1] *= h / 480.0
    g, d, s, _ = synth.sequence(3, width=w, height_px=h, K=K, seed=17, sigma_value=0.1)
    g, d = g.numpy(), d.numpy()
    B = 2
    g8 = [np.stack([np.clip(np.rint(g[(k + b) % 3] * 255), 0, 255).astype(np.uint8) for b in range(B)]) for k in range(3)]
    d16 = [np.stack([np.clip(np.rint(d[(k + b) % 3] * 5000), 0, 65535).astype(np.uint16) for b in range(B)]) for k in range(3)]
    fl = [[dvo.ingest(g8[k][b], d16[k][b]) for b in range(B)] for k in range(3)]
    dev = torch.device("cuda", 0)
    def run(mode):
        bt = dvo.Batch(B, K, w, h, levels, culls, cfg=dvo.default_config(gn_pixels_per_thread=4))
        out = []
        for k in range(3):
            if mode == "raw_host":
                bt.push_raw_host(g8[k], d16[k])
            elif mode == "raw_device":
                tg = torch.from_numpy(g8[k]).to(dev); td = torch.from_numpy(d16[k].view(np.int16)).to(dev); torch.cuda.synchronize()
                bt.push_raw_device(tg.data_ptr(), 1, td.data_ptr())
            elif mode == "float_host":
                bt.push_host(np.stack([f[0] for f in fl[k]]), np.stack([f[1] for f in fl[k]]), np.stack([f[2] for f in fl[k]]))
            else:
                t = [torch.from_numpy(np.stack([f[m] for f in fl[k]])).to(dev) for m in range(3)]; torch.cuda.synchronize()
                bt.push_device(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr())
            if k > 0:
                bt.synchronize()
                out.append((bt.last_poses()[0].copy(), [bt.last_track_log(b)["residual"] for b in range(B)]))
        bt.close()
        return out
    ref = run("raw_device")
    assert np.abs(ref[-1][0]).max() > 1e-6
    for mode in ("raw_host", "float_host", "float_device"):
        got = run(mode)
        for (xa, la), (xb, lb) in zip(ref, got):
            np.testing.assert_array_equal(xa, xb, err_msg=mode)
            for b in range(B):
                for l in range(levels):
                    np.testing.assert_array_equal(la[b][l], lb[b][l], err_msg=mode)
