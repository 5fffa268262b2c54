#!/usr/bin/env python3
"""Diagnostic: teacher-forced (oracle state loaded per frame) tracking on the logicool0 fixture -- where do GPU and oracle part?"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "direct-visual-odometry_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa
import dvo_amd as dvo
import orc
from real_data import K_LOGICOOL, frames_from_fixture, write_keyframe_store

fx = dict(np.load(os.path.join(ROOT, "tests", "golden", "logicool0_excerpt.npz")))
frames = frames_from_fixture(fx)
seed = int(fx["seed_vo"])
ovo = orc.OVO(K_LOGICOOL, 640, 480, seed=seed)
ovo.set_initial_depth(fx["init_depth"], np.full_like(fx["init_depth"], 0.5))
ovo.odometrize(frames[0])
vo = dvo.VisualOdometry(K_LOGICOOL, 640, 480, cfg=dvo.default_config(rng_seed=seed))
path = os.path.join(tempfile.mkdtemp(), "s.dvokf")
for i in range(1, len(frames)):
    kfs = [ovo.keyframe(k) for k in range(ovo.keyframe_count())]
    write_keyframe_store(path, K_LOGICOOL, 640, 480, kfs, latest_id=i - 1)
    vo.load(path)
    oref = kfs[-1]
    oobj = orc.OFrame(frames[i], None, None, K_LOGICOOL, 3, 2)
    xo, lo = orc.track(oobj, oref)
    T, key = vo.odometrize(frames[i])
    To, keyo = ovo.odometrize(frames[i])
    lg = vo.lastTrackLog()
    print("frame %2d key %d/%d |T-To| %.2e iters gpu %s orc %s" % (i, key, keyo, np.abs(T - To).max(), lg["n_iter"], lo["n_iter"]))
    for l in range(3):
        n = min(lg["n_iter"][l], lo["n_iter"][l])
        d = [float(np.abs(lg["xi_after"][l][k] - lo["xi_after"][l][k]).max()) for k in range(n)]
        nv = [int(lg["n_valid"][l][k]) - int(lo["n_valid"][l][k]) for k in range(n)]
        print("   level %d |xi_g - xi_o| per it: %s   n_valid diff %s" % (l, " ".join("%.1e" % v for v in d), nv))
    # operator-level: each oracle iteration's input pose through the GPU op; conditioning of H
    xi = np.zeros(6, np.float32)
    for l in range(3):
        for it in range(lo["n_iter"][l]):
            r = dvo.optimize(oobj.gray(l), oref.gray(l), oref.depth(l), oref.sigma(l), oref.K(l), xi, l, want_mask=True)
            o = orc.optimize(oobj.gray(l), oref.gray(l), oref.depth(l), oref.sigma(l), oref.K(l), xi, l, want_mask=True)
            H = orc.upper_to_full(o["H"])
            back = np.abs(H @ r["xi_update"].astype(np.float64) - o["g"]).max() / (np.abs(H) @ np.abs(r["xi_update"].astype(np.float64)) + np.abs(o["g"])).max()
            if it < 2 or it == lo["n_iter"][l] - 1:
                print("      L%d it %2d mask equal %s  H rel %.1e  upd diff %.1e of |upd| %.1e  backward err %.1e  cond %.1e" % (
                    l, it, bool((r["mask"] == o["mask"]).all()), np.abs(r["H"] - o["H"]).max() / np.abs(o["H"]).max(),
                    np.abs(r["xi_update"] - o["xi_update"]).max(), np.abs(o["xi_update"]).max(), back, np.linalg.cond(H)))
            xi = lo["xi_after"][l][it]
vo.close()
