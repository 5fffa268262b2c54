#!/usr/bin/env python3
"""Diagnostic: GPU mono pipeline vs the oracle on the logicool0 fixture, frame by frame (poses, keyframe flags, iteration
counts, per-iteration agreement of the GPU's track log with the oracle's optimize at the same input pose)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "direct-visual-odometry_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa
import dvo_amd as dvo
import orc
from real_data import K_LOGICOOL, frames_from_fixture

fx = dict(np.load(os.path.join(ROOT, "tests", "golden", "logicool0_excerpt.npz")))
frames = frames_from_fixture(fx)
vo = dvo.VisualOdometry(K_LOGICOOL, 640, 480, cfg=dvo.default_config(rng_seed=int(fx["seed_vo"])))
vo.setInitialDepth(fx["init_depth"], np.full_like(fx["init_depth"], 0.5))
ovo = orc.OVO(K_LOGICOOL, 640, 480, seed=int(fx["seed_vo"]))
ovo.set_initial_depth(fx["init_depth"], np.full_like(fx["init_depth"], 0.5))
for i, g in enumerate(frames):
    if i > 0:
        oref = ovo.keyframe(ovo.keyframe_count() - 1)
        oobj = orc.OFrame(g, None, None, K_LOGICOOL, 3, 2)
        xo, lo = orc.track(oobj, oref)
        # GPU keyframe maps before this frame
        kf = vo.keyframe(vo.keyframeCount() - 1)
        dd = [float(np.abs(kf["depth"] - oref.depth(2)).max()), float((np.abs(kf["depth"] - oref.depth(2)) > 1e-3).mean()),
              float((kf["age"] != oref.age()).mean()), float(np.abs(kf["sigma"] - oref.sigma(2)).max())]
    T, key = vo.odometrize(g)
    To, keyo = ovo.odometrize(g)
    if i > 0:
        lg = vo.lastTrackLog()
        _, xi_g, rel_g = vo.lastFramePose() if not key else (0, vo.keyframeInfo(vo.keyframeCount() - 1)["xi"], vo.keyframeInfo(vo.keyframeCount() - 1)["rel_xi"])
        print("frame %2d key gpu/orc %d/%d  |T-To| %.2e  rel diff %.2e |rel| %.4f  iters gpu %s orc %s  kf-before: depth max %.2e frac>1e-3 %.4f age-mismatch %.5f sigma max %.2e"
              % (i, key, keyo, np.abs(T - To).max(), np.abs(rel_g - xo).max(), np.linalg.norm(xo[:3]), lg["n_iter"], lo["n_iter"], *dd))
        # per-iteration check along the GPU's trajectory
        xi = np.zeros(6, np.float32)
        worst = 0.0
        for l in range(3):
            for it in range(lg["n_iter"][l]):
                o = orc.optimize(oobj.gray(l), oref.gray(l), oref.depth(l), oref.sigma(l), oref.K(l), xi, l)
                nxt = orc.se3_concatenate(xi, o["xi_update"])
                dv = np.abs(lg["xi_after"][l][it] - nxt).max()
                if o["n_valid"] != lg["n_valid"][l][it] or dv > 1e-4:
                    print("     level %d it %d: n_valid gpu %d orc %d, |xi_after diff| %.2e, |upd| %.3e" % (l, it, lg["n_valid"][l][it], o["n_valid"], dv, np.abs(o["xi_update"]).max()))
                worst = max(worst, dv)
                xi = lg["xi_after"][l][it]
        print("     worst per-iteration xi diff %.2e" % worst)
vo.close()
