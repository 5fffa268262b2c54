#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (the reference holds no golden vectors: SURVEY.md §4).

Inputs are stored next to the expected outputs, so neither the GPU box nor a later run depends on the synthetic
generator reproducing bit-identical images.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "direct-visual-odometry_amd"), os.path.join(ROOT, "oracle")]
import orc
from dvo_amd import synth

K640 = synth.K_640
INV = np.float32(-2.0)


def main():
    g, d, s, poses = synth.sequence(4, seed=42, sigma_value=0.5)
    g, d, s = g.numpy(), d.numpy(), s.numpy()
    rng = np.random.RandomState(2024)
    ref = orc.OFrame(g[0], d[0], s[0], K640, 4, 1)
    obj = orc.OFrame(g[1], d[1], s[1], K640, 4, 1)

    # ---- Track::optimize on the 80x60 and 160x120 levels, with INVALID pixels, depth holes, both sigma clamps ----
    out = {}
    for level in (1, 2):
        rg, rd, rs, K = ref.gray(level), ref.depth(level), ref.sigma(level), ref.K(level)
        og = obj.gray(level)
        rg[rng.uniform(size=rg.shape) < 0.02] = INV
        og[rng.uniform(size=og.shape) < 0.02] = INV
        rd[rng.uniform(size=rd.shape) < 0.04] = 0.0
        rs[:, : rs.shape[1] // 2] = 0.004
        rs[:, rs.shape[1] // 2:] = 0.3
        for j, xi in enumerate([np.zeros(6, np.float32), np.array([0.004, -0.003, 0.002, 0.003, -0.002, 0.004], np.float32)]):
            o = orc.optimize(og, rg, rd, rs, K, xi, level, want_mask=True)
            key = "L%d_p%d_" % (level, j)
            out.update({key + "xi": xi, key + "H": o["H"], key + "g": o["g"], key + "sum_r2": o["sum_r2"],
                        key + "n_valid": o["n_valid"], key + "mask": np.packbits(o["mask"]), key + "xi_update": o["xi_update"],
                        key + "residual": o["residual"]})
        out.update({"L%d_obj" % level: og, "L%d_ref" % level: rg, "L%d_depth" % level: rd, "L%d_sigma" % level: rs, "L%d_K" % level: K})
    np.savez_compressed(os.path.join(HERE, "gn_step.npz"), **out)

    # ---- Tracker::track on pre-culled 160x120 frames (levels 3, culls 0 => 40x30, 80x60, 160x120) ----
    c = lambda a: orc.cull_image(a, 2)
    Kc = orc.cull_intrinsic(K640, 2)
    g0, d0, g1 = c(g[0]), c(d[0]), c(g[1])
    s0 = np.full_like(d0, 0.5)
    r = orc.OFrame(g0, d0, s0, Kc, 3, 0)
    o = orc.OFrame(g1, None, None, Kc, 3, 0)
    xi, log = orc.track(o, r)
    np.savez_compressed(os.path.join(HERE, "track.npz"), obj=g1, ref=g0, depth=d0, sigma=s0, K=Kc, xi=xi,
                        n_iter=np.array(log["n_iter"]), residual=np.concatenate(log["residual"]),
                        n_valid=np.concatenate(log["n_valid"]), xi_after=np.concatenate(log["xi_after"]))

    # ---- image ops: warpImage, gradiate, cullImage on an 80x60 level ----
    rg, rd, K = ref.gray(1), ref.depth(1), ref.K(1)
    rg[10:12, 10:14] = INV
    rg[20, 20:24] = 0.0
    rd[3:6, 4:9] = 0.0
    xi = np.array([0.01, -0.006, 0.008, 0.004, -0.003, 0.006], np.float32)
    np.savez_compressed(os.path.join(HERE, "image_ops.npz"), gray=rg, depth=rd, K=K, xi=xi,
                        warped=orc.warp_image(xi, rg, rd, K), gradx=orc.gradiate(rg, True), grady=orc.gradiate(rg, False),
                        cull1=orc.cull_image(rg, 1), cull2=orc.cull_image(rg, 2))

    # ---- mapping: propagate, regularize, Mapper::update at 160x120 ----
    kf0 = orc.OFrame(g[0], d[0], np.full_like(d[0], 0.5), K640, 3, 2, id=0)
    kf1 = orc.OFrame(g[2], d[2], np.full_like(d[2], 0.5), K640, 3, 2, id=2)
    ob = orc.OFrame(g[3], None, None, K640, 3, 2, id=3)
    T01 = np.linalg.inv(poses[2]) @ poses[0]
    T12 = np.linalg.inv(poses[3]) @ poses[2]
    xi1 = orc.se3_log(T01.astype(np.float32)) * np.float32(20)
    rel = orc.se3_log(T12.astype(np.float32)) * np.float32(20)
    kf1.set_pose(xi1, xi1)
    ob.set_pose(orc.se3_concatenate(xi1, rel), rel)
    top_d = kf1.depth(2) + rng.normal(0, 0.05, kf1.depth(2).shape).astype(np.float32)
    top_s = np.full_like(top_d, 0.3)
    age = (rng.uniform(size=top_d.shape) < 0.5).astype(np.float32)
    kf1.update_depth_sigma(top_d, top_s)
    kf1.set_age(age)
    Kt = kf1.K(2)
    pd, ps, pa = orc.propagate(top_d, top_s, age, rel, Kt)
    reg = orc.regularize(top_d, top_s)
    hist_gray = np.stack([kf0.gray(2), kf1.gray(2)])
    hist_xi = np.stack([kf0.xi, kf1.xi])
    obj_gray, obj_xi, obj_rel = ob.gray(2), ob.xi, ob.rel_xi
    valid = orc.mapper_update([kf0, kf1], ob, 11)
    np.savez_compressed(os.path.join(HERE, "mapping.npz"), depth=top_d, sigma=top_s, age=age, K=Kt, rel=rel,
                        prop_depth=pd, prop_sigma=ps, prop_age=pa, regularized=reg, hist_gray=hist_gray, hist_xi=hist_xi,
                        obj_gray=obj_gray, obj_xi=obj_xi, obj_rel=obj_rel, obj_id=3, seed=11,
                        upd_depth=kf1.depth(2), upd_sigma=kf1.sigma(2), upd_age=kf1.age(), valid=valid)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
