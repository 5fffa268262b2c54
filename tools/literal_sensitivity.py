#!/usr/bin/env python3
"""CPU only: how far the canonical arithmetic (deviation D8: shared reciprocals, fmaf chains; D2: SE(3) in double) moves the
reference's per-pixel decisions away from the LITERAL source expressions (oracle/dvo_oracle.h ORC_LIT_*).  Prints the table of
DESIGN.md §3; tests/test_oracle_literal.py asserts its bounds.  TEST INFRASTRUCTURE (uses the oracle)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "direct-visual-odometry_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import orc  # noqa: E402
from dvo_amd import synth  # noqa: E402


def gn_step_stats(obj, ref, xi_by_level, crop=True, mask=orc.LIT_ARITH):
    """one optimize() per level at the given pose, canonical vs literal: mask flips, n_valid, relative step / H difference"""
    rows = []
    for l in range(ref.levels):
        args = (obj.gray(l), ref.gray(l), ref.depth(l), ref.sigma(l), ref.K(l), xi_by_level[l], l)
        a = orc.optimize(*args, crop=crop, want_mask=True)
        with orc.literal(mask):
            b = orc.optimize(*args, crop=crop, want_mask=True)
        Ha = orc.upper_to_full(a["H"]); Hb = orc.upper_to_full(b["H"])
        nu = float(np.linalg.norm(a["xi_update"]))
        rows.append(dict(level=l, n_valid=a["n_valid"], flips=int((a["mask"] != b["mask"]).sum()), dn=int(b["n_valid"] - a["n_valid"]),
                         dH=float(np.abs(Ha - Hb).max() / max(np.abs(Ha).max(), 1e-30)),
                         dupd=float(np.linalg.norm(a["xi_update"] - b["xi_update"]) / max(nu, 1e-30)), upd=nu))
    return rows


def track_stats(obj, ref, crop=True, mask=orc.LIT_ARITH, nudge=0):
    xa, la = orc.track(obj, ref, crop=crop)
    with orc.literal(mask, nudge):
        xb, lb = orc.track(obj, ref, crop=crop)
    return xa, la, xb, lb


def main():
    K = synth.K_640
    g, d, s, _ = synth.sequence(6, seed=42, sigma_value=0.1)
    g, d, s = g.numpy(), d.numpy(), s.numpy()
    print("== SYN-640 sensor depth (Frame(g,d,s,K,4,1)), sigma 0.1: one GN step per level at the oracle's own per-level entry pose ==")
    tot = dict(px=0, flips=0)
    for k in range(5):
        ref = orc.OFrame(g[k], d[k], s[k], K, 4, 1); obj = orc.OFrame(g[k + 1], d[k + 1], s[k + 1], K, 4, 1)
        _, lg = orc.track(obj, ref)
        xi_lv = [np.zeros(6, np.float32)] + [lg["xi_after"][l][-1] for l in range(3)]
        for r in gn_step_stats(obj, ref, xi_lv):
            tot["px"] += r["n_valid"]; tot["flips"] += r["flips"]
            print("pair %d level %d: n_valid %6d  mask flips %3d  dn %+d  dH/|H| %.2e  |dupd|/|upd| %.2e (|upd| %.3g)" %
                  (k, r["level"], r["n_valid"], r["flips"], r["dn"], r["dH"], r["dupd"], r["upd"]))
    print("total: %d mask flips in %d contributing pixels" % (tot["flips"], tot["px"]))
    for name, m in (("ARITH", orc.LIT_ARITH), ("ARITH+SE3", orc.LIT_ARITH | orc.LIT_SE3)):
        for sg in (0.1, 0.5):
            gg, dd, ss, _ = synth.sequence(6, seed=42, sigma_value=sg)
            gg, dd, ss = gg.numpy(), dd.numpy(), ss.numpy()
            out = []
            for k in range(5):
                ref = orc.OFrame(gg[k], dd[k], ss[k], K, 4, 1); obj = orc.OFrame(gg[k + 1], dd[k + 1], ss[k + 1], K, 4, 1)
                xa, la, xb, lb = track_stats(obj, ref, mask=m)
                out.append((float(np.abs(xa - xb).max()), la["n_iter"], lb["n_iter"]))
            print("whole track(), %s literal vs canonical, sigma %.1f:" % (name, sg), ["%.2e %s %s" % o for o in out])
    out = []
    for k in range(5):
        ref = orc.OFrame(g[k], d[k], s[k], K, 4, 1); obj = orc.OFrame(g[k + 1], d[k + 1], s[k + 1], K, 4, 1)
        xa, la, xb, lb = track_stats(obj, ref, mask=0, nudge=1)
        out.append((float(np.abs(xa - xb).max()), la["n_iter"], lb["n_iter"]))
    print("whole track(), canonical vs canonical with a 1-ulp nudge of the first step, sigma 0.1:", ["%.2e %s %s" % o for o in out])


def mono_real():
    """the reference's own webcam frames (data/logicool0 excerpt) through the mono pipeline (track + map), canonical vs literal"""
    from real_data import frames_from_fixture
    fx = np.load(os.path.join(ROOT, "tests", "golden", "logicool0_excerpt.npz"))
    fr = frames_from_fixture(fx)
    K = np.asarray(fx["K"], np.float32)
    res = {}
    for name, m in (("canonical", 0), ("literal", orc.LIT_ARITH), ("literal+se3", orc.LIT_ARITH | orc.LIT_SE3)):
        with orc.literal(m):
            vo = orc.OVO(K, 640, 480, seed=int(fx["seed_vo"]))
            vo.set_initial_depth(fx["init_depth"], np.full_like(fx["init_depth"], 0.5))
            Ts, keys = [], []
            for f in fr:
                T, k = vo.odometrize(f); Ts.append(T); keys.append(k)
        res[name] = (np.array(Ts), np.array(keys))
    Ta, ka = res["canonical"]
    for name in ("literal", "literal+se3"):
        Tb, kb = res[name]
        dpos = np.linalg.norm(Ta[:, :3, 3] - Tb[:, :3, 3], axis=1)
        first = int(np.argmax(ka != kb)) if (ka != kb).any() else -1
        print("logicool0 mono, %s vs canonical: |dt| per frame %s ; first different keyframe decision at frame %d" %
              (name, " ".join("%.1e" % v for v in dpos), first))


if __name__ == "__main__":
    mono_real()
    main()
