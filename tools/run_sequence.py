#!/usr/bin/env python3
"""Run a dataset through the MI355X tracker: the loop of the reference's main.cpp:33-58 / test/sequence.cpp:10-23.

    python tools/run_sequence.py --tum /data/rgbd_dataset_freiburg1_xyz --fx 517.3 --fy 516.5 --cx 318.6 --cy 255.3
    python tools/run_sequence.py --list /data/KINECT_50MM            (the reference's "rgb depth" list files)
    python tools/run_sequence.py --synthetic 40 --out /tmp/syn       (writes a TUM-format directory first)

Frames are read with the built-in PNG reader, uploaded RAW (u8 RGB + u16 depth) and converted on the device
(dvo_vo_odometrize_depth_raw).  Writes <out>/trajectory.txt (TUM format) and prints ATE / RPE when ground truth exists.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "direct-visual-odometry_amd"))
import numpy as np
import torch  # noqa: F401  (first: see tests/conftest.py)

import dvo_amd as dvo
from dvo_amd import synth


def write_synthetic_tum(directory, n, seed=42):
    """A TUM-format directory (rgb/, depth/, rgb.txt, depth.txt, groundtruth.txt) rendered from the synthetic scene."""
    from PIL import Image
    os.makedirs(os.path.join(directory, "rgb"), exist_ok=True)
    os.makedirs(os.path.join(directory, "depth"), exist_ok=True)
    g, d, s, poses = synth.sequence(n, seed=seed)
    with open(os.path.join(directory, "rgb.txt"), "w") as fr, open(os.path.join(directory, "depth.txt"), "w") as fd, \
            open(os.path.join(directory, "groundtruth.txt"), "w") as fg:
        for f in (fr, fd, fg):
            f.write("# synthetic sequence (dvo_amd.synth)\n")
        for i in range(n):
            t = 1000.0 + i / 30.0
            g8 = np.clip(np.rint(g[i].numpy() * 255), 0, 255).astype(np.uint8)
            Image.fromarray(np.stack([g8, g8, g8], -1)).save(os.path.join(directory, "rgb", "%.6f.png" % t))
            Image.fromarray(np.clip(np.rint(d[i].numpy() * 5000), 0, 65535).astype(np.uint16)).save(os.path.join(directory, "depth", "%.6f.png" % t))
            fr.write("%.6f rgb/%.6f.png\n" % (t, t))
            fd.write("%.6f depth/%.6f.png\n" % (t, t))
            T = poses[i]
            R = T[:3, :3]
            qw = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
            qx, qy, qz = (R[2, 1] - R[1, 2]) / (4 * qw), (R[0, 2] - R[2, 0]) / (4 * qw), (R[1, 0] - R[0, 1]) / (4 * qw)
            fg.write("%.6f %.7f %.7f %.7f %.7f %.7f %.7f %.7f\n" % (t, T[0, 3], T[1, 3], T[2, 3], qx, qy, qz, qw))
    return poses


def quat_to_T(p7):
    tx, ty, tz, qx, qy, qz, qw = [float(v) for v in p7]
    T = np.eye(4)
    T[:3, :3] = [[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                 [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                 [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]]
    T[:3, 3] = [tx, ty, tz]
    return T


def run(ds, K, out_dir, sigma_override=None, max_frames=0):
    n = len(ds) if not max_frames else min(len(ds), max_frames)
    e0 = ds.entry(0)
    first = dvo.imread(e0["rgb"])
    h, w = first.shape[:2]
    cfg = dvo.default_config()
    vo = dvo.VisualOdometry(K, w, h, cfg=cfg)
    poses, stamps, gts = [np.eye(4)], [], []
    t_total = 0.0
    for i in range(n):
        e = ds.entry(i)
        rgb = dvo.imread(e["rgb"])
        d16 = dvo.imread(e["depth"])
        if d16.shape != (h, w):
            raise SystemExit("depth %s does not match the colour frame %dx%d (register it first)" % (d16.shape, w, h))
        t0 = time.perf_counter()
        T_rel = vo.odometrizeUsingDepthRaw(rgb, d16).astype(np.float64)
        t_total += time.perf_counter() - t0
        if i:
            poses.append(poses[-1] @ np.linalg.inv(T_rel))   # exp(xi_rel) maps reference-frame points into the new frame
        stamps.append(e["timestamp"])
        gts.append(e["gt"])
    vo.close()
    os.makedirs(out_dir, exist_ok=True)
    P = np.array(poses)
    dvo.write_tum_trajectory(os.path.join(out_dir, "trajectory.txt"), P, timestamps=np.array(stamps))
    res = {"frames": n, "fps_including_h2d_and_sync": n / t_total}
    gts = np.array(gts)
    if np.isfinite(gts).all():
        G = np.array([quat_to_T(p) for p in gts])
        G = np.array([np.linalg.inv(G[0]) @ g for g in G])
        res["ate_rmse_m"] = dvo.ate(P[:, :3, 3], G[:, :3, 3])[0]
        res["rpe_trans_m"], res["rpe_rot_rad"] = dvo.rpe(P, G, delta=1)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tum"); ap.add_argument("--list"); ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--out", default="gpurun_out/run_sequence")
    ap.add_argument("--fx", type=float, default=525.0); ap.add_argument("--fy", type=float, default=525.0)
    ap.add_argument("--cx", type=float, default=319.5); ap.add_argument("--cy", type=float, default=239.5)
    ap.add_argument("--max-frames", type=int, default=0)
    a = ap.parse_args()
    K = np.array([[a.fx, 0, a.cx], [0, a.fy, a.cy], [0, 0, 1]], np.float32)
    if a.synthetic:
        d = os.path.join(a.out, "tum_synthetic")
        write_synthetic_tum(d, a.synthetic)
        ds = dvo.Dataset(d, tum=True)
    elif a.tum:
        ds = dvo.Dataset(a.tum, tum=True)
    elif a.list:
        ds = dvo.Dataset(a.list)
    else:
        raise SystemExit("one of --tum, --list, --synthetic is required")
    print(run(ds, K, a.out, max_frames=a.max_frames))


if __name__ == "__main__":
    main()
