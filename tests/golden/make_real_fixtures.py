#!/usr/bin/env python3
"""Fixtures from the reference's OWN data files (the only externally authored inputs of this path): run here, where
/root/reference exists; the GPU box only sees the committed .npz files.

  logicool0_excerpt.npz  20 frames of data/logicool0 (frames 60, 63, ..., 117: the camera is static before that), converted as
      Loader::getNormalizedUndistortedImages does (src/core/loader.cpp:15-42,55-62: BGR2GRAY, 1/255, nearest remap with
      K = (780, 796, 378, 220), D = (-0.0462, 0.152, -0.00429, 0.0117, -0.0725), INVALID border), then culled by 4 to the
      160 x 120 top level of Frame(gray, K, 3, 2) (include/system/system.hpp:47) and stored as the u8 gray value plus an
      INVALID bit mask (a 640 x 480 frame that decimates to exactly this level is rebuilt by pixel repetition).
      Expected outputs = the CPU oracle's VisualOdometry::odometrize over those frames (keyframe flags, world poses, age
      maps, keyframe depth maps).  ORACLE-DERIVED: parity unpinned (the reference cannot be built here).
  kinect50mm_ir_depth.npz, kinect1deg_ir_depth.npz  four registered (IR, depth) pairs of data/KINECT_50MM (translation, ~50 mm per
      frame) and data/KINECT_1DEG (rotation, ~1 degree per frame), decimated by 2: inputs only.
  kinect50mm_excerpt.npz  one data/KINECT_50MM depth PNG (u16, 512 x 424) decimated by 2 and the matching rgb PNG sampled
      to the same size: inputs only; the expected k_ingest outputs are integer arithmetic restated in numpy by the test.

Only pixel DATA is stored, never reference source text.    python tests/golden/make_real_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "direct-visual-odometry_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import dvo_amd as dvo   # host-side PNG reader only (no GPU needed)
import orc
from real_data import K_LOGICOOL, D_LOGICOOL, frames_from_fixture, undistort_nearest_np, bgr2gray_u8

REF = "/root/reference/data"
FIRST, STRIDE, COUNT = 60, 3, 20
SEED_DEPTH, SEED_VO = 5, 3
MAP_FRAMES = [4, 9, 14, 19]


def main():
    # ---- logicool0 ----
    gray_u8, invalid = [], []
    for k in range(COUNT):
        im = dvo.imread(os.path.join(REF, "logicool0", "%04d.png" % (FIRST + k * STRIDE)))   # [480, 640, 3] R, G, B
        g8 = bgr2gray_u8(im)
        g8u, inv = undistort_nearest_np(g8, K_LOGICOOL, D_LOGICOOL)                           # nearest remap of the u8 values
        gray_u8.append(g8u[::4, ::4].copy())
        invalid.append(inv[::4, ::4].copy())
    gray_u8 = np.stack(gray_u8); invalid = np.stack(invalid)
    rng = np.random.RandomState(SEED_DEPTH)   # stands in for cv::randn(depth, 1.5, 0.5), max(depth, 0.5) (frame.hpp:17-21), D6
    init_depth = np.maximum(rng.normal(1.5, 0.5, (120, 160)), 0.5).astype(np.float32)
    init_sigma = np.full_like(init_depth, 0.5)
    fx = dict(gray_u8=gray_u8, invalid=np.packbits(invalid), init_depth=init_depth, seed_vo=SEED_VO,
              first=FIRST, stride=STRIDE, K=K_LOGICOOL, D=D_LOGICOOL)
    frames = frames_from_fixture(fx)
    vo = orc.OVO(K_LOGICOOL, 640, 480, seed=SEED_VO)
    vo.set_initial_depth(init_depth, init_sigma)
    keys, poses, ages, depths, sigmas, valid, nkf = [], [], [], [], [], [], []
    for g in frames:
        T, key = vo.odometrize(g)
        kf = vo.keyframe(vo.keyframe_count() - 1)
        keys.append(key); poses.append(T); nkf.append(vo.keyframe_count()); valid.append(vo.last_valid_updates())
        ages.append(kf.age().astype(np.uint8))
        if len(keys) - 1 in MAP_FRAMES:   # the depth / sigma maps of the newest keyframe are kept for a few frames only (size)
            depths.append(kf.depth(2)); sigmas.append(kf.sigma(2))
    out = dict(fx)
    out.update(key=np.array(keys), T_world=np.stack(poses).astype(np.float32), n_keyframes=np.array(nkf), valid_updates=np.array(valid),
               age=np.stack(ages), map_frames=np.array(MAP_FRAMES), depth=np.stack(depths), sigma=np.stack(sigmas))
    np.savez_compressed(os.path.join(HERE, "logicool0_excerpt.npz"), **out)
    print("logicool0: keyframes", int(np.sum(keys)), "valid updates", valid, "max age", int(np.max(ages)))

    # ---- KINECT_50MM ----
    d16 = dvo.imread(os.path.join(REF, "KINECT_50MM", "depth05.png"))            # [424, 512] u16
    rgba = dvo.imread(os.path.join(REF, "KINECT_50MM", "rgb05.png"))             # [1080, 1920, 4] u8
    d = d16[::2, ::2].copy()                                                      # 212 x 256
    c = rgba[::5, ::7][:d.shape[0], :d.shape[1]].copy()
    np.savez_compressed(os.path.join(HERE, "kinect50mm_excerpt.npz"), depth16=d, rgba=c)
    print("kinect: depth", d.shape, d.dtype, "zeros %.3f" % (d == 0).mean(), "rgba", c.shape)

    # ---- KINECT_50MM as a sensor-depth sequence: the IR image is registered with the depth image (same sensor), so four
    #      (ir, depth) pairs are a real RGB-D-like input with real holes and noise for the odometrizeUsingDepth path ----
    #      KINECT_1DEG is the same sensor turning ~1 degree per frame: the rotation-dominant counterpart (the reference's README: rotation
    #      tracking "did not work well") ----
    for name, out in (("KINECT_50MM", "kinect50mm_ir_depth.npz"), ("KINECT_1DEG", "kinect1deg_ir_depth.npz")):
        irs, deps = [], []
        for k in (1, 2, 3, 4):
            ir = dvo.imread(os.path.join(REF, name, "ir%02d.png" % k))      # [424, 512] u16
            dd = dvo.imread(os.path.join(REF, name, "depth%02d.png" % k))
            irs.append(np.minimum(ir[::2, ::2] >> 5, 255).astype(np.uint8))           # 256 x 212 u8 gray
            deps.append(dd[::2, ::2].copy())
        np.savez_compressed(os.path.join(HERE, out), gray_u8=np.stack(irs), depth16=np.stack(deps))
        print(name, "ir/depth sequence:", np.stack(irs).shape, "holes %.3f" % (np.stack(deps) == 0).mean())


if __name__ == "__main__":
    main()
