"""Helpers shared by tests/golden/make_real_fixtures.py and the tests that read its fixtures: the integer / index arithmetic of
src/core/loader.cpp restated in numpy (independent of libdvo), and the rebuild of full-size frames from the stored excerpt."""
import numpy as np

K_LOGICOOL = np.array([[780, 0, 378], [0, 796, 220], [0, 0, 1]], np.float32)          # src/core/loader.cpp:17
D_LOGICOOL = np.array([-0.0462, 0.152, -0.00429, 0.0117, -0.0725], np.float32)        # src/core/loader.cpp:18


def bgr2gray_u8(rgb):
    """cv::cvtColor(COLOR_BGR2GRAY) on 8-bit data: fixed-point luma, R 4899 G 9617 B 1868, >> 14 (file order R, G, B[, A])."""
    if rgb.ndim == 2:
        return rgb.astype(np.uint8)
    r, g, b = (rgb[..., i].astype(np.uint32) for i in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)


def undistort_nearest_np(img, K, D):
    """cv::initUndistortRectifyMap(K, D, I, K) + cv::remap(INTER_NEAREST, BORDER_CONSTANT): returns (remapped, invalid mask)."""
    h, w = img.shape
    K = np.asarray(K, np.float64); D = np.asarray(D, np.float64)
    v, u = np.mgrid[0:h, 0:w].astype(np.float64)
    x = (u - K[0, 2]) / K[0, 0]
    y = (v - K[1, 2]) / K[1, 1]
    r2 = x * x + y * y
    rad = 1 + r2 * (D[0] + r2 * (D[1] + r2 * D[4]))
    xd = x * rad + 2 * D[2] * x * y + D[3] * (r2 + 2 * x * x)
    yd = y * rad + D[2] * (r2 + 2 * y * y) + 2 * D[3] * x * y
    mx = np.rint((xd * K[0, 0] + K[0, 2]).astype(np.float32)).astype(np.int64)
    my = np.rint((yd * K[1, 1] + K[1, 2]).astype(np.float32)).astype(np.int64)
    ok = (mx >= 0) & (mx < w) & (my >= 0) & (my < h)
    out = np.zeros_like(img)
    out[ok] = img[my[ok], mx[ok]]
    return out, ~ok


def frames_from_fixture(fx):
    """[n][480][640] float32 frames whose Frame(gray, K, 3, 2) top level is exactly the stored 160 x 120 excerpt:
    gray = u8 * (1/255) (loader.cpp:61), INVALID (-2) on the undistortion border, every pixel repeated 4 x 4."""
    g8 = np.asarray(fx["gray_u8"])
    n, h, w = g8.shape
    inv = np.unpackbits(np.asarray(fx["invalid"]))[: n * h * w].reshape(n, h, w).astype(bool)
    g = g8.astype(np.float32) * np.float32(1.0 / 255.0)
    g[inv] = np.float32(-2.0)
    return np.repeat(np.repeat(g, 4, axis=1), 4, axis=2)


def ingest_np(rgb, d16, scale=1.0 / 5000.0):
    """loader.cpp:137-147 + transform.cpp:60-76 on raw frames: what k_ingest must produce, bit for bit."""
    gray = bgr2gray_u8(rgb).astype(np.float32) * np.float32(1.0 / 255.0)
    depth = d16.astype(np.float32) * np.float32(scale)
    sigma = np.where(d16 > 0, np.float32(0.1), np.float32(1.0)).astype(np.float32)
    gray = np.where(d16 == 0, np.float32(-2.0), gray).astype(np.float32)
    return gray, depth, sigma


def write_keyframe_store(path, K, width, height, keyframes, latest_id, levels=3, culls=2):
    """The binary keyframe store of dvo_vo_save / dvo_vo_load (csrc/dvo_store.cpp) written from oracle frames (orc.OFrame):
    lets a test put the ORACLE's FrameHistory into a dvo_vo handle, so that one odometrize() call can be compared given
    identical inputs (the free-running pipelines drift apart: DESIGN.md §6, sensitivity of the mapping to 1e-8 pose differences)."""
    import struct
    with open(path, "wb") as f:
        f.write(b"DVOKF01\0")
        f.write(struct.pack("<8i", 1, width, height, levels, culls, len(keyframes), latest_id, 0))
        f.write(np.asarray(K, np.float32).reshape(9).tobytes())
        for kf in keyframes:
            f.write(struct.pack("<2i", int(kf.c.id), int(kf.c.ref_index)))
            f.write(np.asarray(kf.xi, np.float32).tobytes())
            f.write(np.asarray(kf.rel_xi, np.float32).tobytes())
            for l in range(levels):
                f.write(np.ascontiguousarray(kf.gray(l), np.float32).tobytes())
            top = levels - 1
            f.write(np.ascontiguousarray(kf.depth(top), np.float32).tobytes())
            f.write(np.ascontiguousarray(kf.sigma(top), np.float32).tobytes())
            f.write(np.ascontiguousarray(kf.age(), np.float32).tobytes())
