"""ctypes binding of libdvo.so (include/dvo.h): the MI355X-native direct-VO hot path.

The names mirror the reference's interface for this path (include/system/system.hpp, include/track/*.hpp,
include/map/implement.hpp, include/core/{convert,transform}.hpp) so the parity tests read like the
reference's own demos.  There is NO CPU fallback: if lib/libdvo.so is missing, or no GPU is visible when a
compute entry point is called, this raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DVO_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "lib", "libdvo.so")  # (override: A/B runs of two builds)
INVALID = np.float32(-2.0)
MAX_LEVELS = 8
MAX_ITERATIONS = 32
FP = C.POINTER(C.c_float)


class DvoError(RuntimeError):
    pass


# status codes of include/dvo.h
DVO_OK, DVO_ERR_BAD_ARGUMENT, DVO_ERR_HIP, DVO_ERR_NO_DEVICE, DVO_ERR_NO_VALID_PIXELS, DVO_ERR_NOT_READY, DVO_ERR_OUT_OF_MEMORY = range(7)


class Config(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("min_update", C.c_float), ("min_residual", C.c_float),
                ("fixed_iterations", C.c_int), ("crop_enable", C.c_int), ("step_default", C.c_float),
                ("step_level1", C.c_float), ("step_level2", C.c_float), ("sigma_min", C.c_float),
                ("sigma_max", C.c_float), ("min_depth", C.c_float), ("keyframe_min_translation", C.c_float),
                ("keyframe_max_frames", C.c_int), ("rng_seed", C.c_uint32), ("device", C.c_int),
                ("stream", C.c_void_p), ("profile", C.c_int), ("gn_pixels_per_thread", C.c_int),
                ("gn_use_lds_patch", C.c_int), ("gn_gather_group", C.c_int), ("track_streams", C.c_int), ("track_adaptive", C.c_int), ("track_fused_tiles", C.c_int),
                ("track_single_launch", C.c_int)]


class TrackLog(C.Structure):
    _fields_ = [("levels", C.c_int), ("n_iter", C.c_int * MAX_LEVELS),
                ("residual", (C.c_float * MAX_ITERATIONS) * MAX_LEVELS),
                ("update_norm", (C.c_float * MAX_ITERATIONS) * MAX_LEVELS),
                ("n_valid", (C.c_int * MAX_ITERATIONS) * MAX_LEVELS),
                ("xi_after", ((C.c_float * 6) * MAX_ITERATIONS) * MAX_LEVELS),
                ("xi_update", ((C.c_float * 6) * MAX_ITERATIONS) * MAX_LEVELS)]

    def to_dict(self):
        L = self.levels
        out = dict(n_iter=[self.n_iter[l] for l in range(L)], residual=[], upd_norm=[], n_valid=[], xi_after=[], xi_update=[])
        for l in range(L):
            n = self.n_iter[l]
            out["xi_update"].append(np.array([self.xi_update[l][i][:] for i in range(n)], np.float32).reshape(n, 6))
            out["residual"].append(np.array(self.residual[l][:n], np.float32))
            out["upd_norm"].append(np.array(self.update_norm[l][:n], np.float32))
            out["n_valid"].append(np.array(self.n_valid[l][:n], np.int32))
            out["xi_after"].append(np.array([self.xi_after[l][i][:] for i in range(n)], np.float32).reshape(n, 6))
        return out


class GnResult(C.Structure):
    _fields_ = [("H", C.c_double * 21), ("g", C.c_double * 6), ("sum_r2", C.c_double), ("n_valid", C.c_int),
                ("xi_update", C.c_float * 6), ("residual", C.c_float), ("xi_next", C.c_float * 6)]


class GnProfile(C.Structure):
    _fields_ = [("gn_ms", C.c_double), ("gn_launches", C.c_uint64), ("gn_pixels", C.c_uint64),
                ("gn_iterations", C.c_uint64)]


class MonoStats(C.Structure):
    _fields_ = [("frames", C.c_int), ("keyframes_created", C.c_int), ("ring_keyframes", C.c_int),
                ("valid_updates_last_frame", C.c_int), ("clamped_pixels", C.c_int)]


class MapProfile(C.Structure):
    _fields_ = [("frames", C.c_uint64), ("depth_update_ms", C.c_double), ("regularize_ms", C.c_double), ("propagate_ms", C.c_double),
                ("update_window_pixels", C.c_uint64), ("map_pixels", C.c_uint64)]


EXPORTS = [
    "dvo_config_default", "dvo_version", "dvo_status_string", "dvo_last_error", "dvo_device_count",
    "dvo_vo_create", "dvo_vo_destroy", "dvo_vo_set_initial_depth", "dvo_vo_init_keyframe", "dvo_vo_odometrize",
    "dvo_vo_odometrize_depth", "dvo_vo_odometrize_raw", "dvo_vo_keyframe_count", "dvo_vo_keyframe_info", "dvo_vo_keyframe_get",
    "dvo_vo_last_frame_pose", "dvo_vo_last_valid_updates", "dvo_vo_last_track_log", "dvo_debug_persist_timeline",
    "dvo_batch_create", "dvo_batch_destroy", "dvo_batch_push_device", "dvo_batch_push_host", "dvo_batch_last_poses",
    "dvo_batch_prefetch_device", "dvo_batch_copy_poses_device", "dvo_batch_last_track_log", "dvo_batch_synchronize", "dvo_batch_profile", "dvo_batch_probe_gn", "dvo_shard_range", "dvo_batch_gather_poses_rccl",
    "dvo_batch_push_raw_device", "dvo_batch_prefetch_raw_device", "dvo_batch_push_raw_host", "dvo_batch_odometrize_raw_device",
    "dvo_batch_odometrize_host", "dvo_batch_odometrize_raw_host",
    "dvo_batch_create_mono", "dvo_batch_set_initial_depth", "dvo_batch_set_initial_depth_device", "dvo_batch_odometrize_device",
    "dvo_batch_world_poses", "dvo_batch_copy_world_poses_device", "dvo_batch_keyframe_get", "dvo_batch_mono_stats", "dvo_batch_profile_mapping",
    "dvo_op_cull_image", "dvo_op_gradient", "dvo_op_warp_image", "dvo_op_pyramid", "dvo_op_gn_step", "dvo_op_track",
    "dvo_op_propagate", "dvo_op_regularize", "dvo_op_depth_update", "dvo_op_se3_exp", "dvo_op_se3_log",
    "dvo_op_se3_concatenate",
    "dvo_png_info", "dvo_png_read", "dvo_dataset_open_tum", "dvo_dataset_open_list", "dvo_dataset_size", "dvo_dataset_entry",
    "dvo_dataset_close", "dvo_op_ingest", "dvo_vo_odometrize_depth_raw", "dvo_op_undistort",
    "dvo_eval_ate", "dvo_eval_rpe", "dvo_pose_inverse", "dvo_traj_write_tum",
    "dvo_vo_save", "dvo_vo_load", "dvo_vo_set_history_limit", "dvo_op_visualize", "dvo_ppm_write",
    "dvo_selftest_reciprocal", "dvo_selftest_sqrt", "dvo_selftest_division", "dvo_selftest_trig",
]

_lib = None


def lib():
    """Load lib/libdvo.so.  Raises (never falls back) when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DvoError("libdvo.so not built: run `make -C direct-visual-odometry_amd` (or __graft_entry__.build()); "
                           "there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.dvo_version.restype = C.c_char_p
        L.dvo_status_string.restype = C.c_char_p
        L.dvo_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def _check(st):
    if st != 0:
        L = lib()
        raise DvoError("%s: %s" % (L.dvo_status_string(st).decode(), L.dvo_last_error().decode()))


def default_config(**kw):
    c = Config()
    lib().dvo_config_default(C.byref(c))
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def fp(a):
    return a.ctypes.data_as(FP)


def device_count():
    return lib().dvo_device_count()


# ------------------------------------------------------------------ math::se3 (device evaluated)
class se3:
    @staticmethod
    def exp(xi, dev=0):
        xi = f32(xi); T = np.zeros(16, np.float32)
        _check(lib().dvo_op_se3_exp(dev, fp(xi), fp(T)))
        return T.reshape(4, 4)

    @staticmethod
    def log(T, dev=0):
        T = f32(T).reshape(16); xi = np.zeros(6, np.float32)
        _check(lib().dvo_op_se3_log(dev, fp(T), fp(xi)))
        return xi

    @staticmethod
    def concatenate(a, b, dev=0):
        a = f32(a); b = f32(b); o = np.zeros(6, np.float32)
        _check(lib().dvo_op_se3_concatenate(dev, fp(a), fp(b), fp(o)))
        return o


# ------------------------------------------------------------------ Convert / Transform
class Convert:
    @staticmethod
    def cullImage(src, times, dev=0):
        src = f32(src); h, w = src.shape
        dst = np.zeros((h >> times, w >> times), np.float32)
        _check(lib().dvo_op_cull_image(dev, fp(src), w, h, times, fp(dst)))
        return dst

    @staticmethod
    def gradiate(img, x, dev=0):
        img = f32(img); h, w = img.shape
        out = np.zeros((h, w), np.float32)
        _check(lib().dvo_op_gradient(dev, fp(img), w, h, 1 if x else 0, fp(out)))
        return out


class Transform:
    @staticmethod
    def warpImage(xi, gray, depth, K, dev=0):
        xi = f32(xi); gray = f32(gray); depth = f32(depth); K = f32(K).reshape(9)
        h, w = gray.shape
        out = np.zeros((h, w), np.float32)
        _check(lib().dvo_op_warp_image(dev, fp(xi), fp(gray), fp(depth), w, h, fp(K), fp(out)))
        return out


def pyramid(gray, depth, sigma, levels, culls, dev=0):
    """Frame(gray, depth, sigma, K, levels, culls) pyramids (frame.cpp:16-37): lists of per-level arrays."""
    gray = f32(gray); h, w = gray.shape
    d = f32(depth) if depth is not None else None
    s = f32(sigma) if sigma is not None else None
    shapes = [((h >> culls) >> (levels - 1 - i), (w >> culls) >> (levels - 1 - i)) for i in range(levels)]
    go = [np.zeros(sh, np.float32) for sh in shapes]
    do = [np.zeros(sh, np.float32) for sh in shapes]
    so = [np.zeros(sh, np.float32) for sh in shapes]
    arr = lambda lst: (FP * levels)(*[fp(a) for a in lst])
    _check(lib().dvo_op_pyramid(dev, fp(gray), fp(d) if d is not None else None, fp(s) if s is not None else None,
                                w, h, levels, culls, arr(go), arr(do), arr(so)))
    return go, (do if d is not None else None), (so if s is not None else None)


# ------------------------------------------------------------------ Track
def optimize(obj_gray, ref_gray, ref_depth, ref_sigma, K, xi, level, cfg=None, want_mask=False, dev=0):
    """Track::optimize (src/track/optimize.cpp:10-99): one Gauss-Newton step on one level."""
    obj_gray = f32(obj_gray); ref_gray = f32(ref_gray); ref_depth = f32(ref_depth); ref_sigma = f32(ref_sigma)
    K = f32(K).reshape(9); xi = f32(xi)
    h, w = ref_gray.shape
    out = GnResult()
    mask = np.zeros((h, w), np.uint8) if want_mask else None
    _check(lib().dvo_op_gn_step(dev, C.byref(cfg) if cfg is not None else None, fp(obj_gray), fp(ref_gray),
                                fp(ref_depth), fp(ref_sigma), w, h, fp(K), fp(xi), level, C.byref(out),
                                mask.ctypes.data_as(C.c_void_p) if want_mask else None))
    res = dict(H=np.array(out.H[:]), g=np.array(out.g[:]), sum_r2=out.sum_r2, n_valid=out.n_valid,
               xi_update=np.array(out.xi_update[:], np.float32), residual=np.float32(out.residual),
               xi_next=np.array(out.xi_next[:], np.float32))
    if want_mask:
        res["mask"] = mask
    return res


def track(obj_gray, ref_gray, ref_depth, ref_sigma, K, levels, culls, cfg=None, dev=0):
    """Tracker::track (src/track/tracker.cpp:22-85) on full-resolution frames."""
    obj_gray = f32(obj_gray); ref_gray = f32(ref_gray); ref_depth = f32(ref_depth); ref_sigma = f32(ref_sigma)
    K = f32(K).reshape(9)
    h, w = ref_gray.shape
    xi = np.zeros(6, np.float32); log = TrackLog()
    _check(lib().dvo_op_track(dev, C.byref(cfg) if cfg is not None else None, fp(obj_gray), fp(ref_gray),
                              fp(ref_depth), fp(ref_sigma), w, h, fp(K), levels, culls, fp(xi), C.byref(log)))
    return xi, log.to_dict()


# ------------------------------------------------------------------ Map::Implement
class Implement:
    @staticmethod
    def propagate(ref_depth, ref_sigma, ref_age, xi, K, dev=0):
        d = f32(ref_depth); s = f32(ref_sigma); a = f32(ref_age); xi = f32(xi); K = f32(K).reshape(9)
        h, w = d.shape
        od = np.zeros_like(d); os_ = np.zeros_like(d); oa = np.zeros_like(d)
        _check(lib().dvo_op_propagate(dev, fp(d), fp(s), fp(a), w, h, fp(xi), fp(K), fp(od), fp(os_), fp(oa)))
        return od, os_, oa

    @staticmethod
    def regularize(depth, sigma, dev=0):
        d = f32(depth); s = f32(sigma); h, w = d.shape
        out = np.zeros_like(d)
        _check(lib().dvo_op_regularize(dev, fp(d), fp(s), w, h, fp(out)))
        return out


def mapper_update(hist_gray, hist_xi, obj_gray, obj_xi, obj_rel_xi, obj_id, K, ref_depth, ref_sigma, ref_age,
                  cfg=None, dev=0):
    """Mapper::update (src/map/mapper.cpp:76-137).  Returns updated (depth, sigma, age, valid_updates)."""
    n = len(hist_gray)
    grays = [f32(g) for g in hist_gray]
    h, w = grays[0].shape
    garr = (FP * n)(*[fp(g) for g in grays])
    hx = f32(np.asarray(hist_xi).reshape(n, 6))
    og = f32(obj_gray); ox = f32(obj_xi); orx = f32(obj_rel_xi); K = f32(K).reshape(9)
    d = f32(ref_depth).copy(); s = f32(ref_sigma).copy(); a = f32(ref_age).copy()
    v = C.c_int(0)
    _check(lib().dvo_op_depth_update(dev, C.byref(cfg) if cfg is not None else None, n, garr, fp(hx), fp(og), fp(ox),
                                     fp(orx), int(obj_id), fp(K), w, h, fp(d), fp(s), fp(a), C.byref(v)))
    return d, s, a, v.value


# ------------------------------------------------------------------ Core::Loader (dataset front-end) and evaluation
def imread(path):
    """cv::imread(path, IMREAD_UNCHANGED) for PNGs: uint8 / uint16 array [H, W] or [H, W, C] in file channel order (RGB)."""
    w = C.c_int(); h = C.c_int(); ch = C.c_int(); bd = C.c_int()
    _check(lib().dvo_png_info(path.encode(), C.byref(w), C.byref(h), C.byref(ch), C.byref(bd)))
    shape = (h.value, w.value) if ch.value == 1 else (h.value, w.value, ch.value)
    out = np.zeros(shape, np.uint8 if bd.value == 8 else np.uint16)
    _check(lib().dvo_png_read(path.encode(), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.nbytes)))
    return out


class Dataset:
    """TUM RGB-D directory or one of the reference's list files (include/core/loader.hpp:28-52,77-105)."""

    def __init__(self, directory, list_file=None, tum=False, max_dt=0.02):
        self._p = C.c_void_p()
        if tum:
            _check(lib().dvo_dataset_open_tum(directory.encode(), C.c_double(max_dt), C.byref(self._p)))
        else:
            _check(lib().dvo_dataset_open_list(directory.encode(), list_file.encode() if list_file else None, C.byref(self._p)))

    def __len__(self):
        return lib().dvo_dataset_size(self._p)

    def entry(self, i):
        t = C.c_double(); a = C.create_string_buffer(1024); b = C.create_string_buffer(1024); gt = np.zeros(7, np.float32)
        _check(lib().dvo_dataset_entry(self._p, i, C.byref(t), a, b, 1024, fp(gt)))
        return dict(timestamp=t.value, rgb=a.value.decode(), depth=b.value.decode(), gt=gt)

    def close(self):
        if self._p:
            lib().dvo_dataset_close(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ingest(rgb, depth16=None, depth_scale=1.0 / 5000.0, sigma_valid=0.1, sigma_invalid=1.0, invalidate_gray=True, dev=0):
    """k_ingest: raw u8 gray/RGB(A) (+ u16 depth) -> float gray [0,1] (+ depth [m], sigma)."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w = rgb.shape[:2]
    ch = 1 if rgb.ndim == 2 else rgb.shape[2]
    gray = np.zeros((h, w), np.float32)
    if depth16 is None:
        _check(lib().dvo_op_ingest(dev, rgb.ctypes.data_as(C.c_void_p), ch, None, w, h, C.c_float(depth_scale), C.c_float(sigma_valid),
                                   C.c_float(sigma_invalid), 0, fp(gray), None, None))
        return gray
    d16 = np.ascontiguousarray(depth16, np.uint16)
    depth = np.zeros((h, w), np.float32); sigma = np.zeros((h, w), np.float32)
    _check(lib().dvo_op_ingest(dev, rgb.ctypes.data_as(C.c_void_p), ch, d16.ctypes.data_as(C.c_void_p), w, h, C.c_float(depth_scale),
                               C.c_float(sigma_valid), C.c_float(sigma_invalid), 1 if invalidate_gray else 0, fp(gray), fp(depth), fp(sigma)))
    return gray, depth, sigma


def undistort(src, K, D, dev=0):
    """Loader::getNormalizedUndistortedImages (src/core/loader.cpp:15-42)."""
    src = f32(src); K = f32(K).reshape(9); D = f32(D).reshape(5)
    h, w = src.shape
    out = np.zeros_like(src)
    _check(lib().dvo_op_undistort(dev, fp(src), w, h, fp(K), fp(D), fp(out)))
    return out


VIS_GRAY, VIS_DEPTH, VIS_SIGMA, VIS_AGE, VIS_GRADIENT = range(5)


def visualize(mode, a, b=None, dev=0):
    """Draw::visualize{Gray,Depth,Sigma,Age,Gradient} (src/core/draw.cpp:7-100) -> uint8 RGB [H, W, 3]."""
    a = f32(a); h, w = a.shape
    bb = f32(b) if b is not None else None
    out = np.zeros((h, w, 3), np.uint8)
    _check(lib().dvo_op_visualize(dev, int(mode), fp(a), fp(bb) if bb is not None else None, w, h, out.ctypes.data_as(C.c_void_p)))
    return out


def selftest_reciprocal(dev=0):
    """All 2^32 float patterns through the kernels' reciprocal vs the IEEE division: (fast-path inputs, mismatches, first bad bits)."""
    n = C.c_uint64(); bad = C.c_uint64(); first = C.c_uint32()
    _check(lib().dvo_selftest_reciprocal(dev, C.byref(n), C.byref(bad), C.byref(first)))
    return n.value, bad.value, first.value


def selftest_trig(dev=0):
    """The device's SE(3) sin / cos / atan2 kernels vs the math library on 2^24 arguments: largest relative differences (sin, cos, atan2)."""
    a = C.c_double(); b = C.c_double(); c = C.c_double(); n = C.c_uint64()
    _check(lib().dvo_selftest_trig(dev, C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
    return a.value, b.value, c.value, n.value


def selftest_sqrt(dev=0):
    """Every float in [2^-100, 2^100] through the regularize kernels' short square root vs sqrtf: (inputs, mismatches, first bad bits)."""
    n = C.c_uint64(); bad = C.c_uint64(); first = C.c_uint32()
    _check(lib().dvo_selftest_sqrt(dev, C.byref(n), C.byref(bad), C.byref(first)))
    return n.value, bad.value, first.value


def selftest_division(b_first=0, b_stride=1, b_count=1 << 23, dev=0):
    """The regularize kernels' short division vs the IEEE quotient for b_count mantissas of b times all 2^23 of a: (pairs, mismatches, first bad pair)."""
    n = C.c_uint64(); bad = C.c_uint64(); first = C.c_uint64()
    _check(lib().dvo_selftest_division(dev, C.c_uint32(b_first), C.c_uint32(b_stride), C.c_uint32(b_count), C.byref(n), C.byref(bad), C.byref(first)))
    return n.value, bad.value, first.value


def write_ppm(path, rgb):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    _check(lib().dvo_ppm_write(path.encode(), rgb.ctypes.data_as(C.c_void_p), rgb.shape[1], rgb.shape[0]))


def ate(est_xyz, gt_xyz, with_scale=False):
    """Absolute trajectory error (RMSE after Horn alignment).  Returns (rmse, R, t, scale)."""
    e = f32(est_xyz).reshape(-1, 3); g = f32(gt_xyz).reshape(-1, 3)
    rm = C.c_double(); R = np.zeros(9); t = np.zeros(3); s = C.c_double()
    _check(lib().dvo_eval_ate(e.shape[0], fp(e), fp(g), 1 if with_scale else 0, C.byref(rm), R.ctypes.data_as(C.POINTER(C.c_double)),
                              t.ctypes.data_as(C.POINTER(C.c_double)), C.byref(s)))
    return rm.value, R.reshape(3, 3), t, s.value


def rpe(est_T, gt_T, delta=1):
    e = f32(est_T).reshape(-1, 16); g = f32(gt_T).reshape(-1, 16)
    a = C.c_double(); b = C.c_double()
    _check(lib().dvo_eval_rpe(e.shape[0], fp(e), fp(g), delta, C.byref(a), C.byref(b)))
    return a.value, b.value


def pose_inverse(T):
    T = f32(T).reshape(16); o = np.zeros(16, np.float32)
    _check(lib().dvo_pose_inverse(fp(T), fp(o)))
    return o.reshape(4, 4)


def write_tum_trajectory(path, T, timestamps=None):
    T = f32(T).reshape(-1, 16)
    ts = np.ascontiguousarray(timestamps, np.float64) if timestamps is not None else None
    _check(lib().dvo_traj_write_tum(path.encode(), T.shape[0], ts.ctypes.data_as(C.POINTER(C.c_double)) if ts is not None else None, fp(T)))


# ------------------------------------------------------------------ System::VisualOdometry
class VisualOdometry:
    """System::VisualOdometry (include/system/system.hpp:12-104)."""

    def __init__(self, K, width, height, cfg=None):
        K = f32(K).reshape(9)
        self.width, self.height = width, height
        self._p = C.c_void_p()
        _check(lib().dvo_vo_create(fp(K), width, height, C.byref(cfg) if cfg is not None else None, C.byref(self._p)))

    def close(self):
        if self._p:
            lib().dvo_vo_destroy(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def setInitialDepth(self, depth, sigma):
        d = f32(depth); s = f32(sigma)
        _check(lib().dvo_vo_set_initial_depth(self._p, fp(d), fp(s)))

    def initKeyframe(self, gray, depth, sigma):
        g = f32(gray); d = f32(depth); s = f32(sigma)
        _check(lib().dvo_vo_init_keyframe(self._p, fp(g), fp(d), fp(s)))

    def odometrize(self, gray):
        g = f32(gray); T = np.zeros(16, np.float32); key = C.c_int(0)
        _check(lib().dvo_vo_odometrize(self._p, fp(g), fp(T), C.byref(key)))
        return T.reshape(4, 4), bool(key.value)

    def odometrizeRaw(self, rgb_u8):
        """odometrize() fed with a raw uint8 frame [H, W] or [H, W, C] (converted on the device)."""
        rgb = np.ascontiguousarray(rgb_u8, np.uint8)
        ch = 1 if rgb.ndim == 2 else rgb.shape[2]
        T = np.zeros(16, np.float32); key = C.c_int(0)
        _check(lib().dvo_vo_odometrize_raw(self._p, rgb.ctypes.data_as(C.c_void_p), ch, fp(T), C.byref(key)))
        return T.reshape(4, 4), bool(key.value)

    def odometrizeUsingDepth(self, gray, depth, sigma):
        g = f32(gray); d = f32(depth); s = f32(sigma); T = np.zeros(16, np.float32)
        _check(lib().dvo_vo_odometrize_depth(self._p, fp(g), fp(d), fp(s), fp(T)))
        return T.reshape(4, 4)

    def odometrizeUsingDepthRaw(self, rgb_u8, depth_u16, depth_scale=1.0 / 5000.0):
        rgb = np.ascontiguousarray(rgb_u8, np.uint8); d16 = np.ascontiguousarray(depth_u16, np.uint16)
        ch = 1 if rgb.ndim == 2 else rgb.shape[2]
        T = np.zeros(16, np.float32)
        _check(lib().dvo_vo_odometrize_depth_raw(self._p, rgb.ctypes.data_as(C.c_void_p), ch, d16.ctypes.data_as(C.c_void_p),
                                                 C.c_float(depth_scale), fp(T)))
        return T.reshape(4, 4)

    def save(self, path):
        _check(lib().dvo_vo_save(self._p, path.encode()))

    def load(self, path):
        _check(lib().dvo_vo_load(self._p, path.encode()))

    def setHistoryLimit(self, n):
        _check(lib().dvo_vo_set_history_limit(self._p, int(n)))

    def keyframeCount(self):
        return lib().dvo_vo_keyframe_count(self._p)

    def keyframeInfo(self, index):
        i = C.c_int(); l = C.c_int(); w = C.c_int(); h = C.c_int()
        xi = np.zeros(6, np.float32); rel = np.zeros(6, np.float32)
        _check(lib().dvo_vo_keyframe_info(self._p, index, C.byref(i), C.byref(l), C.byref(w), C.byref(h), fp(xi), fp(rel)))
        return dict(id=i.value, levels=l.value, width=w.value, height=h.value, xi=xi, rel_xi=rel)

    def keyframe(self, index, level=None):
        info = self.keyframeInfo(index)
        top = info["levels"] - 1
        if level is None:
            level = top
        sh = (info["height"] >> (top - level), info["width"] >> (top - level))
        g = np.zeros(sh, np.float32); d = np.zeros(sh, np.float32); s = np.zeros(sh, np.float32)
        a = np.zeros(sh, np.float32) if level == top else None
        K = np.zeros(9, np.float32)
        _check(lib().dvo_vo_keyframe_get(self._p, index, level, fp(g), fp(d), fp(s), fp(a) if a is not None else None, fp(K)))
        return dict(gray=g, depth=d, sigma=s, age=a, K=K.reshape(3, 3), **info)

    def lastFramePose(self):
        i = C.c_int(); xi = np.zeros(6, np.float32); rel = np.zeros(6, np.float32)
        _check(lib().dvo_vo_last_frame_pose(self._p, C.byref(i), fp(xi), fp(rel)))
        return i.value, xi, rel

    def lastValidUpdates(self):
        return lib().dvo_vo_last_valid_updates(self._p)

    def lastTrackLog(self):
        log = TrackLog()
        _check(lib().dvo_vo_last_track_log(self._p, C.byref(log)))
        return log.to_dict()


# ------------------------------------------------------------------ batched tracking (n_seq sequences per GPU)
class Batch:
    def __init__(self, n_seq, K, width, height, levels=4, culls=1, cfg=None):
        K = f32(K).reshape(9)
        self.n_seq, self.width, self.height, self.levels, self.culls = n_seq, width, height, levels, culls
        self._p = C.c_void_p()
        _check(lib().dvo_batch_create(n_seq, fp(K), width, height, levels, culls,
                                      C.byref(cfg) if cfg is not None else None, C.byref(self._p)))

    def close(self):
        if self._p:
            lib().dvo_batch_destroy(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def push_device(self, gray_ptr, depth_ptr, sigma_ptr):
        """Device pointers (ints, e.g. torch.Tensor.data_ptr()) to [n_seq, H, W] float32."""
        _check(lib().dvo_batch_push_device(self._p, C.c_void_p(gray_ptr), C.c_void_p(depth_ptr), C.c_void_p(sigma_ptr)))

    def prefetch_device(self, gray_ptr, depth_ptr, sigma_ptr):
        """Build the pyramids of the frame that the NEXT push_device will receive, on a side stream (the buffers must be complete)."""
        _check(lib().dvo_batch_prefetch_device(self._p, C.c_void_p(gray_ptr), C.c_void_p(depth_ptr), C.c_void_p(sigma_ptr)))

    def push_host(self, gray, depth, sigma):
        g = f32(gray); d = f32(depth); s = f32(sigma)
        assert g.shape == (self.n_seq, self.height, self.width)
        _check(lib().dvo_batch_push_host(self._p, fp(g), fp(d), fp(s)))

    def push_raw_device(self, rgb_ptr, channels, depth16_ptr, depth_scale=1.0 / 5000.0):
        """Device pointers to raw frames: [n_seq, H, W(, C)] uint8 and [n_seq, H, W] uint16 (converted inside the pyramid kernel)."""
        _check(lib().dvo_batch_push_raw_device(self._p, C.c_void_p(rgb_ptr), int(channels), C.c_void_p(depth16_ptr), C.c_float(depth_scale)))

    def prefetch_raw_device(self, rgb_ptr, channels, depth16_ptr, depth_scale=1.0 / 5000.0):
        _check(lib().dvo_batch_prefetch_raw_device(self._p, C.c_void_p(rgb_ptr), int(channels), C.c_void_p(depth16_ptr), C.c_float(depth_scale)))

    def push_raw_host(self, rgb_u8, depth_u16, depth_scale=1.0 / 5000.0):
        rgb = np.ascontiguousarray(rgb_u8, np.uint8); d16 = np.ascontiguousarray(depth_u16, np.uint16)
        ch = 1 if rgb.ndim == 3 else rgb.shape[3]
        assert rgb.shape[:3] == (self.n_seq, self.height, self.width) and d16.shape == (self.n_seq, self.height, self.width)
        _check(lib().dvo_batch_push_raw_host(self._p, rgb.ctypes.data_as(C.c_void_p), ch, d16.ctypes.data_as(C.c_void_p), C.c_float(depth_scale)))

    def last_poses(self):
        xi = np.zeros((self.n_seq, 6), np.float32); T = np.zeros((self.n_seq, 16), np.float32)
        _check(lib().dvo_batch_last_poses(self._p, fp(xi), fp(T)))
        return xi, T.reshape(self.n_seq, 4, 4)

    def copy_poses_device(self, xi_ptr, T_ptr=0):
        """Async D2D copy of the last poses into device memory (ints = device pointers, 0 = skip)."""
        _check(lib().dvo_batch_copy_poses_device(self._p, C.c_void_p(xi_ptr or None), C.c_void_p(T_ptr or None)))

    def last_track_log(self, seq):
        log = TrackLog()
        _check(lib().dvo_batch_last_track_log(self._p, seq, C.byref(log)))
        return log.to_dict()

    def synchronize(self):
        _check(lib().dvo_batch_synchronize(self._p))

    def profile(self, reset=False):
        p = GnProfile()
        _check(lib().dvo_batch_profile(self._p, C.byref(p), 1 if reset else 0))
        return dict(gn_ms=p.gn_ms, gn_launches=p.gn_launches, gn_pixels=p.gn_pixels, gn_iterations=p.gn_iterations)

    def probe_gn(self, level, n_launches):
        ms = C.c_float(); px = C.c_uint64()
        _check(lib().dvo_batch_probe_gn(self._p, level, n_launches, C.byref(ms), C.byref(px)))
        return ms.value, px.value


class MonoBatch:
    """n_seq mono sequences per GPU: System::VisualOdometry::odometrize (track + Mapper::estimate + regularize, system.hpp:44-74,
    src/map/mapper.cpp:16-144) for every sequence per call, keyframe decisions on the device (dvo_batch_create_mono)."""

    def __init__(self, n_seq, K, width, height, ring_keyframes=8, cfg=None):
        K = f32(K).reshape(9)
        self.n_seq, self.width, self.height = n_seq, width, height
        self._p = C.c_void_p()
        _check(lib().dvo_batch_create_mono(n_seq, fp(K), width, height, ring_keyframes,
                                           C.byref(cfg) if cfg is not None else None, C.byref(self._p)))

    def close(self):
        if self._p:
            lib().dvo_batch_destroy(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def setInitialDepth(self, depth, sigma):
        d = f32(depth); s = f32(sigma)
        assert d.shape == (self.height // 4, self.width // 4)
        _check(lib().dvo_batch_set_initial_depth(self._p, fp(d), fp(s)))

    def setInitialDepthDevice(self, depth_ptr, sigma_ptr):
        _check(lib().dvo_batch_set_initial_depth_device(self._p, C.c_void_p(depth_ptr), C.c_void_p(sigma_ptr)))

    def odometrize_device(self, gray_ptr):
        """Device pointer (int) to [n_seq, H, W] float32 gray frames."""
        _check(lib().dvo_batch_odometrize_device(self._p, C.c_void_p(gray_ptr)))

    def odometrize_raw_device(self, rgb_ptr, channels):
        """Device pointer (int) to [n_seq, H, W(, C)] uint8 frames (gray / RGB / RGBA)."""
        _check(lib().dvo_batch_odometrize_raw_device(self._p, C.c_void_p(rgb_ptr), int(channels)))

    def odometrize_host(self, frames):
        """Host frames [n_seq, H, W]: float32 gray, or uint8 gray / [n_seq, H, W, C] uint8 colour (raw, converted on the device)."""
        a = np.ascontiguousarray(frames)
        assert a.shape[:3] == (self.n_seq, self.height, self.width)
        if a.dtype == np.uint8:
            _check(lib().dvo_batch_odometrize_raw_host(self._p, a.ctypes.data_as(C.c_void_p), 1 if a.ndim == 3 else a.shape[3]))
        else:
            a = f32(a)
            _check(lib().dvo_batch_odometrize_host(self._p, fp(a)))

    def world_poses(self):
        xi = np.zeros((self.n_seq, 6), np.float32); T = np.zeros((self.n_seq, 16), np.float32); key = np.zeros(self.n_seq, np.int32)
        _check(lib().dvo_batch_world_poses(self._p, fp(xi), fp(T), key.ctypes.data_as(C.c_void_p)))
        return xi, T.reshape(self.n_seq, 4, 4), key.astype(bool)

    def copy_world_poses_device(self, xi_ptr=0, T_ptr=0, key_ptr=0):
        _check(lib().dvo_batch_copy_world_poses_device(self._p, C.c_void_p(xi_ptr or None), C.c_void_p(T_ptr or None), C.c_void_p(key_ptr or None)))

    def keyframe(self, seq, level=2):
        sh = ((self.height // 4) >> (2 - level), (self.width // 4) >> (2 - level))
        g = np.zeros(sh, np.float32); d = np.zeros(sh, np.float32); s = np.zeros(sh, np.float32)
        a = np.zeros(sh, np.float32) if level == 2 else None
        xi = np.zeros(6, np.float32); i = C.c_int(); n = C.c_int(); v = C.c_int()
        _check(lib().dvo_batch_keyframe_get(self._p, seq, level, fp(g), fp(d), fp(s), fp(a) if a is not None else None, fp(xi),
                                            C.byref(i), C.byref(n), C.byref(v)))
        return dict(gray=g, depth=d, sigma=s, age=a, xi=xi, id=i.value, n_keyframes=n.value, valid_updates=v.value)

    def last_track_log(self, seq):
        log = TrackLog()
        _check(lib().dvo_batch_last_track_log(self._p, seq, C.byref(log)))
        return log.to_dict()

    def synchronize(self):
        _check(lib().dvo_batch_synchronize(self._p))

    def stats(self, seq):
        """per-sequence counters (dvo_mono_stats): frames, keyframes created, ring size, valid updates of the last frame and the
        cumulative number of pixels whose birth keyframe had left the ring (searched against the oldest retained one instead)"""
        st = MonoStats()
        _check(lib().dvo_batch_mono_stats(self._p, seq, C.byref(st)))
        return {k: getattr(st, k) for k, _ in MonoStats._fields_}

    def profile_mapping(self, reset=False):
        p = MapProfile()
        _check(lib().dvo_batch_profile_mapping(self._p, C.byref(p), 1 if reset else 0))
        return {k: getattr(p, k) for k, _ in MapProfile._fields_}

    def profile(self, reset=False):
        p = GnProfile()
        _check(lib().dvo_batch_profile(self._p, C.byref(p), 1 if reset else 0))
        return dict(gn_ms=p.gn_ms, gn_launches=p.gn_launches, gn_pixels=p.gn_pixels, gn_iterations=p.gn_iterations)
