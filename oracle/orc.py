"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
PARITY STATUS: parity unpinned (see oracle/dvo_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
MAX_LEVELS = 8
MAX_ITER = 15
INVALID = np.float32(-2.0)


def build(force=False):
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "dvo_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB


def _cpu_has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            return " fma " in f.read().replace("\n", " ")
    except OSError:
        return True


class Outcome(C.Structure):
    _fields_ = [("H", C.c_double * 21), ("g", C.c_double * 6), ("sum_r2", C.c_double), ("n_valid", C.c_int),
                ("xi_update", C.c_float * 6), ("residual", C.c_float)]


class Frame(C.Structure):
    _fields_ = [("levels", C.c_int), ("culls", C.c_int), ("w", C.c_int * MAX_LEVELS), ("h", C.c_int * MAX_LEVELS),
                ("K", (C.c_float * 9) * MAX_LEVELS),
                ("gray", C.POINTER(C.c_float) * MAX_LEVELS), ("depth", C.POINTER(C.c_float) * MAX_LEVELS),
                ("sigma", C.POINTER(C.c_float) * MAX_LEVELS), ("age", C.POINTER(C.c_float)),
                ("id", C.c_int), ("ref_index", C.c_int), ("xi", C.c_float * 6), ("rel_xi", C.c_float * 6)]


class TrackLog(C.Structure):
    _fields_ = [("n_iter", C.c_int * MAX_LEVELS), ("residual", (C.c_float * MAX_ITER) * MAX_LEVELS),
                ("upd_norm", (C.c_float * MAX_ITER) * MAX_LEVELS), ("n_valid", (C.c_int * MAX_ITER) * MAX_LEVELS),
                ("xi_after", ((C.c_float * 6) * MAX_ITER) * MAX_LEVELS)]


_lib = None
FP = C.POINTER(C.c_float)


def lib():
    global _lib
    if _lib is None:
        if not _cpu_has_fma():
            raise RuntimeError("oracle is built with -mfma; this CPU has no FMA")
        build()
        L = C.CDLL(_LIB)
        L.orc_get_pixel.restype = C.c_float
        L.orc_get_subpixel.restype = C.c_float
        L.orc_get_subpixel_dense.restype = C.c_float
        L.orc_rng_depth.restype = C.c_float
        L.orc_frame_create.restype = C.POINTER(Frame)
        L.orc_vo_create.restype = C.c_void_p
        L.orc_vo_keyframe.restype = C.POINTER(Frame)
        L.orc_vo_last_frame.restype = C.POINTER(Frame)
        L.orc_get_pixel.argtypes = [FP, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_get_subpixel.argtypes = [FP, C.c_int, C.c_int, C.c_float, C.c_float]
        L.orc_get_subpixel_dense.argtypes = [FP, C.c_int, C.c_int, C.c_float, C.c_float]
        L.orc_rng_depth.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_gaussian_update.argtypes = [FP, FP, C.c_float, C.c_float, C.c_float]
        L.orc_gaussian_fuse.argtypes = [FP, FP, C.c_float, C.c_float]
        L.orc_pose_from_xi.argtypes = [FP, C.c_float, FP]
        L.orc_vo_create.argtypes = [FP, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int]
        for name in ("orc_vo_destroy", "orc_vo_set_initial_depth", "orc_vo_init_keyframe", "orc_vo_odometrize",
                     "orc_vo_odometrize_depth", "orc_vo_keyframe_count", "orc_vo_keyframe", "orc_vo_last_frame",
                     "orc_vo_last_valid_updates"):
            getattr(L, name).argtypes = None
        _lib = L
    return _lib


def set_threads(n):
    """Threads of the forEach bodies: 1 = the sequential oracle; n > 1 = row-parallel (cv::Mat::forEach), bench baseline only."""
    lib().orc_set_threads(int(n))


LIT_ARITH, LIT_SE3 = 1, 2


def set_literal(mask):
    """Arithmetic mode of the oracle (dvo_oracle.h): 0 = canonical (D8 order, SE(3) in double); LIT_ARITH = per-pixel expressions as
    the reference source writes them; LIT_SE3 = float-literal se3.cpp.  Sensitivity measurements only."""
    lib().orc_set_literal(int(mask))


def set_tracker_params(step3=None, min_residual=-1.0, min_update=-1.0):
    """override the reference's step / stop literals (bench.py's converging side leg only); no arguments = back to the reference's"""
    if step3 is None:
        lib().orc_set_tracker_params(None, C.c_float(min_residual), C.c_float(min_update))
    else:
        st = f32(step3)
        lib().orc_set_tracker_params(fp(st), C.c_float(min_residual), C.c_float(min_update))


def set_nudge_ulps(n):
    """orc_track moves the first component of its first xi_update by n ulps (sensitivity probe; 0 = off)."""
    lib().orc_set_nudge_ulps(int(n))


class literal:
    """with orc.literal(orc.LIT_ARITH): ... -- never while other threads run oracle calls."""

    def __init__(self, mask=LIT_ARITH, nudge=0):
        self.mask, self.nudge = mask, nudge

    def __enter__(self):
        set_literal(self.mask); set_nudge_ulps(self.nudge)

    def __exit__(self, *a):
        set_literal(0); set_nudge_ulps(0)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def fp(a):
    return a.ctypes.data_as(FP)


# ---- SE(3) -----------------------------------------------------------------
def se3_exp(xi, lit=False):
    xi = f32(xi); T = np.zeros(16, np.float32)
    (lib().orc_se3_exp_f32lit if lit else lib().orc_se3_exp)(fp(xi), fp(T))
    return T.reshape(4, 4)


def se3_log(T, lit=False):
    T = f32(T).reshape(16); xi = np.zeros(6, np.float32)
    (lib().orc_se3_log_f32lit if lit else lib().orc_se3_log)(fp(T), fp(xi))
    return xi


def se3_concatenate(a, b, lit=False):
    a = f32(a); b = f32(b); o = np.zeros(6, np.float32)
    (lib().orc_se3_concatenate_f32lit if lit else lib().orc_se3_concatenate)(fp(a), fp(b), fp(o))
    return o


def pose_from_xi(xi, sign=1.0):
    xi = f32(xi); Rt = np.zeros(12, np.float32)
    lib().orc_pose_from_xi(fp(xi), C.c_float(sign), fp(Rt))
    return Rt


# ---- image primitives --------------------------------------------------------
def cull_image(src, times):
    src = f32(src); h, w = src.shape
    dst = np.zeros((h >> times, w >> times), np.float32)
    lib().orc_cull_image(fp(src), w, h, times, fp(dst))
    return dst


def cull_intrinsic(K, times):
    K = f32(K).reshape(9); o = np.zeros(9, np.float32)
    lib().orc_cull_intrinsic(fp(K), times, fp(o))
    return o.reshape(3, 3)


def gradiate(img, xdir):
    img = f32(img); h, w = img.shape
    out = np.zeros((h, w), np.float32)
    lib().orc_gradiate(fp(img), w, h, 1 if xdir else 0, fp(out))
    return out


def get_subpixel(img, px, py, dense=False):
    img = f32(img); h, w = img.shape
    fn = lib().orc_get_subpixel_dense if dense else lib().orc_get_subpixel
    return np.float32(fn(fp(img), w, h, C.c_float(px), C.c_float(py)))


def warp(Rt, px, py, d, K):
    Rt = f32(Rt); K = f32(K).reshape(9); p = np.zeros(2, np.float32)
    lib().orc_warp(fp(Rt), C.c_float(px), C.c_float(py), C.c_float(d), fp(K), fp(p))
    return p


def warp_image(xi, gray, depth, K):
    xi = f32(xi); gray = f32(gray); depth = f32(depth); K = f32(K).reshape(9)
    h, w = gray.shape
    out = np.zeros((h, w), np.float32)
    lib().orc_warp_image(fp(xi), fp(gray), fp(depth), w, h, fp(K), fp(out))
    return out


# ---- optimize / track --------------------------------------------------------
def upper_to_full(H21):
    H = np.zeros((6, 6)); k = 0
    for i in range(6):
        for j in range(i, 6):
            H[i, j] = H[j, i] = H21[k]; k += 1
    return H


def optimize(obj_gray, ref_gray, ref_depth, ref_sigma, K, xi, level, crop=True, variant=0, want_mask=False):
    obj_gray = f32(obj_gray); ref_gray = f32(ref_gray); ref_depth = f32(ref_depth); ref_sigma = f32(ref_sigma)
    K = f32(K).reshape(9); xi = f32(xi)
    h, w = ref_gray.shape
    out = Outcome()
    mask = np.zeros((h, w), np.uint8) if want_mask else None
    lib().orc_optimize(fp(obj_gray), fp(ref_gray), fp(ref_depth), fp(ref_sigma), w, h, fp(K), fp(xi), level,
                       1 if crop else 0, variant, C.byref(out), mask.ctypes.data_as(C.c_void_p) if want_mask else None)
    res = dict(H=np.array(out.H[:]), g=np.array(out.g[:]), sum_r2=out.sum_r2, n_valid=out.n_valid,
               xi_update=np.array(out.xi_update[:], np.float32), residual=np.float32(out.residual))
    if want_mask:
        res["mask"] = mask
    return res


def solve6(H21, g):
    H21 = np.ascontiguousarray(H21, np.float64); g = np.ascontiguousarray(g, np.float64)
    x = np.zeros(6, np.float32)
    lib().orc_solve6(H21.ctypes.data_as(C.POINTER(C.c_double)), g.ctypes.data_as(C.POINTER(C.c_double)), fp(x))
    return x


def lsq_svd(A, B):
    A = f32(A); B = f32(B); x = np.zeros(6, np.float32)
    lib().orc_lsq_svd(fp(A), fp(B), A.shape[0], fp(x))
    return x


class OFrame:
    """Owning wrapper of orc_frame (System::Frame, include/system/frame.hpp:72-144)."""

    def __init__(self, gray, depth, sigma, K, levels, culls, id=0, ptr=None, own=True):
        if ptr is None:
            gray = f32(gray); h, w = gray.shape
            d = f32(depth) if depth is not None else None
            s = f32(sigma) if sigma is not None else None
            K = f32(K).reshape(9)
            ptr = lib().orc_frame_create(fp(gray), fp(d) if d is not None else None, fp(s) if s is not None else None,
                                         w, h, fp(K), levels, culls, id)
        self.ptr = ptr
        self.own = own

    def __del__(self):
        if self.own and self.ptr:
            lib().orc_frame_destroy(self.ptr)
            self.ptr = None

    @property
    def c(self):
        return self.ptr.contents

    @property
    def levels(self):
        return self.c.levels

    def size(self, level):
        return self.c.w[level], self.c.h[level]

    def K(self, level):
        return np.array(self.c.K[level][:], np.float32).reshape(3, 3)

    def _map(self, arr, level):
        w, h = self.size(level)
        return np.ctypeslib.as_array(arr[level], shape=(h, w)).copy()

    def gray(self, level):
        return self._map(self.c.gray, level)

    def depth(self, level):
        return self._map(self.c.depth, level)

    def sigma(self, level):
        return self._map(self.c.sigma, level)

    def age(self):
        w, h = self.size(self.levels - 1)
        return np.ctypeslib.as_array(self.c.age, shape=(h, w)).copy()

    @property
    def xi(self):
        return np.array(self.c.xi[:], np.float32)

    @property
    def rel_xi(self):
        return np.array(self.c.rel_xi[:], np.float32)

    def set_pose(self, xi, rel_xi):
        for i in range(6):
            self.c.xi[i] = float(xi[i]); self.c.rel_xi[i] = float(rel_xi[i])

    def set_age(self, age):
        a = f32(age); w, h = self.size(self.levels - 1)
        C.memmove(self.c.age, a.ctypes.data, 4 * w * h)

    def update_depth_sigma(self, d, s):
        d = f32(d); s = f32(s)
        lib().orc_frame_update_depth_sigma(self.ptr, fp(d), fp(s))


def track(obj, ref, crop=True, variant=0, fixed_iters=0):
    xi = np.zeros(6, np.float32); log = TrackLog()
    lib().orc_track(obj.ptr, ref.ptr, 1 if crop else 0, variant, fixed_iters, fp(xi), C.byref(log))
    L = ref.levels
    out = dict(n_iter=[log.n_iter[l] for l in range(L)], residual=[], upd_norm=[], n_valid=[], xi_after=[])
    for l in range(L):
        n = log.n_iter[l]
        out["residual"].append(np.array(log.residual[l][:n], np.float32))
        out["upd_norm"].append(np.array(log.upd_norm[l][:n], np.float32))
        out["n_valid"].append(np.array(log.n_valid[l][:n], np.int32))
        out["xi_after"].append(np.array([log.xi_after[l][i][:] for i in range(n)], np.float32).reshape(n, 6))
    return xi, out


# ---- mapping -----------------------------------------------------------------
def rng_depth(seed, frame_id, pixel):
    return np.float32(lib().orc_rng_depth(seed, frame_id, pixel))


def gaussian_update(depth, sigma, d, s, reset):
    a = C.c_float(depth); b = C.c_float(sigma)
    ok = lib().orc_gaussian_update(C.byref(a), C.byref(b), C.c_float(d), C.c_float(s), C.c_float(reset))
    return np.float32(a.value), np.float32(b.value), bool(ok)


def gaussian_fuse(depth, sigma, d, s):
    a = C.c_float(depth); b = C.c_float(sigma)
    ok = lib().orc_gaussian_fuse(C.byref(a), C.byref(b), C.c_float(d), C.c_float(s))
    return np.float32(a.value), np.float32(b.value), bool(ok)


def propagate(ref_depth, ref_sigma, ref_age, xi, K):
    d = f32(ref_depth); s = f32(ref_sigma); a = f32(ref_age); xi = f32(xi); K = f32(K).reshape(9)
    h, w = d.shape
    od = np.zeros_like(d); os_ = np.zeros_like(d); oa = np.zeros_like(d)
    lib().orc_propagate(fp(d), fp(s), fp(a), w, h, fp(xi), fp(K), fp(od), fp(os_), fp(oa))
    return od, os_, oa


def regularize(depth, sigma):
    d = f32(depth); s = f32(sigma); h, w = d.shape
    out = np.zeros_like(d)
    lib().orc_regularize(fp(d), fp(s), w, h, fp(out))
    return out


def implement_update(obj_gray, born_gray, r_xi, K, qx, qy, depth, sigma):
    og = f32(obj_gray); bg = f32(born_gray); h, w = og.shape
    gx = gradiate(bg, True); gy = gradiate(bg, False)
    r_xi = f32(r_xi); K = f32(K).reshape(9)
    nd = C.c_float(); ns = C.c_float()
    lib().orc_implement_update(fp(og), fp(bg), fp(gx), fp(gy), w, h, fp(r_xi), fp(K), int(qx), int(qy),
                               C.c_float(depth), C.c_float(sigma), C.byref(nd), C.byref(ns))
    return np.float32(nd.value), np.float32(ns.value)


def mapper_update(history, obj, seed):
    """history: list of OFrame, oldest first; history[-1] is the ref keyframe (updated in place)."""
    arr = (C.POINTER(Frame) * len(history))(*[f.ptr for f in history])
    return lib().orc_mapper_update(arr, len(history), obj.ptr, C.c_uint32(seed))


def need_new_frame(rel_xi, id, ref_id):
    r = f32(rel_xi)
    return bool(lib().orc_need_new_frame(fp(r), id, ref_id))


class OVO:
    """System::VisualOdometry (include/system/system.hpp:12-104) on the oracle."""

    def __init__(self, K, w, h, seed=0, crop=True, variant=0):
        K = f32(K).reshape(9)
        self.w, self.h = w, h
        self.p = C.c_void_p(lib().orc_vo_create(fp(K), w, h, C.c_uint32(seed), 1 if crop else 0, variant))

    def __del__(self):
        if self.p:
            lib().orc_vo_destroy(self.p); self.p = None

    def set_initial_depth(self, depth, sigma):
        d = f32(depth); s = f32(sigma)
        lib().orc_vo_set_initial_depth(self.p, fp(d), fp(s))

    def init_keyframe(self, gray, depth, sigma):
        g = f32(gray); d = f32(depth); s = f32(sigma)
        lib().orc_vo_init_keyframe(self.p, fp(g), fp(d), fp(s))

    def odometrize(self, gray):
        g = f32(gray); T = np.zeros(16, np.float32)
        key = lib().orc_vo_odometrize(self.p, fp(g), fp(T))
        return T.reshape(4, 4), bool(key)

    def odometrize_depth(self, gray, depth, sigma):
        g = f32(gray); d = f32(depth); s = f32(sigma); T = np.zeros(16, np.float32)
        lib().orc_vo_odometrize_depth(self.p, fp(g), fp(d), fp(s), fp(T))
        return T.reshape(4, 4)

    def keyframe_count(self):
        return lib().orc_vo_keyframe_count(self.p)

    def keyframe(self, i):
        return OFrame(None, None, None, None, 0, 0, ptr=lib().orc_vo_keyframe(self.p, i), own=False)

    def last_frame(self):
        p = lib().orc_vo_last_frame(self.p)
        return OFrame(None, None, None, None, 0, 0, ptr=p, own=False) if p else None

    def last_valid_updates(self):
        return lib().orc_vo_last_valid_updates(self.p)
