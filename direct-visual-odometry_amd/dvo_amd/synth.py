"""Synthetic RGB-D sequences with exact ground truth (SYN-640 / SYN-1080).

A textured height field  Z = 1.5 + 0.3 sin(2 pi X / 2.0) cos(2 pi Y / 1.5)  [m, world frame] is
ray-cast from a pinhole camera; gray = band-limited texture of the hit point in [0.05, 0.95]
(never <= 0, so the getSubpixel `last > 0` quirk is only hit when a test provokes it), depth = z in the
camera frame, sigma = 0.1 (what Transform::mapDepthtoGray assigns, src/core/transform.cpp:75).

Everything is torch (CPU or ROCm device): bench.py renders directly into HBM, tests render on CPU.
There is no network, so TUM fr1/fr2 cannot be fetched; this generator stands in for them.
"""
import math

import numpy as np
import torch

K_640 = np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]], np.float32)
K_1080 = np.array([[1575.0, 0, 959.5], [0, 1575.0, 539.5], [0, 0, 1]], np.float32)


def _texture_params(seed=1234, octaves=6, waves_per_octave=6):
    rng = np.random.RandomState(seed)
    freqs, phases, amps = [], [], []
    for o in range(octaves):
        base = 1.5 * (2.0 ** o)  # cycles per metre
        for _ in range(waves_per_octave):
            ang = rng.uniform(0, 2 * math.pi)
            f = base * rng.uniform(0.8, 1.25)
            freqs.append((f * math.cos(ang), f * math.sin(ang)))
            phases.append(rng.uniform(0, 2 * math.pi))
            amps.append(0.5 ** (0.7 * o))
    return np.array(freqs), np.array(phases), np.array(amps)


_TEX = _texture_params()


def texture(X, Y):
    """Band-limited texture in [0.05, 0.95] at world (X, Y) (float64 tensors)."""
    freqs, phases, amps = _TEX
    acc = torch.zeros_like(X)
    for (fx, fy), ph, a in zip(freqs, phases, amps):
        acc = acc + a * torch.sin(2 * math.pi * (fx * X + fy * Y) + ph)
    acc = acc / float(np.sum(amps))  # in [-1, 1]
    return 0.5 + 0.45 * torch.tanh(2.5 * acc) / math.tanh(2.5)


def height(X, Y):
    a, lx, ly = 0.3, 2.0, 1.5
    sx, cx = torch.sin(2 * math.pi * X / lx), torch.cos(2 * math.pi * X / lx)
    sy, cy = torch.sin(2 * math.pi * Y / ly), torch.cos(2 * math.pi * Y / ly)
    Z = 1.5 + a * sx * cy
    dZdX = a * (2 * math.pi / lx) * cx * cy
    dZdY = -a * (2 * math.pi / ly) * sx * sy
    return Z, dZdX, dZdY


def se3_exp_np(xi):
    """4x4 exp of a twist (v, w), float64 (same closed form as src/math/se3.cpp:70-98)."""
    xi = np.asarray(xi, np.float64)
    v, w = xi[:3], xi[3:]
    th = np.linalg.norm(w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    T = np.eye(4)
    if th < 1e-12:
        T[:3, :3] = np.eye(3) + W
        T[:3, 3] = v
        return T
    A, B, Cc = math.sin(th) / th, (1 - math.cos(th)) / th ** 2, (th - math.sin(th)) / th ** 3
    T[:3, :3] = np.eye(3) + A * W + B * W @ W
    T[:3, 3] = (np.eye(3) + B * W + Cc * W @ W) @ v
    return T


def render(T_wc, K, width, height_px, device="cpu", newton_iters=10):
    """Render (gray, depth) float32 [H, W] for camera pose T_wc (world <- camera, 4x4)."""
    dev = torch.device(device)
    T = torch.as_tensor(np.asarray(T_wc, np.float64), device=dev)
    fx, fy, cx, cy = float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2])
    u = torch.arange(width, dtype=torch.float64, device=dev)
    v = torch.arange(height_px, dtype=torch.float64, device=dev)
    vv, uu = torch.meshgrid(v, u, indexing="ij")
    rx, ry = (uu - cx) / fx, (vv - cy) / fy
    R, t = T[:3, :3], T[:3, 3]
    dx = R[0, 0] * rx + R[0, 1] * ry + R[0, 2]
    dy = R[1, 0] * rx + R[1, 1] * ry + R[1, 2]
    dz = R[2, 0] * rx + R[2, 1] * ry + R[2, 2]
    lam = (1.5 - t[2]) / dz
    for _ in range(newton_iters):
        X, Y = t[0] + lam * dx, t[1] + lam * dy
        Z, zx, zy = height(X, Y)
        F = t[2] + lam * dz - Z
        dF = dz - (zx * dx + zy * dy)
        lam = lam - F / dF
    X, Y = t[0] + lam * dx, t[1] + lam * dy
    gray = texture(X, Y).to(torch.float32)
    depth = lam.to(torch.float32)  # camera-frame z, since the ray has z = 1
    return gray, depth


def render_batch(T_wcs, K, width, height_px, device="cpu", newton_iters=10):
    """render() for N poses at once: returns (gray, depth) float32 [N, H, W].

    The same elementwise float64 operations in the same order as render() -- the results are bit-identical -- but each torch
    kernel covers N frames, so rendering thousands of frames is no longer launch-bound."""
    dev = torch.device(device)
    T = torch.as_tensor(np.asarray(T_wcs, np.float64), device=dev).reshape(-1, 4, 4)
    fx, fy, cx, cy = float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2])
    u = torch.arange(width, dtype=torch.float64, device=dev)
    v = torch.arange(height_px, dtype=torch.float64, device=dev)
    vv, uu = torch.meshgrid(v, u, indexing="ij")
    rx, ry = ((uu - cx) / fx)[None], ((vv - cy) / fy)[None]
    e = lambda i, j: T[:, i, j].reshape(-1, 1, 1)
    dx = e(0, 0) * rx + e(0, 1) * ry + e(0, 2)
    dy = e(1, 0) * rx + e(1, 1) * ry + e(1, 2)
    dz = e(2, 0) * rx + e(2, 1) * ry + e(2, 2)
    t0, t1, t2 = e(0, 3), e(1, 3), e(2, 3)
    lam = (1.5 - t2) / dz
    for _ in range(newton_iters):
        X, Y = t0 + lam * dx, t1 + lam * dy
        Z, zx, zy = height(X, Y)
        F = t2 + lam * dz - Z
        dF = dz - (zx * dx + zy * dy)
        lam = lam - F / dF
    X, Y = t0 + lam * dx, t1 + lam * dy
    return texture(X, Y).to(torch.float32), lam.to(torch.float32)


def trajectory(n_frames, seed=42, sigma_t=0.005, sigma_r_deg=0.3):
    """World<-camera poses: T_0 = I, T_k = T_{k-1} exp(xi_k), xi_k ~ N(0, diag(sigma_t^2, sigma_r^2))."""
    rng = np.random.RandomState(seed)
    poses = [np.eye(4)]
    for _ in range(1, n_frames):
        xi = np.concatenate([rng.normal(0, sigma_t, 3), rng.normal(0, math.radians(sigma_r_deg), 3)])
        poses.append(poses[-1] @ se3_exp_np(xi))
    return poses


def sequence(n_frames, width=640, height_px=480, K=None, seed=42, device="cpu", sigma_value=0.1, **kw):
    """Returns gray, depth, sigma tensors [n, H, W] float32 and the list of GT poses (world <- camera)."""
    if K is None:
        K = K_640 if width == 640 else K_1080
    poses = trajectory(n_frames, seed=seed, **kw)
    grays, depths = [], []
    for T in poses:
        g, d = render(T, K, width, height_px, device=device)
        grays.append(g)
        depths.append(d)
    gray = torch.stack(grays)
    depth = torch.stack(depths)
    sigma = torch.full_like(depth, sigma_value)
    return gray, depth, sigma, poses
