#!/usr/bin/env python3
"""bench.py -- tracked frames/sec of the MI355X-native direct-VO hot path (BASELINE.json metric).

One "step" = every sequence of the batch ingests its next 640x480 frame (u8 gray + u16 sensor depth, already resident in HBM),
builds the 4-level pyramid and tracks it against its previous frame with the reference's coarse-to-fine Gauss-Newton loop
(odometrizeUsingDepth, include/system/system.hpp:77-93) -> one relative pose per sequence, left in HBM.
value = sequences x steps x ranks / wall time.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

The default N = 1 run also carries, outside the timed region and in the same JSON line:
  roofline      k_track_gn, hipEvents around every launch of an identical pass
  cpu_baseline  the CPU oracle on this box's host cores, bounded sample of the same workload
  accuracy      vs the synthetic ground truth, vs the CPU oracle on the same pairs, the oracle's own sensitivity
                (literal source arithmetic; a 1-ulp nudge) and a per-iteration GPU / oracle parity check that FAILS the run
  value_incl_h2d  the SURVEY.md §8(d) definition: every frame streamed from pinned host memory, every pose read back
  secondary     single-stream dvo_vo handles; BASELINE configs[2] (mono tracking + inverse-depth filter, 8192 sequences) and
                configs[3] (SYN-1080 dense alignment, 128 sequences) as short legs; a converging (gain 1) configuration

Independent sequences shard one batch per GPU (weak scaling); the only collective is the RCCL all_gather of the pose arrays
after the timed steps (BASELINE config 5), timed separately as gather_ms.
"""
import argparse
import copy
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "direct-visual-odometry_amd"))

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_ISSUE_PEAK = 39.3e12    # FP32 lane-instructions/s at one vector instruction per four cycles: 16 lanes x 1024 SIMDs x 2.4 GHz
# 16 B per evaluated pixel and Gauss-Newton iteration = obj_gray + ref_gray + ref_depth + ref_sigma.  For raw
# sensor frames sigma is the constant of transform.cpp:75 wherever a pixel can contribute, so no sigma / weight map exists and the
# kernel's algorithmic input is 12 B per pixel: `roofline.achieved` uses the bytes of the mode that ran (never more than the
# kernel has to move), the 16-B contract figure is reported beside it.
GN_BYTES_CONTRACT = 16


def gn_bytes_per_pixel(const_weight):
    return 12 if const_weight else 16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="sequences tracked concurrently per GPU (0 = 16384 for syn640 with raw input, 4096 with float "
                                                         "maps, 8192 for syn640-mono, 128 for syn1080)")
    ap.add_argument("--frames", type=int, default=6, help="distinct frames per sequence kept in HBM (ping-pong order)")
    ap.add_argument("--workload", default="syn640", choices=["syn640", "syn1080", "syn640-mono"],
                    help="syn640: sensor-depth tracking (BASELINE configs[1], the headline); syn1080: configs[3]; "
                         "syn640-mono: mono tracking + inverse-depth filter (configs[2])")
    ap.add_argument("--ring", type=int, default=8, help="syn640-mono: keyframes kept per sequence")
    ap.add_argument("--mono-init", default="random", choices=["random", "gt"],
                    help="syn640-mono: depth of the first keyframes -- random = the reference's N(1.5, 0.5) >= 0.5 (frame.hpp:17-21); "
                         "gt = the rendered depth of frame 0 + N(0, 0.1) noise, sigma 0.5 (an initialised map: the stereo updates then succeed)")
    ap.add_argument("--input", default="raw", choices=["raw", "float"],
                    help="what is resident in HBM per frame: raw = u8 gray + u16 depth as a sensor / cv::imread delivers them (loader.cpp:137-147), "
                         "converted inside the pyramid kernel; float = float32 gray + depth + sigma maps (round 1's form)")
    ap.add_argument("--fixed-iters", type=int, default=0, help="0 = the reference's early exit; N = exactly N per level")
    ap.add_argument("--sigma", type=float, default=0.1, help="sensor sigma (src/core/transform.cpp:75)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--streams", type=int, default=0, help="sub-batches tracked on concurrent HIP streams (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--prefetch", action="store_true", help="build the pyramid of frame k+1 on the library's low-priority side stream while frame k "
                    "tracks (dvo_batch_prefetch_*): +1 % frames/s, but k_track_gn then shares the chip with k_pyramid and its per-launch "
                    "time -- the roofline figure -- reads 7 % longer; default: every pyramid in order on the tracking stream")
    ap.add_argument("--no-prefetch", action="store_true", help="(the default since round 2; kept for old command lines)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the side legs (single stream, mono batch, SYN-1080, converging configuration)")
    ap.add_argument("--pcie-steps", type=int, default=300, help="frames streamed per sequence in the PCIe-inclusive measurement (0 = skip)")
    ap.add_argument("--pcie-batch", type=int, default=4096, help="sequences of the PCIe-inclusive measurement (pinned host copies of 3 frames each)")
    ap.add_argument("--step-scale", type=float, default=1.0, help="multiplies the reference's per-level step literals (optimize.cpp:22-26); "
                    "1 = the reference.  Used by the converging side leg only")
    ap.add_argument("--min-residual", type=float, default=-1.0, help="stop threshold on the mean squared residual (tracker.cpp:16); < 0 = the reference's 5e-3")
    return ap.parse_args()


def ring_index(k, n):
    """0,1,..,n-1,n-2,..,1,0,1,..: consecutive frames are always neighbours of the trajectory."""
    period = 2 * (n - 1)
    r = k % period
    return r if r < n else period - r


def _spawn_ranks(n):
    """`python bench.py --gpus N` outside torchrun: start N ranks under torch.distributed.run as a CHILD process (never an exec),
    before anything in this process has touched the GPU, and leave with its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


class Env:
    """what every leg needs: ranks, device, stream, the (lazily imported) oracle"""

    def __init__(self, a):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != a.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or plain `python bench.py --gpus %d`)"
                             % (a.gpus, self.world, a.gpus, a.gpus))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (libdvo has no CPU fallback)")
        # DVO_BENCH_REHEARSE=1: multi-rank control flow on ONE GPU (every rank on cuda:0, gloo collectives on CPU tensors);
        # only for rehearsing the N > 1 path on a 1-GPU box, never for reported numbers.
        self.rehearse = os.environ.get("DVO_BENCH_REHEARSE") == "1"
        if self.rehearse:
            self.local = 0
        elif torch.cuda.device_count() < self.world:
            raise SystemExit("bench.py: %d ranks but %d visible GPUs" % (self.world, torch.cuda.device_count()))
        torch.cuda.set_device(self.local)
        self.dev = torch.device("cuda", self.local)
        self.cdev = torch.device("cpu") if self.rehearse else self.dev     # where collective payloads live
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo") if self.rehearse else dist.init_process_group("nccl", device_id=self.dev)  # nccl = RCCL over xGMI
            self.dist = dist
        self.stream = torch.cuda.current_stream().cuda_stream
        self.solo = self.rank == 0 and self.world == 1
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        self.ncore = max(1, min(ncpu, 16))      # the box's CPU share for one GPU
        self._orc = None

    def barrier(self):
        torch.cuda.synchronize()
        if self.dist:
            self.dist.barrier()
        torch.cuda.synchronize()

    @property
    def orc(self):
        """the CPU oracle: checker and cpu_baseline only, never on the timed path"""
        if self._orc is None:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import orc
            self._orc = orc
        return self._orc


def diff_stats(dx, within=1e-3):
    dx = np.asarray(dx, np.float64)
    if dx.size == 0:
        return {"pairs": 0}
    return {"pairs": int(dx.size), "pairs_within_1e-3": int(np.sum(dx < within)), "median_abs_pose_diff": float(np.median(dx)),
            "p90_abs_pose_diff": float(np.percentile(dx, 90)), "max_abs_pose_diff": float(np.max(dx))}


def render_sequences(env, B, F, W, H, K, want_depth=True, n_gt=32):
    """synthetic sequences rendered straight into HBM as raw sensor frames: u8 gray [F][B][H][W] (+ u16 depth, 1/5000 m units)"""
    from dvo_amd import synth
    dev = env.dev
    gray8 = torch.empty((F, B, H, W), dtype=torch.uint8, device=dev)
    depth16 = torch.empty((F, B, H, W), dtype=torch.int16, device=dev) if want_depth else None   # (bit pattern of uint16)
    gt_poses = []
    chunk = max(1, (96 if W == 640 else 12) // F)  # sequences rendered per call (float64 temporaries: ~20 x chunk x F frames)
    for b0 in range(0, B, chunk):
        b1 = min(B, b0 + chunk)
        poses = [synth.trajectory(F, seed=42 + 1000 * env.rank + b) for b in range(b0, b1)]
        for b, p in zip(range(b0, b1), poses):
            if b < n_gt:
                gt_poses.append(p)
        Ts = np.stack([p[f] for p in poses for f in range(F)])
        g, d = synth.render_batch(Ts, K, W, H, device=dev, newton_iters=6)
        g = g.reshape(b1 - b0, F, H, W).permute(1, 0, 2, 3); d = d.reshape(b1 - b0, F, H, W).permute(1, 0, 2, 3)
        gray8[:, b0:b1] = torch.clamp(torch.round(g * 255.0), 0, 255).to(torch.uint8)
        if want_depth:
            depth16[:, b0:b1] = torch.clamp(torch.round(d * 5000.0), 0, 65535).to(torch.int32).to(torch.int16)
    torch.cuda.synchronize()
    return gray8, depth16, gt_poses


def host_raw_frames(gray8, depth16, fsel, bsel):
    """float32 numpy (gray, depth, sigma) of raw frames: exactly what k_pyramid_raw4 computes from them (loader.cpp:137-147, transform.cpp:60-76)"""
    g8 = gray8[fsel, bsel].cpu().numpy(); d16 = depth16[fsel, bsel].cpu().numpy().view(np.uint16)
    gg = g8.astype(np.float32) * np.float32(1.0 / 255.0)
    gg[d16 == 0] = np.float32(-2.0)
    return gg, d16.astype(np.float32) * np.float32(1.0 / 5000.0), np.where(d16 > 0, np.float32(0.1), np.float32(1.0)).astype(np.float32)


def rel_errors(xis, pairs, poses):
    """exp(xi) maps reference-frame points into the new frame: T_rel = inv(P_new) P_ref for world<-camera poses P"""
    from dvo_amd import synth
    et, er = [], []
    for xi, (fr, fo) in zip(xis, pairs):
        E = synth.se3_exp_np(np.asarray(xi, np.float64)) @ np.linalg.inv(np.linalg.inv(poses[fo]) @ poses[fr])
        et.append(float(np.dot(E[:3, 3], E[:3, 3])))
        er.append(float(np.arccos(np.clip((np.trace(E[:3, :3]) - 1.0) / 2.0, -1.0, 1.0)) ** 2))
    return et, er


def backward_error(orc, H21, g, x):
    H = orc.upper_to_full(np.asarray(H21, np.float64)); g = np.asarray(g, np.float64); x = np.asarray(x, np.float64)
    return float(np.abs(H @ x - g).max() / max((np.abs(H) @ np.abs(x) + np.abs(g)).max(), 1e-300))


# ======================================================================================================================
# sensor-depth tracking: the headline (syn640), SYN-1080, and the converging side leg
# ======================================================================================================================
def run_depth(a, env, role="headline", data=None):
    """role: 'headline' = everything; 'secondary' = value + roofline + accuracy only (short legs of the default run)"""
    import dvo_amd as dvo
    from dvo_amd import synth
    dev, rank, world = env.dev, env.rank, env.world
    head = role == "headline"
    if a.workload == "syn640":
        W, H, K, levels, culls = 640, 480, synth.K_640, 4, 1          # Frame(g,d,s,K,4,1), system.hpp:82
    else:
        W, H, K, levels, culls = 1920, 1080, synth.K_1080, 5, 0
        if a.fixed_iters == 0:
            a.fixed_iters = 10
    raw = a.input == "raw" and abs(a.sigma - 0.1) < 1e-9   # (raw frames carry the sensor sigma of transform.cpp:75; other sigmas need float maps)
    if a.batch <= 0:
        # the latency-bound parts of a step (coarse levels, solves, the tail of each level) amortise over more sequences per launch:
        # 251 k / 263 k / 270 k frames/s at 4096 / 8192 / 16384.  16384 raw sequences = 90 GB of input frames + 80 GB of pyramids.
        a.batch = (16384 if raw else 4096) if a.workload == "syn640" else 128
    B, F = a.batch, max(2, a.frames)
    crop = 1 if a.workload == "syn640" else 0
    tracker_over = {}
    if abs(a.step_scale - 1.0) > 1e-12:
        tracker_over.update(step_default=2.0 * a.step_scale, step_level1=1.5 * a.step_scale, step_level2=1.0 * a.step_scale)
    if a.min_residual >= 0:
        tracker_over.update(min_residual=a.min_residual)

    # ---- frames in HBM ----------------------------------------------------------------------------------------------
    t_gen = time.time()
    if data is None:
        gray8, depth16, gt_poses = render_sequences(env, B, F, W, H, K)
        data = {"gray8": gray8, "depth16": depth16, "gt_poses": gt_poses, "F": F}
    gray8, depth16, gt_poses = data["gray8"], data["depth16"], data["gt_poses"]
    assert gray8.shape[1] >= B and gray8.shape[0] == F
    if not raw:   # float maps derived from the same raw frames on the device (what Loader::getNormalizedImages would hand over)
        gray = gray8[:, :B].to(torch.float32) * (1.0 / 255.0)
        depth = (depth16[:, :B].to(torch.int32) & 0xFFFF).to(torch.float32) * (1.0 / 5000.0)
        gray[depth == 0] = -2.0
        sigma = torch.where(depth > 0, torch.full_like(depth, a.sigma), torch.ones_like(depth))
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen

    def host_frames(fsel, bsel):
        if raw:
            return host_raw_frames(gray8, depth16, fsel, bsel)
        return gray[fsel, bsel].cpu().numpy(), depth[fsel, bsel].cpu().numpy(), sigma[fsel, bsel].cpu().numpy()

    def make_cfg(**kw):
        base = dict(device=env.local, stream=env.stream, fixed_iterations=a.fixed_iters, track_streams=a.streams, crop_enable=crop)
        base.update(tracker_over); base.update(kw)
        return dvo.default_config(**base)

    def push(bt, k, out=None, nb=None):
        # frames are resident and complete.  Default: each push builds its pyramid on the tracking stream.  --prefetch: the pyramid of
        # frame k+1 is built on the library's side stream while frame k tracks (dvo_batch_prefetch_device); every step from k = 1 on
        # then prefetches exactly one frame and consumes the one prefetched by the step before.  Either way K timed steps contain K
        # pyramid builds.  (nb: a batch over the first nb sequences -- frame f's first nb images are contiguous.)
        f = ring_index(k, F)
        if a.prefetch and k >= 1 and nb is None:
            fn = ring_index(k + 1, F)
            if raw:
                bt.prefetch_raw_device(gray8[fn].data_ptr(), 1, depth16[fn].data_ptr())
            else:
                bt.prefetch_device(gray[fn].data_ptr(), depth[fn].data_ptr(), sigma[fn].data_ptr())
        if raw:
            bt.push_raw_device(gray8[f].data_ptr(), 1, depth16[f].data_ptr())
        else:
            bt.push_device(gray[f].data_ptr(), depth[f].data_ptr(), sigma[f].data_ptr())
        if out is not None:
            bt.copy_poses_device(out.data_ptr())

    # ---- the timed region ---------------------------------------------------------------------------------------------
    batch = dvo.Batch(B, K, W, H, levels, culls, cfg=make_cfg())
    poses_out = torch.zeros((a.steps, B, 6), dtype=torch.float32, device=dev)
    push(batch, 0)                          # first frame: reference only
    for k in range(a.warmup):
        push(batch, 1 + k)
    env.barrier()
    t0 = time.perf_counter()
    for k in range(a.steps):
        push(batch, 1 + a.warmup + k, poses_out[k])
    env.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=env.cdev)
    gather_ms = 0.0
    if world > 1:
        env.dist.all_reduce(tmax, op=env.dist.ReduceOp.MAX)
        # config 5: gather every rank's poses over RCCL/xGMI (tens of KB: latency bound)
        from dvo_amd import shard
        local_poses = poses_out.permute(1, 0, 2).contiguous().to(env.cdev)  # [sequence][frame][6]
        torch.cuda.synchronize()
        tg = time.perf_counter()
        allp, lens = shard.gather_poses(local_poses)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - tg) * 1e3
        assert allp.shape[0] == B * world and int(lens.min().item()) == a.steps
    dt = float(tmax.item())
    fps = B * a.steps * world / dt
    log0 = batch.last_track_log(0)
    finite = bool(torch.isfinite(poses_out).all().item())
    batch.close()   # (its pyramids are freed before the next pass allocates its own)

    out = {
        "metric": "tracked frames/sec (640x480 semi-dense) at 1 GPU" if a.workload == "syn640" else "tracked frames/sec (1920x1080 dense)",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "SYN-640 (stand-in for TUM fr1/desk: no dataset offline), 640x480, sensor depth, "
                               "frame-to-frame tracking, 4-level pyramid (320x240 top), reference thresholds"
                   if a.workload == "syn640" else "SYN-1080 dense alignment (BASELINE configs[3]), 5-level pyramid, fixed iterations, no crop",
                   "input": "raw u8 gray + u16 depth (3 B/px), converted inside k_pyramid" if raw else "float32 gray + depth + sigma maps (12 B/px)",
                   "sequences_per_gpu": B, "frames_in_hbm_per_sequence": F, "sigma": a.sigma,
                   "pyramid": ("frame k+1's pyramid built on a low-priority side stream while frame k tracks (--prefetch)" if a.prefetch else
                               "every frame's pyramid built on the tracking stream ahead of its tracking"),
                   "fixed_iterations": a.fixed_iters, "iterations_per_level_seq0": log0["n_iter"],
                   "poses_finite": finite, "gather_ms": gather_ms, "datagen_s": round(t_gen, 2)},
    }
    if tracker_over:
        out["config"]["non_reference_tracker_constants"] = tracker_over
    elif a.workload == "syn640":
        out["config"]["note"] = ("sigma = %.1f: the reference's weight step/sigma multiplies only the residual and its gradients are un-halved "
                                 "(optimize.cpp:83-89, convert.cpp:58,71), so the update is %.0fx the Gauss-Newton step at levels 0 and 3" %
                                 (a.sigma, 2.0 / max(0.01, min(0.5, a.sigma)) / 2.0))

    # ---- accuracy of the timed steps against the synthetic ground truth (BASELINE metric: "ATE RMSE vs reference") ------
    step_pairs = [(ring_index(a.warmup + k, F), ring_index(1 + a.warmup + k, F)) for k in range(a.steps)]
    NS = min(len(gt_poses), B) if head else min(len(gt_poses), B, 8 if a.workload == "syn640" else 2)
    if rank == 0 and NS > 0:
        xs = poses_out[:, :NS].cpu().numpy()  # [steps][sample][6]
        et, er = [], []
        for b in range(NS):
            t_, r_ = rel_errors(xs[:, b], step_pairs, gt_poses[b])
            et += t_; er += r_
        out["accuracy"] = {"rel_translation_rmse_m": float(np.sqrt(np.mean(et))), "rel_rotation_rmse_rad": float(np.sqrt(np.mean(er))),
                           "rel_translation_median_m": float(np.sqrt(np.median(et))), "rel_translation_p90_m": float(np.sqrt(np.percentile(et, 90))),
                           "sample": "%d sequences x %d timed frame pairs vs the synthetic ground truth "
                                     "(per-frame motion ~ N(0, 5 mm / 0.3 deg))" % (NS, a.steps)}

    # ---- PCIe-inclusive rate: the same steps fed from pinned HOST buffers, every pose read back (SURVEY.md §8d fps definition) ----
    if head and a.pcie_steps > 0:
        PB = min(B, a.pcie_batch)
        # (fixed launch schedule: the adaptive one keeps the host inside push() until the GPU is nearly done with the step, so the
        #  next frame's transfer would not be queued in time to overlap it)
        hb = dvo.Batch(PB, K, W, H, levels, culls, cfg=make_cfg(track_adaptive=-1))
        nh = min(F, 3)
        if raw:
            host = [(gray8[f, :PB].cpu().pin_memory(), depth16[f, :PB].cpu().pin_memory()) for f in range(nh)]
        else:
            host = [(gray[f, :PB].cpu().pin_memory(), depth[f, :PB].cpu().pin_memory(), sigma[f, :PB].cpu().pin_memory()) for f in range(nh)]
        ring = 8   # device / pinned slots for the poses in flight (the read-back of step k is asynchronous)
        xi_dev = torch.zeros((ring, PB, 6), dtype=torch.float32, device=dev)
        xi_pin = torch.zeros((ring, PB, 6), dtype=torch.float32).pin_memory()

        def hpush(k):
            fr = host[ring_index(k, len(host))]
            if raw:
                hb.push_raw_host(fr[0].numpy(), fr[1].numpy().view(np.uint16))
            else:
                hb.push_host(fr[0].numpy(), fr[1].numpy(), fr[2].numpy())
        hpush(0); hpush(1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(a.pcie_steps):
            hpush(2 + k)                                                 # H2D on the library's copy stream: overlaps the tracking of the step before
            hb.copy_poses_device(xi_dev[k % ring].data_ptr())
            xi_pin[k % ring].copy_(xi_dev[k % ring], non_blocking=True)  # D2H of every pose of the step, asynchronous, same stream
        hb.synchronize()
        torch.cuda.synchronize()
        t_h = time.perf_counter() - t1
        full_frames = os.environ.get("DVO_UPLOAD_FULL_FRAMES") is not None
        pcie_b = (3 * W * (H >> culls if (culls > 0 and H % (1 << culls) == 0 and not full_frames) else H)) if raw else 12 * W * H
        out["value_incl_h2d"] = PB * a.pcie_steps / t_h
        out["incl_h2d"] = {"sequences": PB, "frames_streamed_per_sequence": a.pcie_steps, "seconds": round(t_h, 2),
                           "bytes_per_frame": (3 if raw else 12) * W * H, "bytes_over_pcie_per_frame": pcie_b,
                           "pcie_GBps": PB * a.pcie_steps * pcie_b / t_h / 1e9,
                           "note": "SURVEY.md §8(d) fps definition (H2D of every frame + D2H of every pose inside the timed loop), the loop shape of "
                                   "main.cpp:44-50 for %d sequences at once; the rate is bound by the host link, not by the batch size; raw host "
                                   "frames: only the rows the pyramid keeps (every 2^culls-th: Convert::cullImage) are transferred, by one "
                                   "strided copy per buffer; host frames cycle through %d pinned buffers per sequence" % (PB, nh),
                           "input": ("pinned host u8 gray + u16 depth (dvo_batch_push_raw_host)" if raw else
                                     "pinned host float32 gray + depth + sigma (dvo_batch_push_host)") + ", every pose copied back to pinned host memory per step; "
                                    "transfers of step k+1 overlap the tracking of step k (copy stream, two staging slots)"}
        hb.close()
        del host, xi_pin

    # ---- roofline of the dominant kernel (k_track_gn): HIP events around every launch of an identical pass ----
    if not a.no_roofline:
        pb = dvo.Batch(B, K, W, H, levels, culls, cfg=make_cfg(profile=1))
        push(pb, 0)
        for k in range(a.warmup):
            push(pb, 1 + k)
        pb.profile(reset=True)
        for k in range(a.steps):
            push(pb, 1 + a.warmup + k)
        pr = pb.profile()
        top_ms, top_px = pb.probe_gn(levels - 1, 20)
        pb.close()
        if pr["gn_launches"] > 0 and pr["gn_ms"] > 0:
            bpp = gn_bytes_per_pixel(raw)
            px_per_launch = pr["gn_pixels"] / pr["gn_launches"]
            ms_per_launch = pr["gn_ms"] / pr["gn_launches"]
            achieved = bpp * px_per_launch / (ms_per_launch * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "k_track_gn",
                               "avg_launch_us": ms_per_launch * 1e3, "launches": pr["gn_launches"],
                               "algorithmic_bytes_per_pixel": bpp,
                               "algorithmic_bytes_per_launch": bpp * px_per_launch,
                               "bytes_note": ("raw sensor frames: obj gray + ref gray + ref depth = 12 B per evaluated pixel and iteration; sigma is the "
                                              "constant of transform.cpp:75 for every pixel that can contribute, so no sigma / weight map is read"
                                              if raw else "obj gray + ref gray + ref depth + weight (sigma) map = 16 B per evaluated pixel and iteration"),
                               "contract_16B_per_pixel": {"achieved": GN_BYTES_CONTRACT * px_per_launch / (ms_per_launch * 1e-3) / 1e9,
                                                          "frac": GN_BYTES_CONTRACT * px_per_launch / (ms_per_launch * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                          "note": "SURVEY.md §8(d)'s per-pixel figure (with a sigma map), for comparison with earlier rounds"},
                               "gn_share_of_step_time": pr["gn_ms"] / (dt * 1e3),
                               "top_level_probe": {"avg_launch_us": top_ms * 1e3,
                                                   "achieved": bpp * top_px / (top_ms * 1e-3) / 1e9,
                                                   "frac": bpp * top_px / (top_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}
            # HBM bytes per launch come from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over this same command
            # (tools/pmc_traffic.sh): a process cannot profile itself, so the committed summary for the matching
            # workload/batch is quoted here, and null is reported when there is none.
            for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*k_track_gn_pmc*.txt")), key=_natural, reverse=True):
                if "exp_" in os.path.basename(fn):
                    continue
                valu = waves = None
                for line in open(fn):  # the last dispatch block of the file is the finest-level probe launch
                    t = line.split()
                    if len(t) == 2 and t[0] == "SQ_INSTS_VALU": valu = float(t[1])
                    if len(t) == 2 and t[0] == "SQ_WAVES": waves = float(t[1])
                if valu and waves:
                    out["roofline"]["valu_instructions_per_pixel"] = {"value": valu / waves / 4.0,  # 4 pixels per lane (PPT)
                                                                      "source": "profiles/" + os.path.basename(fn)}
                    break
            if head and (a.fixed_iters == 0 or a.workload == "syn1080"):
                for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json")), key=_natural, reverse=True):
                    with open(fn) as fh:
                        tr = json.load(fh)
                    if tr.get("kernel") == "k_track_gn" and tr.get("sequences_per_gpu") == B and tr.get("workload", "syn640") == a.workload:
                        out["roofline"]["traffic"] = tr["traffic_bytes_per_launch"]
                        out["roofline"]["traffic_source"] = "profiles/" + os.path.basename(fn)
                        break

    # ---- parity sample (oracle = checker): a small batch over the first sequences, logs read back after every step ----------
    parity_fail = None
    gpu_iters = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and NS > 0 and tracker_over:   # the oracle takes the same non-reference constants
        env.orc.set_tracker_params([tracker_over.get("step_default", 2.0), tracker_over.get("step_level1", 1.5), tracker_over.get("step_level2", 1.0)],
                                   tracker_over.get("min_residual", -1.0), -1.0)
    if rank == 0 and world == 1 and not a.no_cpu_baseline and NS > 0:
        orc = env.orc
        from concurrent.futures import ThreadPoolExecutor
        # same tile size as the big batch (auto picks 4 pixels per thread there): then a sequence's bits do not depend on its batch
        sb = dvo.Batch(NS, K, W, H, levels, culls, cfg=make_cfg(gn_pixels_per_thread=4))
        xs_small = np.zeros((a.steps, NS, 6), np.float32)
        logs = [[None] * NS for _ in range(a.steps)]
        push(sb, 0, nb=NS)
        for k in range(a.warmup):
            push(sb, 1 + k, nb=NS)
        for k in range(a.steps):
            push(sb, 1 + a.warmup + k, nb=NS)
            xs_small[k] = sb.last_poses()[0]
            for b in range(NS):
                logs[k][b] = sb.last_track_log(b)
        sb.close()
        xs_big = poses_out[:, :NS].cpu().numpy()
        same = bool(np.array_equal(xs_small.view(np.uint32), xs_big.view(np.uint32)))
        gpu_iters = [[logs[k][b]["n_iter"][:levels] for k in range(a.steps)] for b in range(NS)]
        cap = a.fixed_iters if a.fixed_iters > 0 else 15
        NP = min(NS, 4 if a.workload == "syn640" else 1)     # sequences whose every iteration goes through the oracle
        steps_checked = range(a.steps) if a.workload == "syn640" else range(min(a.steps, 2))
        gh, dh, sh = host_frames(slice(None), slice(0, NP))

        def check_seq(b):
            n_it = n_bad_valid = 0
            worst_back = worst_comp = 0.0
            for k in steps_checked:
                fr, fo = step_pairs[k]
                ref = orc.OFrame(gh[fr, b], dh[fr, b], sh[fr, b], K, levels, culls)
                obj = orc.OFrame(gh[fo, b], dh[fo, b], sh[fo, b], K, levels, culls)
                lg = logs[k][b]
                xi = np.zeros(6, np.float32)
                for l in range(levels):
                    for it in range(lg["n_iter"][l]):
                        o = orc.optimize(obj.gray(l), ref.gray(l), ref.depth(l), ref.sigma(l), ref.K(l), xi, l, crop=bool(crop))
                        n_it += 1
                        if o["n_valid"] != int(lg["n_valid"][l][it]):
                            n_bad_valid += 1
                        upd = lg["xi_update"][l][it]
                        if o["n_valid"] > 0:
                            worst_back = max(worst_back, backward_error(orc, o["H"], o["g"], upd))
                        nxt = orc.se3_concatenate(xi, upd)
                        if np.all(np.isfinite(nxt)):
                            worst_comp = max(worst_comp, float(np.abs(nxt - lg["xi_after"][l][it]).max() / max(1.0, float(np.abs(nxt).max()))))
                        xi = lg["xi_after"][l][it]      # follow the GPU's trajectory: parity of every step given its input
            return n_it, n_bad_valid, worst_back, worst_comp
        t_p = time.perf_counter()
        with ThreadPoolExecutor(env.ncore) as ex:
            res = list(ex.map(check_seq, range(NP)))
        ok = same and all(r[1] == 0 for r in res) and max(r[2] for r in res) <= 2e-6 and max(r[3] for r in res) <= 5e-7
        out.setdefault("accuracy", {})["per_iteration_parity"] = {
            "ok": bool(ok), "iterations_checked": int(sum(r[0] for r in res)), "contributing_pixel_count_mismatches": int(sum(r[1] for r in res)),
            "max_backward_error": float(max(r[2] for r in res)), "backward_error_bound": 2e-6,
            "max_composition_error_rel": float(max(r[3] for r in res)),
            "small_batch_equals_big_batch_bitwise": same, "seconds": round(time.perf_counter() - t_p, 1),
            "sample": "every Gauss-Newton iteration of %d sequences x %d timed frame pairs: the CPU oracle's optimize() at the GPU's own input "
                      "pose -> contributing-pixel count equal, the GPU's xi_update within the stated backward error of the oracle's normal "
                      "equations, pose composition exact to rounding; and the %d-sequence batch re-run of the first %d sequences gives the "
                      "%d-sequence batch's poses bit for bit" % (NP, len(list(steps_checked)), NS, NS, B)}
        if not ok:
            parity_fail = out["accuracy"]["per_iteration_parity"]

    # ---- secondary (N = 1 only, a few seconds): the single-sequence drop-in entry points, host frames in, pose out per call --
    if head and env.solo and a.workload == "syn640" and not a.no_secondary:
        g0, d0, s0 = host_frames(slice(None), 0)
        n_sec = 200
        vo = dvo.VisualOdometry(K, W, H, cfg=dvo.default_config(device=env.local))
        for k in range(3):
            f = ring_index(k, F); vo.odometrizeUsingDepth(g0[f], d0[f], s0[f])
        t1 = time.perf_counter()
        for k in range(n_sec):
            f = ring_index(3 + k, F); vo.odometrizeUsingDepth(g0[f], d0[f], s0[f])
        single_depth = n_sec / (time.perf_counter() - t1)
        vo.close()
        # (the per-iteration log is read back on demand: the iteration count comes from an untimed repeat of the same frames)
        vo = dvo.VisualOdometry(K, W, H, cfg=dvo.default_config(device=env.local))
        its = 0
        for k in range(3 + n_sec):
            f = ring_index(k, F); vo.odometrizeUsingDepth(g0[f], d0[f], s0[f])
            if k >= 3:
                its += sum(vo.lastTrackLog()["n_iter"])
        vo.close()
        # the same loop fed with what the sensor delivers (u8 gray + u16 depth: dvo_vo_odometrize_depth_raw)
        g8h = gray8[:, 0].cpu().numpy(); d16h = depth16[:, 0].cpu().numpy().view(np.uint16)
        vo = dvo.VisualOdometry(K, W, H, cfg=dvo.default_config(device=env.local))
        for k in range(3):
            f = ring_index(k, F); vo.odometrizeUsingDepthRaw(g8h[f], d16h[f])
        t1 = time.perf_counter()
        for k in range(n_sec):
            f = ring_index(3 + k, F); vo.odometrizeUsingDepthRaw(g8h[f], d16h[f])
        single_raw = n_sec / (time.perf_counter() - t1)
        vo.close()
        vo = dvo.VisualOdometry(K, W, H, cfg=dvo.default_config(device=env.local, rng_seed=1))
        di = d0[0][::4, ::4].copy()
        vo.setInitialDepth(di, np.full_like(di, 0.5))
        for k in range(3):
            vo.odometrize(g0[ring_index(k, F)])
        t1 = time.perf_counter(); keys = 0
        for k in range(n_sec):
            _, key = vo.odometrize(g0[ring_index(3 + k, F)]); keys += int(key)
        single_mono = n_sec / (time.perf_counter() - t1)
        vo.close()
        # ... and the reference's own loop from C++ (test/sequence.cpp:10-23 through include/dvo.hpp, no Python in the timed region):
        # lib/single_stream_bench as a child process on the same six frames of sequence 0
        cpp = None
        exe = os.path.join(ROOT, "direct-visual-odometry_amd", "lib", "single_stream_bench")
        if os.path.exists(exe):
            import re
            import subprocess
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                fn = os.path.join(td, "frames.f32")
                np.stack([g0, d0, s0], axis=1).astype(np.float32).tofile(fn)
                Kf = np.asarray(K, np.float32).reshape(3, 3)
                r = subprocess.run([exe, fn, str(F), str(W), str(H), repr(float(Kf[0, 0])), repr(float(Kf[1, 1])), repr(float(Kf[0, 2])), repr(float(Kf[1, 2])), "400"],
                                   capture_output=True, text=True, timeout=300)
            cpp = {}
            for line in r.stdout.splitlines():
                m = re.match(r"^(.*?)\s+([\d.]+) frames/s\s+([\d.]+) us/frame\s+\(([\d.]+) GN", line)
                if m:
                    key = "float_pageable" if "pageable" in m.group(1) else ("float_pinned" if "pinned" in m.group(1) else "raw_u8_u16")
                    cpp[key] = {"fps": float(m.group(2)), "us_per_frame": float(m.group(3)), "gn_iterations_per_frame": float(m.group(4))}
                m = re.match(r"^(odometrize\S*) \(mono.*?\s+([\d.]+) frames/s\s+([\d.]+) us/frame\s+\((\d+) keyframes in (\d+) frames", line)
                if m:   # the reference's main loop (main.cpp:49: vo.odometrize(gray) per frame): tracking + Mapper::estimate + regularize
                    cpp["mono_raw_u8" if "Raw" in m.group(1) else "mono_float_pageable"] = {
                        "fps": float(m.group(2)), "us_per_frame": float(m.group(3)), "keyframes": int(m.group(4)), "frames": int(m.group(5))}
            if not cpp:
                cpp = {"error": (r.stderr or r.stdout)[-300:]}
        out["secondary"] = {"single_stream_odometrizeUsingDepth_fps": single_depth, "single_stream_odometrizeUsingDepthRaw_fps": single_raw,
                            "single_stream_cpp": cpp,
                            "single_stream_odometrize_mono_track_map_fps": single_mono,
                            "depth_gn_iterations_per_frame": its / n_sec, "mono_keyframes": keys, "frames": n_sec,
                            "note": "one dvo_vo handle, 640x480 host frames in / pose out per call (PCIe and launch latency included); the *_fps fields "
                                    "are timed through the ctypes binding (numpy -> ctypes adds ~0.1 ms per call), single_stream_cpp is the same loop "
                                    "from C++ through include/dvo.hpp: the reference's own API (system.hpp:77-93)"}

    # ---- CPU baseline: the oracle on this box's host cores, bounded sample of the same workload ----------
    if env.solo and not a.no_cpu_baseline and NS > 0:
        orc = env.orc
        from concurrent.futures import ThreadPoolExecutor
        ncore = env.ncore
        gh, dh, sh = host_frames(slice(None), slice(0, NS))  # [F][NS][H][W]

        def seq_job(b):
            res = []
            for (fr, fo) in step_pairs:
                ref = orc.OFrame(gh[fr, b], dh[fr, b], sh[fr, b], K, levels, culls)
                obj = orc.OFrame(gh[fo, b], dh[fo, b], sh[fo, b], K, levels, culls)
                xi, lg = orc.track(obj, ref, crop=bool(crop), variant=0, fixed_iters=a.fixed_iters)
                res.append((xi, lg["n_iter"]))
            return res

        def oracle_pass():
            with ThreadPoolExecutor(ncore) as ex:
                return list(ex.map(seq_job, range(NS)))
        # The SAME sample the GPU accuracy was taken on: every timed frame pair of the first NS sequences through the (hoisted)
        # oracle, sequence-parallel on host threads (ctypes releases the GIL).
        t_o = time.perf_counter()
        o_res = oracle_pass()
        t_o = time.perf_counter() - t_o
        xs = poses_out[:, :NS].cpu().numpy()
        per_seq, o_et, o_er, dx, below = [], [], [], [], []
        cap = a.fixed_iters if a.fixed_iters > 0 else 15
        for b in range(NS):
            ox = [r[0] for r in o_res[b]]
            et_o, er_o = rel_errors(ox, step_pairs, gt_poses[b])
            et_g, er_g = rel_errors(xs[:, b], step_pairs, gt_poses[b])
            o_et += et_o; o_er += er_o
            d_ = np.abs(np.asarray(ox) - xs[:, b]).max(axis=1)   # |xi_gpu - xi_oracle|_inf per frame pair
            dx += list(d_)
            for k in range(a.steps):
                o_ok = max(o_res[b][k][1][:levels]) < cap
                g_ok = gpu_iters is not None and max(gpu_iters[b][k]) < cap
                below.append(bool(o_ok and g_ok))
            per_seq.append({"seq": b, "gpu_rmse_m": float(np.sqrt(np.mean(et_g))), "oracle_rmse_m": float(np.sqrt(np.mean(et_o))),
                            "max_abs_pose_diff": float(d_.max())})
        acc = out["accuracy"]
        acc["cpu_oracle_rel_translation_rmse_m"] = float(np.sqrt(np.mean(o_et)))
        acc["cpu_oracle_rel_rotation_rmse_rad"] = float(np.sqrt(np.mean(o_er)))
        acc["cpu_oracle_rel_translation_median_m"] = float(np.sqrt(np.median(o_et)))
        gvo = diff_stats(dx)
        gvo["worst_gpu_sequences"] = sorted(per_seq, key=lambda r: -r["gpu_rmse_m"])[:4]
        gvo["oracle_seconds"] = round(t_o, 1)
        if a.fixed_iters == 0 and gpu_iters is not None:
            sel = np.asarray(below, bool)
            gvo["pairs_where_both_stop_before_the_iteration_cap"] = diff_stats(np.asarray(dx)[sel])
        gvo["note"] = ("same %d sequences x %d frame pairs through the CPU oracle, whole Tracker::track calls (|xi_gpu - xi_oracle|_inf); compare with "
                       "oracle_self_sensitivity: the oracle against itself under perturbations no two implementations share" % (NS, len(step_pairs)))
        acc["gpu_vs_oracle"] = gvo
        if head and a.fixed_iters == 0:
            # How far does the oracle move under (a) the LITERAL source arithmetic instead of the canonical order product and oracle share
            #  and (b) one unit in the last place on the very first update?  If these spreads look like gpu_vs_oracle,
            # whole-call disagreement is the iteration's own sensitivity (the 10x over-relaxed step), not a GPU discrepancy.
            ox0 = np.asarray([[r[0] for r in o_res[b]] for b in range(NS)])
            sens = {}
            for name, (mask, nudge) in (("literal_arithmetic", (orc.LIT_ARITH, 0)), ("literal_arithmetic_and_float_se3", (orc.LIT_ARITH | orc.LIT_SE3, 0)),
                                        ("one_ulp_nudge_of_first_update", (0, 1))):
                with orc.literal(mask, nudge):
                    p_res = oracle_pass()
                px = np.asarray([[r[0] for r in p_res[b]] for b in range(NS)])
                d_ = np.abs(px - ox0).max(axis=2).reshape(-1)
                st = diff_stats(d_)
                et_p = []
                for b in range(NS):
                    et_p += rel_errors(px[b], step_pairs, gt_poses[b])[0]
                st["rel_translation_rmse_m"] = float(np.sqrt(np.mean(et_p)))
                sens[name] = st
            sens["note"] = ("the canonical oracle vs the oracle with the reference's per-pixel expressions evaluated exactly as written (true "
                            "divisions, no shared reciprocal, no fma; + se3.cpp in float), and vs itself with the first xi_update of every call "
                            "moved by 1 ulp: same %d x %d pairs, same statistics as gpu_vs_oracle" % (NS, len(step_pairs)))
            acc["oracle_self_sensitivity"] = sens
        if head:
            def run(variant, budget):
                """frames/s of the oracle on sequence 0's frames: pyramid + track per frame, as one GPU step does per sequence"""
                n, t_start = 0, time.perf_counter()
                ref = orc.OFrame(gh[0, 0], dh[0, 0], sh[0, 0], K, levels, culls)
                k = 0
                while time.perf_counter() - t_start < budget:
                    f = ring_index(1 + k, F)
                    obj = orc.OFrame(gh[f, 0], dh[f, 0], sh[f, 0], K, levels, culls)
                    orc.track(obj, ref, crop=bool(crop), variant=variant, fixed_iters=a.fixed_iters)
                    ref = obj
                    n += 1; k += 1
                return n / (time.perf_counter() - t_start), n
            one_fps, n1 = run(1, a.cpu_seconds * 0.6)                 # faithful, 1 thread
            orc.set_threads(ncore)
            all_fps, na = run(1, a.cpu_seconds * 0.4)                 # faithful, forEach bodies row-parallel over the box's cores
            orc.set_threads(1)
            hoisted_fps, nh = run(0, max(2.0, a.cpu_seconds / 5))
            out["cpu_baseline"] = {"value": all_fps, "unit": "frames/s", "cores": ncore, "kind": "port",
                                   "sample": "%d frame pairs of sequence 0 (same frames, pyramid + track), oracle 'faithful' variant (per-pixel se3 exp, "
                                             "materialised warpImage, Nx6 stack + SVD least squares) with the forEach bodies row-parallel over %d "
                                             "threads as cv::Mat::forEach (optimize.cpp:28, transform.cpp:39); no cv::Mat heap churn, so it "
                                             "under-states the real reference's cost" % (na, ncore),
                                   "one_core": {"value": one_fps, "cores": 1, "sample": "%d frame pairs, same variant, 1 thread" % n1},
                                   "hoisted_value": hoisted_fps, "hoisted_sample": "%d frame pairs, 1 thread, pose hoisted + 6x6 normal equations" % nh,
                                   "cpu": _cpu_model()}
            if "secondary" in out:
                best = out["secondary"]["single_stream_odometrizeUsingDepth_fps"]
                cppr = out["secondary"].get("single_stream_cpp") or {}
                if "float_pageable" in cppr:
                    best = cppr["float_pageable"]["fps"]
                out["secondary"]["single_stream_vs_cpu_baseline"] = {"single_stream_fps": best, "x_all_cores": best / all_fps, "x_one_core": best / one_fps,
                                                                     "note": "float maps from pageable host memory through the reference's own API (C++ loop "
                                                                             "when lib/single_stream_bench is built) against the CPU oracle's faithful variant"}
    if tracker_over and env._orc is not None:
        env.orc.set_tracker_params()
    return out, data, parity_fail


# ======================================================================================================================
# mono tracking + inverse-depth filter (BASELINE configs[2])
# ======================================================================================================================
def run_mono(a, env, role="headline", gray=None):
    """One step = every sequence runs System::VisualOdometry::odometrize (system.hpp:44-74) on its next gray frame: 3-level pyramid
    (160x120 top), Tracker::track against its newest keyframe, then Mapper::estimate (propagate + new keyframe, or stereo update
    against the keyframe each pixel was born in) and regularize -- dvo_batch_create_mono, keyframe decisions on the device, no host
    round trip.  gray: u8 frames [F][>=B][H][W] already in HBM (the default run hands over the headline's)."""
    import dvo_amd as dvo
    from dvo_amd import synth
    dev, rank, world = env.dev, env.rank, env.world
    head = role == "headline"
    W, H, K = 640, 480, synth.K_640
    B = a.batch if a.batch > 0 else 8192   # 1.43 M / 1.51 M / 1.52 M frames/s at 4096 / 8192 / 16384 sequences (round 2)
    F = max(2, a.frames)
    t_gen = time.time()
    raw = a.input == "raw"
    init_depth = None
    if gray is None:
        gray = torch.empty((F, B, H, W), dtype=torch.uint8 if raw else torch.float32, device=dev)
        init_depth = torch.empty((B, H // 4, W // 4), dtype=torch.float32, device=dev) if a.mono_init == "gt" else None
        chunk = max(1, 96 // F)
        gen = torch.Generator(device=dev); gen.manual_seed(1234 + rank)
        for b0 in range(0, B, chunk):
            b1 = min(B, b0 + chunk)
            Ts = np.stack([synth.trajectory(F, seed=42 + 1000 * rank + b)[f] for b in range(b0, b1) for f in range(F)])
            g, d = synth.render_batch(Ts, K, W, H, device=dev, newton_iters=6)
            g = g.reshape(b1 - b0, F, H, W).permute(1, 0, 2, 3)
            gray[:, b0:b1] = torch.clamp(torch.round(g * 255.0), 0, 255).to(torch.uint8) if raw else g
            if init_depth is not None:
                d0 = d.reshape(b1 - b0, F, H, W)[:, 0, ::4, ::4]
                init_depth[b0:b1] = d0 + 0.1 * torch.randn(d0.shape, generator=gen, device=dev)
    else:
        assert raw and gray.dtype == torch.uint8 and gray.shape[0] == F and gray.shape[1] >= B
    init_sigma = torch.full_like(init_depth, 0.5) if init_depth is not None else None
    torch.cuda.synchronize()

    def odo(mb, f):
        if raw:
            mb.odometrize_raw_device(gray[f].data_ptr(), 1)     # u8 gray as cv::imread + cvtColor deliver it, converted in k_pyramid
        else:
            mb.odometrize_device(gray[f].data_ptr())
    t_gen = time.time() - t_gen
    poses_out = torch.zeros((a.steps, B, 6), dtype=torch.float32, device=dev)
    keys_out = torch.zeros((a.steps, B), dtype=torch.int32, device=dev)

    def run(profile):
        cfg = dvo.default_config(device=env.local, stream=env.stream, profile=profile, rng_seed=1)
        mb = dvo.MonoBatch(B, K, W, H, ring_keyframes=a.ring, cfg=cfg)
        if init_depth is not None:
            mb.setInitialDepthDevice(init_depth.data_ptr(), init_sigma.data_ptr())
        odo(mb, 0)                                           # first frame: first keyframe of every sequence
        for k in range(a.warmup):
            odo(mb, ring_index(1 + k, F))
        if profile:
            mb.profile(reset=True); mb.profile_mapping(reset=True)
        env.barrier()
        t0 = time.perf_counter()
        for k in range(a.steps):
            odo(mb, ring_index(1 + a.warmup + k, F))
            mb.copy_world_poses_device(poses_out[k].data_ptr(), 0, keys_out[k].data_ptr())
        env.barrier()
        return mb, time.perf_counter() - t0

    mb, dt = run(0)
    tmax = torch.tensor([dt], dtype=torch.float64, device=env.cdev)
    if world > 1:
        env.dist.all_reduce(tmax, op=env.dist.ReduceOp.MAX)
    dt = float(tmax.item())
    log0 = mb.last_track_log(0)
    kf0 = mb.keyframe(0)
    vu = [mb.keyframe(b)["valid_updates"] for b in range(min(B, 64))]
    stats = [mb.stats(b) for b in range(min(B, 64))]
    mb.close()
    keys = keys_out.float().mean().item()
    out = {"metric": "tracked frames/sec (640x480 mono: tracking + inverse-depth filter)", "value": B * a.steps * world / dt, "unit": "frames/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "SYN-640 mono (stand-in for TUM fr2/desk: no dataset offline), 640x480 gray only, "
                                  "Frame(gray,K,3,2) pyramid (160x120 top), track + Mapper::estimate + regularize per frame, " +
                                  ("the reference's random initial depth N(1.5, 0.5) >= 0.5 (frame.hpp:17-21)" if a.mono_init == "random" else
                                   "first keyframes initialised with the rendered depth + N(0, 0.1) noise, sigma 0.5"),
                      "input": "raw u8 gray (1 B/px), converted inside k_pyramid" if raw else "float32 gray (4 B/px)",
                      "sequences_per_gpu": B, "frames_in_hbm_per_sequence": F, "keyframe_ring": a.ring,
                      "keyframe_fraction_of_timed_frames": keys, "iterations_per_level_seq0": log0["n_iter"],
                      "keyframes_created_seq0": kf0["n_keyframes"], "mean_valid_updates_last_frame": float(np.mean(vu)),
                      "ring_clamped_pixels_first_64_sequences": int(sum(s["clamped_pixels"] for s in stats)),
                      "ring_note": "FrameHistory is a ring of %d keyframes per sequence (the reference's is unbounded): a pixel older than the ring is "
                                   "searched against the oldest retained keyframe; ring_clamped_pixels counts how often that happened" % a.ring,
                      "poses_finite": bool(torch.isfinite(poses_out).all().item()), "datagen_s": round(t_gen, 2)}}
    if head and a.pcie_steps > 0:   # co-headline: every frame streamed from pinned host memory, every pose copied back
        PB = min(B, a.pcie_batch)
        hb = dvo.MonoBatch(PB, K, W, H, ring_keyframes=a.ring, cfg=dvo.default_config(device=env.local, stream=env.stream, rng_seed=1))
        host = [gray[f, :PB].cpu().pin_memory() for f in range(min(F, 3))]
        ring = 8
        xi_dev = torch.zeros((ring, PB, 6), dtype=torch.float32, device=dev)
        xi_pin = torch.zeros((ring, PB, 6), dtype=torch.float32).pin_memory()
        hb.odometrize_host(host[0].numpy()); hb.odometrize_host(host[1].numpy())
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(a.pcie_steps):
            hb.odometrize_host(host[ring_index(2 + k, len(host))].numpy())
            hb.copy_world_poses_device(xi_dev[k % ring].data_ptr())
            xi_pin[k % ring].copy_(xi_dev[k % ring], non_blocking=True)
        hb.synchronize()
        torch.cuda.synchronize()
        out["value_incl_h2d"] = PB * a.pcie_steps / (time.perf_counter() - t1)
        full_frames = os.environ.get("DVO_UPLOAD_FULL_FRAMES") is not None
        out["incl_h2d"] = {"sequences": PB, "frames_streamed_per_sequence": a.pcie_steps, "bytes_per_frame": (1 if raw else 4) * W * H,
                           "bytes_over_pcie_per_frame": (W * (H >> 2 if (H % 4 == 0 and not full_frames) else H)) if raw else 4 * W * H,
                           "note": "raw host frames: only the rows the pyramid keeps (every 4th: Frame(gray,K,3,2)) are transferred, by one strided copy",
                           "input": "pinned host %s gray (dvo_batch_odometrize_%shost), every world pose copied back per step" % (("u8", "raw_") if raw else ("float32", ""))}
        hb.close()
    if not a.no_roofline:
        pb, _ = run(1)
        pr = pb.profile()
        pm = pb.profile_mapping()
        pb.close()
        if pr["gn_launches"] > 0 and pr["gn_ms"] > 0:
            bpl = GN_BYTES_CONTRACT * pr["gn_pixels"] / pr["gn_launches"]      # (mono frames carry a sigma map: 16 B per pixel)
            mpl = pr["gn_ms"] / pr["gn_launches"]
            ach = bpl / (mpl * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                               "kernel": "k_track_gn", "avg_launch_us": mpl * 1e3, "launches": pr["gn_launches"],
                               "algorithmic_bytes_per_launch": bpl, "gn_share_of_step_time": pr["gn_ms"] / (dt * 1e3),
                               "note": "levels of 40x30 .. 160x120 only: latency-bound launches; this workload's dominant kernels are the mapping "
                                       "kernels below (FP32 issue bound, not HBM bound: SURVEY.md §8d)"}
        if pm["frames"] > 0:
            # k_depth_update against the vector issue peak ("report it against vector-FP32 peak and say so"): instructions per
            # pixel from the committed PMC summary of this kernel (a process cannot profile itself), time and pixels measured live
            valu_px, src = None, None
            for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*mono*mapping_pmc*.txt")), key=_natural, reverse=True):
                grid = valu = None
                for line in open(fn):
                    t = line.split()
                    if line.startswith("dvo::k_depth_update") and grid is None:
                        grid = float([x for x in t if x.startswith("grid=")][0][5:]); continue
                    if grid is not None and valu is None and len(t) == 2 and t[0] == "SQ_INSTS_VALU":
                        valu = float(t[1]); break
                    if grid is not None and line.startswith("dvo::") and not line.startswith("dvo::k_depth_update"):
                        break
                if grid and valu:
                    valu_px, src = valu * 64.0 / grid, "profiles/" + os.path.basename(fn)
                    break
            upd_us = pm["depth_update_ms"] / pm["frames"] * 1e3
            reg_us = pm["regularize_ms"] / pm["frames"] * 1e3
            mk = {"k_depth_update": {"avg_launch_us": upd_us, "pixels_per_launch": pm["update_window_pixels"],
                                     "note": "k_age_table + k_depth_update (Mapper::update, mapper.cpp:76-137); ~1/6 of the sequences skip it on a keyframe frame"},
                  "k_regularize_redecimate": {"avg_launch_us": reg_us, "pixels_per_launch": pm["map_pixels"],
                                              "achieved_GBps": 28.0 * pm["map_pixels"] / (reg_us * 1e-6) / 1e9,
                                              "bytes_note": "12 B read + 16 B written per top-level pixel (depth, sigma in; depth, and the re-decimated depth / sigma / weight levels out)"},
                  "k_propagate_x3": {"avg_launch_us": pm["propagate_ms"] / pm["frames"] * 1e3}}
            if valu_px:
                rate = valu_px * pm["update_window_pixels"] / (upd_us * 1e-6)
                mk["k_depth_update"].update({"bound": "fp32 vector issue", "valu_lane_instructions_per_pixel": valu_px, "source": src,
                                             "achieved_lane_instructions_per_s": rate, "peak_lane_instructions_per_s": VALU_ISSUE_PEAK,
                                             "frac": rate / VALU_ISSUE_PEAK})
            out["mapping_kernels"] = mk
    if env.solo and not a.no_cpu_baseline:
        orc = env.orc
        g0 = gray[:, 0].cpu().numpy()
        if raw:
            g0 = g0.astype(np.float32) * np.float32(1.0 / 255.0)
        if init_depth is not None:
            d0 = init_depth[0].cpu().numpy()
        else:   # the library's default initial map (hash-based N(1.5, 0.5) >= 0.5, seed = cfg.rng_seed): read it back from a 1-sequence batch
            tb = dvo.MonoBatch(1, K, W, H, cfg=dvo.default_config(device=env.local, rng_seed=1))
            g00 = gray[0, :1].contiguous()
            if raw:
                tb.odometrize_raw_device(g00.data_ptr(), 1)
            else:
                tb.odometrize_device(g00.data_ptr())
            d0 = tb.keyframe(0)["depth"].copy()   # (depth of the first keyframe = the initial map: no mapping has run yet)
            tb.close()
        # the same frames of sequence 0 through the oracle's VisualOdometry with the same initial map: trajectory agreement (ATE)
        ovo = orc.OVO(K, W, H, seed=1)
        ovo.set_initial_depth(d0, np.full_like(d0, 0.5))
        order = [0] + [ring_index(1 + k, F) for k in range(a.warmup + a.steps)]
        ores = [ovo.odometrize(g0[f]) for f in order][1 + a.warmup:]
        To = [r[0] for r in ores]
        Tg = [synth.se3_exp_np(x.astype(np.float64)) for x in poses_out[:, 0].cpu().numpy()]
        dpos = np.array([np.linalg.norm(np.linalg.inv(tg)[:3, 3] - np.linalg.inv(np.asarray(to, np.float64))[:3, 3]) for tg, to in zip(Tg, To)])
        keys_g = keys_out[:, 0].cpu().numpy().astype(bool)
        out["accuracy"] = {"trajectory_rmse_vs_cpu_oracle_m": float(np.sqrt(np.mean(dpos ** 2))), "max_m": float(dpos.max()),
                           "first_5_frames_max_m": float(dpos[:5].max()), "frames": len(dpos),
                           "keyframe_decisions_equal": bool(np.array_equal(keys_g, np.array([r[1] for r in ores], bool))),
                           "note": "camera positions of sequence 0 over the timed frames, GPU batch vs the CPU oracle run on the same frames with "
                                   "the same initial map (unaligned); the mapping amplifies last-bit pose differences frame over frame "
                                   "(DESIGN.md §6), so agreement decays along the sequence"}
        # CPU baseline of this workload: the oracle's odometrize on sequence 0's frames (a side leg gets a shorter sample)
        def cpu(budget):
            vo = orc.OVO(K, W, H, seed=1, variant=1)
            vo.set_initial_depth(d0, np.full_like(d0, 0.5))
            vo.odometrize(g0[0])
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < budget:
                vo.odometrize(g0[ring_index(1 + n, F)]); n += 1
            return n / (time.perf_counter() - t0), n
        budget = a.cpu_seconds * 0.5 if head else 2.5
        one, n1 = cpu(budget)
        orc.set_threads(env.ncore)
        allc, na = cpu(budget)
        orc.set_threads(1)
        out["cpu_baseline"] = {"value": allc, "unit": "frames/s", "cores": env.ncore, "kind": "port",
                               "sample": "%d frames of sequence 0 through the oracle's VisualOdometry::odometrize ('faithful' tracker variant, "
                                         "forEach bodies of the tracker row-parallel over %d threads; the mapper loops are sequential)" % (na, env.ncore),
                               "one_core": {"value": one, "cores": 1, "sample": "%d frames, 1 thread" % n1}, "cpu": _cpu_model()}
    return out


def _brief(o, extra=()):
    """what a side leg contributes to the headline's JSON line"""
    keep = ["value", "unit", "ms_per_step", "steps", "warmup", "roofline", "mapping_kernels", "accuracy"] + list(extra)
    r = {k: o[k] for k in keep if k in o}
    r["config"] = {k: v for k, v in o["config"].items() if k not in ("pyramid", "ring_note")}
    return r


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_spawn_ranks(a.gpus))
    env = Env(a)
    import dvo_amd as dvo  # noqa: F401  (fails loudly when libdvo.so is missing: there is no CPU fallback)
    fail = None
    if a.workload == "syn640-mono":
        out = run_mono(a, env)
    else:
        out, data, fail = run_depth(a, env, "headline")
        if a.workload == "syn640" and env.solo and not a.no_secondary and a.input == "raw" and abs(a.sigma - 0.1) < 1e-9:
            sec = out.setdefault("secondary", {})
            t_legs = time.time()
            # BASELINE configs[2]: the mono pipeline on the first 8192 sequences' gray frames (already in HBM)
            am = copy.copy(a); am.workload = "syn640-mono"; am.batch = min(8192, a.batch); am.steps = min(a.steps, 20); am.warmup = min(a.warmup, 3)
            data["depth16_keep"] = data["depth16"]
            sec["mono_batch"] = _brief(run_mono(am, env, role="secondary", gray=data["gray8"]), extra=("metric", "cpu_baseline"))
            cppm = (sec.get("single_stream_cpp") or {}).get("mono_float_pageable")
            cbm = sec["mono_batch"].get("cpu_baseline")
            if cppm and cbm:   # the reference's main loop (main.cpp:49) as one stream, against the CPU oracle's odometrize on the same kind of frames
                sec["single_stream_mono_vs_cpu_baseline"] = {"single_stream_fps": cppm["fps"], "x_all_cores": cppm["fps"] / cbm["value"],
                                                             "x_one_core": cppm["fps"] / cbm["one_core"]["value"],
                                                             "note": "vo.odometrize(gray) per frame from C++ through include/dvo.hpp (float gray from pageable "
                                                                     "host memory, tracking + Mapper::estimate + regularize) against the CPU oracle's "
                                                                     "VisualOdometry::odometrize"}
            # a converging configuration beside the headline (the reference's constants cannot converge at sigma = 0.1): step literals
            # halved and sigma 0.5 -> the update IS the Gauss-Newton step (gain 1 at levels 0 and 3), stop on the update norm only
            ac = copy.copy(a); ac.batch = min(1024, a.batch); ac.sigma = 0.5; ac.input = "float"; ac.step_scale = 0.5; ac.min_residual = 0.0
            ac.steps = min(a.steps, 10); ac.warmup = min(a.warmup, 2)
            oc, _, fc = run_depth(ac, env, "secondary", data=data)
            sec["converging_gain1"] = _brief(oc)
            fail = fail or fc
            del data
            torch.cuda.empty_cache()
            # BASELINE configs[3]: SYN-1080 dense alignment, 5 levels, 10 iterations per level
            a8 = copy.copy(a); a8.workload = "syn1080"; a8.batch = 128; a8.fixed_iters = 10; a8.steps = min(a.steps, 10); a8.warmup = min(a.warmup, 2)
            o8, _, f8 = run_depth(a8, env, "secondary")
            sec["syn1080"] = _brief(o8, extra=("metric",))
            fail = fail or f8
            sec["legs_seconds"] = round(time.time() - t_legs, 1)
    if env.rank == 0:
        print(json.dumps(out))
    if env.dist:
        env.dist.destroy_process_group()
    if fail is not None:
        sys.stderr.write("bench.py: per-iteration GPU / oracle parity FAILED: %s\n" % json.dumps(fail))
        raise SystemExit(3)


def _natural(path):
    """Sort key that orders r01_v9 before r01_v10 (newest profile summary last)."""
    import re
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(path))]


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip() + " (%d logical cores visible)" % os.cpu_count()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
