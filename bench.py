#!/usr/bin/env python3
"""bench.py -- tracked frames/sec of the MI355X-native direct-VO hot path (BASELINE.json metric).

One "step" = every sequence of the batch ingests its next 640x480 frame (gray + sensor depth + sigma, already
resident in HBM), builds the 4-level pyramid and tracks it against its previous frame with the reference's
coarse-to-fine Gauss-Newton loop (odometrizeUsingDepth, include/system/system.hpp:77-93) -> one relative pose
per sequence, left in HBM.  value = sequences x steps x ranks / wall time.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Independent sequences shard one batch per GPU (weak scaling); the only collective is the RCCL all_gather of
the pose arrays after the timed steps (BASELINE config 5), timed separately as gather_ms.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "direct-visual-odometry_amd"))

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
GN_BYTES_PER_PIXEL = 16     # obj_gray + ref_gray + ref_depth + ref_sigma


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="sequences tracked concurrently per GPU (0 = 16384 for syn640 with raw input, 4096 with float "
                                                         "maps or for syn640-mono, 128 for syn1080)")
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
    ap.add_argument("--no-secondary", action="store_true", help="skip the single-stream side measurements")
    ap.add_argument("--pcie-steps", type=int, default=16, help="steps of the PCIe-inclusive side measurement (0 = skip)")
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


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_spawn_ranks(a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or plain `python bench.py --gpus %d`)"
                         % (a.gpus, world, a.gpus, a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libdvo has no CPU fallback)")
    # DVO_BENCH_REHEARSE=1: multi-rank control flow on ONE GPU (every rank on cuda:0, gloo collectives on CPU tensors);
    # only for rehearsing the N > 1 path on a 1-GPU box, never for reported numbers.
    rehearse = os.environ.get("DVO_BENCH_REHEARSE") == "1"
    local = 0 if rehearse else local
    if not rehearse and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks but %d visible GPUs" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = torch.device("cpu") if rehearse else dev     # where collective payloads live
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI

    import dvo_amd as dvo
    from dvo_amd import synth

    if a.workload == "syn640-mono":
        return main_mono(a, rank, local, world, dev, cdev, rehearse)
    if a.workload == "syn640":
        W, H, K, levels, culls = 640, 480, synth.K_640, 4, 1          # Frame(g,d,s,K,4,1), system.hpp:82
    else:
        W, H, K, levels, culls = 1920, 1080, synth.K_1080, 5, 0
        if a.fixed_iters == 0:
            a.fixed_iters = 10
    if a.batch <= 0:
        # the latency-bound parts of a step (coarse levels, solves, the tail of each level) amortise over more sequences per launch:
        # 251 k / 263 k / 270 k frames/s at 4096 / 8192 / 16384.  16384 raw sequences = 90 GB of input frames + 80 GB of pyramids.
        a.batch = (16384 if (a.input == "raw" and abs(a.sigma - 0.1) < 1e-9) else 4096) if a.workload == "syn640" else 128
    B, F = a.batch, max(2, a.frames)

    # ---- synthetic sequences rendered straight into HBM: [F][B][H][W] --------------------------------
    raw = a.input == "raw" and abs(a.sigma - 0.1) < 1e-9   # (raw frames carry the sensor sigma of transform.cpp:75; other sigmas need float maps)
    t_gen = time.time()
    if raw:
        gray8 = torch.empty((F, B, H, W), dtype=torch.uint8, device=dev)
        depth16 = torch.empty((F, B, H, W), dtype=torch.int16, device=dev)   # (bit pattern of uint16: 1/5000 m units, TUM convention)
    else:
        gray = torch.empty((F, B, H, W), dtype=torch.float32, device=dev)
        depth = torch.empty_like(gray)
    gt_poses = []  # world <- camera ground truth of the first sequences (accuracy sample)
    all_poses = []
    for b in range(B):
        poses = synth.trajectory(F, seed=42 + 1000 * rank + b)
        if b < 32:
            gt_poses.append(poses)
        all_poses.append(poses)
    chunk = max(1, (96 if W == 640 else 12) // F)  # sequences rendered per call (float64 temporaries: ~20 x chunk x F frames)
    for b0 in range(0, B, chunk):
        b1 = min(B, b0 + chunk)
        Ts = np.stack([all_poses[b][f] for b in range(b0, b1) for f in range(F)])
        g, d = synth.render_batch(Ts, K, W, H, device=dev, newton_iters=6)
        g = g.reshape(b1 - b0, F, H, W).permute(1, 0, 2, 3); d = d.reshape(b1 - b0, F, H, W).permute(1, 0, 2, 3)
        if raw:
            gray8[:, b0:b1] = torch.clamp(torch.round(g * 255.0), 0, 255).to(torch.uint8)
            depth16[:, b0:b1] = torch.clamp(torch.round(d * 5000.0), 0, 65535).to(torch.int32).to(torch.int16)
        else:
            gray[:, b0:b1] = g; depth[:, b0:b1] = d
    del all_poses
    if not raw:
        sigma = torch.full_like(gray, a.sigma)
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen

    def host_frames(fsel, bsel):
        """float32 numpy (gray, depth, sigma) of frames fsel x sequences bsel, exactly what the device path computes from the input"""
        if not raw:
            return gray[fsel, bsel].cpu().numpy(), depth[fsel, bsel].cpu().numpy(), sigma[fsel, bsel].cpu().numpy()
        g8 = gray8[fsel, bsel].cpu().numpy(); d16 = depth16[fsel, bsel].cpu().numpy().view(np.uint16)
        gg = g8.astype(np.float32) * np.float32(1.0 / 255.0)
        gg[d16 == 0] = np.float32(-2.0)
        return gg, d16.astype(np.float32) * np.float32(1.0 / 5000.0), np.where(d16 > 0, np.float32(0.1), np.float32(1.0)).astype(np.float32)

    stream = torch.cuda.current_stream().cuda_stream
    cfg = dvo.default_config(device=local, stream=stream, fixed_iterations=a.fixed_iters, track_streams=a.streams,
                             crop_enable=1 if a.workload == "syn640" else 0)
    batch = dvo.Batch(B, K, W, H, levels, culls, cfg=cfg)
    poses_out = torch.zeros((a.steps, B, 6), dtype=torch.float32, device=dev)

    def push(bt, k, out=None):
        # frames are resident and complete.  Default: each push builds its pyramid on the tracking stream.  --prefetch: the pyramid of
        # frame k+1 is built on the library's side stream while frame k tracks (dvo_batch_prefetch_device); every step from k = 1 on
        # then prefetches exactly one frame and consumes the one prefetched by the step before.  Either way K timed steps contain K
        # pyramid builds.
        f = ring_index(k, F)
        if a.prefetch and k >= 1:
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

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    push(batch, 0)                          # first frame: reference only
    for k in range(a.warmup):
        push(batch, 1 + k)
    barrier()
    t0 = time.perf_counter()
    for k in range(a.steps):
        push(batch, 1 + a.warmup + k, poses_out[k])
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    gather_ms = 0.0
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # config 5: gather every rank's poses over RCCL/xGMI (tens of KB: latency bound)
        from dvo_amd import shard
        local_poses = poses_out.permute(1, 0, 2).contiguous().to(cdev)  # [sequence][frame][6]
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
    batch.close()   # (its pyramids are freed before the roofline pass allocates its own)

    out = {
        "metric": "tracked frames/sec (640x480 semi-dense) at 1 GPU" if a.workload == "syn640" else "tracked frames/sec (1920x1080 dense)",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "SYN-640 (stand-in for TUM fr1/desk: no dataset offline), 640x480, sensor depth, "
                               "frame-to-frame tracking, 4-level pyramid (320x240 top), reference thresholds"
                   if a.workload == "syn640" else "SYN-1080 dense alignment, 5-level pyramid, fixed iterations",
                   "input": "raw u8 gray + u16 depth (3 B/px), converted inside k_pyramid" if raw else "float32 gray + depth + sigma maps (12 B/px)",
                   "sequences_per_gpu": B, "frames_in_hbm_per_sequence": F, "sigma": a.sigma,
                   "pyramid": ("frame k+1's pyramid built on a low-priority side stream while frame k tracks (--prefetch)" if a.prefetch else
                               "every frame's pyramid built on the tracking stream ahead of its tracking (default; --prefetch overlaps it: "
                               "+1 % frames/s, k_track_gn per-launch time +7 %)"),
                   "fixed_iterations": a.fixed_iters, "iterations_per_level_seq0": log0["n_iter"],
                   "note": "sigma = 0.1 is the reference's sensor-depth constant (transform.cpp:75): the Gauss-Newton step is 10x over-relaxed "
                           "(optimize.cpp:83-89), most sequences run to max_iterations = 15 on every level without converging",
                   "poses_finite": finite, "gather_ms": gather_ms, "datagen_s": round(t_gen, 2)},
    }

    # ---- accuracy of the timed steps against the synthetic ground truth (BASELINE metric: "ATE RMSE vs reference") ------
    # exp(xi) maps reference-frame points into the new frame: T_rel = inv(P_new) P_ref for world<-camera poses P.
    def rel_errors(xis, pairs, poses):
        et, er = [], []
        for xi, (fr, fo) in zip(xis, pairs):
            E = synth.se3_exp_np(np.asarray(xi, np.float64)) @ np.linalg.inv(np.linalg.inv(poses[fo]) @ poses[fr])
            et.append(float(np.dot(E[:3, 3], E[:3, 3])))
            er.append(float(np.arccos(np.clip((np.trace(E[:3, :3]) - 1.0) / 2.0, -1.0, 1.0)) ** 2))
        return et, er

    step_pairs = [(ring_index(a.warmup + k, F), ring_index(1 + a.warmup + k, F)) for k in range(a.steps)]
    if rank == 0:
        xs = poses_out[:, :len(gt_poses)].cpu().numpy()  # [steps][sample][6]
        et, er = [], []
        for b, poses in enumerate(gt_poses):
            t_, r_ = rel_errors(xs[:, b], step_pairs, poses)
            if b == 0:
                seq0_rmse = float(np.sqrt(np.mean(t_)))
            et += t_; er += r_
        out["accuracy"] = {"rel_translation_rmse_m": float(np.sqrt(np.mean(et))), "rel_rotation_rmse_rad": float(np.sqrt(np.mean(er))),
                           "rel_translation_median_m": float(np.sqrt(np.median(et))), "rel_translation_p90_m": float(np.sqrt(np.percentile(et, 90))),
                           "sequence0_rel_translation_rmse_m": seq0_rmse,
                           "sample": "%d sequences x %d timed frame pairs vs the synthetic ground truth "
                                     "(per-frame motion ~ N(0, 5 mm / 0.3 deg))" % (len(gt_poses), a.steps)}

    # ---- PCIe-inclusive rate (reported in config, never `value`): the same steps fed from pinned HOST buffers ----
    if a.pcie_steps > 0:
        PB = min(B, 1024)  # a bounded sample of the batch: the rate is PCIe bound, pinned host copies of everything are not needed
        # (fixed launch schedule: the adaptive one keeps the host inside push() until the GPU is nearly done with the step, so the
        #  next frame's transfer would not be queued in time to overlap it)
        hcfg = dvo.default_config(device=local, stream=stream, fixed_iterations=a.fixed_iters, track_streams=a.streams, track_adaptive=-1,
                                  crop_enable=1 if a.workload == "syn640" else 0)
        hb = dvo.Batch(PB, K, W, H, levels, culls, cfg=hcfg)
        if raw:
            host = [(gray8[f, :PB].cpu().pin_memory(), depth16[f, :PB].cpu().pin_memory()) for f in range(min(F, 3))]
        else:
            host = [(gray[f, :PB].cpu().pin_memory(), depth[f, :PB].cpu().pin_memory(), sigma[f, :PB].cpu().pin_memory()) for f in range(min(F, 3))]
        xi_dev = torch.zeros((a.pcie_steps, PB, 6), dtype=torch.float32, device=dev)
        xi_pin = torch.zeros((a.pcie_steps, PB, 6), dtype=torch.float32).pin_memory()
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
            hpush(2 + k)                                         # H2D on the library's copy stream: overlaps the tracking of the step before
            hb.copy_poses_device(xi_dev[k].data_ptr())
            xi_pin[k].copy_(xi_dev[k], non_blocking=True)        # D2H of every pose of the step, asynchronous, same stream
        hb.synchronize()
        torch.cuda.synchronize()
        incl = PB * a.pcie_steps / (time.perf_counter() - t1)
        # co-headline (SURVEY.md §8d defines fps "including H2D of each gray frame and D2H of each pose"); `value` is HBM-resident
        out["value_incl_h2d"] = incl
        full_frames = os.environ.get("DVO_UPLOAD_FULL_FRAMES") is not None
        out["incl_h2d"] = {"sequences": PB, "frames_streamed_per_sequence": a.pcie_steps, "bytes_per_frame": (3 if raw else 12) * W * H,
                           "bytes_over_pcie_per_frame": (3 * W * (H >> culls if (culls > 0 and H % (1 << culls) == 0 and not full_frames) else H)) if raw else 12 * W * H,
                           "note": "raw host frames: only the rows the pyramid keeps (every 2^culls-th: Convert::cullImage) are transferred, by one strided copy per buffer",
                           "input": ("pinned host u8 gray + u16 depth (dvo_batch_push_raw_host)" if raw else
                                     "pinned host float32 gray + depth + sigma (dvo_batch_push_host)") + ", every pose copied back to pinned host memory per step; "
                                    "transfers of step k+1 overlap the tracking of step k (copy stream, two staging slots)"}
        hb.close()

    # ---- roofline of the dominant kernel (k_track_gn): HIP events around every launch of an identical pass ----
    if not a.no_roofline:
        pcfg = dvo.default_config(device=local, stream=stream, fixed_iterations=a.fixed_iters, profile=1, track_streams=a.streams,
                                  crop_enable=1 if a.workload == "syn640" else 0)
        pb = dvo.Batch(B, K, W, H, levels, culls, cfg=pcfg)
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
            bytes_per_launch = GN_BYTES_PER_PIXEL * pr["gn_pixels"] / pr["gn_launches"]
            ms_per_launch = pr["gn_ms"] / pr["gn_launches"]
            achieved = bytes_per_launch / (ms_per_launch * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "k_track_gn",
                               "avg_launch_us": ms_per_launch * 1e3, "launches": pr["gn_launches"],
                               "algorithmic_bytes_per_launch": bytes_per_launch,
                               "gn_share_of_step_time": pr["gn_ms"] / (dt * 1e3),
                               "top_level_probe": {"avg_launch_us": top_ms * 1e3,
                                                   "achieved": GN_BYTES_PER_PIXEL * top_px / (top_ms * 1e-3) / 1e9}}
            # HBM bytes per launch come from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over this same command
            # (tools/pmc_traffic.sh): a process cannot profile itself, so the committed summary for the matching
            # workload/batch is quoted here, and null is reported when there is none.
            # Instruction count of the committed PMC run, for reference only: the kernel's time does not follow it (DESIGN.md §10 --
            # plain FP32 instructions issue in ~2.9 cycles and beside the general class; the vector ALUs and the L1 / texture path
            # of the gathers are both 80 % busy).
            import glob
            for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*k_track_gn_pmc.txt")), key=_natural, reverse=True):
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
            if a.workload == "syn640" and a.fixed_iters == 0:
                for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json")), key=_natural, reverse=True):
                    with open(fn) as fh:
                        tr = json.load(fh)
                    if tr.get("kernel") == "k_track_gn" and tr.get("sequences_per_gpu") == B:
                        out["roofline"]["traffic"] = tr["traffic_bytes_per_launch"]
                        out["roofline"]["traffic_source"] = "profiles/" + os.path.basename(fn)
                        break

    # ---- secondary (N = 1 only, a few seconds): the single-sequence drop-in entry points, host frames in, pose out per call --
    # BASELINE configs[2] (tracking + inverse-depth filter) and the latency view of configs[1]; never part of `value`.
    if rank == 0 and world == 1 and a.workload == "syn640" and not a.no_secondary:
        g0, d0, s0 = host_frames(slice(None), 0)
        n_sec = 40
        vo = dvo.VisualOdometry(K, W, H, cfg=dvo.default_config(device=local))
        for k in range(3):
            f = ring_index(k, F); vo.odometrizeUsingDepth(g0[f], d0[f], s0[f])
        t1 = time.perf_counter(); its = 0
        for k in range(n_sec):
            f = ring_index(3 + k, F); vo.odometrizeUsingDepth(g0[f], d0[f], s0[f])
            its += sum(vo.lastTrackLog()["n_iter"])
        single_depth = n_sec / (time.perf_counter() - t1)
        vo.close()
        vo = dvo.VisualOdometry(K, W, H, cfg=dvo.default_config(device=local, rng_seed=1))
        di = d0[0][::4, ::4].copy()
        vo.setInitialDepth(di, np.full_like(di, 0.5))
        for k in range(3):
            vo.odometrize(g0[ring_index(k, F)])
        t1 = time.perf_counter(); keys = 0
        for k in range(n_sec):
            _, key = vo.odometrize(g0[ring_index(3 + k, F)]); keys += int(key)
        single_mono = n_sec / (time.perf_counter() - t1)
        vo.close()
        out["secondary"] = {"single_stream_odometrizeUsingDepth_fps": single_depth, "single_stream_odometrize_mono_track_map_fps": single_mono,
                            "depth_gn_iterations_per_frame": its / n_sec, "mono_keyframes": keys, "frames": n_sec,
                            "note": "one dvo_vo handle, 640x480 host frames in / pose out per call (PCIe and launch latency included)"}

    # ---- CPU baseline: the oracle on this box's host cores, bounded sample of the same workload ----------
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import orc
        from concurrent.futures import ThreadPoolExecutor
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        ncore = max(1, min(ncpu, 16))      # the box's CPU share for one GPU
        NS = len(gt_poses)
        gh, dh, sh = host_frames(slice(None), slice(0, NS))  # [F][NS][H][W]
        crop = a.workload == "syn640"

        def run(variant, budget):
            """frames/s of the oracle on sequence 0's frames: pyramid + track per frame, as one GPU step does per sequence"""
            n, t_start = 0, time.perf_counter()
            ref = orc.OFrame(gh[0, 0], dh[0, 0], sh[0, 0], K, levels, culls)
            k = 0
            while time.perf_counter() - t_start < budget:
                f = ring_index(1 + k, F)
                obj = orc.OFrame(gh[f, 0], dh[f, 0], sh[f, 0], K, levels, culls)
                orc.track(obj, ref, crop=crop, variant=variant, fixed_iters=a.fixed_iters)
                ref = obj
                n += 1; k += 1
            return n / (time.perf_counter() - t_start), n

        one_fps, n1 = run(1, a.cpu_seconds * 0.6)                 # faithful, 1 thread
        orc.set_threads(ncore)
        all_fps, na = run(1, a.cpu_seconds * 0.4)                 # faithful, forEach bodies row-parallel over the box's cores
        orc.set_threads(1)
        hoisted_fps, nh = run(0, max(2.0, a.cpu_seconds / 5))

        # The SAME sample the GPU accuracy was taken on: every timed frame pair of the first NS sequences through the (hoisted)
        # oracle, sequence-parallel on host threads (ctypes releases the GIL) -- shows whether the error tail is the algorithm's
        # (the reference's 10x over-relaxed step at sigma = 0.1: optimize.cpp:83-89, transform.cpp:75) or the GPU path's.
        if "accuracy" in out and gt_poses:
            def seq_job(b):
                res = []
                for (fr, fo) in step_pairs:
                    ref = orc.OFrame(gh[fr, b], dh[fr, b], sh[fr, b], K, levels, culls)
                    obj = orc.OFrame(gh[fo, b], dh[fo, b], sh[fo, b], K, levels, culls)
                    xi, lg = orc.track(obj, ref, crop=crop, variant=0, fixed_iters=a.fixed_iters)
                    res.append((xi, lg["n_iter"]))
                return res
            t_o = time.perf_counter()
            with ThreadPoolExecutor(ncore) as ex:
                o_res = list(ex.map(seq_job, range(NS)))
            t_o = time.perf_counter() - t_o
            xs = poses_out[:, :NS].cpu().numpy()
            per_seq, o_et, o_er, dx = [], [], [], []
            for b in range(NS):
                ox = [r[0] for r in o_res[b]]
                et_o, er_o = rel_errors(ox, step_pairs, gt_poses[b])
                et_g, er_g = rel_errors(xs[:, b], step_pairs, gt_poses[b])
                o_et += et_o; o_er += er_o
                d_ = np.abs(np.asarray(ox) - xs[:, b]).max(axis=1)   # |xi_gpu - xi_oracle|_inf per frame pair
                dx += list(d_)
                per_seq.append({"seq": b, "gpu_rmse_m": float(np.sqrt(np.mean(et_g))), "oracle_rmse_m": float(np.sqrt(np.mean(et_o))),
                                "max_abs_pose_diff": float(d_.max())})
            worst = sorted(per_seq, key=lambda r: -r["gpu_rmse_m"])[:4]
            acc = out["accuracy"]
            acc["cpu_oracle_rel_translation_rmse_m"] = float(np.sqrt(np.mean(o_et)))
            acc["cpu_oracle_rel_rotation_rmse_rad"] = float(np.sqrt(np.mean(o_er)))
            acc["cpu_oracle_rel_translation_median_m"] = float(np.sqrt(np.median(o_et)))
            acc["gpu_vs_oracle"] = {"pairs": len(dx), "median_abs_pose_diff": float(np.median(dx)), "p90_abs_pose_diff": float(np.percentile(dx, 90)),
                                    "max_abs_pose_diff": float(np.max(dx)), "pairs_within_1e-3": int(np.sum(np.asarray(dx) < 1e-3)),
                                    "worst_gpu_sequences": worst, "oracle_seconds": round(t_o, 1),
                                    "note": "same %d sequences x %d frame pairs through the CPU oracle; at sigma = 0.1 the reference's iteration is "
                                            "chaotic (gain 10), so whole-call poses agree only where both converge; per-iteration parity at "
                                            "scale is a -m gpu test (tests/test_gpu_parity_scale.py)" % (NS, len(step_pairs))}
        out["cpu_baseline"] = {"value": all_fps, "unit": "frames/s", "cores": ncore, "kind": "port",
                               "sample": "%d frame pairs of sequence 0 (same frames, pyramid + track), oracle 'faithful' variant (per-pixel se3 exp, "
                                         "materialised warpImage, Nx6 stack + SVD least squares) with the forEach bodies row-parallel over %d "
                                         "threads as cv::Mat::forEach (optimize.cpp:28, transform.cpp:39)" % (na, ncore),
                               "one_core": {"value": one_fps, "cores": 1, "sample": "%d frame pairs, same variant, 1 thread" % n1},
                               "hoisted_value": hoisted_fps, "hoisted_sample": "%d frame pairs, 1 thread, pose hoisted + 6x6 normal equations" % nh,
                               "cpu": _cpu_model()}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def main_mono(a, rank, local, world, dev, cdev, rehearse):
    """BASELINE configs[2]: 640x480 mono tracking + inverse-depth filter.  One step = every sequence runs
    System::VisualOdometry::odometrize (system.hpp:44-74) on its next gray frame: 3-level pyramid (160x120 top), Tracker::track
    against its newest keyframe, then Mapper::estimate (propagate + new keyframe, or stereo update against the keyframe each
    pixel was born in) and regularize -- dvo_batch_create_mono, keyframe decisions on the device, no host round trip."""
    import dvo_amd as dvo
    from dvo_amd import synth
    if world > 1:
        import torch.distributed as dist
    W, H, K = 640, 480, synth.K_640
    B = a.batch if a.batch > 0 else 8192   # 1.43 M / 1.51 M / 1.52 M frames/s at 4096 / 8192 / 16384 sequences (round 2)
    F = max(2, a.frames)
    t_gen = time.time()
    raw = a.input == "raw"
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
    init_sigma = torch.full_like(init_depth, 0.5) if init_depth is not None else None
    torch.cuda.synchronize()

    def odo(mb, f):
        if raw:
            mb.odometrize_raw_device(gray[f].data_ptr(), 1)     # u8 gray as cv::imread + cvtColor deliver it, converted in k_pyramid
        else:
            mb.odometrize_device(gray[f].data_ptr())
    t_gen = time.time() - t_gen
    stream = torch.cuda.current_stream().cuda_stream
    poses_out = torch.zeros((a.steps, B, 6), dtype=torch.float32, device=dev)
    keys_out = torch.zeros((a.steps, B), dtype=torch.int32, device=dev)

    def run(profile):
        cfg = dvo.default_config(device=local, stream=stream, profile=profile, rng_seed=1)
        mb = dvo.MonoBatch(B, K, W, H, ring_keyframes=a.ring, cfg=cfg)
        if init_depth is not None:
            mb.setInitialDepthDevice(init_depth.data_ptr(), init_sigma.data_ptr())
        odo(mb, 0)                                           # first frame: first keyframe of every sequence
        for k in range(a.warmup):
            odo(mb, ring_index(1 + k, F))
        if profile:
            mb.profile(reset=True)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(a.steps):
            odo(mb, ring_index(1 + a.warmup + k, F))
            mb.copy_world_poses_device(poses_out[k].data_ptr(), 0, keys_out[k].data_ptr())
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return mb, time.perf_counter() - t0

    mb, dt = run(0)
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    log0 = mb.last_track_log(0)
    kf0 = mb.keyframe(0)
    vu = [mb.keyframe(b)["valid_updates"] for b in range(min(B, 64))]
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
                      "poses_finite": bool(torch.isfinite(poses_out).all().item()), "datagen_s": round(t_gen, 2)}}
    if a.pcie_steps > 0:   # co-headline: every frame streamed from pinned host memory, every pose copied back
        PB = min(B, 1024)
        hb = dvo.MonoBatch(PB, K, W, H, ring_keyframes=a.ring, cfg=dvo.default_config(device=local, stream=stream, rng_seed=1))
        host = [gray[f, :PB].cpu().pin_memory() for f in range(min(F, 3))]
        xi_dev = torch.zeros((a.pcie_steps, PB, 6), dtype=torch.float32, device=dev)
        xi_pin = torch.zeros((a.pcie_steps, PB, 6), dtype=torch.float32).pin_memory()
        hb.odometrize_host(host[0].numpy()); hb.odometrize_host(host[1].numpy())
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(a.pcie_steps):
            hb.odometrize_host(host[ring_index(2 + k, len(host))].numpy())
            hb.copy_world_poses_device(xi_dev[k].data_ptr())
            xi_pin[k].copy_(xi_dev[k], non_blocking=True)
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
        pb.close()
        if pr["gn_launches"] > 0 and pr["gn_ms"] > 0:
            bpl = GN_BYTES_PER_PIXEL * pr["gn_pixels"] / pr["gn_launches"]
            mpl = pr["gn_ms"] / pr["gn_launches"]
            ach = bpl / (mpl * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                               "kernel": "k_track_gn", "avg_launch_us": mpl * 1e3, "launches": pr["gn_launches"],
                               "algorithmic_bytes_per_launch": bpl, "gn_share_of_step_time": pr["gn_ms"] / (dt * 1e3),
                               "note": "levels of 40x30 .. 160x120 only: latency-bound launches; the mapping kernels' figures are in "
                                       "profiles/r02_mono_*"}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import orc
        g0 = gray[:, 0].cpu().numpy()
        if raw:
            g0 = g0.astype(np.float32) * np.float32(1.0 / 255.0)

        if init_depth is not None:
            d0 = init_depth[0].cpu().numpy()
        else:   # the library's default initial map (hash-based N(1.5, 0.5) >= 0.5, seed = cfg.rng_seed): read it back from a 1-sequence batch
            tb = dvo.MonoBatch(1, K, W, H, cfg=dvo.default_config(device=local, rng_seed=1))
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
        To = [ovo.odometrize(g0[f])[0] for f in order][1 + a.warmup:]
        Tg = [synth.se3_exp_np(x.astype(np.float64)) for x in poses_out[:, 0].cpu().numpy()]
        dpos = np.array([np.linalg.norm(np.linalg.inv(tg)[:3, 3] - np.linalg.inv(np.asarray(to, np.float64))[:3, 3]) for tg, to in zip(Tg, To)])
        out["accuracy"] = {"trajectory_rmse_vs_cpu_oracle_m": float(np.sqrt(np.mean(dpos ** 2))), "max_m": float(dpos.max()),
                           "first_5_frames_max_m": float(dpos[:5].max()), "frames": len(dpos),
                           "note": "camera positions of sequence 0 over the timed frames, GPU batch vs the CPU oracle run on the same frames with "
                                   "the same initial map (unaligned); the mapping amplifies last-bit pose differences frame over frame "
                                   "(DESIGN.md §6), so agreement decays along the sequence"}

        def cpu(budget):
            vo = orc.OVO(K, W, H, seed=1, variant=1)
            vo.set_initial_depth(d0, np.full_like(d0, 0.5))
            vo.odometrize(g0[0])
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < budget:
                vo.odometrize(g0[ring_index(1 + n, F)]); n += 1
            return n / (time.perf_counter() - t0), n
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        ncore = max(1, min(ncpu, 16))
        one, n1 = cpu(a.cpu_seconds * 0.5)
        orc.set_threads(ncore)
        allc, na = cpu(a.cpu_seconds * 0.5)
        orc.set_threads(1)
        out["cpu_baseline"] = {"value": allc, "unit": "frames/s", "cores": ncore, "kind": "port",
                               "sample": "%d frames of sequence 0 through the oracle's VisualOdometry::odometrize ('faithful' tracker variant, "
                                         "forEach bodies of the tracker row-parallel over %d threads; the mapper loops are sequential)" % (na, ncore),
                               "one_core": {"value": one, "cores": 1, "sample": "%d frames, 1 thread" % n1}, "cpu": _cpu_model()}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


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
