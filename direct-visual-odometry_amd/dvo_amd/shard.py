"""Multi-GPU sharding of independent sequences (BASELINE config 5).

The path shards ACROSS sequences only: frame t of a sequence tracks against state produced by frames < t
(include/system/system.hpp:48,57,67), so one sequence never spans GPUs.  Each rank owns a contiguous block of
sequences and tracks them with no communication; the single collective is the end-of-run gather of the pose
arrays (tens of KB per rank: latency bound, the xGMI link bandwidth is irrelevant at this size).

Works with backend "nccl" (= RCCL over xGMI on ROCm) for device tensors and "gloo" for CPU tensors (tests).
"""
import torch
import torch.distributed as dist


def assign_sequences(n_sequences, world_size):
    """Contiguous block distribution: returns [(first, count)] per rank; counts differ by at most one."""
    if n_sequences < 0 or world_size < 1:
        raise ValueError("bad arguments")
    base, extra = divmod(n_sequences, world_size)
    out, first = [], 0
    for r in range(world_size):
        cnt = base + (1 if r < extra else 0)
        out.append((first, cnt))
        first += cnt
    return out


def gather_poses(local_poses, lengths=None, group=None):
    """All-gather per-sequence pose arrays.

    local_poses: float32 tensor [n_local_sequences, max_frames_local, D] (D = 6 twists or 16 matrix entries);
    lengths:     optional int tensor/list [n_local_sequences] of valid frames per sequence (ragged sequences).
    Returns (poses [n_total, max_frames, D], lengths [n_total]) on every rank, sequences in global order.
    Ranks may own different numbers of sequences and frames: everything is padded to the maxima, plus one
    length word per sequence, exactly as SURVEY.md §8e describes.
    """
    if not (dist.is_available() and dist.is_initialized()):
        n = local_poses.shape[0]
        ln = torch.as_tensor(lengths if lengths is not None else [local_poses.shape[1]] * n, dtype=torch.int64)
        return local_poses, ln
    world = dist.get_world_size(group)
    dev = local_poses.device
    n_local, f_local, D = local_poses.shape
    if lengths is None:
        lengths = [f_local] * n_local
    ln_local = torch.as_tensor(lengths, dtype=torch.int64, device=dev)
    # 1) shapes: (n_local, f_local) of every rank
    shape = torch.tensor([n_local, f_local], dtype=torch.int64, device=dev)
    shapes = [torch.zeros_like(shape) for _ in range(world)]
    dist.all_gather(shapes, shape, group=group)
    n_max = int(max(s[0].item() for s in shapes))
    f_max = int(max(s[1].item() for s in shapes))
    # 2) padded payload: poses + one length word per sequence
    pad = torch.zeros((n_max, f_max, D), dtype=local_poses.dtype, device=dev)
    pad[:n_local, :f_local] = local_poses
    lpad = torch.zeros((n_max,), dtype=torch.int64, device=dev)
    lpad[:n_local] = ln_local
    all_p = [torch.zeros_like(pad) for _ in range(world)]
    all_l = [torch.zeros_like(lpad) for _ in range(world)]
    dist.all_gather(all_p, pad, group=group)
    dist.all_gather(all_l, lpad, group=group)
    poses = torch.cat([all_p[r][: int(shapes[r][0].item())] for r in range(world)], dim=0)
    lens = torch.cat([all_l[r][: int(shapes[r][0].item())] for r in range(world)], dim=0)
    return poses, lens
