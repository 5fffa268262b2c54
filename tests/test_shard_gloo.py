"""world_size-2 gloo test of the multi-GPU sharding host logic (dvo_amd/shard.py): sequence assignment and the
padded pose all-gather.  Runs on CPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dvo_amd import shard


def test_assign_sequences():
    assert shard.assign_sequences(8, 8) == [(i, 1) for i in range(8)]
    assert shard.assign_sequences(10, 4) == [(0, 3), (3, 3), (6, 2), (8, 2)]
    assert shard.assign_sequences(2, 4) == [(0, 1), (1, 1), (2, 0), (2, 0)]
    assert shard.assign_sequences(0, 2) == [(0, 0), (0, 0)]
    with pytest.raises(ValueError):
        shard.assign_sequences(3, 0)


def test_gather_without_process_group_is_identity():
    p = torch.arange(24, dtype=torch.float32).reshape(2, 2, 6)
    out, ln = shard.gather_poses(p, [2, 1])
    assert out is p and ln.tolist() == [2, 1]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, cnt = shard.assign_sequences(n_total, world)[rank]
    frames = 5 + 2 * rank                      # ragged: ranks hold different numbers of frames
    local = torch.zeros((cnt, frames, 6))
    lens = []
    for i in range(cnt):
        n = frames - (i % 2)
        lens.append(n)
        for f in range(n):
            local[i, f] = torch.tensor([first + i, f, 0, 0, 0, rank], dtype=torch.float32)
    poses, ln = shard.gather_poses(local, lens)
    q.put((rank, poses.numpy(), ln.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gather_poses_world_size_2():
    world, n_total = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    res.sort(key=lambda t: t[0])
    np.testing.assert_array_equal(res[0][1], res[1][1])        # every rank holds the same gathered result
    poses, ln = res[0][1], res[0][2]
    assert poses.shape == (5, 7, 6)                             # padded to the longest sequence
    assign = shard.assign_sequences(n_total, world)
    for r, (first, cnt) in enumerate(assign):
        for i in range(cnt):
            s = first + i
            n = (5 + 2 * r) - (i % 2)
            assert ln[s] == n
            for f in range(n):
                assert poses[s, f, 0] == s and poses[s, f, 1] == f and poses[s, f, 5] == r
            assert not poses[s, n:].any()


def test_c_abi_shard_range_equals_the_python_split():
    """dvo_shard_range (include/dvo.h): the block of sequences a rank owns, for a C++ host -- the same split as shard.assign_sequences."""
    import ctypes as C
    import dvo_amd as dvo
    from dvo_amd import shard
    L = dvo.lib()
    for n in (0, 1, 7, 8, 9, 16384, 100003):
        for world in (1, 2, 3, 4, 8):
            want = shard.assign_sequences(n, world)
            for r in range(world):
                first, count = C.c_int(), C.c_int()
                assert L.dvo_shard_range(n, world, r, C.byref(first), C.byref(count)) == 0
                assert (first.value, count.value) == want[r], (n, world, r)
    f, c = C.c_int(), C.c_int()
    assert L.dvo_shard_range(8, 0, 0, C.byref(f), C.byref(c)) == dvo.DVO_ERR_BAD_ARGUMENT
    assert L.dvo_shard_range(8, 2, 2, C.byref(f), C.byref(c)) == dvo.DVO_ERR_BAD_ARGUMENT
    assert L.dvo_batch_gather_poses_rccl(None, None, 1, None) == dvo.DVO_ERR_BAD_ARGUMENT
