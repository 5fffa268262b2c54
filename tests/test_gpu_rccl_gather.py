"""dvo_batch_gather_poses_rccl (include/dvo.h): the path's one collective from the C ABI, on a single-rank RCCL communicator (the
8-GPU run is the driver's: SURVEY.md §8e).  The communicator is made through librccl's own C API via ctypes -- what a C++ host does
with ncclGetUniqueId / ncclCommInitRank."""
import ctypes as C

import numpy as np
import pytest

import dvo_amd as dvo
from util import K640, frames

pytestmark = pytest.mark.gpu


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def test_gather_over_a_single_rank_communicator_returns_the_batch_poses():
    import torch
    try:
        rccl = C.CDLL("librccl.so")
    except OSError:
        pytest.skip("librccl.so not loadable")
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        g, d, s, _ = frames(4, sigma=0.5)
        B = 3
        bt = dvo.Batch(B, K640, 640, 480, 4, 1)
        for step in range(2):
            bt.push_host(np.stack([g[step + b % 2] for b in range(B)]), np.stack([d[step + b % 2] for b in range(B)]),
                         np.stack([s[step + b % 2] for b in range(B)]))
        want = bt.last_poses()[0]
        out = torch.zeros((1, B, 6), dtype=torch.float32, device="cuda")
        assert dvo.lib().dvo_batch_gather_poses_rccl(bt._p, comm, 1, C.c_void_p(out.data_ptr())) == 0
        bt.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy()[0], want)
        assert np.abs(want).max() > 0
        bt.close()
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
