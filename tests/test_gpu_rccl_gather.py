"""dvo_batch_gather_poses_rccl (include/dvo.h): the path's one collective from the C ABI, on a single-rank RCCL communicator (the
8-GPU run is the driver's: SURVEY.md §8e).  The work happens in a child process (tests/rccl_gather_child.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_gather_over_a_single_rank_communicator_returns_the_batch_poses():
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_gather_child.py")
    r = subprocess.run([sys.executable, child], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    if "SKIP" in r.stdout:
        pytest.skip(r.stdout.strip())
    assert "OK gathered" in r.stdout
