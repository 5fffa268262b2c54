import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "direct-visual-odometry_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

# Load order matters on a ROCm box: torch ships its own libamdhip64; import it BEFORE libdvo.so pulls in a HIP
# runtime, exactly as bench.py does, so one process never initialises two different runtimes.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timeout: per-test timeout (pytest-timeout)")


def _has_gpu():
    try:
        if torch is not None and torch.cuda.is_available():
            return True
    except Exception:
        pass
    try:
        import dvo_amd
        return dvo_amd.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
