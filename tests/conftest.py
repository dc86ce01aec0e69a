import os
import sys

import numpy as np
import pytest

try:        # torch first, whatever the test order: its wheel carries private copies of the ROCm runtime, and a process
    import torch  # noqa: F401  that loads libvamp_hip.so BEFORE torch ends up with two runtimes
except Exception:       # noqa: BLE001  (CPU-only tests do not need it)
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session", params=[0, 64, 16, 256, 65], ids=["pack-auto", "pack-64", "pack-16", "pack-256", "pack-64t"])
def hip_ctx(request):
    """fp64 / accurate-wofz context on device 0 (GPU tests only), once per walker packing:
    automatic, one walker per wavefront, four walkers per wavefront, one walker per workgroup, one
    walker per wavefront with its own Taylor tables (<= 8 lines)."""
    import vamp_amd
    ctx = vamp_amd.HipContext(device=0)
    ctx.set_packing(request.param)
    ctx.packing_request = request.param
    yield ctx
    ctx.close()
