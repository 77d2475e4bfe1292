import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def hip():
    """The ctypes binding; GPU tests must run the real HIP library (no fallback exists)."""
    import torch
    from micro_diffusion_amd import hip as _hip
    assert torch.cuda.is_available(), "gpu-marked test started without a GPU"
    _hip.lib()
    return _hip
