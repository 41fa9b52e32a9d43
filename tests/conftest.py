import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """Session fixture for -m gpu tests: fails (does not skip) when the HIP library cannot run —
    a GPU test that silently passes on a fallback is worse than a red one."""
    assert _have_gpu(), "this test needs a ROCm device"
    import a_link_amd  # noqa: F401
    from a_link_amd import _abi
    _abi.init(0)
    return _abi
