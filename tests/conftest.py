import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_sees_the_gpu_first(request):
    """GPU runs only: torch initialises its HIP runtime BEFORE the library makes its first HIP call -- the other way round
    torch reports "No HIP GPUs are available" and the tests that hand torch device buffers to the C ABI would skip."""
    expr = request.config.getoption("-m") or ""
    if "not gpu" not in expr and os.path.exists("/dev/kfd"):   # (a GPU box; the CPU suite never touches the GPU)
        try:
            import torch

            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:   # no torch / no GPU: the tests that need it say so themselves
            pass
    yield
