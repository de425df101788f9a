import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # CPU tier: the SAD surfaces of the emulated ABI are an exhaustive search on the host (tests/support/la_emul.c over the oracle) — 4096 vectors per
    # block at the default range of 32 dominate the emulated encodes.  The bindings read the range from the environment and the bitstream does not depend
    # on it (a vector outside a window is computed by the C function), so the emulated encoders search +-12; on a GPU box the default stays.
    if not os.path.exists("/dev/kfd"):
        os.environ.setdefault("X265HIP_SADPLANES_RANGE", "12")


@pytest.fixture(scope="session")
def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
