import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def checker():
    """Strongest CPU checker available: the real reference (.so prebuilt from /root/reference) or the C restatement."""
    import oracle
    oracle.build(("port",))
    return oracle.best()


@pytest.fixture(scope="session")
def port():
    import oracle
    oracle.build(("port",))
    return oracle.port()


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (gpu tests never fall back to CPU)")
    import feathercnn_amd
    feathercnn_amd.load_library()  # fails loudly if the HIP extension is missing
    return torch.device("cuda:0")
