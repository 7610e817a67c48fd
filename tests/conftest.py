import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure), built on demand with gcc."""
    from oracle import capi

    capi.lib()
    return capi


@pytest.fixture(scope="session")
def prl():
    """The product's Python host layer over libprlib_hip.so; builds the library if it is missing."""
    import __graft_entry__ as ge

    ge.build_hip_library()
    import prlib_amd

    return prlib_amd


@pytest.fixture(scope="session")
def cuda_device(prl):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return torch.device("cuda:0")
