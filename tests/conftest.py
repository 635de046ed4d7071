import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle_kernels():
    import oracle
    oracle.lib()
    return oracle.OracleKernels()


@pytest.fixture(scope="session")
def hip_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from nesie_amd import _lib
    _lib.load()  # a missing libnesie_hip.so must fail the GPU tier loudly
    return torch.device("cuda:0")
