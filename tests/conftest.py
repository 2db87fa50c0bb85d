import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible (tests never fall back to CPU)")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def no_probe_left_behind():
    """scda_amd.probe only ever exposes a Probe inside a trainer step or a with-block: nothing can be left behind -- checked anyway"""
    yield
    import sys
    if "scda_amd.probe" in sys.modules:
        assert not sys.modules["scda_amd.probe"].active(), "a scda_amd.probe.Probe is still installed after the test"
