import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
os.environ["SCDA_ALLOW_TEST_HOOKS"] = "1"     # the parity tests steer the product path's three test hooks (train_step.active_test_hooks)


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
def no_test_hook_left_behind():
    """a hook one test forgets to clear would silently change every later test of the process"""
    yield
    import sys
    if "scda_amd.train_step" in sys.modules:
        left = sys.modules["scda_amd.train_step"].active_test_hooks()
        assert not left, "test left product-path hooks installed: %s" % left
