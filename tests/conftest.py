import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# the persistent auction reports a timed-out team barrier instead of returning garbage (debug aid)
os.environ.setdefault("SN_EMD_CHECK", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")
