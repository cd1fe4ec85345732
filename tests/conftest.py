import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# the persistent auction reports a timed-out team barrier instead of returning garbage (debug aid)
os.environ.setdefault("SN_EMD_CHECK", "1")
# the library reads its tuning / test knobs once per process; the tests switch them inside one process (SN_KNOB, common.hpp)
os.environ.setdefault("SN_KNOBS_PER_CALL", "1")
if os.environ.get("AB_LIB"):   # run the suite against an A/B build of the library (tools/build_variant.sh)
    import sparenet_amd._lib as _ab
    _ab.LIB_PATH = os.path.abspath(os.environ["AB_LIB"])


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
