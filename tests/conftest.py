import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle

    pyoracle.build()
    return pyoracle


@pytest.fixture(params=["leader-group-major", "slot-major"])
def row_layout(request):
    """Mencius contexts keep their per-slot rows leader-group-major in HBM unless FPX_F_SLOT_MAJOR_ROWS is set
    (include/fpx.h); FPX_SLOT_MAJOR=1 in the environment is the same switch for contexts the test does not build
    itself.  Tests that use this fixture run under both layouts: every result must be the same."""
    old = os.environ.pop("FPX_SLOT_MAJOR", None)
    if request.param == "slot-major":
        os.environ["FPX_SLOT_MAJOR"] = "1"
    yield request.param
    os.environ.pop("FPX_SLOT_MAJOR", None)
    if old is not None:
        os.environ["FPX_SLOT_MAJOR"] = old
