import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the product library and the oracle are built (both compile on CPU)."""
    import __graft_entry__ as ge

    ge.build(quiet=True)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def read_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return f.read()
