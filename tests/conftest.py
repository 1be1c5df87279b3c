import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: wall-clock bounds on an MI355X; run ONLY when asked for by name "
                                       "(-m perf): a slower-clocked box must not turn the parity record red")


def pytest_collection_modifyitems(config, items):
    """`perf` tests assert times.  They are collected by `-m perf` alone: under any other selection (`-m gpu`, the
    parity gate; `-m "not gpu"`, the CPU suite) they are deselected, so no clock speed can fail either."""
    if "perf" in (config.getoption("-m") or ""):
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("perf") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


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
