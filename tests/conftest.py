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


# Modules whose tests are about plumbing AROUND the hot path (the bench line, the cgo shim's call sequence, the
# cross-process gather): collected after every comparison with the oracle, so that under `pytest -x` nothing about
# a profiler, a launcher or a transport can leave a parity test unreached.
_PLUMBING_MODULES = ("test_gpu_bench", "test_go_shim_harness", "test_gpu_ipc_gather")


def _perf_asked_for_by_name(markexpr):
    """True when the -m expression selects a test BECAUSE it carries `perf`: it accepts a test marked {perf, gpu}
    and rejects one marked {gpu} alone.  (`-m gpu` accepts both: the parity gate, perf stays out.  `-m "gpu and not
    perf"` rejects the first: pytest's own filter drops them.)"""
    if not markexpr:
        return False
    try:
        from _pytest.mark.expression import Expression

        e = Expression.compile(markexpr)

        def holds(marks):
            try:
                return bool(e.evaluate(lambda name, **kw: name in marks))
            except TypeError:  # (older pytest: matcher takes the name only)
                return bool(e.evaluate(lambda name: name in marks))

        return holds({"perf", "gpu"}) and not holds({"gpu"})
    except Exception:
        return markexpr.strip() == "perf"


def pytest_collection_modifyitems(config, items):
    """`perf` tests assert times.  They are collected by `-m perf` alone: under any other selection (`-m gpu`, the
    parity gate; `-m "not gpu"`, the CPU suite) they are deselected, so no clock speed can fail either.  And the
    plumbing modules go last (stable: everything else keeps pytest's order)."""
    if not _perf_asked_for_by_name(config.getoption("-m") or ""):
        keep, drop = [], []
        for it in items:
            (drop if it.get_closest_marker("perf") else keep).append(it)
        if drop:
            config.hook.pytest_deselected(items=drop)
            items[:] = keep
    items.sort(key=lambda it: 1 if getattr(it.module, "__name__", "").rsplit(".", 1)[-1] in _PLUMBING_MODULES else 0)


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
