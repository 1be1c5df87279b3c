"""The output formatter on the device (ff_kernels_fmt.hpp, ff_format_distances_device) against the host's
ff_format_float -- itself held to the oracle's restatement of Go's rule (tests/test_host_cpu.py) and to
std::to_chars (csrc/fmt_selftest.cpp).  Replaces the loop of frcfrc/frcfrc.go:58-62."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from frackyfrac_amd import _lib as L
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def device_text(values: np.ndarray) -> bytes:
    import torch

    v = torch.from_numpy(np.ascontiguousarray(values, dtype=np.float64)).cuda()
    n = int(v.numel())
    cap = int(L.lib().ff_text_bound(n))
    assert cap == 25 * n
    text = torch.full((max(cap, 1) + 64,), 0x7E, dtype=torch.uint8, device="cuda")  # '~': never printed
    nbytes = ctypes.c_size_t(0)
    err = L.errbuf()
    L.check(L.lib().ff_format_distances_device(ctypes.c_void_p(v.data_ptr()), n, ctypes.c_void_p(text.data_ptr()),
                                               ctypes.byref(nbytes), None, err, L.ERRLEN), err)
    torch.cuda.synchronize()
    host = text.cpu().numpy()
    assert (host[nbytes.value:] == 0x7E).all(), "wrote past the text's end"
    return host[:nbytes.value].tobytes()


def host_text(values: np.ndarray) -> bytes:
    buf = ctypes.create_string_buffer(40)
    out = []
    f = L.lib().ff_format_float
    for x in values:
        n = f(float(x), buf)
        out.append(buf.raw[:n])
    return b"\n".join(out) + b"\n" if out else b""


def test_specials_edges_and_every_layout():
    vals = [0.0, -0.0, float("nan"), float("inf"), float("-inf"), 1.0, -1.0, 0.5, 0.1, 1e-5, 9.999e-5, 1e-4, 123456.0, 999999.0,
            1e6, 1e21, 1e22, 1e23, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, 0.3333333333333333,
            2 / 3, 1e100, 1e-100, 123.456, 100000.0, 1234567.0, 0.000123, 12345678901234567890.0, 4.35, 0.30000000000000004]
    vals += [float(np.ldexp(1.0, e)) for e in range(-1074, 1024, 7)]
    vals += [-v for v in vals[5:40]]
    v = np.array(vals, dtype=np.float64)
    got = device_text(v)
    assert got == host_text(v)
    # and the oracle's Go %v restatement says the same
    assert got.decode() == "".join(O.format_go_float(float(x)) + "\n" for x in v)


@pytest.mark.parametrize("n", [1, 3, 1023, 1024, 1025, 4096 + 17, 300_001])
def test_block_boundaries_and_ragged_counts(n):
    rng = np.random.default_rng(n)
    v = rng.random(n)
    v[rng.random(n) < 0.05] = 0.0   # lines of one character next to lines of twenty
    v[rng.random(n) < 0.05] = np.nan
    assert device_text(v) == host_text(v)


def test_random_bit_patterns_and_quotients():
    rng = np.random.default_rng(7)
    bits = rng.integers(0, 2 ** 64, size=400_000, dtype=np.uint64)
    v = bits.view(np.float64)
    assert device_text(v) == host_text(v)
    a = rng.integers(0, 2 ** 31, size=400_000).astype(np.float64)
    b = rng.integers(1, 2 ** 31, size=400_000).astype(np.float64)
    q = a / (a + b)  # what a distance is
    assert device_text(q) == host_text(q)


def test_a_pass_worth_of_values_reads_back_as_the_same_doubles():
    """2^23 values: several scan rounds of fmt_scan_kernel; every line parses back to its value."""
    rng = np.random.default_rng(11)
    v = rng.random(1 << 23)
    text = device_text(v)
    back = np.array(text.split(b"\n")[:-1], dtype=np.float64)
    assert back.shape == v.shape and np.array_equal(back, v)
    assert len(text) == sum(len(t) + 1 for t in text.split(b"\n")[:-1])


def test_empty_and_bad_arguments():
    nbytes = ctypes.c_size_t(5)
    err = L.errbuf()
    assert L.lib().ff_format_distances_device(None, 0, None, ctypes.byref(nbytes), None, err, L.ERRLEN) == 0 and nbytes.value == 0
    assert L.lib().ff_format_distances_device(None, 4, None, ctypes.byref(nbytes), None, err, L.ERRLEN) == L.FF_ERR_ARG
    assert L.lib().ff_text_bound(0) == 0 and L.lib().ff_text_bound(-3) == 0
