"""CPU emulation of the FIXED32 arithmetic (staging, integer sums, refine rule) against
the oracle, on the low-diversity inputs that correlate the rounding residuals: equal
branch lengths with repeated counts.  A design tool that uses the oracle as its checker (hence under tests/; a
script, not collected by pytest), not a product path: the numbers it
printed decided the staging rule of stage_fixed32_kernel (per-branch shared dither) and
the refine threshold of finish_fixed32_kernel (DESIGN.md "Arithmetic").

    python tests/emulate_fixed32.py [n_samples] [n_leaves] [density]
"""
import math
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from frackyfrac_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

LIMIT = 2147483647.0


def dither_u(b):
    """The per-branch offset in [0, 1): a 32-bit integer hash of the branch id (the
    device code of ff_kernels_stage.hpp uses the same constants)."""
    x = np.asarray(b, dtype=np.uint64) + np.uint64(1)
    x = (x * np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x ^= x >> np.uint64(32)
    x = (x * np.uint64(0xD6E8FEB86659FD93)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x ^= x >> np.uint64(32)
    return (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def emulate(ft, ip, on, weighted, mode):
    N = len(ip) - 1
    B = ft.n
    ids, ab = on["id"].astype(np.int64), on["abnd"]
    ln = ft.dist
    wex = np.zeros(N)
    for s in range(N):
        a, b = ip[s], ip[s + 1]
        wex[s] = np.sum(ln[ids[a:b]] * (ab[a:b] if weighted else 1.0))
    nnz_max = int(np.max(np.diff(ip)))
    if weighted:
        _, ex = math.frexp((LIMIT - nnz_max - 2.0) / wex.max())
    else:
        _, ex = math.frexp((LIMIT - B - 2.0) / ln.sum())
    e = ex - 1
    u = dither_u(np.arange(B))
    Q = np.zeros((N, B), dtype=np.int64)
    for s in range(N):
        a, b = ip[s], ip[s + 1]
        v = np.ldexp(ln[ids[a:b]] * (ab[a:b] if weighted else 1.0), e)
        if mode == "rint":
            q = np.rint(v)
        else:
            q = np.floor(v + u[ids[a:b]])
        Q[s, ids[a:b]] = q.astype(np.int64)
    W = Q.sum(axis=1)
    assert W.max() <= LIMIT
    out, queued = [], 0
    k = np.diff(ip)
    for i in range(N):
        for j in range(i):
            U = int(np.abs(Q[i] - Q[j]).sum())
            if mode == "rint":
                den = float(W[i] + W[j])
                sigma = math.sqrt((k[i] + k[j]) / 12.0)
            else:
                den = math.ldexp(wex[i] + wex[j], e)
                sigma = math.sqrt((k[i] + k[j]) / 6.0)
            if weighted:
                d = U / den
            else:
                d = 2.0 * U / (den + U) if mode != "rint" else U / (U + (W[i] + W[j] - U) // 2)
            if U * 0.5e-6 < 6.0 * sigma + 1.0:
                queued += 1
            out.append(d)
    return np.array(out), queued


def case(name, ns, nl, dens, lengths, counts, leave, weighted=True):
    tree, ptr, idx, val = synth.make(ns, nl, dens, 5)
    rng = np.random.default_rng(11)
    if lengths is not None:
        tree.branch_len[:] = lengths
        tree.branch_len[0] = 0.0
    if counts == "ones":
        val = np.ones_like(val)
    elif counts == "low":
        r = rng.random(len(val))
        val = np.where(r < 0.70, 1.0, np.where(r < 0.91, 2.0, 3.0 + np.floor(3 * rng.random(len(val)))))
    ft = O.FlatTree(tree.names, tree.branch_len, tree.size, tree.parent)
    ip, on = O.flatten_samples(ft, ptr, idx, val, 2 if leave else 0)
    want = O.unifrac_dists(ip, on, ft.dist, weighted)
    for mode in ("rint", "dither"):
        got, queued = emulate(ft, ip, on, weighted, mode)
        rel = np.max(np.abs(got - want) / np.abs(want))
        print("%-46s %-6s worst rel err %.2e  queued %d/%d" % (name, mode, rel, queued, len(want)))


if __name__ == "__main__":
    ns = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    nl = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    dens = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
    case("unit lengths, counts 1, normalised", ns, nl, dens, 1.0, "ones", False)
    case("unit lengths, counts 1, -l", ns, nl, dens, 1.0, "ones", True)
    case("lengths 0.1, counts mostly 1-2, normalised", ns, nl, dens, 0.1, "low", False)
    case("lengths 0.1, counts mostly 1-2, -l", ns, nl, dens, 0.1, "low", True)
    case("unit lengths, counts mostly 1-2, normalised", ns, nl, dens, 1.0, "low", False)
    case("synthetic dyadic lengths, counts 1..1000", ns, nl, dens, None, None, False)
    case("unweighted, lengths 0.1", ns, nl, dens, 0.1, "ones", False, weighted=False)
    case("unweighted, lengths {0.1,0.2,0.3}", ns, nl, dens,
         np.random.default_rng(3).integers(1, 4, size=2 * nl - 1) / 10.0, "ones", False, weighted=False)
