"""Parity at extreme aspect ratios on a GPU box -- tens of thousands of samples on a tree of two or twenty leaves, two or
three samples on a tree of hundreds of thousands -- through every precision, weighted and unweighted, against the oracle on
the first, middle and last 20,000 pairs (bit for bit for EXACT64 and unweighted, 1e-6 relative for weighted FIXED32).
Usage: python tests/fuzz_extreme_gpu.py   (a script, not collected by pytest; 36 cases, clean at round 4 HEAD)"""
import sys, os, numpy as np, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import frackyfrac_amd as ff
from frackyfrac_amd import synth
from oracle import oracle as O
bad = 0
for (n, leaves, dens) in [(30000, 20, 0.3), (3, 400000, 0.5), (20001, 2, 1.0), (2, 2, 1.0), (9000, 64, 0.05), (40, 200000, 0.01)]:
    tree, ptr, idx, val = synth.make(n, leaves, dens, 77)
    T = ff.parse_newick(tree.newick())
    nodes = ff.flatten_leaf_csr(T, ptr, idx, val)
    ft = O.FlatTree(tree.names, tree.branch_len, tree.size, tree.parent)
    ip, on = O.flatten_samples(ft, ptr, idx, val, 0)
    P = ff.num_pairs(n)
    for weighted in (True, False):
        for prec in ("fixed32", "exact64", "auto"):
            t0 = time.time()
            try:
                got = ff.unifrac_dists(nodes, weighted, precision=prec)
            except ff.FFError as e:
                if "FIXED32 not applicable" in str(e): print(n, leaves, weighted, prec, "not applicable"); continue
                raise
            ok = True
            for a in sorted({0, max(0, P // 2 - 10000), max(0, P - 20000)}):
                b = min(P, a + 20000)
                want = O.unifrac_dists(ip, on, ft.dist, weighted, nthreads=16, pair_begin=a, pair_end=b)
                g = got[a:b]
                if prec == "exact64" or not weighted:
                    ok &= bool(np.array_equal(g, want, equal_nan=True))
                else:
                    rel = np.abs(g - want) / np.where(want == 0, 1, np.abs(want))
                    ok &= bool(np.array_equal(np.isnan(g), np.isnan(want))) and (rel[~np.isnan(rel)].size == 0 or np.nanmax(rel) <= 1e-6)
            if not ok: bad += 1
            print(n, leaves, dens, "w" if weighted else "u", prec, "ok" if ok else "MISMATCH", "%.2fs" % (time.time() - t0), flush=True)
print("bad", bad)
