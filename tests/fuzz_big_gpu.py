"""Randomised parity sweep at sizes where the main rounds of the schedules are in play (3,000 to
9,000 samples): random shapes, densities, shard counts, weighted and unweighted, FIXED32 against
the oracle on 200,000 sampled pairs per case (1e-6 relative / bit-exact for exact-length unweighted); a third of
the cases with log-normal or digit-edge integer branch lengths; unweighted cases with log-normal lengths also under
precision auto (EXACT64 on pair_exact_unw_kernel) and every weighted case also in EXACT64
(pair_exact64_skip_kernel), bit for bit.
Usage: python tests/fuzz_big_gpu.py SEED CASES   (a script, not collected by pytest)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import frackyfrac_amd as ff
from frackyfrac_amd import synth
from oracle import oracle as O

seed0, ncase = int(sys.argv[1]), int(sys.argv[2])
bad = 0
t0 = time.time()
for case in range(ncase):
    rng = np.random.default_rng(seed0 + case)
    n = int(rng.integers(3000, 9000))
    leaves = int(rng.choice([300, 1000, 3000]))
    dens = float(rng.choice([0.02, 0.1, 0.4]))
    weighted = bool(rng.random() < 0.6)
    world = int(rng.choice([1, 1, 2, 3]))
    tree, ptr, idx, val = synth.make(n, leaves, dens, int(rng.integers(1, 1 << 30)))
    # a third of the cases with lengths that are not short binary fractions: log-normal over up to six orders of
    # magnitude (unweighted: graded rows on the matrix cores -- long branches as several rows, three planes and two)
    # or, unweighted only, integers at the edges of the signed digits' ranges (exact: compared bit for bit)
    lengths = str(rng.choice(["dyadic", "dyadic", "lognormal", "edges" if not weighted else "lognormal"]))
    if lengths == "lognormal":
        bl = rng.lognormal(-3.0, float(rng.choice([0.5, 1.5, 2.5, 3.5])), len(tree.branch_len))
        bl[rng.random(len(bl)) < 0.02] = 0.0
        bl[0] = 0.0
        tree.branch_len = bl
    elif lengths == "edges":
        kmax = 63 + 128 * (128 + 256 * 127)
        bl = rng.choice(np.array([1, 63, 64, 65, 16383, 16384, 16447, 16448, 32767, 32768, 32769, 65536, kmax - 1, kmax,
                                  kmax + 1, 3 * kmax + 5], dtype=np.float64), len(tree.branch_len),
                        p=np.array([20, 20, 20, 20, 10, 10, 10, 10, 5, 5, 5, 5, 1, 1, 1, 1]) / 144.0)
        bl[0] = 0.0
        tree.branch_len = bl
    T = ff.parse_newick(tree.newick())
    nodes = ff.flatten_leaf_csr(T, ptr, idx, val)
    P = ff.num_pairs(n)
    got = np.full(P, np.nan)
    try:
        for r in range(world):
            ff.unifrac_dists(nodes, weighted, precision="fixed32", rank=r, world=world, out=got)
    except ff.FFError as e:
        if "FIXED32 not applicable" in str(e):  # (sample weights too far apart for one scale: auto would take EXACT64)
            print("case", case, "n", n, lengths, "FIXED32 not applicable", flush=True)
            continue
        raise
    # unweighted with lengths off the binary grid: what the engine does by itself (auto -> EXACT64 on
    # pair_exact_unw_kernel), compared bit for bit
    # weighted: EXACT64 as well (pair_exact64_skip_kernel at the height the plan picks for each shard), bit for bit
    exact = None
    if weighted or lengths == "lognormal":
        exact = np.full(P, np.nan)
        for r in range(world):
            ff.unifrac_dists(nodes, weighted, precision="exact64" if weighted else "auto", rank=r, world=world, out=exact)
    ft = O.FlatTree(tree.names, tree.branch_len, tree.size, tree.parent)
    ip, on = O.flatten_samples(ft, ptr, idx, val, 0)
    ok = not np.isnan(got).any()
    for q in range(4):
        a = int((P - 50_000) * q // 3)
        want = O.unifrac_dists(ip, on, ft.dist, weighted, nthreads=16, pair_begin=a, pair_end=a + 50_000)
        g = got[a:a + 50_000]
        if exact is not None:
            ok = ok and bool(np.array_equal(exact[a:a + 50_000], want, equal_nan=True))
        if weighted or lengths == "lognormal":
            rel = np.abs(g - want) / np.where(want == 0, 1, np.abs(want))
            ok = ok and bool(np.nanmax(rel) <= 1e-6) and bool(np.array_equal(np.isnan(g), np.isnan(want)))
        else:
            ok = ok and bool(np.array_equal(g, want, equal_nan=True))
    if not ok:
        bad += 1
        print("CASE", seed0 + case, "n", n, "leaves", leaves, "dens", dens, "weighted", weighted, "world", world, "MISMATCH", flush=True)
    print("case", case, "n", n, "leaves", leaves, "w", weighted, lengths, "world", world, "%.0fs" % (time.time() - t0), "bad", bad, flush=True)
print("done", ncase, "bad", bad)
