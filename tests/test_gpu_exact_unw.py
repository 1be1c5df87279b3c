"""pair_exact_unw_kernel: unweighted UniFrac with the reference's bits for ANY branch lengths
(unifracDistUnweighted, frcfrc/unifrac.go:144-171), and the policy that makes it the default when the
lengths are off the binary grid.  Everything here is `np.array_equal` against the oracle: the bar for
unweighted is bit-exact (BASELINE.json north_star).  Needs an MI355X: `pytest -m gpu`."""
import os
import subprocess

import numpy as np
import pytest

import frackyfrac_amd as ff
from frackyfrac_amd import _lib as L
from frackyfrac_amd import synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu

HOST_THREADS = max(1, min(16, len(os.sched_getaffinity(0))))  # oracle threads (the checker only)
FIXED32, EXACT64 = 1, 2
K_EXACT, K_MFMA, K_MFMA_SMALL, K_EXACT_UNW = 1, 2, 4, 5


def same_bits(got, want):
    """NaN where the reference has NaN (0/0: two empty samples, unifrac.go:169), the same bits elsewhere."""
    return np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[~np.isnan(want)], want[~np.isnan(want)])


def problem(ns, nl, dens, seed, lengths="lognormal", sigma=1.5):
    """A synthetic table on a tree whose branch lengths are NOT short binary fractions."""
    tree, ptr, idx, val = synth.make(ns, nl, dens, seed)
    rng = np.random.default_rng(seed + 1)
    if lengths == "lognormal":
        bl = rng.lognormal(-3.0, sigma, len(tree.branch_len))
    elif lengths == "decimal":      # what a Newick file holds: a few decimal digits
        bl = np.round(rng.uniform(0.0, 2.0, len(tree.branch_len)), 5)
    else:
        bl = np.asarray(tree.branch_len, dtype=np.float64).copy()
    bl[0] = 0.0
    tree.branch_len = bl
    nodes = ff.flatten_leaf_csr(ff.parse_newick(tree.newick()), ptr, idx, val)
    ft = O.FlatTree(tree.names, tree.branch_len, tree.size, tree.parent)
    ip, on = O.flatten_samples(ft, ptr, idx, val, 0)
    return nodes, ip, on, ft


def test_auto_takes_the_exact_kernel_for_lengths_off_the_binary_grid_and_only_then():
    """FF_PRECISION_AUTO, a problem past the 2^32 pair-branches under which everything is EXACT64 anyway:
    unweighted + log-normal or decimal lengths -> EXACT64 on pair_exact_unw_kernel (bit-exact is the bar for
    unweighted); unweighted + the generator's dyadic lengths -> FIXED32 on the matrix cores (bit-exact too, 30x
    faster); weighted -> FIXED32 whatever the lengths (its bar is 1e-6); an explicit fixed32 is honoured."""
    n, nl = 1500, 3000
    for lengths, want_prec, want_kernel in (("lognormal", EXACT64, K_EXACT_UNW), ("decimal", EXACT64, K_EXACT_UNW),
                                            ("dyadic", FIXED32, K_MFMA)):
        nodes, ip, on, ft = problem(n, nl, 0.1, 7, lengths)
        assert ff.num_pairs(n) * nodes.n_branches > 2 ** 32
        plan = ff.Plan(nodes, False, precision="auto")
        info = plan.info
        assert info.precision == want_prec and info.kernel in (want_kernel, K_MFMA_SMALL if want_kernel == K_MFMA else -1), lengths
        got = plan.run_host()
        plan.close()
        a = ff.num_pairs(n) // 2
        want = O.unifrac_dists(ip, on, ft.dist, False, nthreads=HOST_THREADS, pair_begin=a, pair_end=a + 50_000)
        assert np.array_equal(got[a:a + 50_000], want), lengths
        pw = ff.Plan(nodes, True, precision="auto")
        assert pw.info.precision == FIXED32
        pw.close()
        pf = ff.Plan(nodes, False, precision="fixed32")
        assert pf.info.precision == FIXED32 and pf.info.kernel in (K_MFMA, K_MFMA_SMALL)
        pf.close()


@pytest.mark.parametrize("name", ["C3", "C5"])
def test_unweighted_full_size_lognormal_lengths_bit_exact(name):
    """BASELINE configs at stated size, unweighted, branch lengths as a real phylogeny has them: AUTO resolves to
    EXACT64 on pair_exact_unw_kernel and 400,000 pairs in four ranges spread over the triangle -- the first rows,
    two from the middle, the last rows next to the diagonal -- are the oracle's bits."""
    cfg = synth.CONFIGS[name]
    n = cfg["n_samples"]
    nodes, ip, on, ft = problem(n, cfg["n_leaves"], cfg["density"], cfg["seed"])
    plan = ff.Plan(nodes, False, precision="auto")
    info = plan.info
    assert (info.precision, info.kernel) == (EXACT64, K_EXACT_UNW)
    got = plan.run_host()
    assert plan.audit() == (0, 0, 0.0)          # nothing to audit: no rounding was staged
    plan.close()
    assert not np.isnan(got).any() and got.min() >= 0.0 and got.max() <= 1.0
    P = ff.num_pairs(n)
    for a in (0, P // 3, 2 * P // 3, P - 100_000):
        want = O.unifrac_dists(ip, on, ft.dist, False, nthreads=HOST_THREADS, pair_begin=a, pair_end=a + 100_000)
        assert np.array_equal(got[a:a + 100_000], want), (name, a)


@pytest.mark.parametrize("ns,nl,dens", [(2, 3, 1.0), (9, 40, 0.3), (65, 33, 0.5), (130, 700, 0.05), (200, 64, 0.9),
                                         (513, 1000, 0.1)])
def test_every_pair_every_variant(monkeypatch, ns, nl, dens):
    """Small shapes, every pair, through every way the engine has of computing them exactly: the unweighted
    kernel with tiles of one and of two column groups, the weighted kernel's arithmetic (FF_EXACT_UNW=0: l * |x - y|
    and l * (x * y) with x, y in {0, 1}), and the oracle.  Rows that are not a multiple of 8 or 32, samples that
    are not a multiple of 64, a single pair."""
    nodes, ip, on, ft = problem(ns, nl, dens, 100 + ns)
    want = O.unifrac_dists(ip, on, ft.dist, False, nthreads=HOST_THREADS)
    for jmax in ("1", "2"):
        monkeypatch.setenv("FF_XU_JMAX", jmax)
        plan = ff.Plan(nodes, False, precision="exact64")
        assert plan.info.kernel == K_EXACT_UNW
        assert same_bits(plan.run_host(), want), jmax
        plan.close()
    monkeypatch.delenv("FF_XU_JMAX")
    monkeypatch.setenv("FF_EXACT_UNW", "0")
    plan = ff.Plan(nodes, False, precision="exact64")
    assert plan.info.kernel == K_EXACT
    assert same_bits(plan.run_host(), want)
    plan.close()


def test_shards_tile_the_pair_space_with_the_same_bits():
    """Row shards (the ranks of a multi-GPU run, the passes of the CLI): each writes its own contiguous slots and
    together they are the one-shard result; a re-targeted plan (ff_plan_set_shard) gives the same."""
    nodes, ip, on, ft = problem(700, 900, 0.12, 5)
    want = O.unifrac_dists(ip, on, ft.dist, False, nthreads=HOST_THREADS)
    for world in (2, 5):
        parts = []
        plan = ff.Plan(nodes, False, precision="exact64", rank=0, world=world)
        for rank in range(world):
            if rank:
                plan.set_shard(rank, world)
            assert plan.info.kernel == K_EXACT_UNW
            parts.append(plan.run_host())
        plan.close()
        assert np.array_equal(np.concatenate(parts), want), world


def test_samples_without_flat_nodes_and_replicates():
    """Both samples empty: 0/0 = NaN; one empty: 1 exactly; identical samples: 0 exactly (unifrac.go:169)."""
    T = ff.parse_newick("((a:0.1,b:0.2):0.3,(c:0.7,d:1e-3):0.05);")
    names = {nm: k for k, nm in enumerate(T.names)}
    rows = [["a", "c"], [], ["a", "c"], ["b"], [], ["a", "b", "c", "d"]]
    ptr = np.cumsum([0] + [len(r) for r in rows]).astype(np.int64)
    idx = np.array([names[x] for r in rows for x in r], dtype=np.int64)
    val = np.ones(len(idx))
    nodes = ff.flatten_leaf_csr(T, ptr, idx, val)
    got = ff.unifrac_dists(nodes, False, precision="exact64")
    k = lambda i, j: i * (i - 1) // 2 + j
    assert np.isnan(got[k(4, 1)]) and got[k(1, 0)] == 1.0 and got[k(2, 0)] == 0.0 and got[k(4, 3)] == 1.0
    ft = O.FlatTree(T.names, T.branch_len, T.subtree_size, T.parent)
    ip, on = O.flatten_samples(ft, ptr, idx, val, 0)
    assert same_bits(got, O.unifrac_dists(ip, on, ft.dist, False))


@pytest.mark.parametrize("kind", ["negative", "inf", "nan", "huge_and_tiny", "minus_zero"])
def test_lengths_no_arithmetic_may_touch(kind):
    """The kernel forms its operands by masking the length's BITS, never by multiplying with 0 or 1: a negative, an
    infinite, a NaN, a subnormal length arrives in the sums exactly where the reference adds it and nowhere else
    (l * 0 would turn an infinite length of a branch NEITHER sample has into NaN for every pair)."""
    tree, ptr, idx, val = synth.make(150, 120, 0.2, 77)
    rng = np.random.default_rng(3)
    bl = rng.lognormal(-2.0, 1.0, len(tree.branch_len))
    bl[0] = 0.0
    pick = rng.choice(np.arange(1, len(bl)), 6, replace=False)
    if kind == "negative":
        bl[pick] = -bl[pick]
    elif kind == "inf":
        bl[pick[:2]] = np.inf
    elif kind == "nan":
        bl[pick[:2]] = np.nan
    elif kind == "huge_and_tiny":
        bl[pick[:3]] = [1e300, 5e-324, 1e-310]
    else:
        bl[pick] = -0.0
    T = ff.parse_newick(tree.newick())
    nodes0 = ff.flatten_leaf_csr(T, ptr, idx, val)
    nodes = ff.FlatNodes(nodes0.indptr, nodes0.branch_id, nodes0.abnd, bl)
    ft = O.FlatTree(tree.names, bl, tree.size, tree.parent)
    ip, on = O.flatten_samples(ft, ptr, idx, val, 0)
    want = O.unifrac_dists(ip, on, bl, False, nthreads=HOST_THREADS)
    plan = ff.Plan(nodes, False, precision="auto")
    assert plan.info.kernel == K_EXACT_UNW
    got = plan.run_host()
    plan.close()
    assert same_bits(got, want)
    if kind in ("inf", "nan"):
        assert np.isnan(want).any() and not np.isnan(want).all()   # (the case bites, and not everywhere)


@pytest.mark.parametrize("precision", ["auto", "exact64"])
@pytest.mark.parametrize("kind", ["negative", "inf", "nan", "huge_and_tiny", "minus_zero"])
def test_weighted_lengths_no_arithmetic_may_touch(kind, precision, monkeypatch):
    """The same for WEIGHTED (unifrac.go:178-203 never touches a branch neither sample has): the dense EXACT64 kernels
    would add l * |0 - 0| or l * 0 for it -- NaN for an infinite or NaN length where the reference prints a finite
    distance.  Such a tree goes to the literal walk (plan_build); negative, huge, subnormal and -0 lengths stay on the
    dense kernel, whose operations are the reference's.  Big enough for the skip kernel's tile heights."""
    monkeypatch.setenv("FF_X_TILE_H", "12")   # (a shard this small would take 4-row tiles of pair_exact64_kernel)
    tree, ptr, idx, val = synth.make(700, 300, 0.15, 78)
    rng = np.random.default_rng(4)
    bl = rng.lognormal(-2.0, 1.0, len(tree.branch_len))
    bl[0] = 0.0
    pick = rng.choice(np.arange(1, len(bl)), 6, replace=False)
    if kind == "negative":
        bl[pick] = -bl[pick]
    elif kind == "inf":
        bl[pick[:2]] = [np.inf, -np.inf]
    elif kind == "nan":
        bl[pick[:2]] = np.nan
    elif kind == "huge_and_tiny":
        bl[pick[:3]] = [1e300, 5e-324, 1e-310]
    else:
        bl[pick] = -0.0
    T = ff.parse_newick(tree.newick())
    nodes0 = ff.flatten_leaf_csr(T, ptr, idx, val)
    nodes = ff.FlatNodes(nodes0.indptr, nodes0.branch_id, nodes0.abnd, bl)
    ft = O.FlatTree(tree.names, bl, tree.size, tree.parent)
    ip, on = O.flatten_samples(ft, ptr, idx, val, 0)
    want = O.unifrac_dists(ip, on, bl, True, nthreads=HOST_THREADS)
    plan = ff.Plan(nodes, True, precision=precision)
    assert plan.info.precision == L.PRECISION_EXACT64
    assert L.KERNEL_NAMES[plan.info.kernel] == ("pair_walk_kernel" if kind in ("inf", "nan") else "pair_exact64_skip_kernel")
    got = plan.run_host()
    plan.close()
    assert same_bits(got, want)
    if kind in ("inf", "nan"):
        assert np.isnan(want).any() and np.isfinite(want).any()   # (the case bites, and not everywhere)
        # sharded, and through the one-call entry point
        parts = []
        for r in range(3):
            p = ff.Plan(nodes, True, precision=precision, rank=r, world=3)
            parts.append(p.run_host())
            p.close()
        assert same_bits(np.concatenate(parts), want)
        assert same_bits(ff.unifrac_dists(nodes, True, precision=precision), want)
        with pytest.raises(L.FFError):
            ff.Plan(nodes, True, precision="fixed32")


@pytest.mark.parametrize("n", [2055, 4097])
def test_all_zero_presence_matrix_and_row_counts_the_tile_height_does_not_divide(n):
    """The two shapes round 4's microbench faulted next to: a presence matrix without a single bit (every
    sample empty: all distances 0 / 0; then half of them empty), and a sample count the tile height does not divide
    (the last tile's rows past the end must neither be written nor counted)."""
    tree, ptr, idx, val = synth.make(n, 400, 0.1, 5)
    T = ff.parse_newick(tree.newick())
    ft = O.FlatTree(tree.names, tree.branch_len, tree.size, tree.parent)
    empty_ptr = np.zeros(n + 1, dtype=np.int64)
    nodes = ff.flatten_leaf_csr(T, empty_ptr, idx[:0], val[:0])
    plan = ff.Plan(nodes, False, precision="exact64")
    assert plan.info.kernel == K_EXACT_UNW
    got = plan.run_host()
    plan.close()
    assert got.shape == (n * (n - 1) // 2,) and np.isnan(got).all()
    # every other sample empty
    keep = np.arange(n) % 2 == 0
    cnt = np.where(keep, np.diff(ptr), 0)
    p2 = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    sel = np.concatenate([np.arange(ptr[s], ptr[s + 1]) for s in range(n) if keep[s]])
    nodes = ff.flatten_leaf_csr(T, p2, idx[sel], val[sel])
    ip, on = O.flatten_samples(ft, p2, idx[sel], val[sel], 0)
    want = O.unifrac_dists(ip, on, ft.dist, False, nthreads=HOST_THREADS)
    for world in (1, 3):
        parts = []
        for r in range(world):
            plan = ff.Plan(nodes, False, precision="exact64", rank=r, world=world)
            assert plan.info.kernel == K_EXACT_UNW
            parts.append(plan.run_host())
            plan.close()
        assert same_bits(np.concatenate(parts), want)


@pytest.mark.parametrize("weighted", [False, True])
def test_a_problem_without_branches(weighted):
    """n_branches == 0 (validate_problem accepts it): every distance is 0 / 0; no kernel may read a branch.  Large
    enough for the weighted skip kernel's tile heights, which read one branch ahead."""
    n = 2100
    nodes = ff.FlatNodes(np.zeros(n + 1, dtype=np.int64), np.zeros(0, dtype=np.int32), np.zeros(0), np.zeros(0))
    for precision in ("exact64", "auto"):
        got = ff.unifrac_dists(nodes, weighted, precision=precision)
        assert got.shape == (n * (n - 1) // 2,) and np.isnan(got).all()


def test_cli_prints_the_reference_bits_for_decimal_lengths(tmp_path):
    """The frcfrc command on a table past the small-problem threshold, unweighted, a tree with decimal branch
    lengths: no "-precision" given, the output is the oracle's text byte for byte and -stats says bit_exact."""
    n, nl = 1200, 3000
    tree, ptr, idx, val = synth.make(n, nl, 0.1, 31)
    bl = np.round(np.random.default_rng(8).uniform(0.001, 1.5, len(tree.branch_len)), 4)
    bl[0] = 0.0
    tree.branch_len = bl
    (tmp_path / "t.tree").write_text(tree.newick())
    lines = []
    for s in range(n):
        lines.append(" ".join("%s:%d" % (tree.names[idx[k]], int(val[k])) for k in range(ptr[s], ptr[s + 1])))
    (tmp_path / "t.sparse").write_text("\n".join(lines) + "\n")
    out = tmp_path / "out.txt"
    r = subprocess.run([L.FRCFRC_PATH, "-s", "-stats", "-i", str(tmp_path / "t.sparse"), "-t", str(tmp_path / "t.tree"),
                        "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert '"precision": "exact64"' in r.stderr and '"bit_exact": true' in r.stderr and "within 1e-6" not in r.stderr
    ft = O.FlatTree(tree.names, bl, tree.size, tree.parent)
    ip, on = O.flatten_samples(ft, ptr, idx, val, 0)
    want = O.unifrac_dists(ip, on, ft.dist, False, nthreads=HOST_THREADS)
    assert out.read_text() == O.format_output(want)
