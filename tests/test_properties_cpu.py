"""Property tests (hypothesis) of the product's host code against the oracle."""
import math

import numpy as np
from hypothesis import given, settings, strategies as st

import frackyfrac_amd as ff
from oracle import oracle as O


@settings(max_examples=400, deadline=None)
@given(st.floats(allow_nan=True, allow_infinity=True))
def test_format_float_any_double(x):
    assert ff.format_float(x) == O.format_go_float(x)
    if math.isfinite(x):
        assert float(ff.format_float(x)) == x  # shortest digits still round-trip


names = st.text(alphabet="abcXYZ019_.-", min_size=1, max_size=5)
lengths = st.one_of(st.none(), st.integers(0, 9), st.floats(0, 10, allow_nan=False).map(lambda f: round(f, 3)))


def tree_texts():
    leaf = st.tuples(names, lengths).map(lambda t: t[0] + ("" if t[1] is None else ":%s" % t[1]))

    def internal(children):
        return st.tuples(st.lists(children, min_size=1, max_size=4), st.one_of(st.just(""), names), lengths).map(
            lambda t: "(" + ",".join(t[0]) + ")" + t[1] + ("" if t[2] is None else ":%s" % t[2]))

    return st.recursive(leaf, internal, max_leaves=25).map(lambda s: s + ";")


@settings(max_examples=200, deadline=None)
@given(tree_texts())
def test_newick_matches_oracle(text):
    t = ff.parse_newick(text)
    ft = O.flatten_tree(O.parse_newick(text))
    assert t.names == ft.names
    assert np.array_equal(t.branch_len, ft.dist)
    assert np.array_equal(t.parent, ft.parent)
    assert np.array_equal(t.subtree_size, ft.size)


tokens = st.text(alphabet="ab:10.5e-+xnNiI ", min_size=0, max_size=8)
lines = st.lists(tokens, min_size=0, max_size=5).map(" ".join)
tables = st.lists(lines, min_size=0, max_size=6).map(lambda ls: "\n".join(ls) + "\n")


def _both(fn_product, fn_oracle, text, nt):
    try:
        want = fn_oracle(text)
        werr = None
    except O.OracleError as e:
        want, werr = None, str(e)
    try:
        got = fn_product(text, nt).to_maps()
        gerr = None
    except ff.FFError as e:
        got, gerr = None, str(e)
    if werr is not None or gerr is not None:
        # range errors are the one message the Python oracle cannot reproduce exactly
        if gerr and "out of range" in gerr:
            return
        assert gerr == werr, (text, gerr, werr)
    else:
        assert got == want, text


@settings(max_examples=300, deadline=None)
@given(tables, st.integers(1, 4))
def test_sparse_loader_matches_oracle(text, nt):
    _both(ff.parse_sparse_abundance, O.parse_sparse_abundance, text, nt)


@settings(max_examples=300, deadline=None)
@given(tables, st.integers(1, 4))
def test_dense_loader_matches_oracle(text, nt):
    _both(ff.parse_abundance, O.parse_abundance, text, nt)


@settings(max_examples=60, deadline=None)
@given(st.integers(2, 40), st.integers(2, 60), st.floats(0.05, 0.9), st.integers(0, 2 ** 31))
def test_stage_a_host_matches_oracle(ns, nl, dens, seed):
    from frackyfrac_amd import synth
    tree, ptr, idx, val = synth.make(ns, nl, dens, seed)
    T = ff.parse_newick(tree.newick())
    got = ff.flatten_leaf_csr(T, ptr, idx, val)
    ft = O.FlatTree(tree.names, tree.branch_len, tree.size, tree.parent)
    ip, nodes = O.flatten_samples(ft, ptr, idx, val, 0)
    assert np.array_equal(got.indptr, ip) and np.array_equal(got.branch_id, nodes["id"])
    assert np.array_equal(got.abnd, nodes["abnd"])


def test_synthetic_generator_follows_its_recipe():
    """splitmix64 -> xoshiro256**.  The published reference vectors of both generators,
    the numpy statement of the recipe against the library's C twin (every sample, any range of
    samples on its own), and the properties the recipe promises."""
    from frackyfrac_amd import synth

    assert synth._splitmix64(0)[1] == 0xE220A8397B1DCDAF
    g = synth.Xoshiro(0)
    g.s = [1, 2, 3, 4]
    assert [g.next() for _ in range(4)] == [11520, 0, 1509978240, 1215971899390074240]
    vec = synth.XoshiroVec(99, np.arange(1, 8, dtype=np.uint64))
    draws = [vec.next() for _ in range(12)]
    for s in range(7):
        ref = synth.Xoshiro(99, s + 1)
        assert [int(d[s]) for d in draws] == [ref.next() for _ in range(12)]
    for ns, nl, dens, seed in ((41, 70, 0.2, 5), (6, 3, 0.01, 9), (300, 150, 0.1, synth.SEED_BASE + 2)):
        tree = synth.yule_tree(nl, seed)
        assert tree.n == 2 * nl - 1 and tree.branch_len[0] == 0 and np.all(tree.branch_len[1:] * 1024 % 1 == 0)
        a = synth.abundances(tree, ns, dens, seed)
        b = synth.abundances_numpy(tree, ns, dens, seed)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
        ptr, idx, val = a
        assert np.all(np.diff(ptr) >= 1) and np.all((val >= 1) & (val <= 1000) & (val % 1 == 0))
        lo, hi = 2, min(ns, 11)
        p2, i2, v2 = synth.abundances(tree, ns, dens, seed, lo, hi)
        assert np.array_equal(p2, ptr[lo:hi + 1] - ptr[lo]) and np.array_equal(i2, idx[ptr[lo]:ptr[hi]])
        assert np.array_equal(v2, val[ptr[lo]:ptr[hi]])
