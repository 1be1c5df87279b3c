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
