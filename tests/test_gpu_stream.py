"""The lazy, ordered, stoppable form of unifracDists (frcfrc/unifrac.go:209-228) through the C ABI:
ff_unifrac_dists_stream / _stream_csr, and the flat-argument twins ff_plan_create_csr / ff_unifrac_dists_csr
a cgo host uses (include/frackyfrac_amd.h).  Needs an MI355X."""
import ctypes

import numpy as np
import pytest

import frackyfrac_amd as ff
from frackyfrac_amd import _lib as L
from frackyfrac_amd import api, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def problem(ns, nl, dens, seed):
    tree, ptr, idx, val = synth.make(ns, nl, dens, seed)
    T = ff.parse_newick(tree.newick())
    nodes = ff.flatten_leaf_csr(T, ptr, idx, val)
    ft = O.FlatTree(tree.names, tree.branch_len, tree.size, tree.parent)
    ip, on = O.flatten_samples(ft, ptr, idx, val, 0)
    return nodes, ip, on, ft


def collect(gen, lo):
    """Pieces must come in ascending, gap-free slot order starting at lo."""
    parts, at, sizes = [], lo, []
    for slot0, d in gen:
        assert slot0 == at
        at += len(d)
        parts.append(d)
        sizes.append(len(d))
    return (np.concatenate(parts) if parts else np.zeros(0)), sizes


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("precision", ["exact64", "fixed32"])
@pytest.mark.parametrize("flat_args", [False, True])
def test_stream_delivers_every_slot_once_in_order(weighted, precision, flat_args):
    nodes, ip, on, ft = problem(333, 900, 0.1, 41)
    want = O.unifrac_dists(ip, on, ft.dist, weighted)
    whole = ff.unifrac_dists(nodes, weighted, precision=precision)
    for chunk in (0, 20000, 777):
        got, sizes = collect(api.unifrac_dists_stream(nodes, weighted, precision=precision, max_pairs_per_chunk=chunk,
                                                      flat_args=flat_args), 0)
        assert got.shape == want.shape
        assert chunk == 0 or max(sizes) <= chunk
        assert np.array_equal(got, whole)  # the same bits as the one-pass entry, however the space is cut
        if precision == "exact64" or not weighted:
            assert np.array_equal(got, want)
        else:
            assert np.max(np.abs(got - want) / want) <= 1e-6


def test_stream_of_one_shard_of_many():
    nodes, ip, on, ft = problem(400, 500, 0.2, 5)
    want = O.unifrac_dists(ip, on, ft.dist, True)
    covered = 0
    for rank in range(3):
        lo, hi = api.shard_slots(400, rank, 3)
        got, _ = collect(api.unifrac_dists_stream(nodes, True, precision="exact64", rank=rank, world=3,
                                                  max_pairs_per_chunk=5000), lo)
        assert np.array_equal(got, want[lo:hi])
        covered += hi - lo
    assert covered == len(want)


def test_stream_stops_when_the_consumer_stops():
    """unifrac.go:221-226.  After the consumer leaves, no further piece is computed or delivered."""
    nodes, ip, on, ft = problem(600, 300, 0.2, 6)
    want = O.unifrac_dists(ip, on, ft.dist, False)
    gen = api.unifrac_dists_stream(nodes, False, precision="exact64", max_pairs_per_chunk=10000)
    slot0, first = next(gen)
    assert slot0 == 0 and 0 < len(first) <= 10000
    assert np.array_equal(first, want[:len(first)])
    gen.close()  # the callback returns 0: ff_unifrac_dists_stream returns FF_OK without the other 17 pieces

    calls = []

    def on_piece(_user, slot_begin, dists, n):
        calls.append((slot_begin, n))
        return 1 if len(calls) < 3 else 0

    p, o, err = nodes.problem(), api._opts(False, "exact64"), L.errbuf()
    rc = L.lib().ff_unifrac_dists_stream(ctypes.byref(p), ctypes.byref(o), 10000, L.DISTS_FN(on_piece), None, err, L.ERRLEN)
    assert rc == 0 and len(calls) == 3
    assert [c[0] for c in calls] == [0, calls[0][1], calls[0][1] + calls[1][1]]


def test_stream_is_lazy_and_validates():
    nodes, *_ = problem(8, 20, 0.5, 7)
    gen = api.unifrac_dists_stream(nodes, True)  # nothing happens until it is ranged over
    del gen
    p, o, err = nodes.problem(), api._opts(True), L.errbuf()
    assert L.lib().ff_unifrac_dists_stream(ctypes.byref(p), ctypes.byref(o), 0, L.DISTS_FN(), None, err, L.ERRLEN) == L.FF_ERR_ARG
    o.rank, o.world = 3, 2
    cb = L.DISTS_FN(lambda *a: 1)
    assert L.lib().ff_unifrac_dists_stream(ctypes.byref(p), ctypes.byref(o), 0, cb, None, err, L.ERRLEN) == L.FF_ERR_ARG
    # one sample: no pairs, no call, no error
    one, *_ = problem(1, 20, 0.5, 8)
    assert list(api.unifrac_dists_stream(one, True)) == []


def test_stream_repeats_a_failing_sub_shard_in_exact64():
    """A table of replicates overflows FIXED32's refinement queue (FF_ERR_PRECISION conditions): the stream delivers
    binary64 results from that sub-shard on, transparently."""
    tree, ptr, idx, val = synth.make(1500, 400, 0.2, 9)
    k = int(ptr[1])
    ptr2 = np.arange(1501, dtype=np.int64) * k  # 1,500 copies of sample 0: every pair is a replicate pair
    idx2, val2 = np.tile(idx[:k], 1500), np.tile(val[:k], 1500)
    nodes = ff.flatten_leaf_csr(ff.parse_newick(tree.newick()), ptr2, idx2, val2)
    got, _ = collect(api.unifrac_dists_stream(nodes, True, precision="fixed32", max_pairs_per_chunk=400000), 0)
    assert got.shape == (1500 * 1499 // 2,) and np.all(got == 0)


@pytest.mark.parametrize("weighted", [False, True])
def test_flat_argument_entry_points_equal_the_struct_ones(weighted):
    nodes, ip, on, ft = problem(200, 300, 0.2, 10)
    want = ff.unifrac_dists(nodes, weighted, precision="fixed32")
    o, err = api._opts(weighted, "fixed32"), L.errbuf()
    out = np.full(len(want), np.nan)
    L.check(L.lib().ff_unifrac_dists_csr(nodes.n_samples, nodes.n_branches, nodes.branch_len.ctypes.data,
                                         nodes.indptr.ctypes.data, nodes.branch_id.ctypes.data, nodes.abnd.ctypes.data,
                                         ctypes.byref(o), out.ctypes.data, err, L.ERRLEN), err)
    assert np.array_equal(out, want)
    h = ctypes.c_void_p()
    L.check(L.lib().ff_plan_create_csr(nodes.n_samples, nodes.n_branches, nodes.branch_len.ctypes.data,
                                       nodes.indptr.ctypes.data, nodes.branch_id.ctypes.data, nodes.abnd.ctypes.data,
                                       ctypes.byref(o), ctypes.byref(h), err, L.ERRLEN), err)
    plan = ff.Plan(None, weighted, _handle=h)
    assert np.array_equal(plan.run_host(), want)
    plan.close()


# ---- the same sequence as TEXT: unifracDists + the loop that prints it (frcfrc.go:58-62), ff_unifrac_text_stream ----

def text_of(nodes, weighted, **kw):
    pieces = []
    n = api.unifrac_text_stream(nodes, weighted, pieces.append, **kw)
    assert n == sum(len(p) for p in pieces) and all(p.endswith(b"\n") for p in pieces)
    return b"".join(pieces).decode(), pieces


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("precision", ["exact64", "fixed32"])
@pytest.mark.parametrize("flat_args", [False, True])
def test_text_stream_is_what_the_printing_loop_writes(weighted, precision, flat_args):
    nodes, ip, on, ft = problem(333, 900, 0.1, 41)
    whole = ff.unifrac_dists(nodes, weighted, precision=precision)
    for chunk in (0, 20000, 777):
        text, _ = text_of(nodes, weighted, precision=precision, max_pairs_per_chunk=chunk, flat_args=flat_args)
        assert text == O.format_output(whole)      # the one-pass entry's doubles, printed as fmt.Fprintln prints them
    if precision == "exact64" or not weighted:
        assert text == O.format_output(O.unifrac_dists(ip, on, ft.dist, weighted))


def test_text_stream_of_one_shard_of_many_and_in_several_pieces_per_sub_shard():
    """3,000 samples: 4.5 M lines, 87 MB of text -- sub-shards larger than a 32-MB slot, so a sub-shard reaches the
    callback in several pieces; shards of a 3-way split tile the whole text."""
    nodes, ip, on, ft = problem(3000, 400, 0.1, 8)
    whole = O.format_output(ff.unifrac_dists(nodes, True, precision="exact64"))
    text, pieces = text_of(nodes, True, precision="exact64")
    assert text == whole and len(pieces) >= 3 and max(len(p) for p in pieces) <= 32 << 20
    parts = [text_of(nodes, True, precision="exact64", rank=r, world=3, max_pairs_per_chunk=1 << 20)[0] for r in range(3)]
    assert "".join(parts) == whole


def test_text_stream_stops_when_the_writer_stops_and_is_lazy():
    nodes, ip, on, ft = problem(700, 300, 0.2, 9)
    whole = O.format_output(ff.unifrac_dists(nodes, False, precision="exact64"))
    got = []

    def writer(b):
        got.append(b)
        return len(got) < 3       # (w.Write fails on the third piece: frcfrc.go:60 `break`)

    n = api.unifrac_text_stream(nodes, False, writer, precision="exact64", max_pairs_per_chunk=10000)
    assert len(got) == 3 and n == sum(len(b) for b in got)
    assert whole.startswith(b"".join(got).decode()) and len(b"".join(got)) < len(whole)
    # a writer that raises: the exception reaches the caller, nothing further is computed
    def bad(_b):
        raise OSError("disk full")
    with pytest.raises(OSError, match="disk full"):
        api.unifrac_text_stream(nodes, False, bad, precision="exact64", max_pairs_per_chunk=10000)
    # nothing to deliver: no callback, no error; a bad problem is an error before anything is staged
    one = ff.FlatNodes(np.array([0, 1], dtype=np.int64), np.array([0], dtype=np.int32), np.array([1.0]), np.array([0.5]))
    assert api.unifrac_text_stream(one, True, got.append) == 0
    broken = ff.FlatNodes(nodes.indptr, nodes.branch_id[::-1].copy(), nodes.abnd, nodes.branch_len)
    with pytest.raises(ff.FFError):
        api.unifrac_text_stream(broken, True, got.append)


def test_text_stream_repeats_a_failing_sub_shard_in_exact64():
    """Replicated samples overflow FIXED32's refinement queue: the sub-shard in which that happens and all later ones
    are computed in binary64, as ff_unifrac_dists_stream does."""
    tree, ptr, idx, val = synth.make(1500, 300, 0.2, 77)
    k = int(ptr[1])
    ptr2 = np.concatenate([[0], np.cumsum([k] * 1500)]).astype(np.int64)       # 1,500 copies of one sample
    idx2, val2 = np.tile(idx[:k], 1500), np.tile(val[:k], 1500)
    T = ff.parse_newick(tree.newick())
    nodes = ff.flatten_leaf_csr(T, ptr2, idx2, val2)
    text, _ = text_of(nodes, True, precision="fixed32", max_pairs_per_chunk=300000)
    assert text == "0\n" * (1500 * 1499 // 2)
