"""Host side of the product (C++ behind the C ABI) against the oracle: Newick reader,
table loaders, species validation, stage A, Go float formatting, sharding, output
writer.  No GPU needed."""
import ctypes
import math
import os
import subprocess

import numpy as np
import pytest

import frackyfrac_amd as ff
from conftest import read_golden
from frackyfrac_amd import _lib as L
from frackyfrac_amd import synth
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_tree_arrays(text):
    ft = O.flatten_tree(O.parse_newick(text))
    return ft


@pytest.mark.parametrize("name", ["uwtd1", "uwtd2", "wtd"])
def test_newick_golden_trees(name):
    text = read_golden(name + ".tree")
    t, ft = ff.parse_newick(text), oracle_tree_arrays(text)
    assert t.names == ft.names
    assert np.array_equal(t.branch_len, ft.dist)
    assert np.array_equal(t.parent, ft.parent)
    assert np.array_equal(t.subtree_size, ft.size)


def test_newick_grammar_extras():
    # whitespace, comments, quoted labels, exponents, unnamed leaves, root length, first tree only
    t = ff.parse_newick(" ( 'it''s a':1e-1 , [c] b:2.5 ,(,c:3)x:4 )root:0.5 ; (z:1);")
    assert t.names == ["root", "it's a", "b", "x", "", "c"]
    assert t.branch_len.tolist() == [0.5, 0.1, 2.5, 4.0, 0.0, 3.0]
    assert t.parent.tolist() == [-1, 0, 0, 0, 3, 3]
    assert t.subtree_size.tolist() == [6, 1, 1, 3, 1, 1]
    single = ff.parse_newick("a;")
    assert single.n == 1 and single.names == ["a"]


@pytest.mark.parametrize("text,msg", [
    ("", "no tree in the given file"),
    ("   \n", "no tree in the given file"),
    ("(a,b", "newick: unexpected end of text, expected ';' at offset 4"),
    ("(a,b));", "newick: unbalanced ')' at offset 5"),
    ("(a:x,b);", "newick: bad branch length at offset 4"),
])
def test_newick_errors(text, msg):
    with pytest.raises(ff.FFError) as e:
        ff.parse_newick(text)
    assert str(e.value) == msg


def test_newick_deep_caterpillar_and_synthetic_roundtrip():
    # 60k-deep caterpillar: the parser and stage A must not recurse
    n = 60000
    text = "(" * n + "t0:1" + "".join(",t%d:1):1" % k for k in range(1, n + 1)) + ";"
    t = ff.parse_newick(text)
    assert t.n == 2 * n + 1
    assert t.subtree_size[0] == 2 * n + 1
    tree, ptr, idx, val = synth.make(3, 500, 0.1, 5)
    T = ff.parse_newick(tree.newick())
    assert T.names == tree.names
    assert np.array_equal(T.parent, tree.parent) and np.array_equal(T.subtree_size, tree.size)
    assert np.array_equal(T.branch_len, tree.branch_len)


LOADER_TEXTS_DENSE = [
    "   aa  bbbb    \n1\t2\n 3  \t  4 \t\n",            # parser_test.go:10
    "a b c\n0 0 0\n1 0 2.5\n",
    "a a b\n1 2 3\n4 0 5\n",                             # duplicate header: last non-zero wins
    "x\n1e3\n+2\n0x1p-2\n",
    "a b\r\n1 2\r\n",
    "a\vb c\n1 2\n",                                     # \v is not in RE2's \s: "a\vb" is ONE species name
]
LOADER_TEXTS_SPARSE = [
    "a:11 b:222  \n  b:32 c:7\n\nd:1\tc:4\ta:10\n",     # parser_test.go:29
    "a:1 a:2 b:3\n",                                      # duplicate: last wins
    "c:d:e::5\n",                                         # split at the last colon
    "\n\n",
    "a:1",                                                # no trailing newline
    "a\vb:1 c:2\n",                                       # vertical tab inside a name
    "a:1\vb:2\n",                                         # ... is no separator: one token, name "a:1\vb"
]


@pytest.mark.parametrize("text", LOADER_TEXTS_DENSE)
def test_dense_loader_matches_oracle(text):
    assert ff.parse_abundance(text).to_maps() == O.parse_abundance(text)


@pytest.mark.parametrize("text", LOADER_TEXTS_SPARSE)
def test_sparse_loader_matches_oracle(text):
    assert ff.parse_sparse_abundance(text).to_maps() == O.parse_sparse_abundance(text)


ERR_DENSE = ["\n1 2\n", "a b\n1\n", "a b\n1 x\n", "a b\n1 -2\n", "a b\n1 2\n\n", "a\ninf\n", "a\nnan\n",
             "a\n1e999\n", "a b\n1 2 3\n", "a\n1_0\n", "a\n-\n", "a\n0x10\n",
             "a b\n1\v2 3\n"]  # a vertical tab does not split: "1\v2" is one token and not a number
ERR_SPARSE = ["a:1 b\n", ":1\n", "a:0\n", "a:nan\n", "a:-1\n", "a:\n", "a:1e999\n", "a:+Inf\n", "b:1 a:x:y\n", "a:1\vb:x\n"]


@pytest.mark.parametrize("text", ERR_DENSE)
def test_dense_loader_errors_match_oracle(text):
    with pytest.raises(O.OracleError) as want:
        O.parse_abundance(text)
    with pytest.raises(ff.FFError) as got:
        ff.parse_abundance(text)
    if "out of range" in str(got.value):  # the oracle's float() cannot tell range errors apart
        assert str(got.value) == 'value #1: strconv.ParseFloat: parsing "1e999": value out of range'
    else:
        assert str(got.value) == str(want.value)
    assert got.value.code == 2


@pytest.mark.parametrize("text", ERR_SPARSE)
def test_sparse_loader_errors_match_oracle(text):
    try:
        O.parse_sparse_abundance(text)
        want = None
    except O.OracleError as e:
        want = str(e)
    with pytest.raises(ff.FFError) as got:
        ff.parse_sparse_abundance(text)
    if "out of range" in str(got.value):
        assert str(got.value) == 'value #1: strconv.ParseFloat: parsing "1e999": value out of range'
    else:
        assert want is not None and str(got.value) == want


def test_parallel_loaders_match_serial():
    """ngoroutines > 1 (parser.go:21,85): same table; the first failing row wins."""
    tree, ptr, idx, val = synth.make(97, 300, 0.1, 13)
    for text, parse in ((synth.sparse_text(tree, ptr, idx, val), ff.parse_sparse_abundance),
                        (synth.dense_text(tree, ptr, idx, val), ff.parse_abundance)):
        want = parse(text).to_maps()
        for nt in (2, 5, 16, 200):
            assert parse(text, nt).to_maps() == want
    bad = "a:1\n" * 50 + "b:x\n" + "a:1\n" * 20 + "c\n" + "a:1\n" * 30
    for nt in (1, 3, 8):
        with pytest.raises(ff.FFError) as e:
            ff.parse_sparse_abundance(bad, nt)
        assert str(e.value) == 'value #1: strconv.ParseFloat: parsing "x": invalid syntax'
    assert ff.parse_abundance("", 4).to_maps() == [] and ff.parse_sparse_abundance("", 4).to_maps() == []


def test_validate_species():
    tree = ff.parse_newick("((a:1,b:1)in:1,c:2);")
    ff.validate_species(ff.parse_sparse_abundance("a:1 in:2\n"), tree)
    with pytest.raises(ff.FFError) as e:
        ff.validate_species(ff.parse_sparse_abundance("a:1\nzz:2.5 a:1\n"), tree)
    assert str(e.value) == 'sample #2 has value 2.5 for species "zz" which is not in the tree'
    assert e.value.code == 3
    # "" is a tree name when a node is unnamed (unifrac.go:70-76) -- but the sparse loader rejects it first
    otree = O.parse_newick("((a:1,b:1)in:1,c:2);")
    with pytest.raises(O.OracleError) as oe:
        O.validate_species([{"a": 1.0}, {"zz": 2.5, "a": 1.0}], otree)
    assert str(oe.value) == str(e.value)


def _flatten_both(tree_text, abnd_text, sparse, leave=False):
    t = ff.parse_newick(tree_text)
    tb = (ff.parse_sparse_abundance if sparse else ff.parse_abundance)(abnd_text)
    got = ff.flatten(tb, t, leave_unnormalized=leave)
    otree = O.parse_newick(tree_text)
    oab = (O.parse_sparse_abundance if sparse else O.parse_abundance)(abnd_text)
    ft = O.flatten_tree(otree)
    ptr, idx, val = O.leaf_csr(oab, ft)
    ip, nodes = O.flatten_samples(ft, ptr, idx, val, 2 if leave else 0)
    return got, ip, nodes, ft


@pytest.mark.parametrize("name", ["uwtd1", "uwtd2", "wtd"])
@pytest.mark.parametrize("sparse", [False, True])
def test_stage_a_golden(name, sparse):
    got, ip, nodes, ft = _flatten_both(read_golden(name + ".tree"), read_golden(name + (".sparse" if sparse else ".dense")), sparse)
    assert np.array_equal(got.indptr, ip)
    assert np.array_equal(got.branch_id, nodes["id"])
    assert np.array_equal(got.abnd, nodes["abnd"])          # bit for bit
    assert np.array_equal(got.branch_len, ft.dist)


def test_stage_a_quirks():
    # multifurcations (order of additions), internal-node keys ignored (Q4), duplicate leaf
    # names, empty samples, zero-length and rooted-length branches, -l (sorted, raw)
    tree = "((a:1,b:0,c:3,a:2)in:1.5,(d:1e-3,(e:7,f:0.1):2)x:0,g:5)r:9;"
    table = "a:0.1 b:0.7 c:1e-9 in:5\n\ne:3 f:1e10 g:2.5\nd:1\n"
    for leave in (False, True):
        got, ip, nodes, ft = _flatten_both(tree, table, True, leave)
        assert np.array_equal(got.indptr, ip)
        assert np.array_equal(got.branch_id, nodes["id"])
        assert np.array_equal(got.abnd, nodes["abnd"])


def test_stage_a_as_the_reference_leaves_it_under_l():
    """-l in the reference skips normalizeFlatNodes and with it the SORT (unifrac.go:57-59,108-110): the lists stay
    in the order the recursion appends them, a node after its subtree (unifrac.go:35-52).  leave_unnormalized =
    "reference" (FF_L_REFERENCE) reproduces those lists: ids, raw values and ORDER as the oracle's mode 1; the same
    entries as the sorted -l lists, and not ascending once a sample reaches an internal node."""
    tree = "((a:1,b:0,c:3,a:2)in:1.5,(d:1e-3,(e:7,f:0.1):2)x:0,g:5)r:9;"
    table = "a:0.1 b:0.7 c:1e-9 in:5\n\ne:3 f:1e10 g:2.5\nd:1\n"
    t = ff.parse_newick(tree)
    tb = ff.parse_sparse_abundance(table)
    got = ff.flatten(tb, t, leave_unnormalized="reference")
    ft = O.flatten_tree(O.parse_newick(tree))
    ptr, idx, val = O.leaf_csr(O.parse_sparse_abundance(table), ft)
    ip, nodes = O.flatten_samples(ft, ptr, idx, val, 1)
    assert np.array_equal(got.indptr, ip) and np.array_equal(got.branch_id, nodes["id"]) and np.array_equal(got.abnd, nodes["abnd"])
    srt = ff.flatten(tb, t, leave_unnormalized=True)
    assert np.array_equal(srt.indptr, got.indptr)
    unsorted = 0
    for s in range(len(ip) - 1):
        a, b = ip[s], ip[s + 1]
        order = np.argsort(got.branch_id[a:b], kind="stable")
        assert np.array_equal(got.branch_id[a:b][order], srt.branch_id[a:b]) and np.array_equal(got.abnd[a:b][order], srt.abnd[a:b])
        unsorted += int(np.any(np.diff(got.branch_id[a:b]) < 0))
    assert unsorted >= 2
    # a synthetic table, the leaf-CSR entry point
    tree2, ptr2, idx2, val2 = synth.make(40, 300, 0.1, 17)
    T2 = ff.parse_newick(tree2.newick())
    g2 = ff.flatten_leaf_csr(T2, ptr2, idx2, val2, leave_unnormalized="reference")
    ft2 = O.FlatTree(tree2.names, tree2.branch_len, tree2.size, tree2.parent)
    ip2, n2 = O.flatten_samples(ft2, ptr2, idx2, val2, 1)
    assert np.array_equal(g2.indptr, ip2) and np.array_equal(g2.branch_id, n2["id"]) and np.array_equal(g2.abnd, n2["abnd"])


@pytest.mark.parametrize("seed,ns,nl,dens", [(1, 64, 200, 0.1), (2, 33, 1000, 0.02), (3, 8, 50, 0.9)])
def test_stage_a_synthetic_matches_oracle(seed, ns, nl, dens):
    tree, ptr, idx, val = synth.make(ns, nl, dens, seed)
    T = ff.parse_newick(tree.newick())
    got = ff.flatten_leaf_csr(T, ptr, idx, val)
    ft = O.FlatTree(tree.names, tree.branch_len, tree.size, tree.parent)
    ip, nodes = O.flatten_samples(ft, ptr, idx, val, 0)
    assert np.array_equal(got.indptr, ip)
    assert np.array_equal(got.branch_id, nodes["id"])
    assert np.array_equal(got.abnd, nodes["abnd"])
    # each sample's normalised flat nodes sum to 1 up to rounding (normalizeFlatNodes)
    sums = np.add.reduceat(got.abnd, got.indptr[:-1])
    assert np.allclose(sums, 1.0, rtol=1e-12)
    # text round trip through both table formats gives the same flat nodes
    for text, parse in ((synth.sparse_text(tree, ptr, idx, val), ff.parse_sparse_abundance),
                        (synth.dense_text(tree, ptr, idx, val), ff.parse_abundance)):
        again = ff.flatten(parse(text), T)
        assert np.array_equal(again.indptr, got.indptr) and np.array_equal(again.abnd, got.abnd)


def test_value_tokens_parse_like_strconv():
    """Every spelling of a value, including the loader's fast path for plain decimals of at
    most 15 digits, must give the correctly rounded binary64 (Python's float() is)."""
    rng = np.random.default_rng(21)
    toks = ["1", "007", "0.5", "123456789012345", "1234567890123456", "0.000000000000001", "99999.9999999999",
            "1e3", "1E-3", "1.", ".5", "0x1p-2", "10", "4.9e-324", "1.7976931348623157e308",
            "2.2250738585072011e-308", "9007199254740993", "0.1", "0.30000000000000004", "123.456", "5e-1"]
    for _ in range(3000):
        nd = int(rng.integers(1, 19))
        digits = "".join(str(int(d)) for d in rng.integers(0, 10, nd))
        cut = int(rng.integers(0, nd + 1))
        tok = digits if cut in (0, nd) else digits[:cut] + "." + digits[cut:]
        if float(tok) > 0:
            toks.append(tok)
    val = lambda tok: float.fromhex(tok) if tok.startswith("0x") else float(tok)
    line = " ".join("s%d:%s" % (i, t) for i, t in enumerate(toks))
    for threads in (1, 3):
        t = ff.parse_sparse_abundance(line + "\n", ngoroutines=threads)
        got = t.to_maps()[0]
        for i, tok in enumerate(toks):
            assert got["s%d" % i] == val(tok), tok


def test_format_float_matches_oracle():
    rng = np.random.default_rng(5)
    vals = [0.0, 1.0, 0.5, 1e-4, 0.00001234, 2 / 3, 19 / 28, 16 / 22, 22 / 36, 1e6, 123456.0, 1234567.0, 1e-5,
            5e-324, 1e21, 1e22, 0.1, 0.3, 1 / 3, 9.999999999999999e-05, 0.000123, math.nan, math.inf, -math.inf, -0.0,
            -2.5, 1.7976931348623157e308, 2.2250738585072014e-308]
    vals += list(rng.random(2000))
    vals += list(10.0 ** rng.uniform(-12, 12, 500))
    vals += [float(np.float32(x)) for x in rng.random(200)]
    for v in vals:
        assert ff.format_float(v) == O.format_go_float(v), repr(v)


def test_format_float_matches_oracle_on_random_bit_patterns():
    """ff_format_float is ff_fmt_core.hpp -- the code the device formatter runs."""
    rng = np.random.default_rng(15)
    for v in rng.integers(0, 2 ** 64, size=20000, dtype=np.uint64).view(np.float64):
        assert ff.format_float(float(v)) == O.format_go_float(float(v)), float(v).hex()


def test_formatter_core_against_to_chars():
    """csrc/fmt_selftest.cpp: the formatter core against std::to_chars' shortest digits under Go's layout rule --
    every power of two and its neighbours, short decimals, integers, quotients, 3 M random bit patterns (1.5e9 were
    run once: DESIGN.md)."""
    import subprocess
    exe = os.path.join(ROOT, "frackyfrac_amd", "lib", "fmt_selftest")
    r = subprocess.run([exe, "3000000", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fmt selftest ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_cpu_quota_is_what_the_cgroup_grants():
    """The frcfrc command's default thread count: never more than the affinity mask or the cgroup's quota."""
    import ctypes
    n = L.lib().ff_cpu_quota()
    assert 1 <= n <= len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            assert n <= -(-int(q) // int(p))
    except OSError:
        pass


def test_write_distances(tmp_path):
    rng = np.random.default_rng(9)
    d = rng.random(100003)
    d[5] = np.nan
    d[7] = 1.0
    d[9] = 0.0
    for threads in (1, 4):
        p = tmp_path / ("out%d.txt" % threads)
        ff.write_distances(str(p), d, threads)
        assert p.read_text() == O.format_output(d)
    # README.md:46-56: the lower triangle loads back with numpy.tril_indices
    back = np.array([float(x) for x in open(tmp_path / "out1.txt")])
    assert np.array_equal(back, d, equal_nan=True)


def test_write_distances_many_rounds(tmp_path):
    """More values than one round of the writer holds (2^18 per thread): parts land at
    the right offsets of a regular file, and a ".gz" output is a sequence of gzip members
    that reads back as one stream."""
    import gzip
    import zlib
    rng = np.random.default_rng(10)
    d = rng.random(3 * (1 << 18) * 2 + 12345)
    d[::1000] = 10.0 ** rng.uniform(-9, -3, len(d[::1000]))  # some %e-formatted values
    want = O.format_output(d)
    ff.write_distances(str(tmp_path / "a.txt"), d, 3)
    assert (tmp_path / "a.txt").read_text() == want
    ff.write_distances(str(tmp_path / "a.txt.gz"), d, 3)
    raw = (tmp_path / "a.txt.gz").read_bytes()
    assert gzip.decompress(raw).decode() == want
    z = zlib.decompressobj(31)
    z.decompress(raw)
    assert len(z.unused_data) > 0                       # more than one member
    # overwriting a longer file truncates it
    ff.write_distances(str(tmp_path / "a.txt"), d[:10], 3)
    assert (tmp_path / "a.txt").read_text() == O.format_output(d[:10])


def test_gzip_io_by_suffix(tmp_path):
    """gostuff/aio (frcfrc.go:93,102): a ".gz" suffix means gzip, on input and on output."""
    import gzip
    rng = np.random.default_rng(11)
    d = rng.random(50000)
    p = tmp_path / "out.txt.gz"
    ff.write_distances(str(p), d, 3)
    assert gzip.open(p, "rt").read() == O.format_output(d)
    tree_gz = tmp_path / "t.tree.gz"
    with gzip.open(tree_gz, "wt") as f:
        f.write(read_golden("uwtd2.tree"))
    t = ff.Tree.read_file(str(tree_gz))
    assert t.names == ff.parse_newick(read_golden("uwtd2.tree")).names
    tab_gz = tmp_path / "t.dense.gz"
    with gzip.open(tab_gz, "wt") as f:
        f.write(read_golden("uwtd2.dense"))
    r = subprocess.run([ff.FRCFRC_PATH, "-t", str(tree_gz), "-i", str(tab_gz), "-l"], capture_output=True, text=True)
    assert r.returncode == 2 and "-l can only be used" in r.stderr      # (argument check comes first)
    # a corrupt .gz is an I/O error, not a crash
    bad = tmp_path / "bad.tree.gz"
    bad.write_bytes(b"\x1f\x8b\x08\x00garbage-not-deflate")
    with pytest.raises(ff.FFError) as e:
        ff.Tree.read_file(str(bad))
    assert e.value.code == 5


def _zstd():
    """libzstd through ctypes: the test's own compressor / decompressor (python has no zstd module here)."""
    import ctypes
    z = ctypes.CDLL("libzstd.so.1")
    z.ZSTD_compressBound.restype = ctypes.c_size_t
    z.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
    z.ZSTD_compress.restype = ctypes.c_size_t
    z.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    z.ZSTD_decompress.restype = ctypes.c_size_t
    z.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    z.ZSTD_isError.argtypes = [ctypes.c_size_t]
    z.ZSTD_findFrameCompressedSize.restype = ctypes.c_size_t
    z.ZSTD_findFrameCompressedSize.argtypes = [ctypes.c_char_p, ctypes.c_size_t]

    def compress(data: bytes) -> bytes:
        buf = ctypes.create_string_buffer(z.ZSTD_compressBound(len(data)))
        n = z.ZSTD_compress(buf, len(buf), data, len(data), 3)
        assert not z.ZSTD_isError(n)
        return buf.raw[:n]

    def decompress_frames(data: bytes, cap: int) -> bytes:
        out, at = b"", 0
        while at < len(data):
            fsz = z.ZSTD_findFrameCompressedSize(data[at:], len(data) - at)
            assert not z.ZSTD_isError(fsz)
            buf = ctypes.create_string_buffer(cap)
            n = z.ZSTD_decompress(buf, cap, data[at:at + fsz], fsz)
            assert not z.ZSTD_isError(n)
            out += buf.raw[:n]
            at += fsz
        return out

    return compress, decompress_frames


def test_zstd_and_bzip2_io_by_suffix(tmp_path):
    """The other codecs gostuff/aio picks by suffix (frcfrc.go:93,102; go.mod:7,12): ".zst" on input and output,
    ".bz2" on input (Go has no bzip2 writer).  The libraries are bound with dlopen (the image ships them without
    headers).  Parity unpinned: aio's source is not in the reference tree."""
    import bz2
    compress, decompress_frames = _zstd()
    tree_text, table_text = read_golden("uwtd2.tree"), read_golden("uwtd2.dense")
    names = ff.parse_newick(tree_text).names
    maps = ff.parse_abundance(table_text).to_maps()
    # input: zstd (one frame, and two frames back to back), bzip2 (one stream, and two)
    for suffix, blobs in ((".zst", [compress(tree_text.encode()), compress(tree_text[:9].encode()) + compress(tree_text[9:].encode())]),
                          (".bz2", [bz2.compress(tree_text.encode()), bz2.compress(tree_text[:9].encode()) + bz2.compress(tree_text[9:].encode())])):
        for k, blob in enumerate(blobs):
            p = tmp_path / ("t%d.tree%s" % (k, suffix))
            p.write_bytes(blob)
            assert ff.Tree.read_file(str(p)).names == names
    for suffix, blob in ((".zst", compress(table_text.encode())), (".bz2", bz2.compress(table_text.encode()))):
        p = tmp_path / ("t.dense" + suffix)
        p.write_bytes(blob)
        h, err = ctypes.c_void_p(), L.errbuf()
        L.check(L.lib().ff_table_read_file(str(p).encode(), 0, ctypes.byref(h), err, L.ERRLEN), err)
        assert ff.Table(h).to_maps() == maps
    # a large text: many decoder rounds
    big = ("%.17g\n" * 200000 % tuple(np.random.default_rng(5).random(200000))).encode()
    (tmp_path / "big.zst").write_bytes(compress(big))
    (tmp_path / "big.bz2").write_bytes(bz2.compress(big))
    for name in ("big.zst", "big.bz2"):
        with pytest.raises(ff.FFError) as e:   # (it is not a tree: but the whole text must have been decoded to say so)
            ff.Tree.read_file(str(tmp_path / name))
        assert e.value.code == 2
    # output: ".zst" = one frame per part, decodable as one text; ".bz2" is refused
    d = np.random.default_rng(12).random(50000)
    out = tmp_path / "out.txt.zst"
    ff.write_distances(str(out), d, 3)
    assert decompress_frames(out.read_bytes(), 1 << 22).decode() == O.format_output(d)
    with pytest.raises(ff.FFError) as e:
        ff.write_distances(str(tmp_path / "out.txt.bz2"), d, 1)
    assert e.value.code == 5 and "bzip2 output is not supported" in str(e.value)
    # corrupt input is an I/O error, not a crash
    for name, blob in (("bad.tree.zst", b"\x28\xb5\x2f\xfdgarbage"), ("bad.tree.bz2", b"BZh9garbage-not-bzip2"),
                       ("cut.tree.zst", compress(big)[:1000]), ("cut.tree.bz2", bz2.compress(big)[:1000])):
        (tmp_path / name).write_bytes(blob)
        with pytest.raises(ff.FFError) as e:
            ff.Tree.read_file(str(tmp_path / name))
        assert e.value.code == 5, name


def test_iter_pairs_and_slots():
    assert list(ff.iter_pairs(4)) == list(O.iter_pairs(4)) == [(1, 0), (2, 0), (2, 1), (3, 0), (3, 1), (3, 2)]
    n = 37
    ii, jj = np.tril_indices(n, -1)
    assert [(int(a), int(b)) for a, b in zip(ii, jj)] == list(ff.iter_pairs(n))
    assert ff.num_pairs(n) == n * (n - 1) // 2 and ff.num_pairs(1) == 0 and ff.num_pairs(0) == 0


@pytest.mark.parametrize("n", [0, 1, 2, 31, 32, 33, 100, 4096, 16384, 11585])
@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_shard_rows_partition(n, world):
    """Shards are contiguous, cover every row once, and hold equal pair counts to
    within one 32-row block."""
    prev = 0
    counts = []
    for r in range(world):
        rb, re = ff.shard_rows(n, r, world)
        assert rb == prev and re >= rb
        if r < world - 1:
            assert re % 32 == 0 or re == n
        prev = re
        sb, se = ff.shard_slots(n, r, world)
        assert (sb, se) == (rb * (rb - 1) // 2 if rb else 0, re * (re - 1) // 2 if re else 0)
        counts.append(se - sb)
    assert prev == n and sum(counts) == ff.num_pairs(n)
    if n >= 1024:
        ideal = ff.num_pairs(n) / world
        assert max(counts) <= ideal + 33 * n   # one 32-row block of slack
    with pytest.raises(ff.FFError):
        ff.shard_rows(10, 2, 2)


def test_compute_without_gpu_fails_loudly():
    """No CPU fallback: on a box without a GPU the compute entry points must error."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    tree = ff.parse_newick(read_golden("wtd.tree"))
    table = ff.parse_abundance(read_golden("wtd.dense"))
    with pytest.raises(ff.FFError) as e:
        ff.unifrac(table, tree, True)
    assert e.value.code == 4 and "no CPU path" in str(e.value)
