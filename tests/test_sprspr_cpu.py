"""The dense -> sparse converter (sprspr/sprspr.go:19-44) behind the C ABI (ff_table_write_sparse)
and as the `sprspr` command: the reference's own test vectors (sprspr_test.go:11-38, lines compared
as sets of tokens because the reference's token order is a Go map's), the oracle's restatement,
and the round trip through the sparse loader."""
import ctypes
import subprocess

import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

import frackyfrac_amd as ff
from conftest import GOLDEN, read_golden
from frackyfrac_amd import _lib as L
from oracle import oracle as O


def sprspr(text):
    r = subprocess.run([L.SPRSPR_PATH], input=text, capture_output=True, text=True)
    return r.returncode, r.stdout, r.stderr


def as_sets(out):
    return [sorted(line.split("\t")) if line else [] for line in out.rstrip("\n").split("\n")]


@pytest.mark.parametrize("dense,want", [
    ("s1\n1.3", "s1:1.3"),
    ("s1\ts2\n0\t4\n3\t0\n", "s2:4\ns1:3"),
    ("s1\ts2\ts3\n4\t3\t2\n5\t0\t8\n0\t0\t10", "s1:4\ts2:3\ts3:2\ns1:5\ts3:8\ns3:10"),
])
def test_reference_vectors(dense, want):
    """sprspr_test.go:15-20."""
    rc, out, err = sprspr(dense)
    assert rc == 0
    assert as_sets(out) == as_sets(want)
    assert as_sets(O.to_sparse(dense)) == as_sets(want)
    assert err.startswith("SparseySparse converts dense format abundance tables to sparse format.\n\nUsage:\n"
                          "sprspr < INPUT_FILE > OUTPUT_FILE\n\nReading standard input...\n")


@pytest.mark.parametrize("name", ["uwtd1", "uwtd2", "wtd"])
def test_golden_dense_files_convert_to_their_sparse_twins(name):
    rc, out, _ = sprspr(read_golden(name + ".dense"))
    assert rc == 0
    assert ff.parse_sparse_abundance(out).to_maps() == ff.parse_sparse_abundance(read_golden(name + ".sparse")).to_maps()
    assert out == O.to_sparse(read_golden(name + ".dense"))


def test_errors_are_the_loaders_and_exit_code_2():
    rc, out, err = sprspr("a b\n1 x\n")
    assert rc == 2 and out == ""
    assert err.endswith('ERROR: value #2: strconv.ParseFloat: parsing "x": invalid syntax\n')
    rc, out, err = sprspr("a b\n1\n")
    assert rc == 2 and "has 1 values, expected 2" in err


def test_c_abi_entry_writes_a_file(tmp_path):
    t = ff.parse_abundance("x y z\n0.5 0 1e-7\n0 0 0\n3 2 1\n")
    err = L.errbuf()
    L.check(L.lib().ff_table_write_sparse(t._h, str(tmp_path / "o.sparse").encode(), err, L.ERRLEN), err)
    text = (tmp_path / "o.sparse").read_text()
    assert text == "x:0.5\tz:1e-07\n\nx:3\ty:2\tz:1\n"          # an all-zero sample is an empty line
    assert ff.parse_sparse_abundance(text).to_maps() == t.to_maps()


names = st.lists(st.text(alphabet="abcdefgXYZ_.0123456789", min_size=1, max_size=6), min_size=1, max_size=6, unique=True)


@settings(max_examples=60, deadline=None)
@given(names, st.data())
def test_round_trip_through_the_sparse_loader(header, data):
    rows = data.draw(st.lists(st.lists(st.one_of(st.just(0.0), st.floats(min_value=1e-9, max_value=1e9, allow_nan=False)),
                                       min_size=len(header), max_size=len(header)), min_size=1, max_size=5))
    dense = " ".join(header) + "\n" + "\n".join(" ".join(repr(v) for v in r) for r in rows) + "\n"
    rc, out, _ = sprspr(dense)
    assert rc == 0
    assert out == O.to_sparse(dense)
    assert ff.parse_sparse_abundance(out).to_maps() == ff.parse_abundance(dense).to_maps()
