"""oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the frackyfrac `frcfrc` path (reference snapshot 2025-01-03)
used as the parity checker for the HIP engine.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Two layers:

* pure-Python restatements of the host-side pieces (table loaders, species
  validation, tree numbering, pair order, Go's float formatting) and of the
  per-pair distances -- literal, slow, for small cases;
* ctypes bindings to oracle/unifrac_oracle.c (built into oracle/_build/ by
  oracle/Makefile) for stage A and the all-pairs merge walk at sizes that
  matter.  tests/test_oracle_golden.py checks both layers against each other
  and against every golden vector the reference holds.

The Newick reader of the reference lives in an un-vendored module
(github.com/fluhus/biostuff v1.0.0, go.mod:6) whose source is absent; the
grammar beyond what the reference's own trees use -- '(' ',' ')' ':' ';', bare
names, decimal lengths -- is therefore "parity unpinned" (SURVEY.md 8c).

Citations are `path:line` relative to the reference root.
"""
from __future__ import annotations

import ctypes
import math
import os
import re
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libunifrac_oracle.so")

# ----------------------------------------------------------------------------
# Loaders -- parser/parser.go
# ----------------------------------------------------------------------------

# parser/parser.go:17 `\S+`: Go's regexp is RE2, whose \s is exactly [\t\n\f\r ] -- a vertical tab
# (or a Unicode space) stays INSIDE a token there, unlike Python's \s
_SPLITTER = re.compile(r"[^\t\n\f\r ]+")


class OracleError(Exception):
    """Stands for a non-nil Go error; str(e) is the message."""


def _go_float(tok: str) -> float:
    """strconv.ParseFloat(tok, 64) for the spellings a table can hold."""
    t = tok.replace("_", "x")  # Go accepts '_' only with base prefixes; reject
    try:
        if any(c.isspace() for c in t):  # Python's float() strips white space, ParseFloat does not
            raise ValueError(tok)
        if t.lower().lstrip("+-") in ("inf", "infinity", "nan"):
            return float(t)
        if t.lower().lstrip("+-").startswith("0x"):
            if "p" not in t.lower():  # Go: hexadecimal mantissa requires a 'p' exponent
                raise ValueError(tok)
            return float.fromhex(t)
        return float(t)
    except ValueError:
        raise OracleError('strconv.ParseFloat: parsing %s: invalid syntax' % go_quote(tok))


def _go_f(f: float) -> str:
    """Go's %f."""
    if math.isnan(f):
        return "NaN"
    if math.isinf(f):
        return "+Inf" if f > 0 else "-Inf"
    return "%f" % f


def _iter_rows(text: str) -> List[str]:
    """parser/parser.go:142-155 iterRows: bufio.Scanner lines (\\n or \\r\\n
    terminated; a final unterminated line counts; an empty final line does not)."""
    rows = text.split("\n")
    if rows and rows[-1] == "":
        rows.pop()
    return [r[:-1] if r.endswith("\r") else r for r in rows]


def _parse_row(row: str, names: Sequence[str]) -> Dict[str, float]:
    """parser/parser.go:59-81 parseRow."""
    parts = _SPLITTER.findall(row)
    if len(parts) != len(names):
        raise OracleError("has %d values, expected %d" % (len(parts), len(names)))
    m: Dict[str, float] = {}
    for i, p in enumerate(parts):
        try:
            f = _go_float(p)
        except OracleError as e:
            raise OracleError("value #%d: %s" % (i + 1, e))
        if math.isnan(f) or math.isinf(f) or f < 0:
            raise OracleError("value #%d: bad value: %s" % (i + 1, _go_f(f)))
        if f == 0:
            continue
        m[names[i]] = f
    return m


def parse_abundance(text: str) -> List[Dict[str, float]]:
    """parser/parser.go:21-57 ParseAbundance (dense table: header row of species
    names, then one row of numbers per sample)."""
    names: Optional[List[str]] = None
    out: List[Dict[str, float]] = []
    for row in _iter_rows(text):
        if names is None:
            parts = _SPLITTER.findall(row)
            if len(parts) == 0:
                raise OracleError("row #1 has 0 values")
            names = parts
            continue
        out.append(_parse_row(row, names))
    return out


def to_sparse(text: str) -> str:
    """sprspr/sprspr.go:19-37 toSparse: every sample of a dense table as name:%g tokens joined by
    tabs, one line per sample.  (The reference ranges over a Go map, so the order of the tokens of a
    line is random there; here it is the header's order -- compare lines as sets, as
    sprspr_test.go:27-33 does.)"""
    return "".join("\t".join("%s:%s" % (k, format_go_float(v)) for k, v in m.items()) + "\n"
                   for m in parse_abundance(text))


def split_sparse(s: str) -> Tuple[str, str]:
    """parser/parser.go:129-140 splitSparse: split at the LAST colon."""
    last = s.rfind(":")
    if last == -1:
        raise OracleError('no colon in "%s"' % s)
    return s[:last], s[last + 1:]


def _parse_sparse_row(row: str) -> Dict[str, float]:
    """parser/parser.go:102-127 parseSparseRow."""
    m: Dict[str, float] = {}
    for i, p in enumerate(_SPLITTER.findall(row)):
        try:
            species, val = split_sparse(p)
        except OracleError as e:
            raise OracleError("value #%d: %s" % (i + 1, e))
        if species == "":
            raise OracleError("value #%d: empty species name" % (i + 1))
        try:
            f = _go_float(val)
        except OracleError as e:
            raise OracleError("value #%d: %s" % (i + 1, e))
        if math.isnan(f) or math.isinf(f) or f < 0:
            raise OracleError("value #%d: bad value: %s" % (i + 1, _go_f(f)))
        if f == 0:
            raise OracleError("value #%d: zeros are not allowed in sparse format" % (i + 1))
        m[species] = f
    return m


def parse_sparse_abundance(text: str) -> List[Dict[str, float]]:
    """parser/parser.go:85-100 ParseSparseAbundance (a blank line is an empty
    sample: parser/parser_test.go:29-35)."""
    return [_parse_sparse_row(r) for r in _iter_rows(text)]


# ----------------------------------------------------------------------------
# Tree -- newick.Node{Name, Distance, Children} as the reference uses it
# ----------------------------------------------------------------------------


@dataclass
class Node:
    name: str = ""
    distance: float = 0.0
    children: List["Node"] = field(default_factory=list)


def parse_newick(text: str) -> Node:
    """First tree of the text (frcfrc/frcfrc.go:109-114 readTree takes the first).
    Grammar: what the reference's trees exercise, plus whitespace."""
    pos = 0
    n = len(text)

    def skip_ws():
        nonlocal pos
        while pos < n and text[pos].isspace():
            pos += 1

    def parse_label() -> str:
        nonlocal pos
        start = pos
        while pos < n and text[pos] not in "(),:;" and not text[pos].isspace():
            pos += 1
        return text[start:pos]

    def parse_node() -> Node:
        nonlocal pos
        node = Node()
        skip_ws()
        if pos < n and text[pos] == "(":
            pos += 1
            while True:
                node.children.append(parse_node())
                skip_ws()
                if pos < n and text[pos] == ",":
                    pos += 1
                    continue
                if pos < n and text[pos] == ")":
                    pos += 1
                    break
                raise OracleError("newick: expected ',' or ')' at offset %d" % pos)
        skip_ws()
        node.name = parse_label()
        skip_ws()
        if pos < n and text[pos] == ":":
            pos += 1
            skip_ws()
            tok = parse_label()
            node.distance = _go_float(tok)
        return node

    skip_ws()
    if pos >= n:
        raise OracleError("no tree in the given file")  # frcfrc.go:113
    root = parse_node()
    skip_ws()
    if pos >= n or text[pos] != ";":
        raise OracleError("newick: expected ';' at offset %d" % pos)
    return root


def pre_order(tree: Node) -> List[Node]:
    """newick.Node.PreOrder as used at frcfrc/unifrac.go:72,129."""
    out: List[Node] = []
    stack = [tree]
    while stack:
        nd = stack.pop()
        out.append(nd)
        stack.extend(reversed(nd.children))
    return out


def tree_names(tree: Node) -> set:
    """frcfrc/unifrac.go:70-76 treeNames."""
    return {nd.name for nd in pre_order(tree)}


def validate_species(abnd: Sequence[Dict[str, float]], tree: Node) -> None:
    """frcfrc/unifrac.go:80-93 validateSpecies.  (Which offending species is
    reported first depends on Go's map order; here: insertion order.)"""
    species = tree_names(tree)
    for i, m in enumerate(abnd):
        for name, val in m.items():
            if name not in species:
                raise OracleError(
                    "sample #%d has value %s for species %s which is not in the tree"
                    % (i + 1, format_go_float(val), go_quote(name)))


def go_quote(s: str) -> str:
    """strconv.Quote / %q: mnemonic escapes for BEL BS FF LF CR TAB VT, backslash-x-hex for the
    other control bytes, printable text as it is."""
    esc = {'"': '\\"', "\\": "\\\\", "\a": "\\a", "\b": "\\b", "\f": "\\f", "\n": "\\n", "\r": "\\r",
           "\t": "\\t", "\v": "\\v"}
    return '"' + "".join(esc.get(c, "\\x%02x" % ord(c) if (ord(c) < 0x20 or ord(c) == 0x7f) else c) for c in s) + '"'


@dataclass
class FlatTree:
    """The tree as enumerateNodes numbers it (frcfrc/unifrac.go:127-133): pre-order,
    root = 0, hence parent[id] < id; treeDists[id] = node.Distance for every node,
    root included (unifrac.go:117-120)."""
    names: List[str]
    dist: np.ndarray      # float64 [n]   treeDists
    size: np.ndarray      # int64   [n]   nodes in the subtree rooted at id
    parent: np.ndarray    # int64   [n]   -1 for the root

    @property
    def n(self) -> int:
        return len(self.names)

    def is_leaf(self) -> np.ndarray:
        return self.size == 1


def flatten_tree(tree: Node) -> FlatTree:
    names: List[str] = []
    dist: List[float] = []
    parent: List[int] = []
    size: List[int] = []
    stack: List[Tuple[Node, int]] = [(tree, -1)]
    order: List[Node] = []
    while stack:
        nd, par = stack.pop()
        my = len(names)
        names.append(nd.name)
        dist.append(nd.distance)
        parent.append(par)
        size.append(1)
        order.append(nd)
        for c in reversed(nd.children):
            stack.append((c, my))
    for i in range(len(names) - 1, 0, -1):
        size[parent[i]] += size[i]
    return FlatTree(names, np.asarray(dist, dtype=np.float64),
                    np.asarray(size, dtype=np.int64), np.asarray(parent, dtype=np.int64))


# ----------------------------------------------------------------------------
# Pure-Python restatement of the hot path (small cases only)
# ----------------------------------------------------------------------------


def abundance_to_flat_nodes_py(abnd: Dict[str, float], tree: Node,
                               enum: Dict[int, int], result: List[Tuple[int, float]]) -> float:
    """frcfrc/unifrac.go:32-53, literally (recursive)."""
    s = 0.0
    for c in tree.children:
        s += abundance_to_flat_nodes_py(abnd, c, enum, result)
    if len(tree.children) == 0:
        a = abnd.get(tree.name, 0.0)
        if a > 0:
            s += a
    if s > 0:
        result.append((enum[id(tree)], s))
    return s


def normalize_flat_nodes_py(nodes: List[Tuple[int, float]]) -> List[Tuple[int, float]]:
    """frcfrc/unifrac.go:56-67."""
    nodes = sorted(nodes, key=lambda t: t[0])
    s = 0.0
    for _, a in nodes:
        s += a
    return [(i, a / s) for i, a in nodes]


def dist_unweighted_py(a, b, tree_dists) -> float:
    """frcfrc/unifrac.go:144-171."""
    result = 0.0
    common = 0.0
    i = j = 0
    while i < len(a) and j < len(b):
        if a[i][0] < b[j][0]:
            result += tree_dists[a[i][0]]
            i += 1
            continue
        if a[i][0] > b[j][0]:
            result += tree_dists[b[j][0]]
            j += 1
            continue
        common += tree_dists[a[i][0]]
        i += 1
        j += 1
    for x in a[i:]:
        result += tree_dists[x[0]]
    for x in b[j:]:
        result += tree_dists[x[0]]
    return _go_div(result, result + common)


def dist_weighted_py(a, b, tree_dists) -> float:
    """frcfrc/unifrac.go:174-205."""
    numer = 0.0
    denom = 0.0
    i = j = 0
    while i < len(a) and j < len(b):
        if a[i][0] < b[j][0]:
            numer += tree_dists[a[i][0]] * a[i][1]
            denom += tree_dists[a[i][0]] * a[i][1]
            i += 1
            continue
        if a[i][0] > b[j][0]:
            numer += tree_dists[b[j][0]] * b[j][1]
            denom += tree_dists[b[j][0]] * b[j][1]
            j += 1
            continue
        numer += tree_dists[a[i][0]] * abs(a[i][1] - b[j][1])
        denom += tree_dists[a[i][0]] * (a[i][1] + b[j][1])
        i += 1
        j += 1
    for x in a[i:]:
        numer += tree_dists[x[0]] * x[1]
        denom += tree_dists[x[0]] * x[1]
    for x in b[j:]:
        numer += tree_dists[x[0]] * x[1]
        denom += tree_dists[x[0]] * x[1]
    return _go_div(numer, denom)


def _go_div(x: float, y: float) -> float:
    """Go float64 division: 0/0 = NaN, x/0 = +-Inf, no exception."""
    if y == 0:
        if x == 0 or math.isnan(x):
            return math.nan
        return math.copysign(math.inf, x) * math.copysign(1.0, y)
    return x / y


def iter_pairs(n: int) -> Iterable[Tuple[int, int]]:
    """common/common.go:21-31 IterPairs, as index pairs (i, j): element 0 of the
    yielded pair is s[i], the HIGHER index (common/common_test.go:9-10)."""
    for i in range(n):
        for j in range(i):
            yield i, j


def unifrac_py(abnd: Sequence[Dict[str, float]], tree: Node, weighted: bool,
               nnorm: bool = False) -> List[float]:
    """frcfrc/unifrac.go:97-124 + :209-228, literally, including the reference's
    behaviour under -l (nnorm): the lists stay in post-order (unifrac.go:108-110)."""
    nodes = pre_order(tree)
    enum = {id(nd): k for k, nd in enumerate(nodes)}
    sets = []
    for a in abnd:
        st: List[Tuple[int, float]] = []
        abundance_to_flat_nodes_py(a, tree, enum, st)
        if not nnorm:
            st = normalize_flat_nodes_py(st)
        sets.append(st)
    tree_dists = [nd.distance for nd in nodes]
    f = dist_weighted_py if weighted else dist_unweighted_py
    return [f(sets[i], sets[j], tree_dists) for i, j in iter_pairs(len(sets))]


# ----------------------------------------------------------------------------
# Go's fmt.Fprintln(w, f) for a float64 -- frcfrc/frcfrc.go:59
# ----------------------------------------------------------------------------


def format_go_float(f: float) -> str:
    """%v of a float64 = strconv.FormatFloat(f, 'g', -1, 64): shortest digits that
    round-trip; %e form when the decimal exponent is < -4 or >= 6 (strconv's
    rule for shortest %g: "use precision 6 for this decision", so 1000000.0
    prints as 1e+06 and 0.0001 stays fixed); the exponent has at least two
    digits.  Distances lie in [0, 1] or are NaN, so only the small-exponent
    branch matters for output files; the large one shows in error messages."""
    f = float(f)
    if math.isnan(f):
        return "NaN"
    if math.isinf(f):
        return "+Inf" if f > 0 else "-Inf"
    if f == 0:
        return "-0" if math.copysign(1.0, f) < 0 else "0"
    sign = "-" if f < 0 else ""
    r = repr(abs(f))
    # digits and decimal exponent from Python's shortest repr
    if "e" in r or "E" in r:
        mant, ex = r.lower().split("e")
        ex = int(ex)
    else:
        mant, ex = r, 0
    if "." in mant:
        ip, fp = mant.split(".")
    else:
        ip, fp = mant, ""
    digits = (ip + fp).lstrip("0")
    # position of the decimal point relative to the first significant digit
    lead_zeros = len(ip + fp) - len((ip + fp).lstrip("0"))
    dp = len(ip) - lead_zeros + ex  # value = 0.d1d2... * 10^dp
    digits = digits.rstrip("0") or "0"
    x = dp - 1
    if x < -4 or x >= 6:
        m = digits[0] + ("." + digits[1:] if len(digits) > 1 else "")
        return "%s%se%s%02d" % (sign, m, "-" if x < 0 else "+", abs(x))
    if dp <= 0:
        return sign + "0." + "0" * (-dp) + digits
    if dp >= len(digits):
        return sign + digits + "0" * (dp - len(digits))
    return sign + digits[:dp] + "." + digits[dp:]


def format_output(dists: Iterable[float]) -> str:
    """frcfrc/frcfrc.go:58-62: one Fprintln per distance."""
    return "".join(format_go_float(float(d)) + "\n" for d in dists)


# ----------------------------------------------------------------------------
# C layer
# ----------------------------------------------------------------------------

FLATNODE = np.dtype([("id", np.int64), ("abnd", np.float64)])
_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError("oracle library not built: run `make -C oracle` "
                               "(or __graft_entry__.build()); expected " + _LIB_PATH)
        L = ctypes.CDLL(_LIB_PATH)
        i64, f64p, vp = ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p
        L.orc_abundance_to_flat_nodes.restype = i64
        L.orc_abundance_to_flat_nodes.argtypes = [i64, vp, vp, vp]
        L.orc_normalize_flat_nodes.restype = None
        L.orc_normalize_flat_nodes.argtypes = [vp, i64]
        L.orc_dist_unweighted.restype = ctypes.c_double
        L.orc_dist_unweighted.argtypes = [vp, i64, vp, i64, vp]
        L.orc_dist_weighted.restype = ctypes.c_double
        L.orc_dist_weighted.argtypes = [vp, i64, vp, i64, vp]
        L.orc_unifrac_dists.restype = ctypes.c_int
        L.orc_unifrac_dists.argtypes = [i64, vp, vp, vp, ctypes.c_int, ctypes.c_int, i64, i64, vp]
        L.orc_flatten_samples.restype = ctypes.c_int
        L.orc_flatten_samples.argtypes = [i64, vp, i64, vp, vp, vp, ctypes.c_int, vp, vp]
        _lib = L
    return _lib


def _p(a: np.ndarray) -> int:
    return a.ctypes.data


def leaf_csr(abnd: Sequence[Dict[str, float]], ft: FlatTree) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Resolve abnd[tree.Name] (unifrac.go:40) for every leaf: CSR of (leaf node id,
    value) per sample; a leaf name that occurs k times in the tree yields k
    entries (each such leaf receives the abundance); keys naming internal nodes
    or positive values of non-leaves are dropped exactly as the reference never
    looks them up (flatNodeOptimization, unifrac.go:18,38-43)."""
    by_name: Dict[str, List[int]] = {}
    leaf = ft.is_leaf()
    for k, nm in enumerate(ft.names):
        if leaf[k]:
            by_name.setdefault(nm, []).append(k)
    ptr = [0]
    idx: List[int] = []
    val: List[float] = []
    for m in abnd:
        for nm, v in m.items():
            if v > 0:
                for k in by_name.get(nm, ()):
                    idx.append(k)
                    val.append(v)
        ptr.append(len(idx))
    return (np.asarray(ptr, dtype=np.int64), np.asarray(idx, dtype=np.int64),
            np.asarray(val, dtype=np.float64))


def flatten_samples(ft: FlatTree, leaf_ptr: np.ndarray, leaf_idx: np.ndarray,
                    leaf_val: np.ndarray, mode: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Stage A through the C restatement.  mode 0 normalise; 1 reference -l
    (post-order, raw); 2 sorted raw.  Returns (indptr int64[N+1], nodes FLATNODE[nnz])."""
    L = lib()
    ns = len(leaf_ptr) - 1
    size = np.ascontiguousarray(ft.size, dtype=np.int64)
    leaf_ptr = np.ascontiguousarray(leaf_ptr, dtype=np.int64)
    leaf_idx = np.ascontiguousarray(leaf_idx, dtype=np.int64)
    leaf_val = np.ascontiguousarray(leaf_val, dtype=np.float64)
    indptr = np.zeros(ns + 1, dtype=np.int64)
    rc = L.orc_flatten_samples(ft.n, _p(size), ns, _p(leaf_ptr), _p(leaf_idx), _p(leaf_val),
                               mode, _p(indptr), None)
    if rc != 0:
        raise MemoryError("orc_flatten_samples")
    nodes = np.zeros(int(indptr[-1]), dtype=FLATNODE)
    rc = L.orc_flatten_samples(ft.n, _p(size), ns, _p(leaf_ptr), _p(leaf_idx), _p(leaf_val),
                               mode, _p(indptr), _p(nodes))
    if rc != 0:
        raise MemoryError("orc_flatten_samples")
    return indptr, nodes


def unifrac_dists(indptr: np.ndarray, nodes: np.ndarray, tree_dists: np.ndarray,
                  weighted: bool, nthreads: int = 1, pair_begin: int = 0,
                  pair_end: Optional[int] = None) -> np.ndarray:
    """frcfrc/unifrac.go:209-228 through the C restatement.  Returns the slots
    [pair_begin, pair_end) of the IterPairs-ordered output."""
    L = lib()
    ns = len(indptr) - 1
    npairs = ns * (ns - 1) // 2
    if pair_end is None:
        pair_end = npairs
    pair_end = min(pair_end, npairs)
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    nodes = np.ascontiguousarray(nodes, dtype=FLATNODE)
    td = np.ascontiguousarray(tree_dists, dtype=np.float64)
    out = np.full(npairs if (pair_begin == 0 and pair_end == npairs) else pair_end, np.nan)
    with np.errstate(all="ignore"):
        rc = L.orc_unifrac_dists(ns, _p(indptr), _p(nodes), _p(td), 1 if weighted else 0,
                                 int(nthreads), int(pair_begin), int(pair_end), _p(out))
    if rc != 0:
        raise RuntimeError("orc_unifrac_dists failed")
    return out[pair_begin:pair_end]


def unifrac(abnd: Sequence[Dict[str, float]], tree: Node, weighted: bool,
            nnorm: bool = False, nthreads: int = 1, reference_l_quirk: bool = False) -> np.ndarray:
    """frcfrc/unifrac.go:97-124 through the C restatement.  With nnorm (-l) the
    default is the evidently intended semantics (sorted, raw counts);
    reference_l_quirk=True reproduces the reference, which skips the sort too."""
    ft = flatten_tree(tree)
    ptr, idx, val = leaf_csr(abnd, ft)
    mode = 0 if not nnorm else (1 if reference_l_quirk else 2)
    indptr, nodes = flatten_samples(ft, ptr, idx, val, mode)
    return unifrac_dists(indptr, nodes, ft.dist, weighted, nthreads)
