"""Deterministic synthetic inputs for tests and bench.py.

Tree: Yule process -- start from a root with two leaves, repeatedly split a
uniformly random current leaf into an internal node with two leaf children until
there are `n_leaves` leaves; B = 2*n_leaves - 1 nodes.  Nodes are numbered in
pre-order (root = 0), leaves are named t<k> in pre-order, internal nodes are
unnamed, the root has length 0 and every other branch length is k/1024 with k
uniform in [1, 1024] (dyadic, so sums of lengths are exact in binary64 in any
order and unweighted results are order-independent).

Abundances: each (sample, leaf) is present independently with probability
`density`; the value is the integer count 1 + floor(999 * u^2), u uniform in
[0, 1).  Every sample is forced to hold at least one leaf.

PRNG: splitmix64 -> xoshiro256**, seed = 0xF4AC0000 + config number.  Any host
can regenerate the inputs from this recipe, no numpy needed:
  * stream(seed, k): a xoshiro256** generator whose four state words are four consecutive
    outputs of splitmix64 started at  seed XOR (0xD1B54A32D192ED03 * k mod 2^64);
    double() = (next() >> 11) * 2^-53.
  * tree = stream(seed, 0): for t = 0 .. n_leaves-3 the leaf to split is
    floor(double() * number_of_current_leaves) in the list of current leaves (the split leaf's
    slot takes its first child, the second child is appended); then, for the nodes in
    pre-order, the root included, length = (1 + floor(double() * 1024)) / 1024 (the root's is
    then set to 0).
  * sample s = stream(seed, s + 1): for the leaves in pre-order two draws each, u1 then u2:
    present iff u1 < density, count = 1 + floor(999 * u2 * u2); after all leaves one more draw
    u3: a sample with no leaf present holds leaf floor(u3 * n_leaves) with the count that leaf
    drew.
Samples have streams of their own, so a process can generate any range of samples
(`abundances(..., begin, end)`): with one process per GPU each rank generates, flattens and
uploads only its share, and the flat nodes are all-gathered (frackyfrac_amd/distributed.py).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

SEED_BASE = 0xF4AC0000
_M64 = (1 << 64) - 1
_STREAM_MUL = 0xD1B54A32D192ED03


def _splitmix64(state: int) -> Tuple[int, int]:
    """(next state, output) of splitmix64."""
    state = (state + 0x9E3779B97F4A7C15) & _M64
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return state, z ^ (z >> 31)


class Xoshiro:
    """xoshiro256** on Python integers (the tree's stream; the reference for XoshiroVec)."""

    def __init__(self, seed: int, stream: int = 0):
        st = (seed ^ ((_STREAM_MUL * stream) & _M64)) & _M64
        self.s = []
        for _ in range(4):
            st, out = _splitmix64(st)
            self.s.append(out)

    def next(self) -> int:
        s = self.s
        x = (s[1] * 5) & _M64
        res = ((((x << 7) | (x >> 57)) & _M64) * 9) & _M64
        t = (s[1] << 17) & _M64
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = ((s[3] << 45) | (s[3] >> 19)) & _M64
        return res

    def double(self) -> float:
        return (self.next() >> 11) * (1.0 / 9007199254740992.0)


class XoshiroVec:
    """One xoshiro256** stream per entry of `streams` (uint64 stream numbers), stepped together."""

    def __init__(self, seed: int, streams: np.ndarray):
        with np.errstate(over="ignore"):
            st = np.uint64(seed) ^ (np.uint64(_STREAM_MUL) * streams.astype(np.uint64))
            self.s = []
            for _ in range(4):
                st = st + np.uint64(0x9E3779B97F4A7C15)
                z = st.copy()
                z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
                z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
                self.s.append(z ^ (z >> np.uint64(31)))

    def next(self) -> np.ndarray:
        s = self.s
        with np.errstate(over="ignore"):
            x = s[1] * np.uint64(5)
            res = ((x << np.uint64(7)) | (x >> np.uint64(57))) * np.uint64(9)
            t = s[1] << np.uint64(17)
        s[2] = s[2] ^ s[0]
        s[3] = s[3] ^ s[1]
        s[1] = s[1] ^ s[2]
        s[0] = s[0] ^ s[3]
        s[2] = s[2] ^ t
        s[3] = (s[3] << np.uint64(45)) | (s[3] >> np.uint64(19))
        return res

    def double(self) -> np.ndarray:
        return (self.next() >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


@dataclass
class SynthTree:
    parent: np.ndarray      # int64 [B], -1 for the root, pre-order numbering
    size: np.ndarray        # int64 [B], nodes in the subtree
    branch_len: np.ndarray  # float64 [B]
    names: List[str]        # "" for internal nodes, t<k> for leaves
    leaf_ids: np.ndarray    # int64 [L] node id of leaf t<k>

    @property
    def n(self) -> int:
        return len(self.parent)

    def newick(self) -> str:
        """Newick text whose pre-order is this numbering."""
        B = self.n
        # children lists
        first_child = np.full(B, -1, dtype=np.int64)
        out: List[str] = []
        # iterative emit using subtree sizes: children of id are id+1, id+1+size[id+1], ...
        stack: List[Tuple[int, int]] = [(0, 0)]  # (node, state) state 0 = open, 1 = close
        size, blen, names = self.size, self.branch_len, self.names
        while stack:
            nd, st = stack.pop()
            if st == 0:
                if size[nd] == 1:
                    out.append(names[nd])
                    if nd != 0 or blen[nd] != 0:
                        out.append(":%s" % _fmt_len(blen[nd]))
                else:
                    out.append("(")
                    stack.append((nd, 1))
                    kids = []
                    c = nd + 1
                    end = nd + size[nd]
                    while c < end:
                        kids.append(c)
                        c += size[c]
                    for idx in range(len(kids) - 1, -1, -1):
                        stack.append((kids[idx], 0))
                        if idx > 0:
                            stack.append((-1, 2))
            elif st == 2:
                out.append(",")
            else:
                out.append(")")
                out.append(names[nd])
                if nd != 0 or blen[nd] != 0:
                    out.append(":%s" % _fmt_len(blen[nd]))
        out.append(";\n")
        return "".join(out)


def _fmt_len(x: float) -> str:
    return repr(float(x))


def yule_tree(n_leaves: int, seed: int) -> SynthTree:
    if n_leaves < 2:
        raise ValueError("need at least 2 leaves")
    rng = Xoshiro(seed, 0)
    # binary tree as child arrays; node 0 = root
    left = [1]
    right = [2]
    left += [-1, -1]
    right += [-1, -1]
    leaves = [1, 2]
    for _ in range(n_leaves - 2):
        k = int(rng.double() * len(leaves))
        nd = leaves[k]
        a = len(left)
        left.extend([-1, -1])
        right.extend([-1, -1])
        left[nd] = a
        right[nd] = a + 1
        leaves[k] = a
        leaves.append(a + 1)
    total = len(left)
    # pre-order renumbering
    order = np.empty(total, dtype=np.int64)   # old id at new position
    newid = np.empty(total, dtype=np.int64)
    parent_new = np.full(total, -1, dtype=np.int64)
    stack = [(0, -1)]
    pos = 0
    while stack:
        nd, par = stack.pop()
        order[pos] = nd
        newid[nd] = pos
        parent_new[pos] = par
        if left[nd] >= 0:
            stack.append((right[nd], pos))
            stack.append((left[nd], pos))
        pos += 1
    size = np.ones(total, dtype=np.int64)
    for i in range(total - 1, 0, -1):
        size[parent_new[i]] += size[i]
    blen = np.array([(1 + int(rng.double() * 1024)) / 1024.0 for _ in range(total)], dtype=np.float64)
    blen[0] = 0.0
    names = [""] * total
    leaf_ids = np.flatnonzero(size == 1)
    for k, nd in enumerate(leaf_ids):
        names[nd] = "t%d" % k
    return SynthTree(parent_new, size, blen, names, leaf_ids.astype(np.int64))


def abundances(tree: SynthTree, n_samples: int, density: float, seed: int, begin: int = 0,
               end: Optional[int] = None, threads: int = 0) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Leaf-value CSR of samples [begin, end) (default: all): (leaf_ptr int64[m+1], leaf_idx int64[nnz]
    node ids, leaf_val float64[nnz]).  Generated by the library's C twin of the recipe
    (csrc/ff_synth.cpp: ff_synth_counts / ff_synth_fill), on `threads` host threads (0: all)."""
    import os

    from . import _lib as L

    if end is None:
        end = n_samples
    m, nl = end - begin, len(tree.leaf_ids)
    if threads <= 0:
        threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    counts = np.zeros(max(m, 1), dtype=np.int64)
    rc = L.lib().ff_synth_counts(nl, float(density), int(seed), begin, end, threads, counts.ctypes.data)
    if rc:
        raise ValueError("ff_synth_counts: bad argument")
    ptr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(counts[:m], out=ptr[1:])
    ordinal = np.zeros(max(int(ptr[-1]), 1), dtype=np.int64)
    val = np.zeros(max(int(ptr[-1]), 1), dtype=np.float64)
    rc = L.lib().ff_synth_fill(nl, float(density), int(seed), begin, end, threads, ptr.ctypes.data, ordinal.ctypes.data,
                               val.ctypes.data)
    if rc:
        raise ValueError("ff_synth_fill: bad argument")
    nnz = int(ptr[-1])
    return ptr, tree.leaf_ids[ordinal[:nnz]].astype(np.int64), val[:nnz]


def abundances_numpy(tree: SynthTree, n_samples: int, density: float, seed: int, begin: int = 0,
                     end: Optional[int] = None, chunk: int = 4096) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """The same table from the numpy statement of the recipe (XoshiroVec): what tests compare the C
    generator with; ten times slower."""
    if end is None:
        end = n_samples
    L = len(tree.leaf_ids)
    ptr = [0]
    idx_parts, val_parts = [], []
    for s0 in range(begin, end, chunk):
        m = min(chunk, end - s0)
        gen = XoshiroVec(seed, np.arange(s0 + 1, s0 + m + 1, dtype=np.uint64))
        present = np.empty((L, m), dtype=bool)
        count = np.empty((L, m), dtype=np.uint16)
        for k in range(L):  # the streams step together: two draws per leaf
            present[k] = gen.double() < density
            u = gen.double()
            count[k] = (1.0 + np.floor(999.0 * u * u)).astype(np.uint16)
        forced = np.minimum((gen.double() * L).astype(np.int64), L - 1)
        present = np.ascontiguousarray(present.T)
        count = np.ascontiguousarray(count.T)
        for r in range(m):
            cols = np.flatnonzero(present[r])
            if len(cols) == 0:
                cols = np.array([forced[r]])
            idx_parts.append(tree.leaf_ids[cols])
            val_parts.append(count[r, cols].astype(np.float64))
            ptr.append(ptr[-1] + len(cols))
    if not idx_parts:
        return np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.float64)
    return (np.asarray(ptr, dtype=np.int64), np.concatenate(idx_parts).astype(np.int64),
            np.concatenate(val_parts).astype(np.float64))


def sparse_text(tree: SynthTree, leaf_ptr, leaf_idx, leaf_val) -> str:
    """The table in the reference's sparse format (name:value tokens, one sample per line)."""
    lines = []
    for s in range(len(leaf_ptr) - 1):
        a, b = leaf_ptr[s], leaf_ptr[s + 1]
        lines.append(" ".join("%s:%d" % (tree.names[leaf_idx[k]], int(leaf_val[k])) for k in range(a, b)))
    return "\n".join(lines) + "\n"


def dense_text(tree: SynthTree, leaf_ptr, leaf_idx, leaf_val) -> str:
    """The table in the reference's dense format (header of species, one row per sample)."""
    L = len(tree.leaf_ids)
    col = {int(nd): k for k, nd in enumerate(tree.leaf_ids)}
    lines = [" ".join(tree.names[nd] for nd in tree.leaf_ids)]
    for s in range(len(leaf_ptr) - 1):
        row = ["0"] * L
        for k in range(leaf_ptr[s], leaf_ptr[s + 1]):
            row[col[int(leaf_idx[k])]] = "%d" % int(leaf_val[k])
        lines.append(" ".join(row))
    return "\n".join(lines) + "\n"


# BASELINE.json configs: (samples, leaves, density, weighted)
CONFIGS = {
    "C2": dict(n_samples=512, n_leaves=2000, density=0.10, weighted=False, seed=SEED_BASE + 2),
    "C3": dict(n_samples=4096, n_leaves=10000, density=0.10, weighted=True, seed=SEED_BASE + 3),
    "C4": dict(n_samples=16384, n_leaves=10000, density=0.10, weighted=True, seed=SEED_BASE + 4),
    "C5": dict(n_samples=8192, n_leaves=50000, density=0.05, weighted=True, seed=SEED_BASE + 5),
}


def make(n_samples: int, n_leaves: int, density: float, seed: int, begin: int = 0, end: Optional[int] = None):
    """(tree, leaf_ptr, leaf_idx, leaf_val) for one configuration (samples [begin, end) of it)."""
    tree = yule_tree(n_leaves, seed)
    ptr, idx, val = abundances(tree, n_samples, density, seed, begin, end)
    return tree, ptr, idx, val
