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

PRNG: numpy's PCG64 seeded with 0xF4AC0000 + config number (Generator streams
are stable across numpy versions for the methods used here).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

SEED_BASE = 0xF4AC0000


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


def yule_tree(n_leaves: int, rng: np.random.Generator) -> SynthTree:
    if n_leaves < 2:
        raise ValueError("need at least 2 leaves")
    # binary tree as child arrays; node 0 = root
    left = [1]
    right = [2]
    left += [-1, -1]
    right += [-1, -1]
    leaves = [1, 2]
    picks = rng.random(n_leaves - 2)
    for t in range(n_leaves - 2):
        k = int(picks[t] * len(leaves))
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
    blen = rng.integers(1, 1025, size=total).astype(np.float64) / 1024.0
    blen[0] = 0.0
    names = [""] * total
    leaf_ids = np.flatnonzero(size == 1)
    for k, nd in enumerate(leaf_ids):
        names[nd] = "t%d" % k
    return SynthTree(parent_new, size, blen, names, leaf_ids.astype(np.int64))


def abundances(tree: SynthTree, n_samples: int, density: float, rng: np.random.Generator,
               chunk: int = 256) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Leaf-value CSR: (leaf_ptr int64[N+1], leaf_idx int64[nnz] node ids, leaf_val float64[nnz])."""
    L = len(tree.leaf_ids)
    ptr = [0]
    idx_parts, val_parts = [], []
    for s0 in range(0, n_samples, chunk):
        m = min(chunk, n_samples - s0)
        present = rng.random((m, L)) < density
        u = rng.random((m, L))
        forced = rng.integers(0, L, size=m)
        for r in range(m):
            cols = np.flatnonzero(present[r])
            if len(cols) == 0:
                cols = np.array([forced[r]])
            vals = 1.0 + np.floor(999.0 * u[r, cols] ** 2)
            idx_parts.append(tree.leaf_ids[cols])
            val_parts.append(vals)
            ptr.append(ptr[-1] + len(cols))
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


def make(n_samples: int, n_leaves: int, density: float, seed: int):
    """(tree, leaf_ptr, leaf_idx, leaf_val) for one configuration."""
    rng = np.random.default_rng(seed)
    tree = yule_tree(n_leaves, rng)
    ptr, idx, val = abundances(tree, n_samples, density, rng)
    return tree, ptr, idx, val
