"""FIXED32 on inputs that correlate rounding residuals: equal branch lengths (unit-length
taxonomy trees, cladograms) with repeated counts (singletons).  With round-to-nearest
staging thousands of branches shared one residual and the error of a weighted distance
grew like k instead of sqrt(k): 2.7e-6 on case (a) below, no pair queued for refinement
(round-1 VERDICT, "What's weak" #1; tests/emulate_fixed32.py reproduces it on the CPU).
The staging now rounds with one offset per branch shared by all samples and divides by
binary64 weights; these tests hold every pair of those inputs to the 1e-6 bar of
unifracDistWeighted (frcfrc/unifrac.go:174-205) against the oracle, through the C ABI
and through the frcfrc command.  Needs an MI355X: `pytest -m gpu`."""
import subprocess

import numpy as np
import pytest

import frackyfrac_amd as ff
from frackyfrac_amd import _lib as L
from frackyfrac_amd import synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu

WEIGHTED_RTOL = 1e-6  # north_star: "within 1e-6 relative for weighted"


def low_diversity(ns, nl, dens, lengths, counts, seed=5):
    """synth.make's tree and presence pattern with every branch length replaced by `lengths`
    (scalar or array; the root keeps 0) and the counts drawn from a narrow set."""
    tree, ptr, idx, val = synth.make(ns, nl, dens, seed)
    rng = np.random.default_rng(11)
    tree.branch_len[:] = lengths
    tree.branch_len[0] = 0.0
    if counts == "ones":
        val = np.ones_like(val)
    elif counts == "low":  # 70 % 1, 21 % 2, the rest 3..5
        r = rng.random(len(val))
        val = np.where(r < 0.70, 1.0, np.where(r < 0.91, 2.0, 3.0 + np.floor(3 * rng.random(len(val)))))
    return tree, ptr, idx, val


def both_sides(tree, ptr, idx, val, leave):
    T = ff.parse_newick(tree.newick())
    nodes = ff.flatten_leaf_csr(T, ptr, idx, val, leave_unnormalized=leave)
    ft = O.FlatTree(tree.names, tree.branch_len, tree.size, tree.parent)
    ip, on = O.flatten_samples(ft, ptr, idx, val, 2 if leave else 0)
    return nodes, ip, on, ft


def rel_err(got, want):
    assert not np.isnan(want).any() and not np.isnan(got).any()
    return np.abs(got - want) / np.abs(want)


CASES = {
    "a_unit_lengths_counts_1": dict(ns=8, nl=10000, dens=0.3, lengths=1.0, counts="ones"),
    "b_tenth_lengths_counts_1_2": dict(ns=8, nl=10000, dens=0.3, lengths=0.1, counts="low"),
    "b2_unit_lengths_counts_1_2": dict(ns=8, nl=10000, dens=0.3, lengths=1.0, counts="low"),
    "c_50k_leaves_5pct": dict(ns=8, nl=50000, dens=0.05, lengths=0.1, counts="low"),
    "d_more_samples": dict(ns=96, nl=4000, dens=0.3, lengths=1.0, counts="ones"),
}


@pytest.mark.parametrize("leave", [False, True], ids=["normalised", "-l"])
@pytest.mark.parametrize("case", sorted(CASES))
def test_fixed32_weighted_low_diversity_every_pair(case, leave):
    nodes, ip, on, ft = both_sides(*low_diversity(**CASES[case]), leave)
    want = O.unifrac_dists(ip, on, ft.dist, True, nthreads=4)
    plan = ff.Plan(nodes, True, precision="fixed32")
    assert plan.info.precision == L.PRECISION_FIXED32
    got = plan.run_host()  # FF_ERR_PRECISION would raise: neither queue overflow nor audit failure
    err = rel_err(got, want)
    assert err.max() <= WEIGHTED_RTOL, (case, leave, float(err.max()), int(err.argmax()))
    n, bad, worst = plan.audit()
    uni, found, chk, headroom = plan.audit_detail()   # (+ the run's pairs just above the refinement rule's bound)
    assert uni == min(4096, len(want)) and n == uni + chk and chk == min(found, 4096) and bad == 0 and worst <= 0.5e-6
    # the audit sample covers every pair here (fewer than 4096): its worst error is the one measured
    # (to the ~1e-15 by which its order of additions differs from the oracle's)
    assert worst <= err.max() * 1.001 + 1e-12
    plan.close()


@pytest.mark.parametrize("case", ["a_unit_lengths_counts_1", "b_tenth_lengths_counts_1_2"])
def test_fixed32_low_diversity_through_the_cli(tmp_path, case):
    """The command with -precision fixed32 (what AUTO selects at scale) on the same inputs,
    normalised and -l, against the oracle's values."""
    tree, ptr, idx, val = low_diversity(**CASES[case])
    (tmp_path / "t.tree").write_text(tree.newick())
    (tmp_path / "t.sparse").write_text(synth.sparse_text(tree, ptr, idx, val))
    for leave in (False, True):
        nodes, ip, on, ft = both_sides(tree, ptr, idx, val, leave)
        want = O.unifrac_dists(ip, on, ft.dist, True)
        r = subprocess.run([L.FRCFRC_PATH, "-w", "-s", *(["-l", "-l-sorted"] if leave else []), "-precision", "fixed32", "-stats",
                            "-i", str(tmp_path / "t.sparse"), "-t", str(tmp_path / "t.tree")],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert '"precision": "fixed32"' in r.stderr
        got = np.array([float(x) for x in r.stdout.split()])
        assert rel_err(got, want).max() <= WEIGHTED_RTOL


@pytest.mark.parametrize("mfma", ["1", "0"])
def test_fixed32_unweighted_few_distinct_lengths(mfma, monkeypatch):
    """Unweighted with lengths off the binary grid and only a few distinct values (every
    internal branch 0.1, every leaf 0.3): round-to-nearest gives all leaves one residual and all
    internal branches another, which does not cancel in result/(result+common).  Both kernels
    (int8 matrix cores, v_sad_u32) must stay within 1e-6 of unifracDistUnweighted
    (unifrac.go:144-171); the CLI says when unweighted is tolerance-grade."""
    monkeypatch.setenv("FF_UNWEIGHTED_MFMA", mfma)
    tree, ptr, idx, val = synth.make(64, 6000, 0.2, 7)
    lengths = np.where(tree.size == 1, 0.3, 0.1)
    tree.branch_len[:] = lengths
    tree.branch_len[0] = 0.0
    nodes, ip, on, ft = both_sides(tree, ptr, idx, val, False)
    want = O.unifrac_dists(ip, on, ft.dist, False, nthreads=4)
    plan = ff.Plan(nodes, False, precision="fixed32")
    assert plan.info.lengths_exact == 0
    assert plan.info.kernel == (4 if mfma == "1" else 0)  # (64 samples: the small-shard matrix-core kernel)
    got = plan.run_host()
    assert rel_err(got, want).max() <= WEIGHTED_RTOL
    n, bad, worst = plan.audit()
    assert n >= len(want) and plan.audit_detail()[0] == len(want) and bad == 0
    plan.close()


def test_fixed32_unweighted_inexact_lengths_keep_their_bits():
    """Branch lengths of a real phylogeny (not short binary fractions): the integer lengths are scaled by
    the largest SAMPLE's sum, not the tree's.  Scaled by the tree's total, a sample that reaches a tenth of
    the tree kept 28 of the 31 bits, most pairs failed the refinement rule and the pass was the binary64
    walk (55 ms at C3's shape instead of 0.5).  Here: within 1e-6 of unifracDistUnweighted
    (unifrac.go:144-171), and only a handful of pairs go to the walk."""
    tree, ptr, idx, val = synth.make(300, 4000, 0.1, 11)
    rng = np.random.default_rng(3)
    tree.branch_len[:] = tree.branch_len * (1.0 + 1e-3 * rng.random(tree.branch_len.shape[0]))
    tree.branch_len[0] = 0.0
    nodes, ip, on, ft = both_sides(tree, ptr, idx, val, False)
    want = O.unifrac_dists(ip, on, ft.dist, False, nthreads=4)
    plan = ff.Plan(nodes, False, precision="fixed32")
    assert plan.info.lengths_exact == 0 and plan.info.kernel in (2, 4) and plan.info.n_digits >= 3
    got = plan.run_host()
    assert rel_err(got, want).max() <= WEIGHTED_RTOL
    queued, cap = plan.refined_pairs()
    assert queued <= len(want) // 50
    # the largest sample's integer sum uses the top bit of the 31 it may
    sums = np.array([np.asarray(ft.dist)[on["id"][ip[s]:ip[s + 1]]].sum() for s in range(300)])
    assert 2.0 ** 30 <= sums.max() * 2.0 ** plan.info.scale_log2 < 2.0 ** 31
    n, bad, worst = plan.audit()
    assert bad == 0
    plan.close()


def test_disjoint_samples_are_at_distance_exactly_one():
    """Samples that share no branch but the root (length 0): numer == denom term by term, so the
    reference returns exactly 1 (unifrac.go:204); FIXED32 must too, although its denominator
    is binary64 and its numerator an integer."""
    tree, ptr, idx, val = synth.make(2, 64, 0.5, 3)
    tree.branch_len[:] = 0.1
    tree.branch_len[0] = 0.0
    left = np.flatnonzero(tree.leaf_ids < 1 + tree.size[1])  # leaves under the root's first child
    right = np.flatnonzero(tree.leaf_ids >= 1 + tree.size[1])
    ptr = np.array([0, len(left), len(left) + len(right)], dtype=np.int64)
    idx = np.concatenate([tree.leaf_ids[left], tree.leaf_ids[right]])
    val = np.concatenate([np.full(len(left), 3.0), np.full(len(right), 7.0)])
    nodes, ip, on, ft = both_sides(tree, ptr, idx, val, False)
    # the root is a flat node of both samples, with length 0
    want = O.unifrac_dists(ip, on, ft.dist, True)
    assert want.tolist() == [1.0]
    for weighted in (True, False):
        got = ff.unifrac_dists(nodes, weighted, precision="fixed32")
        assert got.tolist() == [1.0]


def test_identical_samples_are_at_distance_exactly_zero():
    """Equal values get equal integers whatever the branch's offset: replicates are at 0."""
    tree, ptr, idx, val = low_diversity(2, 3000, 0.3, 0.1, "low")
    a, b = slice(ptr[0], ptr[1]), slice(ptr[1], ptr[2])
    ka, kb = int(ptr[1] - ptr[0]), int(ptr[2] - ptr[1])
    ptr = np.array([0, ka, 2 * ka, 2 * ka + kb], dtype=np.int64)  # samples: A, A again, B
    idx = np.concatenate([idx[a], idx[a], idx[b]])
    val = np.concatenate([val[a], val[a], val[b]])
    nodes, ip, on, ft = both_sides(tree, ptr, idx, val, False)
    want = O.unifrac_dists(ip, on, ft.dist, True)
    got = ff.unifrac_dists(nodes, True, precision="fixed32")
    assert want[0] == 0.0 and got[0] == 0.0
    assert rel_err(got[1:], want[1:]).max() <= WEIGHTED_RTOL


def test_cli_says_when_unweighted_is_tolerance_grade(tmp_path):
    """Unweighted through the command in fixed point with lengths off the binary grid: a line on
    stderr (and "bit_exact": false under -stats) tells the user the values are within 1e-6, not the
    reference's bits; with dyadic lengths, or in exact64, nothing is said and bit_exact is true."""
    tree, ptr, idx, val = synth.make(40, 300, 0.2, 7)
    (tmp_path / "t.sparse").write_text(synth.sparse_text(tree, ptr, idx, val))
    for lengths, precision, noted in ((0.1, "fixed32", True), (None, "fixed32", False), (0.1, "exact64", False)):
        if lengths is not None:
            tree.branch_len[:] = lengths
            tree.branch_len[0] = 0.0
        (tmp_path / "t.tree").write_text(tree.newick())
        r = subprocess.run([L.FRCFRC_PATH, "-s", "-precision", precision, "-stats", "-i", str(tmp_path / "t.sparse"), "-t",
                            str(tmp_path / "t.tree")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert ("Note: branch lengths are not multiples of a power of two" in r.stderr) == noted
        assert ('"bit_exact": false' in r.stderr) == noted and ('"bit_exact": true' in r.stderr) == (not noted)
        tree, _, _, _ = synth.make(40, 300, 0.2, 7)


def test_pairs_just_above_the_refinement_bound_are_the_ones_audited():
    """FIXED32's guarantee is statistical, and the pairs it protects least are those just ABOVE the refinement
    rule's bound U * 1e-6 >= 5 sqrt(k) + 2 (everything under it is re-computed exactly).  Round 3 audited 4,096
    uniformly drawn pairs and so, in a table with a handful of such pairs, none of them.  Here a table is built to
    have them: 300 near-copies of one sample, copy t with a fifth of its leaves scaled by (1 + f_t), f_t a
    geometric ladder from 0.02 to 4 -- the pairs (base, copy t) and (copy t, copy t') sweep U through the
    bound.  Every run must find pairs of headroom in [1, 1.25), compute them (up to 4,096) in binary64, hold what
    it delivered to the audit's bar, and report the run's smallest headroom; and the whole result is within 1e-6
    of the oracle, the planted pairs included."""
    tree, ptr, idx, val = synth.make(1, 1500, 0.4, 5)
    base = np.zeros(tree.n)
    base[idx] = 1000.0 + 50.0 * (np.arange(len(idx)) % 7)
    leaves = np.flatnonzero(base)
    pick = leaves[::5]
    rows = [base]
    for f in np.geomspace(0.02, 4.0, 300):
        r = base.copy()
        r[pick] *= 1.0 + f
        rows.append(r)
    n = len(rows)
    ptr = np.arange(n + 1, dtype=np.int64) * len(leaves)
    idx = np.tile(leaves, n).astype(np.int64)
    val = np.concatenate([r[leaves] for r in rows])
    T = ff.parse_newick(tree.newick())
    nodes = ff.flatten_leaf_csr(T, ptr, idx, val)
    ft = O.FlatTree(tree.names, tree.branch_len, tree.size, tree.parent)
    ip, on = O.flatten_samples(ft, ptr, idx, val, 0)
    want = O.unifrac_dists(ip, on, ft.dist, True, nthreads=4)
    plan = ff.Plan(nodes, True, precision="fixed32")
    assert plan.info.precision == L.PRECISION_FIXED32
    got = plan.run_host()
    queued, cap = plan.refined_pairs()
    checked, failed, worst = plan.audit()
    uniform, found, chk, headroom = plan.audit_detail()
    plan.close()
    assert 0 < queued < len(want)                       # the ladder starts under the bound ...
    assert found > 0 and chk == min(found, 4096)        # ... passes through the band just above it ...
    assert 1.0 <= headroom < 1.25                       # ... and the smallest headroom of the run is in that band
    assert checked == uniform + chk and failed == 0 and worst <= 0.5e-6
    assert rel_err(got, want).max() <= WEIGHTED_RTOL
    # the risk list is per RUN: a second run of the same plan reports the same band (not twice as many)
    plan = ff.Plan(nodes, True, precision="fixed32")
    plan.run_host()
    plan.run_host()
    assert plan.audit_detail()[1] == found
    plan.close()
