"""VERDICT r4 item 8, measured before built: would ordering the samples by a presence signature make the rows of a
12-row tile of pair_exact64_skip_kernel share presence (fewer taken branches per row, more whole-tile X-chains)?
The kernel does not care which sample is which, so the effect on its TIME shows by just permuting the samples of the
input (the distances then belong to the permuted problem; mapping slots back would only be built if this paid).
Orders tried: the generator's; lexicographic by the presence bits of the branches closest to p = 1/2 (the bits that
decide most X/Y switches between neighbouring rows); by the number of flat nodes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import frackyfrac_amd as ff
from frackyfrac_amd import synth

cfg = synth.CONFIGS["C3"]
tree, ptr, idx, val = synth.make(cfg["n_samples"], cfg["n_leaves"], cfg["density"], cfg["seed"])
T = ff.parse_newick(tree.newick())
nodes = ff.flatten_leaf_csr(T, ptr, idx, val)
n, B = nodes.n_samples, nodes.n_branches


def permuted(order):
    cnt = np.diff(nodes.indptr)[order]
    ip = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    sel = np.concatenate([np.arange(nodes.indptr[s], nodes.indptr[s + 1]) for s in order])
    return ff.FlatNodes(ip, nodes.branch_id[sel], nodes.abnd[sel], nodes.branch_len)


def time_it(nd, reps=5):
    plan = ff.Plan(nd, True, precision="exact64")
    out = torch.empty(plan.n_slots, dtype=torch.float64, device="cuda")
    plan.run(out.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.run(out.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    plan.close()
    return dt * 1e3


# presence frequency per branch
freq = np.bincount(nodes.branch_id, minlength=B) / n
mid = np.argsort(np.abs(freq - 0.5))[:48]           # the 48 branches whose presence is closest to a coin flip
pres = np.zeros((n, len(mid)), dtype=np.uint8)
where = {int(b): k for k, b in enumerate(mid)}
for s in range(n):
    ids = nodes.branch_id[nodes.indptr[s]:nodes.indptr[s + 1]]
    hit = np.intersect1d(ids, mid)
    for b in hit:
        pres[s, where[int(b)]] = 1
sig = np.lexsort(pres.T[::-1])
print("presence frequencies of the signature branches: %.2f .. %.2f" % (freq[mid].min(), freq[mid].max()))
print("generator order          %.2f ms" % time_it(nodes))
print("by presence signature    %.2f ms" % time_it(permuted(sig)))
print("by flat-node count       %.2f ms" % time_it(permuted(np.argsort(np.diff(nodes.indptr)))))
# what the order can change at all: the share of (row, branch) cells whose row differs from the row above within a tile
def switches(order):
    # sampled: 2000 random branches
    rng = np.random.default_rng(1)
    bs = rng.choice(B, 2000, replace=False)
    has = np.zeros((n, len(bs)), dtype=bool)
    pos = {int(b): k for k, b in enumerate(bs)}
    for r, s in enumerate(order):
        ids = nodes.branch_id[nodes.indptr[s]:nodes.indptr[s + 1]]
        for b in np.intersect1d(ids, bs):
            has[r, pos[int(b)]] = True
    t = has[: n // 12 * 12].reshape(-1, 12, len(bs))
    return float(np.mean(t[:, 1:, :] != t[:, :-1, :]))
print("X/Y switches per row and branch within 12-row tiles: generator %.3f, signature %.3f" % (switches(np.arange(n)), switches(sig)))
