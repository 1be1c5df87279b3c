"""What a few queued pairs add to a pass: C3 (weighted, and unweighted with lengths off the binary grid) with samples
1..8 replaced by near-copies of sample 0 (28 + pairs go to the binary64 walk), against the same without them."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import frackyfrac_amd as ff  # noqa: E402
from frackyfrac_amd import synth  # noqa: E402

tree, ptr, idx, val = synth.make(4096, 10000, 0.1, synth.CONFIGS["C3"]["seed"])
rng = np.random.default_rng(3)


def timed(nodes, weighted):
    plan = ff.Plan(nodes, weighted, precision="fixed32")
    out = torch.empty(plan.n_slots, dtype=torch.float64, device="cuda")
    for _ in range(3):
        plan.run(out.data_ptr())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        plan.run(out.data_ptr())
    e1.record()
    torch.cuda.synchronize()
    q, _ = plan.refined_pairs()
    plan.close()
    return e0.elapsed_time(e1) / 20, q


for weighted in (True, False):
    if not weighted:
        tree.branch_len = tree.branch_len * (1.0 + 1e-3 * rng.random(tree.branch_len.shape[0]))
    rows = [(idx[ptr[s]:ptr[s + 1]], val[ptr[s]:ptr[s + 1]]) for s in range(4096)]
    plain = ff.flatten_leaf_csr(ff.parse_newick(tree.newick()), ptr, idx, val)
    for s in range(1, 9):
        v = rows[0][1].copy()
        v[rng.integers(0, len(v), 2)] += 1.0
        keep = np.ones(len(v), bool)
        keep[rng.integers(0, len(v), s % 3)] = False
        rows[s] = (rows[0][0][keep], v[keep])
    p2 = np.concatenate([[0], np.cumsum([len(r[0]) for r in rows])]).astype(np.int64)
    reps = ff.flatten_leaf_csr(ff.parse_newick(tree.newick()), p2, np.concatenate([r[0] for r in rows]),
                               np.concatenate([r[1] for r in rows]))
    a, qa = timed(plain, weighted)
    b, qb = timed(reps, weighted)
    print("%s: %.4f ms per pass with %d queued pairs, %.4f ms with %d" % ("weighted" if weighted else "unweighted", a, qa, b, qb), flush=True)
