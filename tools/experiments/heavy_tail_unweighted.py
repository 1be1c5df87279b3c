"""Unweighted FIXED32 at C3's shape with log-normal branch lengths of sigma 2.5 (a few very long branches): 20 passes,
for a kernel trace (rocprofv3 --kernel-trace --stats -- python3 tools/experiments/heavy_tail_unweighted.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import frackyfrac_amd as ff  # noqa: E402
from frackyfrac_amd import synth  # noqa: E402

sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
tree, ptr, idx, val = synth.make(4096, 10000, 0.1, synth.CONFIGS["C3"]["seed"])
rng = np.random.default_rng(5)
rng.random(tree.branch_len.shape[0])
rng.lognormal(-3.0, 1.5, tree.branch_len.shape[0])
bl = rng.lognormal(-3.0, sigma, tree.branch_len.shape[0])
bl[0] = 0.0
tree.branch_len = bl
nodes = ff.flatten_leaf_csr(ff.parse_newick(tree.newick()), ptr, idx, val)
plan = ff.Plan(nodes, False, precision="fixed32")
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
q, cap = plan.refined_pairs()
print("pass %.4f ms, scale 2^%d, %d pairs to the binary64 walk (queue %d), rows %d (%d with three planes)" % (
    e0.elapsed_time(e1) / 20, plan.info.scale_log2, q, cap, plan.info.rows_padded, plan.info.rows_three_planes))
