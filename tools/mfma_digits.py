"""Unweighted FIXED32 with branch lengths that are NOT short dyadic fractions (any real phylogeny):
the integer lengths then take the whole 31-bit budget, three to five base-128 digits instead of
C3's two.  Times the pass for C3's shape with the generator's lengths and with perturbed ones.
    python tools/mfma_digits.py [n_samples] [n_leaves]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import frackyfrac_amd as ff  # noqa: E402
from frackyfrac_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
leaves = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
tree, ptr, idx, val = synth.make(n, leaves, 0.1, synth.CONFIGS["C3"]["seed"])
rng = np.random.default_rng(5)
for label in ("generator lengths (multiples of 1/1024)", "the same times (1 + 1e-3 u), u uniform",
              "log-normal lengths (sigma 1.5), as in a real phylogeny", "log-normal lengths (sigma 2.5)",
              "integers below 2^14 and one of 2^20"):
    nodes = ff.flatten_leaf_csr(ff.parse_newick(tree.newick()), ptr, idx, val)
    plan = ff.Plan(nodes, False, precision="fixed32")
    out = torch.empty(plan.n_slots, dtype=torch.float64, device="cuda")
    for _ in range(3):
        plan.run(out.data_ptr())
    torch.cuda.synchronize()
    plan.timing_collect()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        plan.run(out.data_ptr(), timed=True)
    e1.record()
    torch.cuda.synchronize()
    ms, k = plan.timing_collect()
    print("%-44s digits %d (%d sweep(s) of %d planes)  items %d  kernel + reduce %.4f ms  pass %.4f ms" % (
        label, plan.info.n_digits, plan.info.n_sweeps, plan.info.planes_per_sweep, plan.info.n_items, ms / k,
        e0.elapsed_time(e1) / 20), flush=True)
    q, cap = plan.refined_pairs()
    print("%-44s    scale 2^%d, pairs sent to the binary64 walk %d; staged rows %d, %d of them with three planes" % (
        "", plan.info.scale_log2, q, plan.info.rows_padded, plan.info.rows_three_planes), flush=True)
    if label.startswith("generator"):
        tree.branch_len = tree.branch_len * (1.0 + 1e-3 * rng.random(tree.branch_len.shape[0]))
    elif label.startswith("the same"):
        bl = rng.lognormal(-3.0, 1.5, tree.branch_len.shape[0])
        bl[0] = 0.0
        tree.branch_len = bl
    elif label.endswith("1.5), as in a real phylogeny"):
        bl = rng.lognormal(-3.0, 2.5, tree.branch_len.shape[0])
        bl[0] = 0.0
        tree.branch_len = bl
    else:
        bl = rng.integers(1, 1 << 14, tree.branch_len.shape[0]).astype(np.float64)
        bl[0] = 0.0
        bl[7] = float(1 << 20)
        tree.branch_len = bl
