"""Fixed cost vs per-slab cost of the unweighted MFMA path: times the pair kernel (HIP events
of ff_plan_run_timed) at a fixed sample count for several tree sizes; the
slope over the slab count is the loop, the intercept everything else (launch, digit table,
epilogues).    python tools/mfma_scaling.py [n_samples]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import frackyfrac_amd as ff  # noqa: E402
from frackyfrac_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
pts = []
for leaves in (1250, 2500, 5000, 10000, 20000):
    tree, ptr, idx, val = synth.make(n, leaves, 0.1, 77)
    nodes = ff.flatten_leaf_csr(ff.parse_newick(tree.newick()), ptr, idx, val)
    plan = ff.Plan(nodes, False, precision="fixed32")
    out = torch.empty(plan.n_slots, dtype=torch.float64, device="cuda")
    for _ in range(3):
        plan.run(out.data_ptr())
    torch.cuda.synchronize()
    plan.timing_collect()
    for _ in range(20):
        plan.run(out.data_ptr(), timed=True)
    ms, k = plan.timing_collect()
    slabs = plan.info.rows_padded // 64
    pts.append((slabs, ms / k))
    print("N=%d leaves=%6d slabs=%5d digits=%d items=%d: %.4f ms" % (n, leaves, slabs, plan.info.n_digits, plan.info.n_items, ms / k), flush=True)
    plan.close()
x, y = np.array([p[0] for p in pts], float), np.array([p[1] for p in pts], float)
a, b = np.polyfit(x, y, 1)
print("fit: %.1f us fixed + %.3f us per slab" % (b * 1e3, a * 1e3))
