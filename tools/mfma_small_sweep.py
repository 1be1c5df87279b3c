"""Where pair_common_small_kernel (one launch, 32 x 32 tiles) beats pair_common_mfma_kernel (+ reduce) for
unweighted FIXED32: ms per pass of both over a grid of sample counts and tree sizes, and the work figure
(32 x 32 tiles x k-steps) the plan's threshold S_MAX_WORK is stated in.  Run on the GPU box:
    python tools/mfma_small_sweep.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import frackyfrac_amd as ff  # noqa: E402
from frackyfrac_amd import synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def per_pass(nodes, small):
    os.environ["FF_MFMA_SMALL"] = small
    plan = ff.Plan(nodes, False, precision="fixed32")
    out = torch.empty(plan.n_slots, dtype=torch.float64, device="cuda")
    for _ in range(5):
        plan.run(out.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.run(out.data_ptr(), timed=True)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    ms, n = plan.timing_collect()
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.run(out.data_ptr())   # without the event pair around every launch
    torch.cuda.synchronize()
    wall = min(wall, (time.perf_counter() - t0) / steps * 1e3)
    info = plan.info
    res = out.cpu().numpy()
    plan.close()
    return wall, ms / n, info, res


print("%6s %7s %9s %10s | %9s %9s | %9s %9s | %s" % ("N", "leaves", "tiles32", "work", "small ms", "kernel", "big ms", "kernels", "same bits"))
for n, leaves in ((128, 2000), (256, 2000), (512, 2000), (512, 10000), (768, 2000), (768, 10000), (1024, 2000), (1024, 10000),
                  (1536, 2000), (1536, 10000), (2048, 2000), (2048, 10000), (512, 16000), (3000, 2000)):
    tree, ptr, idx, val = synth.make(n, leaves, 0.1, 1234 + n)
    nodes = ff.flatten_leaf_csr(ff.parse_newick(tree.newick()), ptr, idx, val)
    ws, ks, info_s, rs = per_pass(nodes, "1")
    wb, kb, info_b, rb = per_pass(nodes, "0")
    work = info_s.n_tiles * 2 * (info_s.rows_padded // 64)
    print("%6d %7d %9d %10d | %9.4f %9.4f | %9.4f %9.4f | %s" % (n, leaves, info_s.n_tiles, work, ws, ks, wb, kb,
                                                                  bool(np.array_equal(rs, rb))), flush=True)
