"""RECORD OF AN EXPERIMENT, not a tool of the product: round 3 tried an entry point ff_plan_rebalance (Plan.rebalance)
that re-cut the stream-K remainder of the weighted schedule by the per-wave end stamps of the last completed run --
a SIMD that ended early took the remainder it could have done in the time it idled.  Same bits, and it moved the
measured spread (last SIMD's end over the mean SIMD end) from 2.4-2.8 % to 1.1-1.4 % at C3, but the kernel time by
0.1-0.8 %: what is left of the spread is run-to-run noise (rms 27 us per wave: the last of 1,024 SIMDs is 3 sigma out
every run).  Outputs: profiles/r03_rebalance_experiment*.txt; DESIGN.md 4.1 "Measured balance".  The entry point was
removed again; this script needs it and is kept only to show what was run."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import frackyfrac_amd as ff
from frackyfrac_amd import synth
for wl, n, rank, world in (("C3", None, 0, 1), ("C3", 4800, 0, 1), ("C3", 11584, 3, 8), ("C3", 11584, 7, 8), ("C5", None, 0, 1)):
    cfg = dict(synth.CONFIGS[wl])
    if n: cfg["n_samples"] = n
    tree, ptr, idx, val = synth.make(cfg["n_samples"], cfg["n_leaves"], cfg["density"], cfg["seed"])
    nodes = ff.flatten_leaf_csr(ff.parse_newick(tree.newick()), ptr, idx, val)
    plan = ff.Plan(nodes, True, precision="fixed32", rank=rank, world=world)
    out = torch.empty(plan.n_slots, dtype=torch.float64, device="cuda")
    def timeit(k=30 if wl == "C3" else 5):
        for _ in range(2): plan.run(out.data_ptr())
        torch.cuda.synchronize()
        for _ in range(k): plan.run(out.data_ptr(), timed=True)
        torch.cuda.synchronize()
        ms, c = plan.timing_collect()
        return ms / c
    base = timeit()
    ref = out.cpu().numpy().copy()
    line = "%s N=%d shard %d/%d: kernel %.3f ms items %d" % (wl, cfg["n_samples"], rank, world, base, plan.info.n_items)
    for it in range(4):
        sp = plan.rebalance(0.0)
        t = timeit()
        same = bool(np.array_equal(out.cpu().numpy(), ref))
        line += " | spread %.2f%% -> %.3f ms (%d items, same bits %s)" % (100 * sp, t, plan.info.n_items, same)
    print(line, flush=True)
    plan.close()
