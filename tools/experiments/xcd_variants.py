"""Weighted pair kernel (12 waves per workgroup) with the main rounds' variants: generic split by estimate (default),
XCD-sliced with 2 / 4 / 8 slices (FF_XCD_SLICES forces them where they apply) -- whole problems and row shards."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import frackyfrac_amd as ff
from frackyfrac_amd import synth
cfg = synth.CONFIGS["C3"]
cases = [(n, 0, 1) for n in (3072, 4096, 5120, 5632, 6144, 7168, 8192)] + [(11584, r, 8) for r in range(8)] + [(16384, r, 8) for r in (0, 3, 7)]
cache = {}
for n, rank, world in cases:
    if n not in cache:
        cache.clear()
        tree, ptr, idx, val = synth.make(n, cfg["n_leaves"], cfg["density"], cfg["seed"])
        cache[n] = ff.flatten_leaf_csr(ff.parse_newick(tree.newick()), ptr, idx, val)
    nodes = cache[n]
    line = "N=%5d shard %d/%d:" % (n, rank, world)
    for x in (None, "2", "4", "8"):
        os.environ.pop("FF_XCD_SLICES", None)
        if x: os.environ["FF_XCD_SLICES"] = x
        plan = ff.Plan(nodes, True, precision="fixed32", rank=rank, world=world)
        out = torch.empty(plan.n_slots, dtype=torch.float64, device="cuda")
        for _ in range(2): plan.run(out.data_ptr())
        torch.cuda.synchronize()
        for _ in range(6): plan.run(out.data_ptr(), timed=True)
        torch.cuda.synchronize()
        ms, c = plan.timing_collect()
        line += "  %s %.3f (%d, %dw)" % (x or "default", ms / c, plan.info.n_items, plan.info.n_wave_slots // plan.info.n_compute_units)
        plan.close()
    print(line, flush=True)
