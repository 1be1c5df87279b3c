"""RECORD OF AN EXPERIMENT (round 3): the weighted pair kernel with two and with three waves per SIMD (FF_WAVES_PER_WG = 8 / 12)
on whole problems -- C3, 5,632 and 3,072 samples of its tree, C5 -- after the schedule was rewritten.
"""
import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import frackyfrac_amd as ff
from frackyfrac_amd import synth
for wl, n in (("C3", None), ("C3", 5632), ("C3", 3072), ("C5", None)):
    cfg = dict(synth.CONFIGS[wl])
    if n: cfg["n_samples"] = n
    tree, ptr, idx, val = synth.make(cfg["n_samples"], cfg["n_leaves"], cfg["density"], cfg["seed"])
    nodes = ff.flatten_leaf_csr(ff.parse_newick(tree.newick()), ptr, idx, val)
    line = "%s N=%d:" % (wl, cfg["n_samples"])
    for w in ("8", "12"):
        os.environ["FF_WAVES_PER_WG"] = w
        plan = ff.Plan(nodes, True, precision="fixed32")
        out = torch.empty(plan.n_slots, dtype=torch.float64, device="cuda")
        for _ in range(2): plan.run(out.data_ptr())
        torch.cuda.synchronize()
        k = 20 if wl == "C3" else 4
        for _ in range(k): plan.run(out.data_ptr(), timed=True)
        torch.cuda.synchronize()
        ms, c = plan.timing_collect()
        line += "  %s waves/WG %.3f ms (%d items)" % (w, ms / c, plan.info.n_items)
        plan.close()
    print(line, flush=True)
