"""RECORD OF AN EXPERIMENT (round 3): shards 3, 4 and 1 of 8 of an 11,584-sample problem under the 8-wave kernel and the
12-wave one with plain thirds and with 4 / 8 / no XCD slices -- the run that found the 12-wave kernel's plain thirds 7 and
15 % slower on shards 3 and 4 ("Three waves per SIMD, revisited").  Kernel ms (items).
"""
import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import frackyfrac_amd as ff
from frackyfrac_amd import synth
cfg = synth.CONFIGS["C3"]
n = 11584
tree, ptr, idx, val = synth.make(n, cfg["n_leaves"], cfg["density"], cfg["seed"])
nodes = ff.flatten_leaf_csr(ff.parse_newick(tree.newick()), ptr, idx, val)
for rank in (3, 4, 1):
    line = "shard %d/8:" % rank
    for env in ({"FF_WAVES_PER_WG": "8"}, {"FF_WAVES_PER_WG": "12"}, {"FF_WAVES_PER_WG": "12", "FF_XCD_SLICES": "4"},
                {"FF_WAVES_PER_WG": "12", "FF_XCD_SLICES": "8"}, {"FF_WAVES_PER_WG": "12", "FF_XCD_SLICES": "0"}):
        for k in ("FF_WAVES_PER_WG", "FF_XCD_SLICES"): os.environ.pop(k, None)
        os.environ.update(env)
        plan = ff.Plan(nodes, True, precision="fixed32", rank=rank, world=8)
        out = torch.empty(plan.n_slots, dtype=torch.float64, device="cuda")
        for _ in range(2): plan.run(out.data_ptr())
        torch.cuda.synchronize()
        for _ in range(8): plan.run(out.data_ptr(), timed=True)
        torch.cuda.synchronize()
        ms, c = plan.timing_collect()
        line += "  %s %.3f (%d)" % ("/".join(env.values()), ms / c, plan.info.n_items)
        plan.close()
    print(line, flush=True)
