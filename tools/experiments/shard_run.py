"""Runs the weighted pair kernel of one row shard a few times (for rocprofv3 --pmc): shard_run.py N rank world"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import frackyfrac_amd as ff
from frackyfrac_amd import synth
n, rank, world = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = synth.CONFIGS["C3"]
tree, ptr, idx, val = synth.make(n, cfg["n_leaves"], cfg["density"], cfg["seed"])
nodes = ff.flatten_leaf_csr(ff.parse_newick(tree.newick()), ptr, idx, val)
plan = ff.Plan(nodes, True, precision="fixed32", rank=rank, world=world)
out = torch.empty(plan.n_slots, dtype=torch.float64, device="cuda")
for _ in range(4):
    plan.run(out.data_ptr())
torch.cuda.synchronize()
print("items", plan.info.n_items, "wave slots", plan.info.n_wave_slots)
