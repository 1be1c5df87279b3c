"""How the weighted pair kernel's efficiency depends on the shard's shape: kernel ms and fraction of the vector-ALU
roofline over sample counts (C3's tree), whole problems and shards of a bigger one.  python tools/shape_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import frackyfrac_amd as ff
from frackyfrac_amd import synth

cfg = synth.CONFIGS["C3"]
PEAK = 78.65e12
weighted = "--unweighted" not in sys.argv
PEAK = PEAK if weighted else 5.0e15   # (unweighted: the nominal int8 matrix rate; the timed region includes the reduce)
ns = [int(x) for x in sys.argv[1:] if not x.startswith("--")] or list(range(1024, 8193, 512)) + [4800, 5000, 6000, 7000]
print("%6s %8s %7s %7s %9s %7s" % ("N", "tiles", "items", "ms", "pairs/s", "frac"))
for n in sorted(set(ns)):
    tree, ptr, idx, val = synth.make(n, cfg["n_leaves"], cfg["density"], cfg["seed"])
    nodes = ff.flatten_leaf_csr(ff.parse_newick(tree.newick()), ptr, idx, val)
    plan = ff.Plan(nodes, weighted, precision="fixed32")
    out = torch.empty(plan.n_slots, dtype=torch.float64, device="cuda")
    for _ in range(2):
        plan.run(out.data_ptr())
    torch.cuda.synchronize()
    k = (10 if n <= 4096 else 4) * (1 if weighted else 10)
    for _ in range(k):
        plan.run(out.data_ptr(), timed=True)
    torch.cuda.synchronize()
    ms, c = plan.timing_collect()
    ms /= c
    P = n * (n - 1) // 2
    frac = 2.0 * nodes.n_branches * P / (ms * 1e-3) / PEAK
    print("%6d %8d %7d %7.3f %9.3e %7.3f" % (n, plan.info.n_tiles, plan.info.n_items, ms, P / (ms * 1e-3), frac), flush=True)
    plan.close()
