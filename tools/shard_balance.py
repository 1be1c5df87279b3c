"""Proxy for multi-GPU compute balance on ONE GPU: time every rank's shard of the
weak-scaling problem (N = 4096*sqrt(G) samples) one after the other and compare with the
single-GPU C3 pass.  Excludes the gather.  Usage: shard_balance.py [G ...]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import frackyfrac_amd as ff
from frackyfrac_amd import synth

# (--strong C4 | C5: the stated size of that configuration, row shards over G ranks)
strong = sys.argv[sys.argv.index("--strong") + 1] if "--strong" in sys.argv else None
cfg = synth.CONFIGS[strong or "C3"]
args = [a for a in sys.argv[1:] if a not in ("--strong", strong)]
for G in [int(a) for a in args] or [1, 2, 4, 8]:
    n = cfg["n_samples"] if (G == 1 or strong) else int(round(cfg["n_samples"] * math.sqrt(G) / 32.0)) * 32
    tree, ptr, idx, val = synth.make(n, cfg["n_leaves"], cfg["density"], cfg["seed"])
    T = ff.parse_newick(tree.newick())
    nodes = ff.flatten_leaf_csr(T, ptr, idx, val)
    times, pairs = [], []
    for r in range(G):
        plan = ff.Plan(nodes, True, precision="fixed32", rank=r, world=G)
        out = torch.empty(plan.n_slots, dtype=torch.float64, device="cuda")
        for _ in range(2):
            plan.run(out.data_ptr())
        torch.cuda.synchronize()
        plan.timing_collect()
        for _ in range(5):
            plan.run(out.data_ptr(), timed=True)
        torch.cuda.synchronize()
        ms, k = plan.timing_collect()
        times.append(ms / k)
        pairs.append(plan.n_slots)
        plan.close()
        del out
    P = ff.num_pairs(n)
    print("G=%d N=%d pairs=%d | kernel ms per rank: %s | max %.3f mean %.3f | pairs/s if ranks ran in parallel: %.3e (%.3e per GPU)"
          % (G, n, P, " ".join("%.2f" % t for t in times), max(times), sum(times) / G, P / (max(times) * 1e-3),
             P / (max(times) * 1e-3) / G), flush=True)
