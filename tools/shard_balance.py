"""Proxy for multi-GPU compute balance on ONE GPU: time every rank's shard of a problem one after the other --
the weak-scaling problem (N = 4096*sqrt(G) samples; compare with the single-GPU C3 pass) or, with --strong C4 | C5,
that configuration at its stated size in row shards over G ranks.  Excludes the gather.  Per policy of the weighted
pair kernel's wave count -- auto (the plan's own choice per shard), 8, 12 (FF_WAVES_PER_WG) -- one line: each rank's
kernel ms with the waves per workgroup the plan took in brackets, and max / mean, the figure that bounds scaling.
Usage: shard_balance.py [--strong C4|C5] [--policies auto,8,12] [G ...]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import frackyfrac_amd as ff
from frackyfrac_amd import _lib as L
from frackyfrac_amd import synth


def opt(name, default=None):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


strong = opt("--strong")
policies = opt("--policies", "auto").split(",")
cfg = synth.CONFIGS[strong or "C3"]
skip = {"--strong", strong, "--policies", opt("--policies")}
for G in [int(a) for a in sys.argv[1:] if a not in skip] or [1, 2, 4, 8]:
    n = cfg["n_samples"] if (G == 1 or strong) else int(round(cfg["n_samples"] * math.sqrt(G) / 32.0)) * 32
    tree, ptr, idx, val = synth.make(n, cfg["n_leaves"], cfg["density"], cfg["seed"])
    T = ff.parse_newick(tree.newick())
    nodes = ff.flatten_leaf_csr(T, ptr, idx, val)
    P = ff.num_pairs(n)
    for pol in policies:
        L.lib().ff_tune(b"FF_WAVES_PER_WG", None if pol == "auto" else pol.encode())
        times, waves = [], []
        for r in range(G):
            # a plan per rank, as the ranks of a real run make them: what is staged depends on the shard (the side of
            # pair_low_kernel's blocks of pairs is chosen for the shard's own block count)
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
            waves.append(int(plan.info.n_wave_slots // plan.info.n_compute_units))
            del out
            plan.close()
        print("%s G=%d N=%d pairs=%d waves=%-4s | kernel ms per rank: %s | max %.3f mean %.3f max/mean %.3f | pairs/s if ranks ran "
              "in parallel: %.3e (%.3e per GPU)"
              % (strong or "weak", G, n, P, pol, " ".join("%.2f[%d]" % (t, w) for t, w in zip(times, waves)), max(times),
                 sum(times) / G, max(times) / (sum(times) / G), P / (max(times) * 1e-3), P / (max(times) * 1e-3) / G), flush=True)
    L.lib().ff_tune(b"FF_WAVES_PER_WG", None)
