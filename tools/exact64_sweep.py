"""EXACT64 pair kernel over sample counts and tile heights (FF_X_TILE_H): kernel ms and fraction of the FP64 vector
rate (6 unfused binary64 ops per term against 39.3e12/s).  python tools/exact64_sweep.py [N ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import frackyfrac_amd as ff
from frackyfrac_amd import synth

cfg = synth.CONFIGS["C3"]
PEAK = 39.325e12
ns = [int(x) for x in sys.argv[1:]] or [1024, 1536, 2048, 2560, 3072, 3584, 4096, 4608, 5120, 6144]
print("%6s | default            | %s" % ("N", "  ".join("H=%-2d ms   frac" % h for h in (8, 10, 12, 14, 16))))
for n in ns:
    tree, ptr, idx, val = synth.make(n, cfg["n_leaves"], cfg["density"], cfg["seed"])
    nodes = ff.flatten_leaf_csr(ff.parse_newick(tree.newick()), ptr, idx, val)
    P = n * (n - 1) // 2
    def run(h):
        if h: os.environ["FF_X_TILE_H"] = str(h)
        else: os.environ.pop("FF_X_TILE_H", None)
        plan = ff.Plan(nodes, True, precision="exact64")
        out = torch.empty(plan.n_slots, dtype=torch.float64, device="cuda")
        plan.run(out.data_ptr()); torch.cuda.synchronize()
        for _ in range(2): plan.run(out.data_ptr(), timed=True)
        torch.cuda.synchronize()
        ms, c = plan.timing_collect(); ms /= c
        tiles = plan.info.n_tiles
        plan.close()
        return ms, 6.0 * nodes.n_branches * P / (ms * 1e-3) / PEAK, tiles
    d = run(0)
    line = "%6d | %7.2f %.3f %6d |" % (n, d[0], d[1], d[2])
    for h in (8, 10, 12, 14, 16):
        ms, fr, tiles = run(h)
        line += " %7.2f %.3f " % (ms, fr)
    print(line, flush=True)
