"""RECORD OF AN EXPERIMENT (round 3): kernel time of the weighted pair kernel over sample counts and FORCED splits S of
the generic main rounds -- the data the schedule's makespan estimate was fitted to (ff_schedule.cpp build_schedule;
profiles/r03_sched_split_sweep.txt).  It drove two temporary switches (FF_EXP_PATH=g, FF_EXP_S=<S>) that forced the
generic rounds and their split; they were removed with the experiment, so this script no longer runs as it is."""
import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import frackyfrac_amd as ff
from frackyfrac_amd import synth
cfg = synth.CONFIGS["C3"]
def run(nodes):
    plan = ff.Plan(nodes, True, precision="fixed32")
    out = torch.empty(plan.n_slots, dtype=torch.float64, device="cuda")
    for _ in range(2): plan.run(out.data_ptr())
    torch.cuda.synchronize()
    for _ in range(4): plan.run(out.data_ptr(), timed=True)
    torch.cuda.synchronize()
    ms, c = plan.timing_collect()
    it = plan.info.n_items
    plan.close()
    return ms / c, it
for n in (3584, 4096, 5120, 5632, 7000):
    tree, ptr, idx, val = synth.make(n, cfg["n_leaves"], cfg["density"], cfg["seed"])
    nodes = ff.flatten_leaf_csr(ff.parse_newick(tree.newick()), ptr, idx, val)
    os.environ.pop("FF_EXP_PATH", None); os.environ.pop("FF_EXP_S", None)
    base, it = run(nodes)
    line = "N=%d default %.3f ms (%d items)" % (n, base, it)
    os.environ["FF_EXP_PATH"] = "g"
    for S in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32):
        os.environ["FF_EXP_S"] = str(S)
        try:
            ms, it = run(nodes)
            line += " | S=%d %.3f (%d)" % (S, ms, it)
        except Exception as e:
            line += " | S=%d err" % S
    print(line, flush=True)
