"""Where does pair_common_mfma_kernel's time go?  Times the kernel with parts of its loop body
compiled out (the DIAG template parameter; results are wrong, only the time is of interest).
Needs the diagnostic build of the library:

    make -C frackyfrac_amd/csrc diag
    FF_LIB_PATH=frackyfrac_amd/lib/libfrackyfrac_amd_diag.so python tools/mfma_diag.py [C3|C2|NxL]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import frackyfrac_amd as ff  # noqa: E402
from frackyfrac_amd import synth  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
if wl in synth.CONFIGS:
    c = synth.CONFIGS[wl]
    tree, ptr, idx, val = synth.make(c["n_samples"], c["n_leaves"], c["density"], c["seed"])
else:
    ns, nl = wl.split("x")
    tree, ptr, idx, val = synth.make(int(ns), int(nl), 0.1, 77)
nodes = ff.flatten_leaf_csr(ff.parse_newick(tree.newick()), ptr, idx, val)
plan = ff.Plan(nodes, False, precision="fixed32")
out = torch.empty(plan.n_slots, dtype=torch.float64, device="cuda")
names = {0: "full kernel", 2: "no global loads in the loop", 4: "no fragment building (vector work)",
         8: "no digit reads", 6: "no loads, no fragment building", 14: "MFMAs only"}
for diag in (0, 2, 4, 8, 6, 14):
    os.environ["FF_MFMA_DIAG"] = str(diag)
    for _ in range(3):
        plan.run(out.data_ptr())
    torch.cuda.synchronize()
    plan.timing_collect()
    for _ in range(20):
        plan.run(out.data_ptr(), timed=True)
    ms, n = plan.timing_collect()
    print("DIAG %2d  %-36s %.4f ms" % (diag, names[diag], ms / n), flush=True)
