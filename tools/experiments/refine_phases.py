"""Diagnostic build (make -C frackyfrac_amd/csrc diag): where refine_exact_kernel's workgroup 0 spends its time --
windows' ids into LDS, counting what can be merged, the terms (merge-path searches, loads), the additions, waiting at
the barrier -- over all its pairs, at C3's shape with log-normal lengths of sigma 2.5 (8,338 queued pairs)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("FF_LIB_PATH", os.path.join(ROOT, "frackyfrac_amd", "lib", "libfrackyfrac_amd_diag.so"))
import numpy as np, torch
import frackyfrac_amd as ff
from frackyfrac_amd import synth, _lib as L

tree, ptr, idx, val = synth.make(4096, 10000, 0.1, synth.CONFIGS["C3"]["seed"])
rng = np.random.default_rng(5)
rng.random(tree.branch_len.shape[0])
rng.lognormal(-3.0, 1.5, tree.branch_len.shape[0])
bl = rng.lognormal(-3.0, 2.5, tree.branch_len.shape[0])
bl[0] = 0.0
tree.branch_len = bl
nodes = ff.flatten_leaf_csr(ff.parse_newick(tree.newick()), ptr, idx, val)
plan = ff.Plan(nodes, False, precision="fixed32")
fn = L.lib().ff_debug_small_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int64]
G = 4096
assert fn(None, G) == 0
out = torch.empty(plan.n_slots, dtype=torch.float64, device="cuda")
plan.run(out.data_ptr())
torch.cuda.synchronize()
st = np.zeros((G, 8), dtype=np.uint64)
assert fn(st.ctypes.data, G) == 0
st = st[st[:, :5].sum(axis=1) > 0].astype(np.float64)
tot = st[:, :5].sum(axis=1) / 100.0
print("%d workgroups; total us per workgroup: min %.0f p50 %.0f max %.0f" % (len(st), tot.min(), np.median(tot), tot.max()))
worst = st[np.argmax(tot)]
print("slowest: " + ", ".join("%.1f" % (t / 100.0) for t in worst[:5]))
st = np.median(st, axis=0)
q, cap = plan.refined_pairs()
names = ["ids into LDS", "count", "terms", "additions", "barrier before the next window"]
print("%d queued pairs; median workgroup, us per phase over its pairs: " % q + ", ".join("%s %.1f" % (n, t / 100.0) for n, t in zip(names, st[:5])))
