"""Per-wave end stamps of one shard of a bigger problem (diagnostic build): python shard_stamps.py N rank world [waves]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("FF_LIB_PATH", os.path.join(ROOT, "frackyfrac_amd", "lib", "libfrackyfrac_amd_diag.so"))
os.environ["FF_STAMPS"] = "1"
n, rank, world = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
if len(sys.argv) > 4:
    os.environ["FF_WAVES_PER_WG"] = sys.argv[4]
import numpy as np, torch
import frackyfrac_amd as ff
from frackyfrac_amd import synth, _lib as L
cfg = synth.CONFIGS["C3"]
tree, ptr, idx, val = synth.make(n, cfg["n_leaves"], cfg["density"], cfg["seed"])
nodes = ff.flatten_leaf_csr(ff.parse_newick(tree.newick()), ptr, idx, val)
plan = ff.Plan(nodes, True, precision="fixed32", rank=rank, world=world)
out = torch.empty(plan.n_slots, dtype=torch.float64, device="cuda")
for _ in range(3):
    plan.run(out.data_ptr())
torch.cuda.synchronize()
U = plan.info.n_wave_slots
st = np.zeros(4 * U, dtype=np.uint64)
fn = L.lib().ff_debug_read_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert fn(plan._h, st.ctypes.data) == 0
st = st[:2 * U].reshape(U, 2).astype(np.int64)
end = (st[:, 1] - st[:, 0].min()) / 100.0
wpw = U // plan.info.n_compute_units
print("N=%d shard %d/%d waves/WG %d items %d: wave end us min %.0f p10 %.0f p50 %.0f p90 %.0f max %.0f" % (
    n, rank, world, wpw, plan.info.n_items, end.min(), np.percentile(end, 10), np.median(end), np.percentile(end, 90), end.max()))
wg = end.reshape(-1, wpw).max(axis=1)
print("  per-WG end: p10 %.0f p50 %.0f p90 %.0f max %.0f; by XCD mean %s" % (
    np.percentile(wg, 10), np.median(wg), np.percentile(wg, 90), wg.max(),
    " ".join("%.0f" % wg[np.arange(len(wg)) % 8 == x].mean() for x in range(8))))
late = np.argsort(-wg)[:10]
print("  latest WGs:", " ".join("%d:%.0f" % (g, wg[g]) for g in late))
