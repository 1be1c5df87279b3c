"""Diagnostics (diagnostic build: make -C frackyfrac_amd/csrc diag): phase timeline of pair_common_small_kernel from
in-kernel stamps of the 100 MHz clock -- entry, operands in flight + digit table in LDS, branch sweep, partial tiles
in LDS, distances written -- per workgroup, for C2 (or N LEAVES)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FF_LIB_PATH", os.path.join(ROOT, "frackyfrac_amd", "lib", "libfrackyfrac_amd_diag.so"))
import numpy as np, torch
import frackyfrac_amd as ff
from frackyfrac_amd import synth, _lib as L

n, leaves = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 2000)
tree, ptr, idx, val = synth.make(n, leaves, 0.1, synth.CONFIGS["C2"]["seed"])
nodes = ff.flatten_leaf_csr(ff.parse_newick(tree.newick()), ptr, idx, val)
plan = ff.Plan(nodes, False, precision="fixed32")
assert plan.info.kernel == 4
G = plan.info.n_tiles
fn = L.lib().ff_debug_small_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int64]
assert fn(None, G) == 0
out = torch.empty(plan.n_slots, dtype=torch.float64, device="cuda")
for _ in range(20):
    plan.run(out.data_ptr())
torch.cuda.synchronize()
st = np.zeros((G, 8), dtype=np.uint64)
assert fn(st.ctypes.data, G) == 0
st = st.astype(np.int64)
t0 = st[:, 0].min()
names = ["entry", "operands requested, digits in LDS", "branch sweep done", "partial tiles in LDS", "distances written"]
print("pair_common_small_kernel, %d samples x %d leaves: %d workgroups; us after the first workgroup's entry" % (n, leaves, G))
for k, nm in enumerate(names):
    v = (st[:, k] - t0) / 100.0
    print("  %-36s min %6.2f  p50 %6.2f  max %6.2f" % (nm, v.min(), np.median(v), v.max()))
d = np.diff(st[:, :5], axis=1) / 100.0
print("  phase lengths p50 (us): " + "  ".join("%s %.2f" % (a, b) for a, b in zip(["load+table", "sweep", "to LDS", "finish"], np.median(d, axis=0))))
