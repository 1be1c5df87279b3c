"""Diagnostics: distribution of per-wave run times of pair_sad_kernel.  The stamps are only armed in the
diagnostic build of the library (make -C frackyfrac_amd/csrc diag; FF_STAMPS=1 is read by that build alone)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FF_LIB_PATH", os.path.join(ROOT, "frackyfrac_amd", "lib", "libfrackyfrac_amd_diag.so"))
os.environ["FF_STAMPS"] = "1"
import numpy as np, torch
import frackyfrac_amd as ff
from frackyfrac_amd import synth, _lib as L

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = synth.CONFIGS[wl]
tree, ptr, idx, val = synth.make(cfg["n_samples"], cfg["n_leaves"], cfg["density"], cfg["seed"])
nodes = ff.flatten_leaf_csr(ff.parse_newick(tree.newick()), ptr, idx, val)
plan = ff.Plan(nodes, cfg["weighted"], precision="fixed32")
out = torch.empty(plan.n_slots, dtype=torch.float64, device="cuda")
for _ in range(3):
    plan.run(out.data_ptr())
torch.cuda.synchronize()
U = plan.info.n_wave_slots
st = np.zeros(4 * U, dtype=np.uint64)
fn = L.lib().ff_debug_read_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert fn(plan._h, st.ctypes.data) == 0
clk = st[2 * U:].reshape(U, 2).astype(np.int64)
st = st[:2 * U].reshape(U, 2).astype(np.int64)
t0 = st[:, 0].min()
start = (st[:, 0] - t0) / 100.0   # us
end = (st[:, 1] - t0) / 100.0
print("waves", U, "start us: min %.1f max %.1f | end us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (
    start.min(), start.max(), end.min(), np.percentile(end, 10), np.percentile(end, 50), np.percentile(end, 90), end.max()))
dur = end - start
print("duration us: min %.1f p50 %.1f max %.1f" % (dur.min(), np.median(dur), dur.max()))
wg = end.reshape(-1, 8).max(axis=1)
print("per-WG end: min %.1f p50 %.1f max %.1f" % (wg.min(), np.median(wg), wg.max()))
xcd = np.arange(U // 8) % 8
for x in range(8):
    e = wg[xcd == x]
    print("  xcd-group %d: WG end min %.1f med %.1f max %.1f" % (x, e.min(), np.median(e), e.max()))
# by wave index within WG
for w in range(8):
    e = end.reshape(-1, 8)[:, w]
    print("  wave %d: end med %.1f" % (w, np.median(e)))
cyc = (clk[:, 1] - clk[:, 0]).astype(np.float64)
ghz = cyc / np.maximum(dur, 1e-9) / 1000.0
print("shader clock (s_memtime / s_memrealtime) per wave: min %.3f p50 %.3f max %.3f GHz; cycles p50 %.0f" % (
    ghz.min(), np.median(ghz), ghz.max(), np.median(cyc)))
wpw = U // plan.info.n_compute_units
by_xcd = [np.median(ghz[[w for w in range(U) if (w // wpw) % 8 == x]]) for x in range(8)]
print("by XCD (workgroup g on XCD g mod 8): " + " ".join("%.3f" % v for v in by_xcd))

