"""Diagnostics: what explains when a wave of pair_sad_kernel ends?  Per-wave end stamps of one launch (diagnostic
build, FF_STAMPS) against the wave's own schedule (ff_debug_schedule: rows of main-round items, of wide and narrow
stream-K remainder items, number of items), least squares; then the same per SIMD (waves w and w + 4 of a
workgroup share one).  Run on the GPU box: python tools/wave_cost_fit.py [C3|C4]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FF_LIB_PATH", os.path.join(ROOT, "frackyfrac_amd", "lib", "libfrackyfrac_amd_diag.so"))
os.environ["FF_STAMPS"] = "1"
import numpy as np, torch
import frackyfrac_amd as ff
from frackyfrac_amd import synth, _lib as L

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = dict(synth.CONFIGS[wl])
if len(sys.argv) > 2:
    cfg["n_samples"] = int(sys.argv[2])   # another shape on the same device: are the same workgroups late?
tree, ptr, idx, val = synth.make(cfg["n_samples"], cfg["n_leaves"], cfg["density"], cfg["seed"])
nodes = ff.flatten_leaf_csr(ff.parse_newick(tree.newick()), ptr, idx, val)
plan = ff.Plan(nodes, True, precision="fixed32")
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
os.environ["FF_WAVES_PER_WG"] = str(wpw)
sched = L.lib().ff_debug_schedule
sched.restype = ctypes.c_int64
sched.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                  ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
cap = 1 << 22
items = np.zeros((cap, 8), dtype=np.int32)
iptr = np.zeros(U + 1, dtype=np.int32)
nt = ctypes.c_int64()
n = cfg["n_samples"]
k = sched(0, n, plan.info.rows_padded, 0, n, plan.info.n_compute_units, 2, 1, items.ctypes.data, cap, iptr.ctypes.data, ctypes.byref(nt))
assert k == plan.info.n_items, (k, plan.info.n_items)
feat = np.zeros((U, 5))
for u in range(U):
    for i0, j0, k0, k1, flags, *_ in items[iptr[u]:iptr[u + 1]]:
        rows = k1 - k0
        if flags & 2:
            feat[u, 0] += rows                      # main-round rows (32 x 256)
        elif flags & 4:
            feat[u, 2] += rows                      # remainder, narrow tile (32 x 128)
        else:
            feat[u, 1] += rows                      # remainder, wide tile
        feat[u, 3] += 1                             # items
    feat[u, 4] = 1
names = ["main rows", "remainder rows (wide)", "remainder rows (narrow)", "items", "const"]
print("waves %d, items %d; end us: mean %.1f p50 %.1f max %.1f (kernel ends %.2f %% after its mean wave)" % (
    U, k, end.mean(), np.median(end), end.max(), 100 * (end.max() / end.mean() - 1)))
print("per-wave features: mean", feat.mean(axis=0), "min", feat.min(axis=0), "max", feat.max(axis=0))
coef, *_ = np.linalg.lstsq(feat, end, rcond=None)
res = end - feat @ coef
print("per wave  : " + ", ".join("%s %.5f" % (a, b) for a, b in zip(names, coef)) + " | residual rms %.1f us, max %.1f" % (res.std(), np.abs(res).max()))
# per SIMD: waves w and w + wpw/2... (a workgroup's wave q runs on SIMD q mod 4)
wg = end.reshape(-1, wpw)
fs = feat.reshape(-1, wpw, 5)
simd_end = np.stack([wg[:, q::4].max(axis=1) for q in range(4)], axis=1).ravel()
simd_feat = np.stack([fs[:, q::4, :].sum(axis=1) for q in range(4)], axis=1).reshape(-1, 5)
coef2, *_ = np.linalg.lstsq(simd_feat, simd_end, rcond=None)
res2 = simd_end - simd_feat @ coef2
print("per SIMD  : " + ", ".join("%s %.5f" % (a, b) for a, b in zip(names, coef2)) + " | residual rms %.1f us, max %.1f" % (res2.std(), np.abs(res2).max()))
print("SIMD end us: mean %.1f p50 %.1f max %.1f (kernel ends %.2f %% after its mean SIMD)" % (
    simd_end.mean(), np.median(simd_end), simd_end.max(), 100 * (simd_end.max() / simd_end.mean() - 1)))
# the same launch again: is a wave's deviation its own, run after run?
plan.run(out.data_ptr())
torch.cuda.synchronize()
st2 = np.zeros(4 * U, dtype=np.uint64)
assert fn(plan._h, st2.ctypes.data) == 0
st2 = st2[:2 * U].reshape(U, 2).astype(np.int64)
end2 = (st2[:, 1] - st2[:, 0].min()) / 100.0
print("run-to-run: correlation of per-wave end deviations %.3f; rms difference %.1f us" % (
    np.corrcoef(end - end.mean(), end2 - end2.mean())[0, 1], (end - end2).std()))
xcd = (np.arange(U) // wpw) % 8
print("by XCD: mean end " + " ".join("%.0f" % end[xcd == x].mean() for x in range(8)))
# which waves are last?  (workgroup, wave, features)
order = np.argsort(-end)[:12]
for u in order:
    print("  late wave %5d (wg %3d xcd %d wave %d): end %.1f  main %d wide %d narrow %d items %d" % (
        u, u // wpw, (u // wpw) % 8, u % wpw, end[u], feat[u, 0], feat[u, 1], feat[u, 2], feat[u, 3]))
order = np.argsort(end)[:6]
for u in order:
    print("  early wave %5d (wg %3d xcd %d wave %d): end %.1f  main %d wide %d narrow %d items %d" % (
        u, u // wpw, (u // wpw) % 8, u % wpw, end[u], feat[u, 0], feat[u, 1], feat[u, 2], feat[u, 3]))

wg_dev = (end.reshape(-1, wpw) - feat.reshape(-1, wpw, 5)[:, :, :4] @ coef[:4]).mean(axis=1)   # workgroup mean of (end - fitted work)
print("late workgroups (mean residual, us): " + " ".join("%d:%+.0f" % (g, wg_dev[g] - wg_dev.mean()) for g in np.argsort(-wg_dev)[:16]))
print("early workgroups                   : " + " ".join("%d:%+.0f" % (g, wg_dev[g] - wg_dev.mean()) for g in np.argsort(wg_dev)[:16]))
np.save(os.path.join(ROOT, "gpurun_out", "wg_dev_%s_%d.npy" % (wl, cfg["n_samples"])), wg_dev)
