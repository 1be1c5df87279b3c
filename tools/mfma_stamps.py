"""Phase timeline of pair_common_mfma_kernel: every workgroup stamps item start / loop start /
loop end / accumulators written / item end with the 100 MHz real-time clock.  Diagnostic build only:

    make -C frackyfrac_amd/csrc diag
    FF_LIB_PATH=frackyfrac_amd/lib/libfrackyfrac_amd_diag.so python tools/mfma_stamps.py [C3|C2|NxL] [DIAG]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import frackyfrac_amd as ff  # noqa: E402
from frackyfrac_amd import _lib, synth  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
if len(sys.argv) > 2:  # an ablation of the loop (see tools/mfma_diag.py): 2, 4, 8, 6 or 14
    os.environ["FF_MFMA_DIAG"] = sys.argv[2]
if wl in synth.CONFIGS:
    c = synth.CONFIGS[wl]
    tree, ptr, idx, val = synth.make(c["n_samples"], c["n_leaves"], c["density"], c["seed"])
else:
    ns, nl = wl.split("x")
    tree, ptr, idx, val = synth.make(int(ns), int(nl), 0.1, 77)
if os.environ.get("FF_STAMPS_INEXACT"):  # lengths off the binary grid: three digits, two sweeps
    tree.branch_len = tree.branch_len * (1.0 + 1e-3 * np.random.default_rng(5).random(tree.branch_len.shape[0]))
nodes = ff.flatten_leaf_csr(ff.parse_newick(tree.newick()), ptr, idx, val)
plan = ff.Plan(nodes, False, precision="fixed32")
print("digits %d, items %d" % (plan.info.n_digits, plan.info.n_items))
out = torch.empty(plan.n_slots, dtype=torch.float64, device="cuda")
for _ in range(3):
    plan.run(out.data_ptr())
torch.cuda.synchronize()
G = 256
lib = _lib.lib()
fn = lib.ff_debug_mfma_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int64]
fn.restype = ctypes.c_int
assert fn(None, G) == 0
plan.timing_collect()
plan.run(out.data_ptr(), timed=True)
ms, n = plan.timing_collect()
torch.cuda.synchronize()
st = np.zeros((G, 4, 8), dtype=np.uint64)
assert fn(st.ctypes.data, G) == 0
st = st.astype(np.int64)
t0 = st[:, 0, 0][st[:, 0, 0] > 0].min()
us = lambda x: (x - t0) / 100.0
print("pass (events): %.1f us" % (ms / n * 1000))
first = st[:, 0, 0]
print("first item start: min %.1f  median %.1f  max %.1f us after the earliest" % (us(first.min()), us(np.median(first)), us(first.max())))
ends = st[:, :, 4].max(axis=1)
print("workgroup end:    min %.1f  median %.1f  max %.1f us" % (us(ends.min()), us(np.median(ends)), us(ends.max())))
names = ["prologue (table, first words, set 0)", "loop", "accumulators -> partial tile (or LDS tile)", "drain (or copy-out from LDS)"]
for k in range(4):
    have = st[:, k, 0] > 0
    if not have.any():
        break
    print("item %d (%d workgroups):" % (k, have.sum()))
    for ph in range(4):
        d = (st[have, k, ph + 1] - st[have, k, ph]) / 100.0
        print("   %-40s median %7.2f  max %7.2f us" % (names[ph], np.median(d), d.max()))
    cyc = (st[have, k, 6] - st[have, k, 5]).astype(np.float64)
    dt = (st[have, k, 2] - st[have, k, 1]) / 100.0
    print("   %-40s median %7.0f cycles = %.3f GHz" % ("loop, shader clock (s_memtime)", np.median(cyc), np.median(cyc / dt) / 1000.0))
    if k == 0 and have.all():  # by XCD (workgroup g runs on XCD g mod 8): is the spread a property of the place?
        dt_all = (st[:, 0, 2] - st[:, 0, 1]) / 100.0
        cyc_all = (st[:, 0, 6] - st[:, 0, 5]).astype(np.float64)
        print("   loop by XCD, median us:     " + " ".join("%7.1f" % np.median(dt_all[x::8]) for x in range(8)))
        print("   loop by XCD, median cycles: " + " ".join("%7.0f" % np.median(cyc_all[x::8]) for x in range(8)))
        print("   loop, all workgroups: us min %.1f max %.1f; cycles min %.0f max %.0f" % (
            dt_all.min(), dt_all.max(), cyc_all.min(), cyc_all.max()))
    if k:
        gap = (st[have, k, 0] - st[have, k - 1, 4]) / 100.0
        print("   %-40s median %7.2f us" % ("(gap after previous item)", np.median(gap)))
