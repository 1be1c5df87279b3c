"""End-to-end timing of the frcfrc executable on a synthetic table (sparse text in,
text out).  Usage: cli_scale.py SAMPLES LEAVES DENSITY [dense]"""
import os, subprocess, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from frackyfrac_amd import synth, _lib as L

ns, nl, dens = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
dense = len(sys.argv) > 4 and sys.argv[4] == "dense"
t0 = time.time()
tree, ptr, idx, val = synth.make(ns, nl, dens, 123)
d = tempfile.mkdtemp(prefix="ffcli", dir=os.environ.get("TMPDIR", "/tmp"))
open(d + "/t.tree", "w").write(tree.newick())
text = synth.dense_text(tree, ptr, idx, val) if dense else synth.sparse_text(tree, ptr, idx, val)
open(d + "/t.tab", "w").write(text)
print("generated %d samples, table %.1f MB in %.1fs" % (ns, len(text) / 1e6, time.time() - t0), flush=True)
sums = set()
for p in (None, 1, 16, None):
    out = d + "/out_%s.txt" % p
    if os.path.exists(out):
        os.unlink(out)
    t0 = time.time()
    args = [L.FRCFRC_PATH, "-w", "-t", d + "/t.tree", "-i", d + "/t.tab", "-o", out, "-stats"]
    if p:
        args += ["-p", str(p)]
    if not dense:
        args.insert(1, "-s")
    r = subprocess.run(args, capture_output=True, text=True)
    dt = time.time() - t0
    print("-p %s: rc=%d wall %.2fs; output %.1f MB" % (p, r.returncode, dt, os.path.getsize(out) / 1e6))
    print("   " + r.stderr.strip().replace("\n", " | ")[-700:], flush=True)
    sums.add(subprocess.run(["md5sum", out], capture_output=True, text=True).stdout.split()[0])
    os.unlink(out)
print("outputs identical:", len(sums) == 1)
