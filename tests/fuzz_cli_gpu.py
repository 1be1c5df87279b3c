"""Randomised end-to-end sweep of the frcfrc executable on a GPU box: random small trees and
tables as text (dense and sparse loaders), flags (-w, -l, -p, -gpus, passes), AUTO precision
(binary64 at these sizes) -- stdout must be byte for byte what the oracle prints.
Usage: python tests/fuzz_cli_gpu.py SEED CASES   (a script, not collected by pytest)"""
import os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frackyfrac_amd import synth, _lib as L
from oracle import oracle as O

seed0, ncase = int(sys.argv[1]), int(sys.argv[2])
d = tempfile.mkdtemp(prefix="ffz", dir=os.environ.get("TMPDIR", "/tmp"))
bad = 0
for case in range(ncase):
    rng = np.random.default_rng(seed0 + case)
    n = int(rng.choice([1, 2, 3, 5, 17, 40, 70]))
    leaves = int(rng.choice([2, 3, 9, 40, 200]))
    tree, ptr, idx, val = synth.make(n, leaves, float(rng.choice([0.1, 0.5, 1.0])), int(rng.integers(1, 1 << 30)))
    if rng.random() < 0.5:
        bl = np.round(rng.lognormal(-1.0, 1.0, len(tree.branch_len)), int(rng.integers(1, 8)))
        bl[0] = 0.0
        tree.branch_len = bl
    sparse = bool(rng.random() < 0.5)
    weighted = bool(rng.random() < 0.6)
    unnorm = weighted and bool(rng.random() < 0.3)
    text = synth.sparse_text(tree, ptr, idx, val) if sparse else synth.dense_text(tree, ptr, idx, val)
    nwk = tree.newick()
    open(d + "/t.tree", "w").write(nwk)
    open(d + "/t.tab", "w").write(text)
    otree = O.parse_newick(nwk)
    abnd = O.parse_sparse_abundance(text) if sparse else O.parse_abundance(text)
    lsorted = unnorm and bool(rng.random() < 0.5)     # -l alone is the reference's unsorted-list walk; -l-sorted sorts
    want = O.format_output(O.unifrac(abnd, otree, weighted, nnorm=unnorm, reference_l_quirk=not lsorted))
    args = [L.FRCFRC_PATH, "-t", d + "/t.tree", "-i", d + "/t.tab"]
    if rng.random() < 0.7: args += ["-p", str(int(rng.choice([1, 3])))]   # (else: the CPU quota)
    if sparse: args.append("-s")
    if weighted: args.append("-w")
    if unnorm: args.append("-l")
    if lsorted: args.append("-l-sorted")
    if rng.random() < 0.3: args += ["-gpus", "2"]
    env = dict(os.environ)
    if rng.random() < 0.3: env["FF_CLI_MAX_PAIRS"] = "50"
    # where the text goes: stdout (a pipe), a file, a gzip file -- the device formatter's pipeline behind each
    sink = str(rng.choice(["stdout", "file", "gz"]))
    out_path = d + ("/out.txt.gz" if sink == "gz" else "/out.txt")
    if sink != "stdout": args += ["-o", out_path]
    r = subprocess.run(args, capture_output=True, text=True, env=env)
    got = r.stdout
    if r.returncode == 0 and sink == "file": got = open(out_path).read()
    if r.returncode == 0 and sink == "gz":
        import gzip
        got = gzip.open(out_path, "rt").read()
    if r.returncode != 0 or got != want:
        bad += 1
        print("CASE", seed0 + case, "n", n, "leaves", leaves, "sparse", sparse, "weighted", weighted, "unnorm", unnorm, "rc", r.returncode, r.stderr[-200:], flush=True)
    if case % 50 == 0: print("case", case, "bad", bad, flush=True)
print("done", ncase, "bad", bad)
