"""Randomised check of the streaming entry points on a GPU box: ff_unifrac_dists_stream / _stream_csr against the
one-pass ff_unifrac_dists on random shapes, piece sizes, shards, metrics and precisions -- same bits, every slot once,
ascending -- and an early stop at a random piece.  Usage: python tests/fuzz_stream_gpu.py SEED CASES (a script, not
collected by pytest)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import frackyfrac_amd as ff
from frackyfrac_amd import api, synth

seed0, ncase = int(sys.argv[1]), int(sys.argv[2])
bad = 0
t0 = time.time()
for case in range(ncase):
    rng = np.random.default_rng(seed0 + case)
    n = int(rng.choice([1, 2, 3, 31, 33, 64, 100, 257, 300, 513, 700, 1500]))
    leaves = int(rng.choice([2, 7, 50, 333, 1000, 4000]))
    tree, ptr, idx, val = synth.make(n, leaves, float(rng.choice([0.05, 0.3, 1.0])), int(rng.integers(1, 1 << 30)))
    if rng.random() < 0.4:
        bl = rng.lognormal(-2.0, 1.5, len(tree.branch_len))
        bl[0] = 0.0
        tree.branch_len = bl
    if n > 70 and rng.random() < 0.3:  # replicates: FIXED32 sub-shards fall back to EXACT64 inside the stream
        m, k = int(rng.integers(20, 60)), int(ptr[1])
        ptr2 = np.concatenate([[0], np.cumsum([k] * m + list(np.diff(ptr)[m:]))]).astype(np.int64)
        idx = np.concatenate([np.tile(idx[:k], m), idx[ptr[m]:]])
        val = np.concatenate([np.tile(val[:k], m), val[ptr[m]:]])
        ptr = ptr2
    nodes = ff.flatten_leaf_csr(ff.parse_newick(tree.newick()), ptr, idx, val)
    weighted = bool(rng.random() < 0.5)
    prec = str(rng.choice(["auto", "fixed32", "exact64"]))
    world = int(rng.choice([1, 1, 2, 5]))
    rank = int(rng.integers(0, world))
    chunk = int(rng.choice([0, 1, 7, 100, 5000, 10 ** 6]))
    if chunk == 1 and n > 300:
        chunk = 100
    try:
        want = ff.unifrac_dists(nodes, weighted, precision=prec, rank=rank, world=world)
    except ff.FFError as e:
        if "FIXED32 not applicable" in str(e):
            continue
        raise
    lo, hi = api.shard_slots(n, rank, world)
    at, parts, ok = lo, [], True
    for slot0, d in api.unifrac_dists_stream(nodes, weighted, precision=prec, rank=rank, world=world,
                                             max_pairs_per_chunk=chunk, flat_args=bool(rng.random() < 0.5)):
        ok &= slot0 == at and (chunk == 0 or len(d) <= chunk)
        at += len(d)
        parts.append(d)
    got = np.concatenate(parts) if parts else np.zeros(0)
    ok &= at == hi and np.array_equal(got, want[lo:hi], equal_nan=True)
    # early stop at a random piece: the generator closes, the library returns, nothing hangs
    stop_after, seen = int(rng.integers(1, 4)), 0
    gen = api.unifrac_dists_stream(nodes, weighted, precision=prec, rank=rank, world=world, max_pairs_per_chunk=max(chunk, 50))
    for slot0, d in gen:
        seen += 1
        if seen == stop_after:
            gen.close()
            break
    if not ok:
        bad += 1
        print("CASE", seed0 + case, "n", n, "leaves", leaves, "weighted", weighted, prec, "shard", rank, world, "chunk", chunk,
              "MISMATCH", flush=True)
    if case % 25 == 0:
        print("case", case, "%.0fs" % (time.time() - t0), "bad", bad, flush=True)
print("done", ncase, "bad", bad)
