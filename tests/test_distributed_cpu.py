"""The N > 1 path on CPU: two gloo ranks shard the pair space (ff_shard_rows), each
computes its slice, the slices are gathered to rank 0 with point-to-point send/recv
(frackyfrac_amd/distributed.py) and must reproduce the single-process result.  The
per-rank reduction is injected (the oracle's merge walk) because there is no GPU
here; the sharding, slot arithmetic and gather are the product's."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_samples, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import frackyfrac_amd as ff
    from frackyfrac_amd import synth
    from frackyfrac_amd.distributed import unifrac_dists_sharded
    from oracle import oracle as O

    dist.init_process_group("gloo", rank=rank, world_size=world)
    tree, ptr, idx, val = synth.make(n_samples, 60, 0.2, 99)
    T = ff.parse_newick(tree.newick())
    nodes = ff.flatten_leaf_csr(T, ptr, idx, val)
    onodes = np.zeros(len(nodes.branch_id), dtype=O.FLATNODE)
    onodes["id"], onodes["abnd"] = nodes.branch_id, nodes.abnd

    def compute(nd, weighted, r, w):
        a, b = ff.shard_slots(nd.n_samples, r, w)
        return torch.from_numpy(O.unifrac_dists(nd.indptr, onodes, nd.branch_len, weighted, 1, a, b).copy())

    res = unifrac_dists_sharded(nodes, True, compute=compute)
    if rank == 0:
        want = O.unifrac_dists(nodes.indptr, onodes, nodes.branch_len, True)
        q.put(bool(np.array_equal(res, want)) and len(res) == n_samples * (n_samples - 1) // 2)
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_samples", [70, 5])
def test_two_rank_gather_gloo(n_samples):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_samples, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _worker_chunked(rank, world, port, n_samples, chunks, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import frackyfrac_amd as ff
    from frackyfrac_amd import synth
    from frackyfrac_amd.distributed import gather_slices_chunked
    from oracle import oracle as O

    dist.init_process_group("gloo", rank=rank, world_size=world)
    tree, ptr, idx, val = synth.make(n_samples, 60, 0.2, 99)
    nodes = ff.flatten_leaf_csr(ff.parse_newick(tree.newick()), ptr, idx, val)
    onodes = np.zeros(len(nodes.branch_id), dtype=O.FLATNODE)
    onodes["id"], onodes["abnd"] = nodes.branch_id, nodes.abnd
    produced = []

    def produce(c):  # sub-shard c of this rank = shard rank * chunks + c of world * chunks
        a, b = ff.shard_slots(n_samples, rank * chunks + c, world * chunks)
        produced.append((a, b))
        return torch.from_numpy(O.unifrac_dists(nodes.indptr, onodes, nodes.branch_len, True, 1, a, b).copy())

    full = torch.full((ff.num_pairs(n_samples),), float("nan"), dtype=torch.float64) if rank == 0 else None
    res = gather_slices_chunked(produce, n_samples, rank, world, chunks, 0, full)
    # a rank's sub-shards tile its shard, in order
    a0, b0 = ff.shard_slots(n_samples, rank, world)
    assert produced[0][0] == a0 and produced[-1][1] == b0 and all(x[1] == y[0] for x, y in zip(produced, produced[1:]))
    if rank == 0:
        want = O.unifrac_dists(nodes.indptr, onodes, nodes.branch_len, True)
        q.put(bool(np.array_equal(res.numpy(), want)))
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_samples,chunks", [(2, 130, 3), (3, 70, 2), (2, 9, 4)])
def test_chunked_gather_gloo(world, n_samples, chunks):
    """FF_GATHER=nccl with FF_GATHER_CHUNKS > 1: the receives are posted up front, each rank's
    sub-shards go out as they are produced; the root's array must be the single-process result."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_chunked, args=(r, world, port, n_samples, chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _worker_allgather(rank, world, port, n_samples, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    import frackyfrac_amd as ff
    from frackyfrac_amd import synth
    from frackyfrac_amd.distributed import allgather_flat_nodes, sample_block

    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = sample_block(n_samples, rank, world)
    tree, ptr, idx, val = synth.make(n_samples, 80, 0.15, 424, b, e)   # this rank's samples only
    T = ff.parse_newick(tree.newick())
    mine = ff.flatten_leaf_csr(T, ptr, idx, val)
    assert mine.n_samples == e - b
    got = allgather_flat_nodes(mine)
    tree, ptr, idx, val = synth.make(n_samples, 80, 0.15, 424)
    want = ff.flatten_leaf_csr(T, ptr, idx, val)
    ok = (np.array_equal(got.indptr, want.indptr) and np.array_equal(got.branch_id, want.branch_id)
          and np.array_equal(got.abnd, want.abnd) and np.array_equal(got.branch_len, want.branch_len))
    q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_samples", [(2, 37), (3, 100), (3, 2)])
def test_flat_nodes_of_sample_blocks_allgather_to_the_whole_table(world, n_samples):
    """Input replication: every rank generates and flattens only its block of samples; after one
    all-gather every rank holds exactly the flat nodes a single process computes for all of them."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_allgather, args=(r, world, port, n_samples, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(q.get(timeout=5) for _ in range(world))


def _worker_bounded(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import time

    import torch.distributed as dist

    from frackyfrac_amd.distributed import bounded_barrier, status_all

    dist.init_process_group("gloo", rank=rank, world_size=world)
    # every rank reaches the joint status whatever happened to it locally; the worst code wins everywhere
    q.put(("status", rank, status_all(2 if rank == 1 else 0)))
    q.put(("status", rank, status_all(1 if rank == 0 else 0)))
    q.put(("status", rank, status_all(0)))
    bounded_barrier(None, "a meeting everybody attends", 30.0)
    if rank == 1:
        q.close()
        q.join_thread()
        time.sleep(6.0)     # a peer that is stuck somewhere else: never arrives in time
        os._exit(0)
    t0 = time.monotonic()
    try:
        bounded_barrier(None, "the ipc set-up", 1.5)
        q.put(("barrier", rank, "returned"))
    except RuntimeError as e:
        q.put(("barrier", rank, "%.1f %s" % (time.monotonic() - t0, e)))
    q.close()
    q.join_thread()
    os._exit(0)  # (the group is broken by design: no orderly shutdown)


def test_joint_status_and_bounded_barrier_gloo():
    """The pieces that keep a failing rank from hanging the others (ShardedRun.check_precision,
    unifrac_dists_sharded, the ipc set-up / tear-down): status_all is reached by every rank and returns the worst
    code everywhere; bounded_barrier raises after its timeout when a peer does not come, instead of hanging."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bounded, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    got = []
    while not q.empty():
        got.append(q.get(timeout=5))
    status = sorted(x[1:] for x in got if x[0] == "status")
    assert status == [(0, 0), (0, 1), (0, 2), (1, 0), (1, 1), (1, 2)]
    (b,) = [x for x in got if x[0] == "barrier"]
    assert b[1] == 0 and " s at the ipc set-up; a peer is gone or stuck" in b[2], b
    assert float(b[2].split()[0]) < 5.0


def _worker_failing(rank, world, port, fail_at, q):
    """run_jointly with a stand-in for ShardedRun over gloo: the "nccl"-style point-to-point gather is real, the
    kernels are a constant fill.  Rank 1 fails where `fail_at` says."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["FF_GATHER_TIMEOUT_S"] = "20"
    import torch
    import torch.distributed as dist

    import frackyfrac_amd as ff
    from frackyfrac_amd.distributed import gather_slices, run_jointly

    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 90

    class Run:
        device = None
        closed = None

        def compute_local(self):
            if rank == 1 and fail_at == "compute":
                raise ValueError("boom in the kernels of rank 1")
            a, b = ff.shard_slots(n, rank, world)
            self.local = torch.full((b - a,), float(rank + 1), dtype=torch.float64)

        def sync(self):
            pass

        def gather(self):
            if rank == 1 and fail_at == "gather":
                raise ValueError("boom in the gather of rank 1")   # (it never posts its send)
            return gather_slices(self.local, n, rank, world, 0, None, None)

        def close(self, collective=True):
            self.closed = collective

    run = Run()
    try:
        res = run_jointly(run, world)
        if rank == 0:
            want = np.concatenate([np.full(ff.shard_slots(n, r, world)[1] - ff.shard_slots(n, r, world)[0], r + 1.0)
                                   for r in range(world)])
            q.put((rank, "ok", bool(np.array_equal(res.numpy(), want)), run.closed))
        else:
            q.put((rank, "ok", res is None, run.closed))
    except Exception as e:  # noqa: BLE001
        q.put((rank, type(e).__name__, str(e), run.closed))
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_at", ["none", "compute", "gather"])
def test_a_rank_that_fails_takes_every_rank_out_of_the_exchange_instead_of_hanging_it(fail_at):
    """with the point-to-point transport a rank whose kernels failed skipped its send, the
    root waited in its receive for ever and the others sat in a mismatched all_reduce.  Now the local work comes first,
    then a joint status, and the exchange is entered by all ranks or by none: the failing rank raises its own
    exception, the others a RuntimeError that says where, all within seconds, and everybody closes without the
    collective tear-down.  A rank that fails INSIDE the exchange cannot be waited out jointly; there the peers'
    bounded request waits (FF_GATHER_TIMEOUT_S) end the wait."""
    import torch.multiprocessing as mp

    world, port = 3, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_failing, args=(r, world, port, fail_at, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, kind, msg, closed = q.get(timeout=120)
        got[r] = (kind, msg, closed)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if fail_at == "none":
        assert all(got[r][0] == "ok" and got[r][1] is True and got[r][2] is None for r in range(world)), got
    elif fail_at == "compute":
        assert got[1][0] == "ValueError" and "boom in the kernels" in got[1][1]
        for r in (0, 2):
            assert got[r][0] == "RuntimeError" and "another rank failed in its kernels" in got[r][1], got
        assert all(got[r][2] is False for r in range(world))          # closed without the collective tear-down
    else:
        assert got[1][0] == "ValueError" and "boom in the gather" in got[1][1]
        # (the root either times out in its bounded receive or learns of the failure from the joint status)
        for r in (0, 2):
            assert got[r][0] == "RuntimeError" and ("waited" in got[r][1] or "another rank failed in the gather" in got[r][1]), got
