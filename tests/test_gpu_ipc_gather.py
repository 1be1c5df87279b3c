"""The "ipc" gather transport (frackyfrac_amd/distributed.py) with real device memory:
several processes on the ONE GPU of the box (gloo as the control plane -- RCCL refuses two
ranks on one device), each reducing its row shard and copying the slice into the root's
IPC-mapped result array.  The root's array must equal the single-process result bit for
bit, for several pipelined steps."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_samples, transport, q, chunks=None):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    import frackyfrac_amd as ff
    from frackyfrac_amd import synth
    from frackyfrac_amd.distributed import ShardedRun

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tree, ptr, idx, val = synth.make(n_samples, 300, 0.15, 1234)
    T = ff.parse_newick(tree.newick())
    nodes = ff.flatten_leaf_csr(T, ptr, idx, val)
    run = ShardedRun(nodes, True, rank, world, precision="fixed32", device=0,
                     transport=None if transport == "env" else transport, chunks=chunks)
    if transport in ("auto", "env"):  # fault injected on one rank / switches from the environment: what did every rank take?
        a, b = ff.shard_slots(n_samples, rank, world)
        q.put((rank, run.transport, run.transport_note, run.chunks, run.n_slots == b - a))
        dist.barrier()
        run.close()
        dist.destroy_process_group()
        return
    res = None
    for _ in range(5):  # pipelined: both local buffers of a peer get reused
        res = run.step()
    run.wait()
    if rank == 0:
        want = ff.unifrac_dists(nodes, True, precision="fixed32", device=0)
        got = res.cpu().numpy()
        q.put((run.transport, bool(np.array_equal(got, want)), run.transport_note))
    else:
        assert res is None
    dist.barrier()
    run.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_samples,chunks", [(2, 300, None), (3, 97, None), (2, 700, 3), (3, 400, 2)])
def test_ipc_gather_on_one_device(world, n_samples, chunks):
    """chunks: every rank's shard in that many sub-shards, each copied into the root's array while the next one is
    reduced (round 4: the ipc transport's own pipelining; until then only the RCCL path had it)."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_samples, "ipc", q, chunks)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    transport, equal, note = q.get(timeout=5)
    assert transport == "ipc", note
    assert equal


def test_ipc_failure_on_one_rank_makes_every_rank_fall_back(monkeypatch):
    import torch.multiprocessing as mp

    monkeypatch.setenv("FF_GATHER_FAULT", "1")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 3, port, 64, "auto", q)) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5)[:2] for _ in range(3))
    assert got == [(0, "nccl"), (1, "nccl"), (2, "nccl")]


@pytest.mark.parametrize("env,want_transport,want_chunks,note", [
    ({"FF_GATHER": "ipc"}, "ipc", 1, ""),
    ({"FF_GATHER": "nccl"}, "nccl", 1, ""),
    ({"FF_GATHER_MIN_GBPS": "1e9"}, "nccl", 1, "too slow"),      # a mapping that crawls must not beat RCCL to the job
    ({"FF_GATHER_CHUNKS": "3"}, "ipc", 3, ""),                    # sub-shards pipeline either transport
    ({"FF_GATHER_CHUNKS": "3", "FF_GATHER": "nccl"}, "nccl", 3, ""),
])
def test_gather_switches_from_the_environment(monkeypatch, env, want_transport, want_chunks, note):
    """FF_GATHER / FF_GATHER_MIN_GBPS / FF_GATHER_CHUNKS as a launcher would set them: every rank takes the same
    transport, says why ipc was not used, and its sub-shard plans tile its shard.  (2,000 samples: slices of 8 MB,
    large enough for the bandwidth probe to count.)"""
    import torch.multiprocessing as mp

    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 2000, "env", q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    for rank, transport, why, chunks, tiles in got:
        assert transport == want_transport and chunks == want_chunks and tiles
        assert note in why
