"""Pair-space sharding over the GPUs of one node: one process per GPU
(torch.distributed, backend "nccl" = RCCL over xGMI), each rank reduces the pair
tiles of its contiguous row shard (ff_shard_rows), then ONE exchange step: every
rank sends its slice of the IterPairs-ordered output to the root (point-to-point
send/recv, the slices differ in length), which already holds its own slice in
place.  There is no other collective on the data path: pairs are independent
(frcfrc/unifrac.go:209-228 maps over pairs with no reduction across them).

torch is plumbing here (device buffers, streams, the process group); the
reduction itself is ff_plan_run in the C ABI.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np

from . import api


def gather_slices(local, n_samples: int, rank: int, world: int, root: int = 0, full=None, group=None):
    """local: this rank's distances (torch tensor, shard_slots(n, rank, world) long).
    Returns on root the full [P] tensor (IterPairs order), on other ranks None."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return local
    if rank == root:
        P = api.num_pairs(n_samples)
        if full is None:
            full = torch.empty(P, dtype=local.dtype, device=local.device)
        a, b = api.shard_slots(n_samples, root, world)
        full[a:b].copy_(local)
        ops = []
        for r in range(world):
            if r == root:
                continue
            a, b = api.shard_slots(n_samples, r, world)
            if b > a:
                ops.append(dist.P2POp(dist.irecv, full[a:b], r, group))
        # one group call: the 7 incoming transfers run concurrently, one per xGMI link
        for q in (dist.batch_isend_irecv(ops) if ops else []):
            q.wait()
        return full
    if local.numel() > 0:
        for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, local, root, group)]):
            q.wait()
    return None


class ShardedRun:
    """One rank's share of the hot path: staged plan(s) for its row shard plus the output
    buffers, reusable across steps (bench.py) or used once.

    chunks > 1 (default from FF_GATHER_CHUNKS, 1) cuts the rank's shard into that many
    equal-pair sub-shards (shard rank*chunks+c of world*chunks, still contiguous) that
    run back to back; each finished chunk is handed to RCCL while the next one computes,
    so only the last chunk's transfer is exposed."""

    def __init__(self, nodes: api.FlatNodes, weighted: bool, rank: int, world: int,
                 precision="auto", device: Optional[int] = None, root: int = 0, group=None,
                 chunks: Optional[int] = None):
        import os

        import torch

        self.torch = torch
        self.rank, self.world, self.root, self.group = rank, world, root, group
        self.n_samples = nodes.n_samples
        if not torch.cuda.is_available():
            raise RuntimeError("frackyfrac_amd: no GPU visible; the engine has no CPU path")
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device)
        if chunks is None:
            chunks = int(os.environ.get("FF_GATHER_CHUNKS", "1"))
        self.chunks = max(1, chunks) if world > 1 else 1
        C = self.chunks
        self.plans = [api.Plan(nodes, weighted, precision=precision, device=device, rank=rank * C + c, world=world * C)
                      for c in range(C)]
        self.plan = self.plans[0]
        self.locals = [torch.empty(p.n_slots, dtype=torch.float64, device=self.device) for p in self.plans]
        self.local = self.locals[0]
        self.full = (torch.empty(api.num_pairs(self.n_samples), dtype=torch.float64, device=self.device)
                     if (rank == root and world > 1) else None)

    @property
    def n_slots(self) -> int:
        return sum(p.n_slots for p in self.plans)

    def step(self, timed: bool = False):
        """Reduce this rank's pair tiles, then gather to the root.  Returns the full
        result tensor on the root (the local one when world == 1), None elsewhere."""
        torch = self.torch
        stream = torch.cuda.current_stream(self.device)
        if self.chunks == 1:
            self.plan.run(self.local.data_ptr(), stream.cuda_stream, timed=timed)
            return gather_slices(self.local, self.n_samples, self.rank, self.world, self.root, self.full, self.group)
        import torch.distributed as dist

        C, W = self.chunks, self.world
        reqs = []
        if self.rank == self.root:  # post every receive up front, per peer in chunk order
            ops = []
            for c in range(C):
                for r in range(W):
                    if r == self.root:
                        continue
                    a, b = api.shard_slots(self.n_samples, r * C + c, W * C)
                    if b > a:
                        ops.append(dist.P2POp(dist.irecv, self.full[a:b], r, self.group))
            reqs += dist.batch_isend_irecv(ops) if ops else []
        for c in range(C):
            self.plans[c].run(self.locals[c].data_ptr(), stream.cuda_stream, timed=timed)
            if self.rank == self.root:
                a, b = api.shard_slots(self.n_samples, self.rank * C + c, W * C)
                self.full[a:b].copy_(self.locals[c], non_blocking=True)
            elif self.locals[c].numel() > 0:
                # enqueued behind chunk c's kernels on the communication stream; chunk c+1
                # is launched right after and overlaps with the transfer
                reqs += dist.batch_isend_irecv([dist.P2POp(dist.isend, self.locals[c], self.root, self.group)])
        for q in reqs:
            q.wait()
        return self.full if self.rank == self.root else None

    def timing_collect(self):
        ms = n = 0
        for p in self.plans:
            a, b = p.timing_collect()
            ms += a
            n += b
        return ms, n // max(1, len(self.plans))

    def close(self):
        for p in self.plans:
            p.close()


def unifrac_dists_sharded(nodes: api.FlatNodes, weighted: bool, precision="auto", root: int = 0,
                          group=None, compute: Optional[Callable] = None) -> Optional[np.ndarray]:
    """unifracDists over every rank of the default process group; the root gets
    the complete IterPairs-ordered array, other ranks None.

    `compute(nodes, weighted, rank, world) -> 1-D float64 torch tensor` replaces the
    per-rank GPU reduction; it exists so the sharding/gather logic can be exercised
    on CPU process groups (gloo) in tests -- the product path leaves it None."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if compute is None:
        run = ShardedRun(nodes, weighted, rank, world, precision=precision, root=root, group=group)
        res = run.step()
        torch.cuda.synchronize()
        out = res.cpu().numpy() if res is not None else None
        run.close()
        return out
    local = compute(nodes, weighted, rank, world)
    res = gather_slices(local, nodes.n_samples, rank, world, root, None, group)
    return res.numpy() if res is not None else None
