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
    """One rank's share of the hot path: a staged plan for its row shard plus the
    output buffers, reusable across steps (bench.py) or used once."""

    def __init__(self, nodes: api.FlatNodes, weighted: bool, rank: int, world: int,
                 precision="auto", device: Optional[int] = None, root: int = 0, group=None):
        import torch

        self.torch = torch
        self.rank, self.world, self.root, self.group = rank, world, root, group
        self.n_samples = nodes.n_samples
        if not torch.cuda.is_available():
            raise RuntimeError("frackyfrac_amd: no GPU visible; the engine has no CPU path")
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device)
        self.plan = api.Plan(nodes, weighted, precision=precision, device=device, rank=rank, world=world)
        self.local = torch.empty(self.plan.n_slots, dtype=torch.float64, device=self.device)
        self.full = (torch.empty(api.num_pairs(self.n_samples), dtype=torch.float64, device=self.device)
                     if (rank == root and world > 1) else None)

    def step(self, timed: bool = False):
        """Reduce this rank's pair tiles, then gather to the root.  Returns the full
        result tensor on the root (the local one when world == 1), None elsewhere."""
        torch = self.torch
        stream = torch.cuda.current_stream(self.device)
        self.plan.run(self.local.data_ptr(), stream.cuda_stream, timed=timed)
        return gather_slices(self.local, self.n_samples, self.rank, self.world, self.root, self.full, self.group)

    def close(self):
        self.plan.close()


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
