"""Pair-space sharding over the GPUs of one node: one process per GPU
(torch.distributed, backend "nccl" = RCCL over xGMI), each rank reduces the pair
tiles of its contiguous row shard (ff_shard_rows), then ONE exchange step: every
rank sends its slice of the IterPairs-ordered output to the root (point-to-point
send/recv, the slices differ in length), which already holds its own slice in
place.  There is no other collective on the data path: pairs are independent
(frcfrc/unifrac.go:209-228 maps over pairs with no reduction across them).

Two transports for that exchange (ShardedRun, FF_GATHER=auto|ipc|nccl):
  * "ipc": the root's result array is mapped into every rank (HIP IPC memory handle through
    the C ABI: ff_device_alloc / ff_ipc_export / ff_ipc_open, so a C or Go host has the same
    transport; exchanged once) and each rank copies its finished slice straight into it over its
    xGMI link with a device-to-device hipMemcpyAsync on a side stream -- the copy
    engines move the slice while the rank's CUs already reduce the next batch (two
    local result buffers).  No RCCL kernel competes with the persistent pair kernel
    for CUs.  Checked once at set-up with a known pattern; any failure on any rank
    makes every rank fall back to
  * "nccl": batched send/recv (gather_slices below).

torch is plumbing here (device buffers, streams, the process group); the
reduction itself is ff_plan_run in the C ABI.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np

from . import api


def bounded_barrier(group=None, what: str = "a barrier", timeout_s: float = 120.0) -> None:
    """dist.barrier that gives up: a peer that died (or never arrives) makes this rank raise after
    timeout_s instead of hanging for the process group's own timeout (minutes to forever).  Used at
    the set-up and tear-down points of the "ipc" transport, where a peer may have failed in code of
    its own (mapping another process's memory); not on the data path."""
    import time

    import torch.distributed as dist

    work = dist.barrier(group=group, async_op=True)
    t0 = time.monotonic()
    while not work.is_completed():
        if time.monotonic() - t0 > timeout_s:
            raise RuntimeError("frackyfrac_amd: rank %d waited %.0f s at %s; a peer is gone or stuck -- giving up "
                               "(start the job again; FF_GATHER=nccl avoids the ipc set-up)"
                               % (dist.get_rank(group), timeout_s, what))
        time.sleep(0.002)
    work.wait()


def wait_requests(reqs, what: str, timeout_s: Optional[float] = None, group=None) -> None:
    """Waits for point-to-point requests, but not for ever: a peer that never posts its side (it died, or left the
    exchange through an exception of its own) makes this rank raise after timeout_s (FF_GATHER_TIMEOUT_S, 600 s)
    instead of sitting in the wait until the process group's own timeout.  The bound is this function's own clock on
    BOTH backends.  gloo: Work.wait(timeout) honours it (its send / receive requests complete only inside a wait, so
    they cannot be polled).  RCCL: Work.wait(timeout) only orders the current stream behind the transfer and returns at
    once, so the requests are POLLED -- Work.is_completed, a query of the transfer's event that never blocks -- against
    the deadline, and only a completed request is waited for."""
    import datetime
    import os
    import time

    import torch.distributed as dist

    if timeout_s is None:
        timeout_s = float(os.environ.get("FF_GATHER_TIMEOUT_S", "600"))
    polled = dist.get_backend(group) == "nccl"
    t0 = time.monotonic()
    pause = 0.0002
    for q in reqs:
        while polled and not q.is_completed():
            if time.monotonic() - t0 > timeout_s:
                raise RuntimeError("frackyfrac_amd: waited %.0f s for %s; a peer is gone or never entered the exchange"
                                   % (timeout_s, what))
            time.sleep(pause)
            pause = min(pause * 1.5, 0.005)
        left = max(0.05, timeout_s - (time.monotonic() - t0))
        try:
            ok = q.wait(datetime.timedelta(seconds=left))
        except RuntimeError as e:
            raise RuntimeError("frackyfrac_amd: waited %.0f s for %s; a peer is gone or never entered the exchange (%s)"
                               % (timeout_s, what, str(e).splitlines()[0][:200])) from e
        if ok is False:
            raise RuntimeError("frackyfrac_amd: waited %.0f s for %s; a peer is gone or never entered the exchange"
                               % (timeout_s, what))


def status_all(code: int, group=None, device=None) -> int:
    """Largest `code` over all ranks (one all_reduce that every rank always reaches): 0 = fine,
    1 = FIXED32's guarantee missed somewhere, 2 = some other error somewhere."""
    import torch
    import torch.distributed as dist

    dev = device if (device is not None and dist.get_backend(group) == "nccl") else torch.device("cpu")
    t = torch.tensor([int(code)], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


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
        wait_requests(dist.batch_isend_irecv(ops) if ops else [], "the peers' slices (root)", group=group)
        return full
    if local.numel() > 0:
        wait_requests(dist.batch_isend_irecv([dist.P2POp(dist.isend, local, root, group)]), "the root to take this rank's slice", group=group)
    return None


def allgather_flat_nodes(local: "api.FlatNodes", group=None) -> "api.FlatNodes":
    """Input replication: every rank has flattened ONLY its own contiguous block of
    samples (rank order = sample order) -- a G-th of stage A and of the upload each -- and one
    all-gather over RCCL/xGMI (padded to the largest block) gives every rank the flat nodes of all
    samples, from which it stages the matrix its pair tiles read.  Works on CPU groups (gloo) too."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if world == 1:
        return local
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    n, nnz = local.n_samples, int(local.indptr[-1])
    sizes = torch.zeros(world, 2, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, torch.tensor([[n, nnz]], dtype=torch.int64, device=dev), group=group)
    sizes = sizes.cpu().numpy()
    max_n, max_nnz = int(sizes[:, 0].max()), max(1, int(sizes[:, 1].max()))

    def gather(arr, dtype, width):
        buf = torch.zeros(width, dtype=dtype, device=dev)
        buf[:len(arr)] = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
        out = torch.empty(world * width, dtype=dtype, device=dev)
        dist.all_gather_into_tensor(out, buf, group=group)
        return out.cpu().numpy().reshape(world, width)

    counts = gather(np.diff(local.indptr), torch.int64, max(1, max_n))
    ids = gather(local.branch_id, torch.int32, max_nnz)
    abnd = gather(local.abnd, torch.float64, max_nnz)
    all_counts = np.concatenate([counts[r, :sizes[r, 0]] for r in range(world)])
    indptr = np.zeros(len(all_counts) + 1, dtype=np.int64)
    np.cumsum(all_counts, out=indptr[1:])
    return api.FlatNodes(indptr, np.concatenate([ids[r, :sizes[r, 1]] for r in range(world)]),
                         np.concatenate([abnd[r, :sizes[r, 1]] for r in range(world)]), local.branch_len)


def sample_block(n_samples: int, rank: int, world: int):
    """The contiguous block of samples rank `rank` generates / parses / flattens: [begin, end)."""
    return n_samples * rank // world, n_samples * (rank + 1) // world


def gather_slices_chunked(produce: Callable, n_samples: int, rank: int, world: int, chunks: int, root: int = 0,
                          full=None, group=None, copy_root: Optional[Callable] = None):
    """The gather with every rank's shard cut into `chunks` equal-pair sub-shards (sub-shard c of
    rank r is shard r * chunks + c of world * chunks: still contiguous in IterPairs order) that are
    produced back to back: `produce(c)` returns this rank's finished sub-shard c (a tensor, the work
    enqueued on the current stream) and its transfer to the root is issued at once, so that only
    the last sub-shard's transfer is exposed.  The root posts every receive up front, per peer
    in chunk order.  Returns `full` on the root (allocated if not given), None elsewhere."""
    import torch
    import torch.distributed as dist

    C, W = chunks, world
    reqs = []
    if rank == root:
        ops = []
        for c in range(C):
            for r in range(W):
                if r == root:
                    continue
                a, b = api.shard_slots(n_samples, r * C + c, W * C)
                if b > a:
                    if full is None:
                        raise ValueError("the root needs its result array before the first receive is posted")
                    ops.append(dist.P2POp(dist.irecv, full[a:b], r, group))
        reqs += dist.batch_isend_irecv(ops) if ops else []
    for c in range(C):
        part = produce(c)
        if rank == root:
            a, b = api.shard_slots(n_samples, rank * C + c, W * C)
            if copy_root is not None:
                copy_root(full[a:b], part)
            else:
                full[a:b].copy_(part, non_blocking=True)
        elif part.numel() > 0:
            # enqueued behind sub-shard c's kernels on the communication stream; sub-shard c + 1 is
            # launched right after and overlaps with the transfer
            reqs += dist.batch_isend_irecv([dist.P2POp(dist.isend, part, root, group)])
    wait_requests(reqs, "the sub-shards' transfers", group=group)
    return full if rank == root else None


class ShardedRun:
    """One rank's share of the hot path: staged plan(s) for its row shard plus the output
    buffers, reusable across steps (bench.py) or used once.

    chunks > 1 (default from FF_GATHER_CHUNKS, 1) cuts the rank's shard into that many
    equal-pair sub-shards (shard rank*chunks+c of world*chunks, still contiguous) that
    run back to back; each finished sub-shard is on its way to the root -- by the copy
    engines ("ipc") or by RCCL -- while the next one computes, so only the last
    sub-shard's transfer is exposed."""

    def __init__(self, nodes: api.FlatNodes, weighted: bool, rank: int, world: int,
                 precision="auto", device: Optional[int] = None, root: int = 0, group=None,
                 chunks: Optional[int] = None, transport: Optional[str] = None, ipc_timeout_s: float = 120.0):
        import os

        import torch

        self.torch = torch
        self.rank, self.world, self.root, self.group = rank, world, root, group
        self.ipc_timeout_s = float(ipc_timeout_s)
        self.gather_events = None  # (kernels done, slice delivered) of the last step, for exposed_gather_ms()
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
        if transport is None:
            transport = os.environ.get("FF_GATHER", "auto")
        if transport not in ("auto", "ipc", "nccl"):
            raise ValueError("transport must be auto, ipc or nccl")
        # The root's result array.  For the "ipc" transport it is an allocation of the C ABI's own
        # (ff_device_alloc: the start of a HIP allocation, which is what can be exported), viewed by
        # torch without a copy; otherwise a plain torch tensor.
        self.full, self._full_buf, self._full_note = None, None, ""
        if rank == root and world > 1:
            if transport in ("auto", "ipc"):
                try:
                    self._full_buf = api.DeviceBuffer(api.num_pairs(self.n_samples), device)
                    self.full = self._full_buf.tensor()
                except Exception as e:  # noqa: BLE001 -- the joint verdict of _setup_ipc turns this into "nccl"
                    self._full_note = "alloc: %r" % (e,)
                    if self._full_buf is not None:
                        self._full_buf.free()
                    self._full_buf = None
            if self.full is None:
                self.full = torch.empty(api.num_pairs(self.n_samples), dtype=torch.float64, device=self.device)
        self.transport = "none" if world == 1 else "nccl"
        self.transport_note = ""
        self.ipc_gbps = None
        self._k = 0
        if world > 1 and transport in ("auto", "ipc"):
            if self._setup_ipc():
                self.transport = "ipc"
            elif transport == "ipc":
                raise RuntimeError("frackyfrac_amd: FF_GATHER=ipc but the IPC mapping failed: " + self.transport_note)

    # ---- "ipc" transport -------------------------------------------------------------
    def _flag_all(self, ok: bool) -> bool:
        """True iff ok on every rank (one all_reduce; every rank always takes part)."""
        import torch.distributed as dist

        torch = self.torch
        dev = self.device if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(t.item()) == 1)

    def _setup_ipc(self) -> bool:
        """Maps the root's `full` into this process and proves the mapping with a pattern
        written through it.  The sequence of collectives is the same on every rank whatever
        fails locally, and the verdict is taken jointly."""
        import os

        import torch.distributed as dist

        torch = self.torch
        ok, note = True, ""
        box = [None]
        if self.rank == self.root:
            try:
                if self._full_buf is None:
                    raise RuntimeError(self._full_note or "no exportable buffer")
                self.full.fill_(float("nan"))
                torch.cuda.synchronize(self.device)
                box[0] = self._full_buf.export()  # 64 bytes: hipIpcGetMemHandle through the C ABI
            except Exception as e:  # noqa: BLE001 -- any failure means "use the other transport"
                ok, note = False, "export: %r" % (e,)
        dist.broadcast_object_list(box, src=self.root, group=self.group)
        self.remote = None
        a, b = api.shard_slots(self.n_samples, self.rank, self.world)
        if self.rank != self.root:
            try:
                if box[0] is None:
                    raise RuntimeError("root could not export its buffer")
                if os.environ.get("FF_GATHER_FAULT") == str(self.rank):  # fault injection for the tests
                    raise RuntimeError("injected fault")
                self.remote = api.MappedBuffer(box[0], self.device.index)  # ff_ipc_open
                if b > a:
                    k = min(16, b - a)
                    mark = torch.full((k,), float(self.rank + 1), dtype=torch.float64, device=self.device)
                    st = torch.cuda.current_stream(self.device).cuda_stream
                    api.device_copy_async(self.remote.ptr + 8 * a, mark.data_ptr(), 8 * k, st)
                    api.device_copy_async(self.remote.ptr + 8 * (b - k), mark.data_ptr(), 8 * k, st)
                torch.cuda.synchronize(self.device)
            except Exception as e:  # noqa: BLE001
                ok, note = False, "open/write: %r" % (e,)
        torch.cuda.synchronize(self.device)
        bounded_barrier(self.group, "the ipc set-up (peers mapping the root's result array)", self.ipc_timeout_s)
        if self.rank == self.root and ok:
            torch.cuda.synchronize(self.device)
            for r in range(self.world):
                if r == self.root:
                    continue
                ra, rb = api.shard_slots(self.n_samples, r, self.world)
                if rb > ra:
                    k = min(16, rb - ra)
                    got = torch.cat([self.full[ra:ra + k], self.full[rb - k:rb]]).cpu()
                    if not bool((got == float(r + 1)).all()):
                        ok, note = False, "pattern of rank %d did not arrive" % r
        all_ok = self._flag_all(ok)
        if not all_ok:
            self._drop_mapping()
            self.transport_note = note or "another rank failed"
            return False
        # bandwidth probe: all peers copy their whole slice at once, as in a step.  A mapping
        # that works but crawls (staged through the host, say) must not beat RCCL to the job.
        gbps = float("inf")
        # (what a copy of a step moves at once: the rank's first sub-shard -- its whole shard with chunks = 1 --,
        # which is what self.local holds)
        a, b = api.shard_slots(self.n_samples, self.rank * self.chunks, self.world * self.chunks)
        try:
            st = torch.cuda.current_stream(self.device).cuda_stream
            if self.rank != self.root and b > a:
                api.device_copy_async(self.remote.ptr + 8 * a, self.local.data_ptr(), 8 * (b - a), st)  # first touch
                torch.cuda.synchronize(self.device)
            bounded_barrier(self.group, "the ipc bandwidth probe", self.ipc_timeout_s)
            if self.rank != self.root and b > a:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                api.device_copy_async(self.remote.ptr + 8 * a, self.local.data_ptr(), 8 * (b - a), st)
                e1.record()
                torch.cuda.synchronize(self.device)
                gbps = 8.0 * (b - a) / (e0.elapsed_time(e1) * 1e-3) / 1e9
        except Exception as e:  # noqa: BLE001
            gbps, note = 0.0, "probe: %r" % (e,)
        self.ipc_gbps = gbps
        min_gbps = float(os.environ.get("FF_GATHER_MIN_GBPS", "15"))
        fast = gbps >= min_gbps or (b - a) * 8 < (4 << 20)    # (tiny slices only measure latency)
        if not self._flag_all(fast):
            self._drop_mapping()
            self.transport_note = note or ("ipc copies too slow (%.1f GB/s on this rank; FF_GATHER_MIN_GBPS=%g)"
                                           % (gbps, min_gbps))
            return False
        self.side = torch.cuda.Stream(device=self.device)
        if self.chunks == 1:  # two buffers taken in turn: step k + 1's kernels do not wait for step k's copy
            self.locals = [self.local, torch.empty_like(self.local)] if self.rank != self.root else [self.local]
        # (chunks > 1: a buffer per sub-shard; sub-shard c of step k + 1 waits for its own copy of step k, which went
        # out while the sub-shards behind it were still being reduced)
        self.copy_done = [None] * max(2, self.chunks)
        return True

    def _step_ipc(self, timed: bool):
        torch = self.torch
        main = torch.cuda.current_stream(self.device)
        C = self.chunks
        if self.rank == self.root:
            for c, p in enumerate(self.plans):  # the root's slice is produced in place
                a, b = api.shard_slots(self.n_samples, self.rank * C + c, self.world * C)
                if b > a:
                    p.run(self.full.data_ptr() + 8 * a, main.cuda_stream, timed=timed)
            return self.full
        for c, p in enumerate(self.plans):
            a, b = api.shard_slots(self.n_samples, self.rank * C + c, self.world * C)
            if b <= a:
                continue
            if C == 1:
                q = self._k & 1
                self._k += 1
            else:
                q = c
            if self.copy_done[q] is not None:
                main.wait_event(self.copy_done[q])  # the copy that last read this buffer
            p.run(self.locals[q].data_ptr(), main.cuda_stream, timed=timed)
            ready = torch.cuda.Event(enable_timing=True)
            ready.record(main)
            self.side.wait_event(ready)
            # over this rank's own xGMI link, by the copy engines, while its CUs reduce the next sub-shard / step
            api.device_copy_async(self.remote.ptr + 8 * a, self.locals[q].data_ptr(), 8 * (b - a), self.side.cuda_stream)
            done = torch.cuda.Event(enable_timing=True)
            done.record(self.side)
            self.copy_done[q] = done
            self.gather_events = (ready, done)
        return None

    def sync(self):
        """This rank's own streams are drained (no collective)."""
        self.torch.cuda.synchronize(self.device)

    def wait(self):
        """Every step issued so far is complete on the root: each rank drains its own
        streams (compute, copies, RCCL), then all meet."""
        import torch.distributed as dist

        self.torch.cuda.synchronize(self.device)
        if self.world > 1:
            dist.barrier(group=self.group)

    def check_precision(self):
        """After wait(): does every rank's last run keep FIXED32's promise (refinement queue not
        overflowed, audit sample within its bar -- Plan.check_precision)?  The verdict is taken
        jointly: if any rank's shard fails, FFError(FF_ERR_PRECISION) is raised on EVERY rank, so
        that all of them re-stage in EXACT64 together (unifrac_dists_sharded does)."""
        from ._lib import FF_ERR_PRECISION, FFError

        msg, other = "", None
        try:
            self.torch.cuda.synchronize(self.device)
            for p in self.plans:
                try:
                    p.check_precision()
                except FFError as e:
                    if e.code != FF_ERR_PRECISION:
                        raise
                    msg = str(e)
        except Exception as e:  # noqa: BLE001 -- whatever happened here, the collective below must be reached
            other = e
        worst = 2 if other is not None else (1 if msg else 0)
        if self.world > 1:
            worst = status_all(worst, self.group, self.device)
        if other is not None:
            raise other
        if worst == 2:
            raise RuntimeError("frackyfrac_amd: another rank failed while checking its results")
        if worst == 1:
            raise FFError(FF_ERR_PRECISION, msg or "another rank's shard missed the FIXED32 tolerance")

    def exposed_gather_ms(self) -> float:
        """After wait(): how long the last step's slice took from "this rank's kernels are done" to "its
        slice is in the root's array" -- the part of the gather no compute of THIS step hides (the next
        step's kernels may still overlap it).  0 on the root and for world 1."""
        if self.gather_events is None:
            return 0.0
        a, b = self.gather_events
        return float(a.elapsed_time(b))

    @property
    def n_slots(self) -> int:
        return sum(p.n_slots for p in self.plans)

    def step(self, timed: bool = False):
        """Reduce this rank's pair tiles, then gather to the root.  Returns the full
        result tensor on the root (the local one when world == 1), None elsewhere.
        Asynchronous like any stream work; wait() makes the root's tensor complete (with the
        "ipc" transport the peers' slices land from THEIR streams, so a synchronize on the
        root alone is not enough)."""
        torch = self.torch
        if self.transport == "ipc":
            return self._step_ipc(timed)
        stream = torch.cuda.current_stream(self.device)
        if self.chunks == 1:
            self.plan.run(self.local.data_ptr(), stream.cuda_stream, timed=timed)
            if self.world == 1:
                return self.local
            ready = torch.cuda.Event(enable_timing=True)
            ready.record(stream)
            res = gather_slices(self.local, self.n_samples, self.rank, self.world, self.root, self.full, self.group)
            done = torch.cuda.Event(enable_timing=True)
            done.record(stream)  # (the requests' wait() made this stream wait for the RCCL transfers)
            self.gather_events = (ready, done)
            return res
        def produce(c):
            self.plans[c].run(self.locals[c].data_ptr(), stream.cuda_stream, timed=timed)
            return self.locals[c]

        return gather_slices_chunked(produce, self.n_samples, self.rank, self.world, self.chunks, self.root, self.full,
                                     self.group)

    def compute_local(self, timed: bool = False):
        """The first half of a step with NO collective in it: this rank's kernels are enqueued (every sub-shard into
        its own buffer) and, with the "ipc" transport, the copy of its slice into the root's mapped array -- a DMA of
        its own, which no peer waits for.  A caller that must survive a failure on one rank (unifrac_dists_sharded)
        runs this, takes the joint status, and only then enters gather()."""
        torch = self.torch
        if self.transport == "ipc":
            return self._step_ipc(timed)
        stream = torch.cuda.current_stream(self.device)
        for p, buf in zip(self.plans, self.locals):
            p.run(buf.data_ptr(), stream.cuda_stream, timed=timed)
        return self.local if self.world == 1 else None

    def gather(self):
        """The second half: the point-to-point exchange of the "nccl" transport (every rank must enter it: a rank
        that stays out leaves the root waiting in its receive).  Nothing to do for "ipc" (the slices travel by
        themselves) and for one rank.  Returns what step() returns."""
        if self.world == 1:
            return self.local
        if self.transport == "ipc":
            return self.full if self.rank == self.root else None
        if self.chunks == 1:
            return gather_slices(self.local, self.n_samples, self.rank, self.world, self.root, self.full, self.group)
        return gather_slices_chunked(lambda c: self.locals[c], self.n_samples, self.rank, self.world, self.chunks,
                                     self.root, self.full, self.group)

    def timing_collect(self):
        ms, _, n = self.timing_collect_parts()
        return ms, n

    def timing_collect_parts(self):
        """(ms of the pair reduction, of which the rare rows' kernel, launches) summed over the sub-shard plans."""
        ms = rare = n = 0
        for p in self.plans:
            a, r, b = p.timing_collect_parts()
            ms += a
            rare += r
            n += b
        return ms, rare, n // max(1, len(self.plans))

    def _drop_mapping(self):
        if getattr(self, "remote", None) is not None:
            self.torch.cuda.synchronize(self.device)
            self.remote.close()  # ff_ipc_close
        self.remote = None

    def close(self, collective: bool = True):
        """Collective when the "ipc" transport is in use: every peer unmaps the root's array before
        the root frees it.  The meeting is bounded (ipc_timeout_s), and a caller that is unwinding
        from an error of its own passes collective=False: it must not wait for peers that may be
        waiting for it somewhere else."""
        if getattr(self, "plans", None) is None:
            return
        if self.transport == "ipc":
            self._drop_mapping()
            if collective:
                bounded_barrier(self.group, "the ipc tear-down (peers unmapping the root's result array)",
                                self.ipc_timeout_s)
        if self._full_buf is not None:
            self.torch.cuda.synchronize(self.device)
            self.full = None
            self._full_buf.free()
            self._full_buf = None
        for p in self.plans:
            p.close()
        self.plans = None


def run_jointly(run, world: int, group=None):
    """One pass of a sharded run in which EVERY rank reaches EVERY collective whatever fails locally:

        local work (compute_local + a device sync; no collective inside)   -> joint status
        the exchange (gather: point-to-point, entered by all or by none)   -> joint status
        the precision verdict (check_precision: joint by itself)

    A rank that fails raises its own exception AFTER the joint status that reports it; the others raise a
    RuntimeError naming the step.  (Until round 4 the exchange of the "nccl" transport sat inside the first try: a
    rank whose kernels failed skipped its send, the root waited in its receive for ever and the other ranks sat in a
    mismatched all_reduce.)  `run` is a ShardedRun, or anything with its compute_local / gather / check_precision /
    close / sync methods (the CPU tests drive this with a gloo group and a stand-in).  Returns gather()'s value."""
    def joint(failure, what):
        worst = status_all(2 if failure is not None else 0, group, getattr(run, "device", None)) if world > 1 else (2 if failure else 0)
        if worst:
            # (every rank has drained its device by now: no copy into the root's array is in flight, nobody needs to
            # wait for anybody before unmapping / freeing)
            run.close(collective=False)
            raise failure if failure is not None else RuntimeError("frackyfrac_amd: another rank failed in %s" % what)

    failure = None
    try:
        run.compute_local()
        run.sync()
    except Exception as e:  # noqa: BLE001
        failure = e
    joint(failure, "its kernels")
    res = None
    try:
        res = run.gather()
        run.sync()
    except Exception as e:  # noqa: BLE001
        failure = e
    joint(failure, "the gather")
    return res


def unifrac_dists_sharded(nodes: api.FlatNodes, weighted: bool, precision="auto", root: int = 0,
                          group=None, compute: Optional[Callable] = None) -> Optional[np.ndarray]:
    """unifracDists over every rank of the default process group; the root gets
    the complete IterPairs-ordered array, other ranks None.

    `compute(nodes, weighted, rank, world) -> 1-D float64 torch tensor` replaces the
    per-rank GPU reduction; it exists so the sharding/gather logic can be exercised
    on CPU process groups (gloo) in tests -- the product path leaves it None."""
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if compute is None:
        from ._lib import FF_ERR_PRECISION, FFError

        for attempt in (precision, "exact64"):
            run = ShardedRun(nodes, weighted, rank, world, precision=attempt, root=root, group=group)
            res = run_jointly(run, world, group)   # (closes the run and raises on every rank if any rank failed)
            try:
                # joint by itself: every rank drains its device inside a try, then ONE all_reduce of the verdicts --
                # which is also the point at which every peer's slice has landed in the root's array ("ipc")
                run.check_precision()
            except FFError as e:
                run.close()
                if e.code != FF_ERR_PRECISION or attempt == "exact64":
                    raise
                # a shard somewhere holds mostly replicates (or failed its audit): every rank
                # stages again in binary64, as ff_unifrac_dists does for one device
                continue
            except Exception:
                run.close(collective=False)
                raise
            out = res.cpu().numpy() if res is not None else None
            run.close()
            return out
    local = compute(nodes, weighted, rank, world)
    res = gather_slices(local, nodes.n_samples, rank, world, root, None, group)
    return res.numpy() if res is not None else None
