"""Host-side mirror of the reference's interface for the UniFrac path.

Names follow frackyfrac: `unifrac` (frcfrc/unifrac.go:97), `unifrac_dists`
(unifracDists, unifrac.go:209), `parse_abundance` / `parse_sparse_abundance`
(parser/parser.go:21,85), `validate_species` (unifrac.go:80), `iter_pairs`
(common.IterPairs, common/common.go:21).  Everything here is a thin ctypes layer
over the C ABI (include/frackyfrac_amd.h); the arithmetic happens in the HIP
kernels of frackyfrac_amd/csrc/ff_dev_run.hip (C ABI: ff_device.hip).
"""
from __future__ import annotations

import ctypes
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

from . import _lib as L
from ._lib import FFError, ff_options, ff_plan_info, ff_problem


def _opts(weighted: bool, precision="auto", device: int = -1, rank: int = 0, world: int = 1,
          unsorted_walk: bool = False) -> ff_options:
    o = ff_options()
    L.lib().ff_options_default(ctypes.byref(o))
    o.weighted = 1 if weighted else 0
    o.precision = L.PRECISION_NAMES[precision] if isinstance(precision, str) else int(precision)
    o.device = device
    o.rank = rank
    o.world = world
    o.flags = L.FLAG_UNSORTED_WALK if unsorted_walk else 0
    return o


def _l_mode(leave_unnormalized) -> int:
    """The -l argument of the entry points that flatten: False / True, or "reference" for what the reference really
    does under -l -- lists neither divided nor SORTED (unifrac.go:57-59,108-110; FF_L_REFERENCE)."""
    if leave_unnormalized == "reference":
        return L.L_REFERENCE
    return 1 if leave_unnormalized else 0


def num_pairs(n: int) -> int:
    return int(L.lib().ff_num_pairs(n))


def iter_pairs(n: int) -> Iterator[Tuple[int, int]]:
    """common.IterPairs as index pairs: (i, j) for i in 0..n-1, j in 0..i-1."""
    for i in range(n):
        for j in range(i):
            yield i, j


def shard_rows(n: int, rank: int, world: int) -> Tuple[int, int]:
    rb, re = ctypes.c_int64(), ctypes.c_int64()
    rc = L.lib().ff_shard_rows(n, rank, world, ctypes.byref(rb), ctypes.byref(re))
    if rc:
        raise FFError(rc, "bad shard %d of %d" % (rank, world))
    return rb.value, re.value


def shard_slots(n: int, rank: int, world: int) -> Tuple[int, int]:
    rb, re = shard_rows(n, rank, world)
    return rb * (rb - 1) // 2 if rb else 0, re * (re - 1) // 2 if re else 0


def _release(obj, free_name: str) -> None:
    """Frees obj._h once.  At interpreter shutdown the module globals may already be gone;
    the process is ending, so the handle is simply dropped then."""
    h = getattr(obj, "_h", None)
    if not h:
        return
    obj._h = None
    lib = getattr(L, "lib", None) if L is not None else None
    if lib is not None:
        getattr(lib(), free_name)(h)


class Tree:
    """newick.Node tree flattened in enumerateNodes' pre-order numbering."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def parse(cls, text: str) -> "Tree":
        data = text.encode("utf-8")
        h, err = ctypes.c_void_p(), L.errbuf()
        L.check(L.lib().ff_tree_parse(data, len(data), ctypes.byref(h), err, L.ERRLEN), err)
        return cls(h)

    @classmethod
    def read_file(cls, path: str) -> "Tree":
        h, err = ctypes.c_void_p(), L.errbuf()
        L.check(L.lib().ff_tree_read_file(path.encode(), ctypes.byref(h), err, L.ERRLEN), err)
        return cls(h)

    def __del__(self):
        _release(self, "ff_tree_free")

    @property
    def n(self) -> int:
        return int(L.lib().ff_tree_num_nodes(self._h))

    def _arr(self, fn, dtype) -> np.ndarray:
        n = self.n
        if n == 0:
            return np.zeros(0, dtype=dtype)
        ptr = fn(self._h)
        return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(np.ctypeslib.as_ctypes_type(dtype))),
                                     shape=(n,)).copy()

    @property
    def branch_len(self) -> np.ndarray:
        return self._arr(L.lib().ff_tree_branch_len, np.float64)

    @property
    def parent(self) -> np.ndarray:
        return self._arr(L.lib().ff_tree_parent, np.int64)

    @property
    def subtree_size(self) -> np.ndarray:
        return self._arr(L.lib().ff_tree_subtree_size, np.int64)

    @property
    def names(self) -> List[str]:
        f = L.lib().ff_tree_name
        return [f(self._h, k).decode("utf-8") for k in range(self.n)]


class Table:
    """[]map[string]float64 as the loaders return it."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def _parse(cls, fn, text: str) -> "Table":
        data = text.encode("utf-8")
        h, err = ctypes.c_void_p(), L.errbuf()
        L.check(fn(data, len(data), ctypes.byref(h), err, L.ERRLEN), err)
        return cls(h)

    def __del__(self):
        _release(self, "ff_table_free")

    def __len__(self) -> int:
        return int(L.lib().ff_table_num_samples(self._h))

    def to_maps(self) -> List[Dict[str, float]]:
        out = []
        lib = L.lib()
        name, val = ctypes.c_char_p(), ctypes.c_double()
        for s in range(len(self)):
            m = {}
            for k in range(lib.ff_table_sample_size(self._h, s)):
                lib.ff_table_sample_entry(self._h, s, k, ctypes.byref(name), ctypes.byref(val))
                m[name.value.decode("utf-8")] = val.value
            out.append(m)
        return out


def _parse_mt(text: str, sparse: bool, ngoroutines: int) -> Table:
    data = text.encode("utf-8")
    h, err = ctypes.c_void_p(), L.errbuf()
    L.check(L.lib().ff_table_parse_mt(data, len(data), 1 if sparse else 0, ngoroutines, ctypes.byref(h), err, L.ERRLEN), err)
    return Table(h)


def parse_abundance(text: str, ngoroutines: int = 1) -> Table:
    """parser.ParseAbundance (parser/parser.go:21)."""
    return _parse_mt(text, False, ngoroutines)


def parse_sparse_abundance(text: str, ngoroutines: int = 1) -> Table:
    """parser.ParseSparseAbundance (parser/parser.go:85)."""
    return _parse_mt(text, True, ngoroutines)


def parse_newick(text: str) -> Tree:
    return Tree.parse(text)


def validate_species(table: Table, tree: Tree) -> None:
    """validateSpecies (frcfrc/unifrac.go:80-93); raises FFError with its message."""
    err = L.errbuf()
    L.check(L.lib().ff_validate_species(table._h, tree._h, err, L.ERRLEN), err)


class FlatNodes:
    """Stage-A output: the [][]flatNode of unifrac.go:99-116 as CSR, plus treeDists."""

    def __init__(self, indptr, branch_id, abnd, branch_len):
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        self.branch_id = np.ascontiguousarray(branch_id, dtype=np.int32)
        self.abnd = np.ascontiguousarray(abnd, dtype=np.float64)
        self.branch_len = np.ascontiguousarray(branch_len, dtype=np.float64)

    @property
    def n_samples(self) -> int:
        return len(self.indptr) - 1

    @property
    def n_branches(self) -> int:
        return len(self.branch_len)

    def problem(self) -> ff_problem:
        p = ff_problem()
        p.n_samples = self.n_samples
        p.n_branches = self.n_branches
        p.branch_len = self.branch_len.ctypes.data
        p.indptr = self.indptr.ctypes.data
        p.branch_id = self.branch_id.ctypes.data
        p.abnd = self.abnd.ctypes.data
        return p

    @classmethod
    def _from_handle(cls, h) -> "FlatNodes":
        p = ff_problem()
        L.lib().ff_flat_problem(h, ctypes.byref(p))
        n, b = p.n_samples, p.n_branches

        def arr(ptr, ct, count):
            if count == 0:
                return np.zeros(0, dtype=ct)
            return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(np.ctypeslib.as_ctypes_type(ct))),
                                         shape=(count,)).copy()

        indptr = arr(p.indptr, np.int64, n + 1)
        nnz = int(indptr[-1]) if n >= 0 and len(indptr) else 0
        out = cls(indptr, arr(p.branch_id, np.int32, nnz), arr(p.abnd, np.float64, nnz),
                  arr(p.branch_len, np.float64, b))
        L.lib().ff_flat_free(h)
        return out


def flatten(table: Table, tree: Tree, leave_unnormalized: bool = False) -> FlatNodes:
    """Stage A (unifrac.go:32-67,99-116) on the host."""
    h, err = ctypes.c_void_p(), L.errbuf()
    L.check(L.lib().ff_flatten(table._h, tree._h, _l_mode(leave_unnormalized), ctypes.byref(h), err, L.ERRLEN), err)
    return FlatNodes._from_handle(h)


def flatten_leaf_csr(tree: Tree, leaf_ptr, leaf_idx, leaf_val, leave_unnormalized: bool = False) -> FlatNodes:
    leaf_ptr = np.ascontiguousarray(leaf_ptr, dtype=np.int64)
    leaf_idx = np.ascontiguousarray(leaf_idx, dtype=np.int64)
    leaf_val = np.ascontiguousarray(leaf_val, dtype=np.float64)
    h, err = ctypes.c_void_p(), L.errbuf()
    L.check(L.lib().ff_flatten_leaf_csr(tree._h, len(leaf_ptr) - 1, leaf_ptr.ctypes.data, leaf_idx.ctypes.data,
                                        leaf_val.ctypes.data, _l_mode(leave_unnormalized),
                                        ctypes.byref(h), err, L.ERRLEN), err)
    return FlatNodes._from_handle(h)


def flatten_device(tree: Tree, leaf_ptr, leaf_idx, leaf_val, leave_unnormalized: bool = False) -> FlatNodes:
    """Stage A on the GPU (ff_flatten_device); returns the flat nodes on the host."""
    leaf_ptr = np.ascontiguousarray(leaf_ptr, dtype=np.int64)
    leaf_idx = np.ascontiguousarray(leaf_idx, dtype=np.int64)
    leaf_val = np.ascontiguousarray(leaf_val, dtype=np.float64)
    h, err = ctypes.c_void_p(), L.errbuf()
    L.check(L.lib().ff_flatten_device(tree._h, len(leaf_ptr) - 1, leaf_ptr.ctypes.data, leaf_idx.ctypes.data,
                                      leaf_val.ctypes.data, _l_mode(leave_unnormalized),
                                      ctypes.byref(h), err, L.ERRLEN), err)
    return FlatNodes._from_handle(h)


def unifrac_dists(nodes: FlatNodes, weighted: bool, precision="auto", device: int = -1,
                  rank: int = 0, world: int = 1, out: Optional[np.ndarray] = None, unsorted_walk: bool = False) -> np.ndarray:
    """unifracDists (frcfrc/unifrac.go:209): all pair distances of this shard in
    IterPairs order, computed on the GPU.  Returns the full-length array; slots
    outside the shard keep NaN (or the contents of `out`)."""
    n = nodes.n_samples
    if out is None:
        out = np.full(num_pairs(n), np.nan, dtype=np.float64)
    p, o, err = nodes.problem(), _opts(weighted, precision, device, rank, world, unsorted_walk), L.errbuf()
    L.check(L.lib().ff_unifrac_dists(ctypes.byref(p), ctypes.byref(o), out.ctypes.data, err, L.ERRLEN), err)
    return out


def unifrac_text_stream(nodes: FlatNodes, weighted: bool, write, precision="auto", device: int = -1, rank: int = 0,
                        world: int = 1, max_pairs_per_chunk: int = 0, flat_args: bool = False,
                        unsorted_walk: bool = False) -> int:
    """unifracDists and the loop that prints it (unifrac.go:209-228 + frcfrc.go:58-62) as one lazy sequence of TEXT
    (ff_unifrac_text_stream): `write(bytes)` receives pieces of whole lines, in order, formatted on the device; a
    `write` that returns False stops the computation.  Returns the number of bytes handed over."""
    state = {"error": None, "bytes": 0}

    def on_text(_user, text, n):
        try:  # (nothing may leave a ctypes callback: see unifrac_dists_stream)
            keep_going = write(ctypes.string_at(text, n))
            state["bytes"] += int(n)
            return 0 if keep_going is False else 1
        except BaseException as e:  # noqa: BLE001
            state["error"] = e
            return 0

    cb = L.TEXT_FN(on_text)
    p, o, err = nodes.problem(), _opts(weighted, precision, device, rank, world, unsorted_walk), L.errbuf()
    if flat_args:
        rc = L.lib().ff_unifrac_text_stream_csr(nodes.n_samples, nodes.n_branches, nodes.branch_len.ctypes.data,
                                                nodes.indptr.ctypes.data, nodes.branch_id.ctypes.data, nodes.abnd.ctypes.data,
                                                ctypes.byref(o), int(max_pairs_per_chunk), cb, None, err, L.ERRLEN)
    else:
        rc = L.lib().ff_unifrac_text_stream(ctypes.byref(p), ctypes.byref(o), int(max_pairs_per_chunk), cb, None, err, L.ERRLEN)
    if state["error"] is not None:
        raise state["error"]
    L.check(rc, err)
    return state["bytes"]


def unifrac_dists_stream(nodes: FlatNodes, weighted: bool, precision="auto", device: int = -1, rank: int = 0,
                         world: int = 1, max_pairs_per_chunk: int = 0, flat_args: bool = False
                         ) -> Iterator[Tuple[int, np.ndarray]]:
    """unifracDists as the reference has it (frcfrc/unifrac.go:209-228): a lazy, ordered sequence
    that stops computing when the consumer stops.  Yields (slot_begin, distances) pieces of at most
    max_pairs_per_chunk consecutive slots in IterPairs order (ff_unifrac_dists_stream).  Nothing
    is staged before the first next(); closing the generator early stops the remaining
    sub-shards.  flat_args: through ff_unifrac_dists_stream_csr, as a cgo host calls it.

    The C entry point calls back on the calling thread; a generator cannot be resumed from
    inside a C callback, so this wrapper runs the call on a helper thread and hands pieces over
    one at a time (the callback blocks until the consumer asks for the next piece)."""
    import queue
    import threading

    pieces: "queue.Queue" = queue.Queue(maxsize=1)
    resume = threading.Semaphore(0)
    state = {"stop": False, "error": None, "delivered": 0}

    def on_piece(_user, slot_begin, dists, n):
        # An exception that leaves a ctypes callback is printed and SWALLOWED, and the callback then returns 0 --
        # "the consumer stopped" to the C side, which ends with FF_OK: a truncated sequence that looks complete.
        # So nothing may leave: what goes wrong here (MemoryError on the copy of a 256 MB piece) is kept for the
        # consumer, and 0 stops the computation.
        try:
            pieces.put((int(slot_begin), np.ctypeslib.as_array(dists, shape=(int(n),)).copy()))
            state["delivered"] += int(n)
        except BaseException as e:  # noqa: BLE001
            state["error"] = e
            return 0
        resume.acquire()
        return 0 if state["stop"] else 1

    cb = L.DISTS_FN(on_piece)
    p, o, err = nodes.problem(), _opts(weighted, precision, device, rank, world), L.errbuf()

    def call():
        try:
            if flat_args:
                rc = L.lib().ff_unifrac_dists_stream_csr(
                    nodes.n_samples, nodes.n_branches, nodes.branch_len.ctypes.data, nodes.indptr.ctypes.data,
                    nodes.branch_id.ctypes.data, nodes.abnd.ctypes.data, ctypes.byref(o), int(max_pairs_per_chunk),
                    cb, None, err, L.ERRLEN)
            else:
                rc = L.lib().ff_unifrac_dists_stream(ctypes.byref(p), ctypes.byref(o), int(max_pairs_per_chunk), cb,
                                                     None, err, L.ERRLEN)
            pieces.put(("done", rc))
        except BaseException as e:  # pragma: no cover
            pieces.put(("done", e))

    th = threading.Thread(target=call, daemon=True)
    th.start()
    try:
        while True:
            item = pieces.get()
            if item[0] == "done":
                if state["error"] is not None:
                    raise state["error"]
                if isinstance(item[1], BaseException):
                    raise item[1]
                L.check(item[1], err)
                a, b = shard_slots(nodes.n_samples, rank, world)
                if state["delivered"] != b - a:   # (the consumer did not stop, no error: every slot must have come)
                    raise RuntimeError("ff_unifrac_dists_stream delivered %d of the shard's %d distances"
                                       % (state["delivered"], b - a))
                return
            yield item
            resume.release()
    finally:
        state["stop"] = True
        resume.release()
        th.join()


def unifrac(table: Table, tree: Tree, weighted: bool, leave_unnormalized: bool = False,
            precision="auto", device: int = -1) -> np.ndarray:
    """unifrac (frcfrc/unifrac.go:97): flatten on the host, distances on the GPU."""
    out = np.full(num_pairs(len(table)), np.nan, dtype=np.float64)
    o, err = _opts(weighted, precision, device), L.errbuf()
    L.check(L.lib().ff_unifrac(table._h, tree._h, ctypes.byref(o), _l_mode(leave_unnormalized),
                               out.ctypes.data, err, L.ERRLEN), err)
    return out


class Plan:
    """Staged inputs resident in HBM + tile schedule (ff_plan)."""

    def __init__(self, nodes: Optional[FlatNodes], weighted: bool, precision="auto", device: int = -1,
                 rank: int = 0, world: int = 1, _handle=None, unsorted_walk: bool = False):
        if _handle is None:
            p, o, err = nodes.problem(), _opts(weighted, precision, device, rank, world, unsorted_walk), L.errbuf()
            self._h = ctypes.c_void_p()
            L.check(L.lib().ff_plan_create(ctypes.byref(p), ctypes.byref(o), ctypes.byref(self._h), err, L.ERRLEN), err)
        else:
            self._h = _handle
        self.info = ff_plan_info()
        L.lib().ff_plan_info_get(self._h, ctypes.byref(self.info))

    @classmethod
    def from_leaves(cls, tree: "Tree", leaf_ptr, leaf_idx, leaf_val, weighted: bool, leave_unnormalized: bool = False,
                    precision="auto", device: int = -1, rank: int = 0, world: int = 1) -> "Plan":
        """Stage A on the device, then staging (ff_plan_create_from_leaves)."""
        leaf_ptr = np.ascontiguousarray(leaf_ptr, dtype=np.int64)
        leaf_idx = np.ascontiguousarray(leaf_idx, dtype=np.int64)
        leaf_val = np.ascontiguousarray(leaf_val, dtype=np.float64)
        o, err, h = _opts(weighted, precision, device, rank, world), L.errbuf(), ctypes.c_void_p()
        L.check(L.lib().ff_plan_create_from_leaves(tree._h, len(leaf_ptr) - 1, leaf_ptr.ctypes.data,
                                                   leaf_idx.ctypes.data, leaf_val.ctypes.data,
                                                   _l_mode(leave_unnormalized), ctypes.byref(o),
                                                   ctypes.byref(h), err, L.ERRLEN), err)
        return cls(None, weighted, _handle=h)

    def close(self):
        _release(self, "ff_plan_destroy")

    __del__ = close

    @property
    def n_slots(self) -> int:
        return int(self.info.slot_end - self.info.slot_begin)

    def set_shard(self, rank: int, world: int) -> None:
        """Re-targets the staged plan at shard `rank` of `world` (ff_plan_set_shard): same staged
        matrix, new schedule and accumulators; n_slots and info change."""
        err = L.errbuf()
        L.check(L.lib().ff_plan_set_shard(self._h, int(rank), int(world), err, L.ERRLEN), err)
        L.lib().ff_plan_info_get(self._h, ctypes.byref(self.info))

    def run_host(self) -> np.ndarray:
        """The current shard's distances as a host array (ff_plan_run_host): blocking, no device
        memory on the caller's side."""
        out = np.empty(self.n_slots, dtype=np.float64)
        err = L.errbuf()
        L.check(L.lib().ff_plan_run_host(self._h, out.ctypes.data, err, L.ERRLEN), err)
        return out

    def run(self, d_out_ptr: int, stream: int = 0, timed: bool = False) -> None:
        """One pass of the hot path; d_out_ptr is a device pointer to n_slots doubles,
        stream a hipStream_t handle (0 = null stream).  Asynchronous."""
        err = L.errbuf()
        fn = L.lib().ff_plan_run_timed if timed else L.lib().ff_plan_run
        L.check(fn(self._h, ctypes.c_void_p(stream), ctypes.c_void_p(d_out_ptr), err, L.ERRLEN), err)

    def timing_collect(self) -> Tuple[float, int]:
        """(summed ms of the pair-tile kernel, launches) over the timed runs since the
        last call; synchronises on their events."""
        ms, n = ctypes.c_double(), ctypes.c_int32()
        rc = L.lib().ff_plan_timing_collect(self._h, ctypes.byref(ms), ctypes.byref(n))
        if rc:
            raise FFError(rc, "ff_plan_timing_collect failed")
        return ms.value, n.value

    def timing_collect_parts(self) -> Tuple[float, float, int]:
        """(summed ms of the pair reduction, of which the rare rows' kernel, launches) over the timed runs since the
        last call (ff_plan_timing_collect_parts); synchronises on their events."""
        ms, rare, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int32()
        rc = L.lib().ff_plan_timing_collect_parts(self._h, ctypes.byref(ms), ctypes.byref(rare), ctypes.byref(n))
        if rc:
            raise FFError(rc, "ff_plan_timing_collect_parts failed")
        return ms.value, rare.value, n.value


    def refined_pairs(self) -> Tuple[int, int]:
        """(pairs the last completed run re-computed exactly, queue capacity)."""
        q, c = ctypes.c_int64(), ctypes.c_int64()
        rc = L.lib().ff_plan_refined_pairs(self._h, ctypes.byref(q), ctypes.byref(c))
        if rc:
            raise FFError(rc, "ff_plan_refined_pairs failed")
        return q.value, c.value


    def audit(self) -> Tuple[int, int, float]:
        """(audited pairs, how many of them the last completed run missed by more than 0.5e-6
        relative, worst relative error over the sample) -- ff_plan_audit."""
        n, bad, worst = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_double()
        rc = L.lib().ff_plan_audit(self._h, ctypes.byref(n), ctypes.byref(bad), ctypes.byref(worst))
        if rc:
            raise FFError(rc, "ff_plan_audit failed")
        return n.value, bad.value, worst.value

    def audit_detail(self) -> Tuple[int, int, int, float]:
        """(pairs of the uniform sample, pairs of the last run with a headroom under 1.25 over the refinement rule's
        bound, how many of those were computed in binary64, the run's smallest headroom) -- ff_plan_audit_detail."""
        u, found, chk, h = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_double()
        rc = L.lib().ff_plan_audit_detail(self._h, ctypes.byref(u), ctypes.byref(found), ctypes.byref(chk), ctypes.byref(h))
        if rc:
            raise FFError(rc, "ff_plan_audit_detail failed")
        return u.value, found.value, chk.value, h.value

    def check_precision(self) -> None:
        """Raises FFError(FF_ERR_PRECISION) unless the last completed run keeps FIXED32's
        promise: refinement queue not overflowed, audit sample within its bar."""
        queued, cap = self.refined_pairs()
        if queued > cap:
            raise FFError(L.FF_ERR_PRECISION, "%d nearly identical pairs, %d can be re-computed exactly: "
                          "stage this problem with precision='exact64'" % (queued, cap))
        n, bad, worst = self.audit()
        if bad:
            raise FFError(L.FF_ERR_PRECISION, "%d of %d audited pairs missed the tolerance (worst %.2e): "
                          "stage this problem with precision='exact64'" % (bad, n, worst))


class DeviceBuffer:
    """`n` float64 of device memory owned through the C ABI (ff_device_alloc): exportable to the
    other processes of the node (ff_ipc_export) and viewable as a torch tensor without a copy
    (__cuda_array_interface__)."""

    def __init__(self, n: int, device: int):
        self.n, self.device = int(n), int(device)
        p, err = ctypes.c_void_p(), L.errbuf()
        L.check(L.lib().ff_device_alloc(self.device, 8 * max(self.n, 1), ctypes.byref(p), err, L.ERRLEN), err)
        self.ptr = int(p.value)

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self.n,), "typestr": "<f8", "data": (self.ptr, False), "version": 2}

    def tensor(self):
        import torch

        return torch.as_tensor(self, device=torch.device("cuda", self.device))

    def export(self) -> bytes:
        h, err = (ctypes.c_ubyte * 64)(), L.errbuf()
        L.check(L.lib().ff_ipc_export(ctypes.c_void_p(self.ptr), h, err, L.ERRLEN), err)
        return bytes(h)

    def free(self) -> None:
        if getattr(self, "ptr", 0):
            err = L.errbuf()
            ptr, self.ptr = self.ptr, 0
            L.check(L.lib().ff_device_free(ctypes.c_void_p(ptr), err, L.ERRLEN), err)


class MappedBuffer:
    """Another process's DeviceBuffer mapped into this one (ff_ipc_open)."""

    def __init__(self, handle: bytes, device: int):
        h = (ctypes.c_ubyte * 64).from_buffer_copy(handle)
        p, err = ctypes.c_void_p(), L.errbuf()
        L.check(L.lib().ff_ipc_open(h, int(device), ctypes.byref(p), err, L.ERRLEN), err)
        self.ptr = int(p.value)

    def close(self) -> None:
        if getattr(self, "ptr", 0):
            err = L.errbuf()
            ptr, self.ptr = self.ptr, 0
            L.check(L.lib().ff_ipc_close(ctypes.c_void_p(ptr), err, L.ERRLEN), err)


def device_copy_async(dst_ptr: int, src_ptr: int, nbytes: int, stream: int = 0) -> None:
    """ff_device_copy_async: device-to-device, local or into a mapped peer buffer, on `stream`."""
    err = L.errbuf()
    L.check(L.lib().ff_device_copy_async(ctypes.c_void_p(dst_ptr), ctypes.c_void_p(src_ptr), int(nbytes),
                                         ctypes.c_void_p(stream), err, L.ERRLEN), err)


def format_float(f: float) -> str:
    """fmt.Fprintln's rendering of a float64, without the newline."""
    buf = ctypes.create_string_buffer(40)
    n = L.lib().ff_format_float(float(f), buf)
    return buf.raw[:n].decode("ascii")


def write_distances(path: Optional[str], d: np.ndarray, threads: int = 1) -> None:
    d = np.ascontiguousarray(d, dtype=np.float64)
    err = L.errbuf()
    L.check(L.lib().ff_write_distances(path.encode() if path else None, d.ctypes.data, len(d), threads, err, L.ERRLEN), err)


def frcfrc_main(argv: List[str]) -> int:
    """The whole `frcfrc` command in-process; argv[0] is the program name."""
    arr = (ctypes.c_char_p * len(argv))(*[a.encode() for a in argv])
    return int(L.lib().ff_frcfrc_main(len(argv), arr))
