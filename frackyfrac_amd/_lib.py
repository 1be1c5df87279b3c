"""ctypes binding of libfrackyfrac_amd.so (C ABI: include/frackyfrac_amd.h).

The library is built in-tree by `make -C frackyfrac_amd/csrc` (or
`__graft_entry__.build()`) into frackyfrac_amd/lib/.  There is no Python or CPU
fallback: if the library is missing, importing the compute API raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# FF_LIB_PATH: another build of the same library (tools/mfma_diag.py times ablated kernels from one)
LIB_PATH = os.environ.get("FF_LIB_PATH") or os.path.join(_HERE, "lib", "libfrackyfrac_amd.so")
FRCFRC_PATH = os.path.join(_HERE, "lib", "frcfrc")
SPRSPR_PATH = os.path.join(_HERE, "lib", "sprspr")

FF_OK = 0
FF_ERR_ARG, FF_ERR_PARSE, FF_ERR_SPECIES, FF_ERR_DEVICE, FF_ERR_IO, FF_ERR_INTERNAL, FF_ERR_PRECISION = 1, 2, 3, 4, 5, 6, 7
PRECISION_AUTO, PRECISION_FIXED32, PRECISION_EXACT64 = 0, 1, 2
PRECISION_NAMES = {"auto": 0, "fixed32": 1, "exact64": 2}


class FFError(RuntimeError):
    """A non-zero ff_status; .code is the status, str(e) the library's message
    (the text the reference would print after "ERROR: ")."""

    def __init__(self, code: int, msg: str):
        super().__init__(msg)
        self.code = code


class ff_problem(ctypes.Structure):
    _fields_ = [("n_samples", c_int64), ("n_branches", c_int64), ("branch_len", c_void_p),
                ("indptr", c_void_p), ("branch_id", c_void_p), ("abnd", c_void_p)]


class ff_options(ctypes.Structure):
    _fields_ = [("weighted", c_int32), ("precision", c_int32), ("device", c_int32),
                ("rank", c_int32), ("world", c_int32), ("flags", c_int32), ("reserved", c_int32 * 2)]


class ff_plan_info(ctypes.Structure):
    _fields_ = [("precision", c_int32), ("scale_log2", c_int32), ("lengths_exact", c_int32),
                ("n_compute_units", c_int32), ("n_samples", c_int64), ("n_branches", c_int64),
                ("ld", c_int64), ("rows_padded", c_int64), ("row_begin", c_int64),
                ("row_end", c_int64), ("slot_begin", c_int64), ("slot_end", c_int64),
                ("n_tiles", c_int64), ("n_items", c_int64), ("n_wave_slots", c_int64),
                ("staged_bytes", c_double), ("elements", c_double), ("kernel", c_int32), ("n_digits", c_int32),
                ("n_rows", c_int64), ("n_sweeps", c_int32), ("planes_per_sweep", c_int32),
                ("rows_three_planes", c_int64), ("audit_checked", c_int64), ("audit_failed", c_int64),
                ("audit_worst_rel_err", c_double), ("audit_min_headroom", c_double), ("active_fraction", c_double), ("rare_rows", c_int64), ("rare_updates", c_double)]

# ff_dists_fn: int (*)(void *user, int64_t slot_begin, const double *dists, int64_t n)
DISTS_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_int64, POINTER(c_double), c_int64)
# ff_text_fn: int (*)(void *user, const char *text, size_t n_bytes)
TEXT_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_size_t)

KERNEL_NAMES = {0: "pair_sad_kernel", 1: "pair_exact64_kernel", 2: "pair_common_mfma_kernel",
                3: "pair_sad_sparse_kernel", 4: "pair_common_small_kernel", 5: "pair_exact_unw_kernel", 6: "pair_walk_kernel", 7: "pair_exact64_skip_kernel"}
FLAG_UNSORTED_WALK = 1
L_REFERENCE = 2


# name -> (restype, argtypes); exactly the symbols include/frackyfrac_amd.h declares
SIGNATURES = {
    "ff_options_default": (None, [POINTER(ff_options)]),
    "ff_num_pairs": (c_int64, [c_int64]),
    "ff_shard_rows": (c_int, [c_int64, c_int32, c_int32, POINTER(c_int64), POINTER(c_int64)]),
    "ff_unifrac_dists": (c_int, [POINTER(ff_problem), POINTER(ff_options), c_void_p, c_char_p, c_size_t]),
    "ff_plan_create": (c_int, [POINTER(ff_problem), POINTER(ff_options), POINTER(c_void_p), c_char_p, c_size_t]),
    "ff_plan_create_csr": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(ff_options),
                                   POINTER(c_void_p), c_char_p, c_size_t]),
    "ff_unifrac_dists_csr": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(ff_options),
                                     c_void_p, c_char_p, c_size_t]),
    "ff_unifrac_dists_stream": (c_int, [POINTER(ff_problem), POINTER(ff_options), c_int64, DISTS_FN, c_void_p,
                                        c_char_p, c_size_t]),
    "ff_unifrac_dists_stream_csr": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                            POINTER(ff_options), c_int64, DISTS_FN, c_void_p, c_char_p, c_size_t]),
    "ff_plan_destroy": (None, [c_void_p]),
    "ff_plan_info_get": (c_int, [c_void_p, POINTER(ff_plan_info)]),
    "ff_plan_run": (c_int, [c_void_p, c_void_p, c_void_p, c_char_p, c_size_t]),
    "ff_plan_run_timed": (c_int, [c_void_p, c_void_p, c_void_p, c_char_p, c_size_t]),
    "ff_plan_timing_collect": (c_int, [c_void_p, POINTER(c_double), POINTER(c_int32)]),
    "ff_plan_timing_collect_parts": (c_int, [c_void_p, POINTER(c_double), POINTER(c_double), POINTER(c_int32)]),
    "ff_plan_refined_pairs": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64)]),
    "ff_plan_audit": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_double)]),
    "ff_plan_audit_detail": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64), POINTER(c_double)]),
    "ff_device_alloc": (c_int, [c_int32, c_size_t, POINTER(c_void_p), c_char_p, c_size_t]),
    "ff_device_free": (c_int, [c_void_p, c_char_p, c_size_t]),
    "ff_ipc_export": (c_int, [c_void_p, c_void_p, c_char_p, c_size_t]),
    "ff_ipc_open": (c_int, [c_void_p, c_int32, POINTER(c_void_p), c_char_p, c_size_t]),
    "ff_ipc_close": (c_int, [c_void_p, c_char_p, c_size_t]),
    "ff_device_copy_async": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_char_p, c_size_t]),
    "ff_plan_set_shard": (c_int, [c_void_p, c_int32, c_int32, c_char_p, c_size_t]),
    "ff_plan_run_host": (c_int, [c_void_p, c_void_p, c_char_p, c_size_t]),
    "ff_tree_parse": (c_int, [c_char_p, c_size_t, POINTER(c_void_p), c_char_p, c_size_t]),
    "ff_tree_read_file": (c_int, [c_char_p, POINTER(c_void_p), c_char_p, c_size_t]),
    "ff_tree_free": (None, [c_void_p]),
    "ff_tree_num_nodes": (c_int64, [c_void_p]),
    "ff_tree_branch_len": (c_void_p, [c_void_p]),
    "ff_tree_parent": (c_void_p, [c_void_p]),
    "ff_tree_subtree_size": (c_void_p, [c_void_p]),
    "ff_tree_name": (c_char_p, [c_void_p, c_int64]),
    "ff_table_parse_dense": (c_int, [c_char_p, c_size_t, POINTER(c_void_p), c_char_p, c_size_t]),
    "ff_table_parse_sparse": (c_int, [c_char_p, c_size_t, POINTER(c_void_p), c_char_p, c_size_t]),
    "ff_table_read_file": (c_int, [c_char_p, c_int, POINTER(c_void_p), c_char_p, c_size_t]),
    "ff_table_parse_mt": (c_int, [c_char_p, c_size_t, c_int, c_int, POINTER(c_void_p), c_char_p, c_size_t]),
    "ff_table_read_file_mt": (c_int, [c_char_p, c_int, c_int, POINTER(c_void_p), c_char_p, c_size_t]),
    "ff_table_free": (None, [c_void_p]),
    "ff_table_num_samples": (c_int64, [c_void_p]),
    "ff_table_sample_size": (c_int64, [c_void_p, c_int64]),
    "ff_table_sample_entry": (c_int, [c_void_p, c_int64, c_int64, POINTER(c_char_p), POINTER(c_double)]),
    "ff_table_write_sparse": (c_int, [c_void_p, c_char_p, c_char_p, c_size_t]),
    "ff_sprspr_main": (c_int, [c_int, POINTER(c_char_p)]),
    "ff_validate_species": (c_int, [c_void_p, c_void_p, c_char_p, c_size_t]),
    "ff_flatten": (c_int, [c_void_p, c_void_p, c_int, POINTER(c_void_p), c_char_p, c_size_t]),
    "ff_flatten_leaf_csr": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int,
                                    POINTER(c_void_p), c_char_p, c_size_t]),
    "ff_flatten_device": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int,
                                  POINTER(c_void_p), c_char_p, c_size_t]),
    "ff_plan_create_from_leaves": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int,
                                           POINTER(ff_options), POINTER(c_void_p), c_char_p, c_size_t]),
    "ff_flat_free": (None, [c_void_p]),
    "ff_flat_problem": (None, [c_void_p, POINTER(ff_problem)]),
    "ff_unifrac": (c_int, [c_void_p, c_void_p, POINTER(ff_options), c_int, c_void_p, c_char_p, c_size_t]),
    "ff_format_float": (c_int, [c_double, c_char_p]),
    "ff_write_distances": (c_int, [c_char_p, c_void_p, c_int64, c_int, c_char_p, c_size_t]),
    "ff_cpu_quota": (c_int, []),
    "ff_unifrac_text_stream": (c_int, [POINTER(ff_problem), POINTER(ff_options), c_int64, TEXT_FN, c_void_p, c_char_p, c_size_t]),
    "ff_unifrac_text_stream_csr": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(ff_options), c_int64,
                                           TEXT_FN, c_void_p, c_char_p, c_size_t]),
    "ff_text_bound": (c_size_t, [c_int64]),
    "ff_format_distances_device": (c_int, [c_void_p, c_int64, c_void_p, POINTER(c_size_t), c_void_p, c_char_p, c_size_t]),
    "ff_frcfrc_main": (c_int, [c_int, POINTER(c_char_p)]),
    "ff_tune": (c_int, [c_char_p, c_char_p]),
    "ff_synth_counts": (c_int, [c_int64, c_double, ctypes.c_uint64, c_int64, c_int64, c_int, c_void_p]),
    "ff_synth_fill": (c_int, [c_int64, c_double, ctypes.c_uint64, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "ff_version": (c_char_p, []),
}

_lib = None


def lib() -> ctypes.CDLL:
    """Loads the in-tree shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "frackyfrac_amd: %s not found. Build it with `make -C frackyfrac_amd/csrc` "
                "(or __graft_entry__.build()); there is no fallback path." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the ABI and the header drift apart
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


ERRLEN = 1024


def errbuf():
    return ctypes.create_string_buffer(ERRLEN)


def check(rc: int, err) -> None:
    if rc != FF_OK:
        raise FFError(rc, err.value.decode("utf-8", "replace"))
