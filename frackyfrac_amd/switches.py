"""The ONE list of the engine's switches (environment variables, or ff_tune(name, value) through the C ABI).

INTEGRATION.md section 3 is generated from it (`python -m frackyfrac_amd.switches`), and tests/test_switches_cpu.py
holds it against the sources: every "FF_..." literal the library, the Python layer or bench.py reads is listed here,
every listed name is read somewhere, every tuning switch is exercised by a test, and the table in INTEGRATION.md is
the generated one.  kind: "tuning" (changes how, never what: every value gives the same results), "diagnostic"
(read by the diagnostic build only: `make diag`), "hook" (fault injection and the like, for the tests)."""

SWITCHES = [
    # name, default, kind, what
    ("FF_UNWEIGHTED_MFMA", "1", "tuning", "unweighted FIXED32 on the int8 matrix cores (0: the vector-ALU kernel; same integers)"),
    ("FF_MFMA_SMALL", "unset", "tuning", "1 / 0 forces / forbids the one-launch kernel for shards smaller than one round of the matrix-core kernel (unset: by size)"),
    ("FF_MFMA_GRADED", "1", "tuning", "lengths of more than two base-128 digits: rows staged by length with signed digits, one sweep of three planes then two (0: base-128 digits in branch order, two planes per sweep, two or three sweeps; same integers)"),
    ("FF_MFMA_PRIVATE_MB", "2048", "tuning", "memory up to which every item of the matrix-core schedule owns a partial tile (0: only the remainder's ranges)"),
    ("FF_WAVES_PER_WG", "unset", "tuning", "8 / 12 forces the two- / three-waves-per-SIMD weighted pair kernel (unset: 12 for a shard that begins at row 0 and holds 2.75 or more tiles per CU, about 3,300 samples up, and for any shard of 200,000 or more (tile, branch row) units per CU; else 8)"),
    ("FF_XCD_SLICES", "unset", "tuning", "unset: the schedule builds the plain rounds and the XCD-sliced ones with 2, 4, 8 slices and keeps the best by estimated makespan; 0 forbids sliced rounds, 2 / 4 / 8 asks for that slicing wherever it applies"),
    ("FF_COMPACT", "1", "tuning", "stage only the branches some sample touches"),
    ("FF_SPARSE", "1", "tuning", "the sparse-aware weighted kernel where enough (32-sample block, branch) cells are empty"),
    ("FF_SPARSE_MIN", "0.28", "tuning", "the share of empty cells from which the sparse-aware kernel is taken"),
    ("FF_SPARSE_SPLIT", "unset", "tuning", "weighted FIXED32: 1 / 0 forces / forbids keeping the rows few samples reach out of the staged matrix and reducing them by pair_low_kernel over the pairs that both have them (unset: when that is estimated to save 30 % or more; same integers)"),
    ("FF_LOW_TILE", "unset", "tuning", "128 / 112 / 96 / 80 / 64 forces the side of pair_low_kernel's blocks of pairs (unset: the one whose blocks fill their rounds of two workgroups per CU best)"),
    ("FF_REFINE", "1", "tuning", "FIXED32: exact re-computation of nearly equal pairs (0: tests of what it protects against)"),
    ("FF_AUDIT", "1", "tuning", "FIXED32: the run-time audit -- the uniform sample of a shard's pairs and the run's pairs just above the refinement bound, against binary64"),
    ("FF_EXACT_UNW", "1", "tuning", "EXACT64 unweighted on pair_exact_unw_kernel (0: the weighted kernel's arithmetic on presence as 1.0 / 0.0; same bits, three times the time)"),
    ("FF_XU_JMAX", "unset", "tuning", "1 / 2 forces the width of pair_exact_unw_kernel's tiles in 64-sample column groups (unset: 2, 1 for shards of fewer than six such tiles per CU)"),
    ("FF_X_SKIP", "1", "tuning", "EXACT64 weighted on pair_exact64_skip_kernel, which takes the reference's shortcut for branches a row has not (0: pair_exact64_kernel, six operations for every term; same bits)"),
    ("FF_X_TILE_H", "by size", "tuning", "rows of a weighted EXACT64 pair tile: 4, 8, 10, 12, 14 or 16 (any height: same bits)"),
    ("FF_GATHER", "auto", "tuning", "multi-GPU gather transport: ipc (copies into rank 0's mapped array) / nccl (RCCL send / recv); auto: ipc when it can be set up and is fast enough"),
    ("FF_GATHER_CHUNKS", "1", "tuning", "sub-shards per rank, each on its way to the root while the next one is reduced (either transport)"),
    ("FF_GATHER_MIN_GBPS", "15", "tuning", "the probe rate under which ipc is not used"),
    ("FF_GATHER_TIMEOUT_S", "600", "tuning", "how long a rank waits for a peer's side of the point-to-point gather before it gives up"),
    ("FF_CLI_MAX_PAIRS", "2^25", "tuning", "pairs per pass of the `frcfrc` command"),
    ("FF_LIB_PATH", "-", "tuning", "another build of the library for the Python layer (the diagnostic build)"),
    ("FF_STAMPS", "0", "diagnostic", "per-wave clock stamps of the weighted pair kernel (tools/wave_stamps.py)"),
    ("FF_MFMA_DIAG", "0", "diagnostic", "ablated variants of the matrix-core kernel, for timing only (tools/mfma_diag.py)"),
    ("FF_GATHER_FAULT", "-", "hook", "the rank whose ipc mapping fails (tests/test_gpu_ipc_gather.py)"),
    ("FF_BENCH_TRACE_IMPORTS", "-", "hook", "bench.py's launcher reports whether it imported torch (tests/test_bench_cpu.py)"),
    ("FF_CLI_FAST_EXIT", "-", "hook", "set by the frcfrc executable for its own ff_frcfrc_main call: the command's last frees are left to the exiting process"),
    ("FF_SHIM_PRECISION", "-", "hook", "precision of the cgo shim's call sequence as the C harness runs it (tests/harness/go_shim_sequence.c)"),
]

BEGIN, END = "<!-- switches: generated by `python -m frackyfrac_amd.switches` -->", "<!-- /switches -->"


def markdown() -> str:
    rows = ["| switch | default | what |", "|---|---|---|"]
    rows += ["| `%s` | %s | %s |" % (n, d, w) for n, d, k, w in SWITCHES if k == "tuning"]
    diag = ", ".join("`%s` (%s)" % (n, w) for n, d, k, w in SWITCHES if k == "diagnostic")
    hooks = ", ".join("`%s` (%s)" % (n, w) for n, d, k, w in SWITCHES if k == "hook")
    return "\n".join([BEGIN] + rows + ["", "Read by the diagnostic build only (`make diag`): %s." % diag, "",
                                        "Hooks of the tests: %s." % hooks, END])


if __name__ == "__main__":
    print(markdown())
