#!/usr/bin/env python3
"""bench.py -- throughput of the UniFrac pair reduction (the hot path) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3] [--precision fixed32]
                    [--scaling weak|strong] [--no-secondary] [--no-cpu-baseline] [--no-live-traffic]

A "step" is one pass of the hot path over one batch of synthetic input: the pair
kernels over this rank's row shard of the staged matrix (already resident in HBM)
plus, for N > 1, the gather of the result slices to rank 0.  N = 1 runs BASELINE.json's
headline configuration C3 (weighted UniFrac, 4096 samples x 10k-leaf tree).

N > 1: one rank per GPU under torch.distributed.run.  Started WITHOUT that launcher
(`python bench.py --gpus 8 ...`, the way the N = 1 run is started) this process launches
it itself: it starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a
child BEFORE importing torch or touching HIP, relays rank 0's JSON line and exit code, and
never uses the GPU.  With nothing else asked for the N > 1 line leads with BASELINE configs[3]
-- C4, 16,384 samples, pair tiles sharded over the N GPUs: strong scaling by construction --
and carries the weak-scaled C3 (samples = 4096*sqrt(N), every GPU keeps C3's pair count) beside
it as `weak_scaling`; `--scaling weak|strong` with a `--workload` runs just that.  `value` is
all ranks' pairs / max-over-ranks time.

`secondary` carries the other claimed numbers under the same clock, each with its own
`config.workload`, `dtype`, `ms_per_step` and `roofline`:
  N = 1: C3 in EXACT64 (the reference's binary64 roundings; also `reference_width` of the
         primary record), C3 unweighted (int8 matrix cores), C2 (BASELINE configs[1]), C4 and
         C5 on one GPU, C3 unweighted with log-normal branch lengths on request (fixed32,
         graded digit planes) and as the engine runs it by itself (EXACT64, the reference's bits);
  N > 1: BASELINE configs[4] -- C5 at its stated size, row shards over the N GPUs -- with the
         gather transport that ran.

Prints ONE JSON line on rank 0; `roofline` describes the dominant kernel, timed with HIP
events around every launch of the timed region on the stream it is launched on
(ff_plan_run_timed / ff_plan_timing_collect).  At N = 1 the primary record's `roofline.traffic`
-- the dominant kernel's bytes beyond L2 per launch -- is counted on the box the bench runs on:
with every timing done, two child runs of this script under `rocprofv3 --pmc` (FETCH_SIZE,
WRITE_SIZE: separate passes, counters only); `--no-live-traffic`, a bench that is itself under a
profiler, or a failed pass fall back on the committed counts of profiles/traffic.json and say so.
"""
import argparse
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# this pool's driver only supports dmabuf IPC (cross-process device memory: RCCL, the "ipc" gather)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402

# MI355X ceilings (/opt/skills/guides/MI355X_MICROARCH.md, chip table)
SPARSE_REGIME = ("8192x50000@0.01", "8192x50000@0.002")   # C5's tree and sample count at real-table densities
HBM_PEAK_GBPS = 8000.0            # HBM3E spec
HBM_COPY_GBPS = 6290.0            # measured copy ceiling (same guide)
VALU_PEAK_TLANEOPS = 78.65        # 157.3 TFLOP/s FP32 vector / 2 flops per lane-op
MFMA_I8_PEAK_TOPS = 5000.0        # int8 MFMA issues at 2x the dense bf16 rate (~2.5 PFLOP/s)
LDS_PEAK_TADDS = 9.83             # ds_add_u32 moves an address and a data dword per lane like ds_write_b32: 4 cycles per
                                  # wave-instruction (same guide, LDS table) = 16 lane-adds per CU and clock x 256 CUs x 2.4 GHz
                                  # (measured: 15.8 per CU and clock, tools/microbench/lds_add_rate.hip, profiles/r05_lds_add_rate.txt)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def host_cores():
    """CPU share of this process: the cgroup quota if there is one (the GPU box gives a
    1-GPU job 16 CPUs of a 256-thread host), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(math.ceil(float(quota) / float(period)))))
    except Exception:
        pass
    return n


def cpu_baseline(nodes, weighted, budget_s=18.0):
    """The oracle's merge walk (C restatement of frcfrc/unifrac.go:144-228) on the host
    cores, over a bounded prefix of the pairs in IterPairs order: one thread (the reference's
    default, frcfrc.go:24 `-p 1`) and all cores, median of 3 runs each."""
    from oracle import oracle as O

    cores = host_cores()
    onodes = np.zeros(len(nodes.branch_id), dtype=O.FLATNODE)
    onodes["id"] = nodes.branch_id
    onodes["abnd"] = nodes.abnd
    n = nodes.n_samples
    P = n * (n - 1) // 2

    def timed(count, threads):
        t0 = time.perf_counter()
        O.unifrac_dists(nodes.indptr, onodes, nodes.branch_len, weighted, nthreads=threads, pair_begin=0, pair_end=count)
        return max(time.perf_counter() - t0, 1e-6)

    def leg(threads):
        # rows get longer as the prefix grows (later rows hold more pairs of the same cost), so
        # cost per pair is flat: size the sample for a sixth of the budget from a probe's rate
        probe = min(P, 2000 * threads)
        dt = timed(probe, threads)
        count = int(min(P, max(probe, probe / dt * budget_s / 6.0)))
        runs = sorted(timed(count, threads) for _ in range(3))
        return count, statistics.median(runs), runs

    c1, t1, r1 = leg(1)
    ca, ta, ra = leg(cores)
    return {"value": ca / ta, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "first %d of %d pairs in IterPairs order, median of 3 runs (%.2f / %.2f / %.2f s), "
                      "oracle/unifrac_oracle.c merge walk (C restatement of the reference algorithm, %d threads)"
                      % (ca, P, ra[0], ra[1], ra[2], cores),
            "single_thread": {"value": c1 / t1, "unit": "pairs/s", "cores": 1,
                              "sample": "first %d of %d pairs, median of 3 runs (%.2f / %.2f / %.2f s)"
                                        % (c1, P, r1[0], r1[1], r1[2])}}


# ---- end to end: the command a user runs ---------------------------------------------------

def frcfrc_end_to_end(workload, cores):
    """The `frcfrc` executable on the workload as FILES: synthetic table as sparse text + Newick in, one distance per
    line out (frcfrc/frcfrc.go:29-67), timed from outside (wall) and by the command's own phase table (-stats), with
    the reference's default flags (no -p), with -p 1 and with -p <cores>.  SURVEY 8(d): "report end-to-end and host
    prep separately" -- this is the whole of it, never `value`."""
    import hashlib
    import shutil
    import tempfile

    from frackyfrac_amd import _lib as L
    from frackyfrac_amd import synth

    cfg = synth.CONFIGS[workload]
    t0 = time.perf_counter()
    tree, ptr, idx, val = synth.make(cfg["n_samples"], cfg["n_leaves"], cfg["density"], cfg["seed"])
    d = tempfile.mkdtemp(prefix="ff_e2e_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        with open(os.path.join(d, "t.tree"), "w") as f:
            f.write(tree.newick())
        text = synth.sparse_text(tree, ptr, idx, val)
        with open(os.path.join(d, "t.tab"), "w") as f:
            f.write(text)
        entry = {"workload": workload, "samples": cfg["n_samples"], "leaves": cfg["n_leaves"],
                 "pairs": cfg["n_samples"] * (cfg["n_samples"] - 1) // 2,
                 "input": "sparse text, %.1f MB (+ Newick), generated in %.1f s" % (len(text) / 1e6, time.perf_counter() - t0),
                 "runs": []}
        del text
        digests = set()
        # (the command's first run on a box pays for what no later one does -- the library's code objects read from a cold
        # file cache, the input files' first read: it is reported apart as `cold_start`, default flags, and the three
        # runs below all follow it)
        for flags in (None, [], ["-p", "1"], ["-p", str(cores)]):
            cold, flags = flags is None, flags or []
            out = os.path.join(d, "out.txt")
            if os.path.exists(out):
                os.unlink(out)  # (truncating the previous run's gigabytes is not part of a run)
            cmd = [L.FRCFRC_PATH, "-s", "-w" if cfg["weighted"] else "-w=false", "-t", os.path.join(d, "t.tree"),
                   "-i", os.path.join(d, "t.tab"), "-o", out, "-stats"] + flags
            t0 = time.perf_counter()
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            wall = time.perf_counter() - t0
            run = {"flags": " ".join(flags) or "(default)", "rc": r.returncode, "wall_s": wall}
            stats = [ln for ln in r.stderr.splitlines() if ln.startswith("{")]
            if cold:
                entry["cold_start"] = {"flags": "(default)", "rc": r.returncode, "wall_s": wall,
                                       "seconds": json.loads(stats[-1])["seconds"] if r.returncode == 0 and stats else None}
                continue
            if r.returncode == 0 and stats:
                st = json.loads(stats[-1])
                run.update({"threads": st.get("threads"), "passes": st.get("passes"), "seconds": st["seconds"],
                            "detail": st.get("detail"), "precision": st.get("precision")})
                k = (st.get("detail") or {}).get("kernels")
                if k:
                    run["distances_over_kernels"] = st["seconds"]["distances"] / k
                h = hashlib.md5()
                n_lines = 0
                with open(out, "rb") as f:
                    for blk in iter(lambda: f.read(1 << 24), b""):
                        h.update(blk)
                        n_lines += blk.count(b"\n")
                digests.add(h.hexdigest())
                run["output_MB"] = os.path.getsize(out) / 1e6
                run["lines"] = n_lines
            else:
                run["stderr"] = r.stderr[-400:]
            entry["runs"].append(run)
        entry["outputs_identical"] = len(digests) == 1
        entry["lines_ok"] = all(r.get("lines") == entry["pairs"] for r in entry["runs"])
        return entry
    finally:
        shutil.rmtree(d, ignore_errors=True)


# ---- self-launch ------------------------------------------------------------------------

def launch_command(n_gpus, argv, port):
    """The child a plain `python bench.py --gpus N` starts: torch.distributed.run with one rank
    per GPU, this file and the same arguments."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n_gpus, argv):
    """Runs the N-rank job as a child and relays rank 0's JSON line and the exit code.  The
    parent imports neither torch nor the engine and never touches the GPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = launch_command(n_gpus, argv, port)
    log("bench.py: launching %d ranks: %s" % (n_gpus, " ".join(cmd)))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    for ln in r.stdout.splitlines():
        if not ln.startswith("{"):
            log(ln)
    if lines:
        print(lines[-1], flush=True)
    if os.environ.get("FF_BENCH_TRACE_IMPORTS"):  # (tests)
        log("parent imported torch: %s" % ("torch" in sys.modules))
    return r.returncode if r.returncode else (0 if lines else 1)


# ---- one measurement ----------------------------------------------------------------------

class Ctx:
    pass


def make_problem(ctx, workload, n_samples=None):
    """Synthetic inputs of one configuration (frackyfrac_amd/synth.py), stage A on the host."""
    import frackyfrac_amd as ff
    from frackyfrac_amd import synth

    if workload in synth.CONFIGS:
        cfg = dict(synth.CONFIGS[workload])
        name = workload
    else:
        # SAMPLESxLEAVES[@DENSITY], e.g. 2048x5000 or 8192x50000@0.01
        shape, _, dens = workload.lower().partition("@")
        ns, nl = shape.split("x")
        cfg = dict(n_samples=int(ns), n_leaves=int(nl), density=float(dens) if dens else 0.10, weighted=True,
                   seed=synth.SEED_BASE + 77)
        name = workload.lower() if dens else "custom"   # (with a density: also the key of its counters in profiles/traffic.json)
    if n_samples is not None:
        cfg["n_samples"] = n_samples
    t0 = time.perf_counter()
    # N > 1: every rank generates and flattens only its own block of samples (the generator has a
    # stream per sample); one all-gather replicates the flat nodes
    from frackyfrac_amd.distributed import allgather_flat_nodes, sample_block
    b, e = sample_block(cfg["n_samples"], ctx.rank, ctx.world)
    tree, ptr, idx, val = synth.make(cfg["n_samples"], cfg["n_leaves"], cfg["density"], cfg["seed"], b, e)
    T = ff.parse_newick(tree.newick())
    nodes = ff.flatten_leaf_csr(T, ptr, idx, val)        # stage A on the host
    if ctx.world > 1:
        nodes = allgather_flat_nodes(nodes)
    cfg["prep_s"] = time.perf_counter() - t0
    cfg["name"] = name
    return cfg, nodes


def with_lognormal_lengths(cfg, nodes):
    """The same flat nodes with branch lengths as a real phylogeny has them: not short binary fractions but spread
    over orders of magnitude (log-normal, sigma 1.5; zero-length branches stay zero).  Unweighted FIXED32 then stages
    graded digit planes."""
    import frackyfrac_amd as ff

    bl = np.random.default_rng(cfg["seed"]).lognormal(-3.0, 1.5, nodes.n_branches)
    bl[nodes.branch_len == 0.0] = 0.0
    return ff.FlatNodes(nodes.indptr, nodes.branch_id, nodes.abnd, bl)


def barrier(ctx):
    # every rank first drains its own streams (with the "ipc" transport a peer's slice
    # reaches the root from the PEER's copy stream), then all meet
    ctx.torch.cuda.synchronize()
    if ctx.world > 1:
        ctx.dist.barrier()
    ctx.torch.cuda.synchronize()


def roofline_of(info, B, n_samples, shard_pairs, kernel_ms, launches, weighted, traffic, rare_ms=None):
    """Roofline of the dominant kernel from the ALGORITHMIC work of one launch."""
    from frackyfrac_amd._lib import KERNEL_NAMES

    kernel = int(info.kernel)
    elem_bytes = 4 if info.precision == 1 else 8
    # per pair 8 + (elem*N*B + 4*B + 4*N)/P bytes: one f64 result + the pair's share of one
    # compulsory read of the staged matrix, lengths and row sums
    alg_bytes = 8.0 * shard_pairs + elem_bytes * float(n_samples) * B + 4.0 * B + 4.0 * n_samples
    sec = max(kernel_ms, 1e-9) * 1e-3
    hbm = {"bound": "hbm", "achieved": alg_bytes / sec / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
           "frac": alg_bytes / sec / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes": alg_bytes}
    if traffic and traffic.get("traffic"):
        # what the counters saw move (2*FETCH_SIZE + WRITE_SIZE per launch: of the profiled build here; main() replaces
        # the primary record's by a measurement of this box, live_traffic) over THIS run's kernel time: the
        # rocprof-reported rate SURVEY 8(d) asks for, against the spec and the measured-copy ceiling
        gbps = traffic["traffic"] / sec / 1e9
        hbm["measured_GBps"] = gbps
        hbm["measured_frac_of_8000"] = gbps / HBM_PEAK_GBPS
        hbm["measured_frac_of_6290"] = gbps / HBM_COPY_GBPS
        hbm["traffic_ratio"] = traffic["traffic"] / alg_bytes   # counter bytes / algorithmic bytes: re-reads
    kname = KERNEL_NAMES[kernel]
    if kernel == 0 and info.n_wave_slots == 12 * info.n_compute_units:
        kname = "pair_sad_kernel12"  # (the three-waves-per-SIMD variant the plan picks for whole triangles from ~3,300 samples)
    common = {"kernel": kname, "kernel_ms": kernel_ms, "launches": launches, "hbm": hbm}
    if int(getattr(info, "rare_rows", 0)) > 0:
        # the rows few samples reach are reduced by pair_low_kernel over the pairs that both have them; kernel_ms
        # is both kernels' (one event pair around the two launches); the fraction stays priced on ALL 2*B lane-ops per pair,
        # so the work that is skipped shows as a fraction that may pass 1
        common["kernels"] = [kname, "pair_low_kernel"]
        common["rare_rows"] = int(info.rare_rows)
        common["frac_note"] = ("priced on ALL 2*B lane-ops per pair (SURVEY 8d); %d of the %d staged rows are kept out of the matrix and "
                               "reduced by pair_low_kernel over the pairs that both have them, so fewer lane-ops are issued than "
                               "counted: the fraction may pass 1 (each kernel on its own rows: `parts`, from an event between the two launches; "
                               "profiles/r05_*_kernel_stats.csv)" % (int(info.rare_rows), int(info.n_rows)))
        if rare_ms and 0 < rare_ms < kernel_ms:
            # each kernel on its own rows, from the event the plan records between the two launches
            # (ff_plan_timing_collect_parts): the matrix rows' kernel against the vector ALU's peak for ITS 2 lane-ops per
            # row and pair; the rare rows' kernel has no such count (its work is the pairs that both have a row)
            matrix_rows = int(info.n_rows) - int(info.rare_rows)
            matrix_ms = kernel_ms - rare_ms
            common["parts"] = [
                {"kernel": kname, "rows": matrix_rows, "ms": matrix_ms,
                 "frac": 2.0 * matrix_rows * shard_pairs / (matrix_ms * 1e-3) / 1e12 / VALU_PEAK_TLANEOPS},
                {"kernel": "pair_low_kernel", "rows": int(info.rare_rows), "ms": rare_ms}]
            upd = float(getattr(info, "rare_updates", 0.0))
            if upd > 0:
                # the rare rows' kernel on ITS work: an update = one min + one add into the block's LDS accumulator
                # (sum over the rare rows of n_r (n_r - 1) / 2), against the LDS's 32 lane-adds per CU and clock
                rate = upd / (rare_ms * 1e-3)
                common["parts"][1].update({"updates": upd, "bound": "lds", "achieved": rate / 1e12, "peak": LDS_PEAK_TADDS,
                                           "unit": "T update/s", "frac": rate / 1e12 / LDS_PEAK_TADDS})
    # the binding floor of THIS launch (one rank's shard): its algorithmic work (2*B per pair) at the unit's peak
    peak_ops = {2: MFMA_I8_PEAK_TOPS, 4: MFMA_I8_PEAK_TOPS, 1: VALU_PEAK_TLANEOPS / 2, 7: VALU_PEAK_TLANEOPS / 2,
                5: VALU_PEAK_TLANEOPS / 2}.get(kernel, VALU_PEAK_TLANEOPS) * 1e12
    common["floor_ms"] = 2.0 * B * shard_pairs / peak_ops * 1e3
    common.update(traffic or {"traffic": None})
    if kernel in (2, 4):
        # unweighted on the matrix cores: one multiply-add per branch and pair is the
        # algorithmic work (the base-128 digit passes are the implementation's)
        achieved = 2.0 * B * shard_pairs / sec / 1e12
        return dict({"bound": "mfma", "achieved": achieved, "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s",
                     "frac": achieved / MFMA_I8_PEAK_TOPS, "digits": int(info.n_digits),
                     "algorithmic": "2*B int8 MAC-ops per pair, B=%d, %d pairs per launch" % (B, shard_pairs)}, **common)
    if kernel in (1, 7):
        # EXACT64 weighted (7: pair_exact64_skip_kernel, which issues 1 / H + 2 (1 - d) + 6 d operations per term at row
        # density d -- 3.3 at C3 -- for the same bits).  SURVEY 8(d) prices the f64 variant like the f32 one: 2*B lane-ops per pair, at the FP64
        # vector peak (MI355X: 78.6 TFLOP/s FP64 vector = 39.3e12 ops/s) -> `frac`.  The reference's ROUNDINGS need
        # six unfused binary64 operations per branch and pair (numer: sub, mul by |.|, add; denom: add, mul, add;
        # unifrac.go:191-192), which no bit-exact kernel can go under -> `frac_unfused6`, the builder's count.
        peak = VALU_PEAK_TLANEOPS / 2
        achieved = 2.0 * B * shard_pairs / sec / 1e12
        return dict({"bound": "valu", "achieved": achieved, "peak": peak, "unit": "T f64 op/s", "frac": achieved / peak,
                     "achieved_unfused6": 3.0 * achieved, "frac_unfused6": 3.0 * achieved / peak,
                     "algorithmic": "2*B binary64 lane-ops per pair (SURVEY 8d; frac_unfused6: against the 6*B unfused operations of "
                                    "the reference's both-present case, unifrac.go:191-192, for every term -- "
                                    "pair_exact64_skip_kernel issues fewer, so it may pass 1), B=%d, %d pairs per launch"
                                    % (B, shard_pairs)},
                    **common)
    if kernel == 5:
        # EXACT64 unweighted: the reference's two running sums are two binary64 ADDITIONS per branch and pair
        # (result or common, unifrac.go:151-167) -- 2*B per pair at the FP64 vector rate.  The kernel issues fewer
        # (a branch a row has not adds to one sum only) plus 5/8 of an instruction per term to prepare operands.
        peak = VALU_PEAK_TLANEOPS / 2
        achieved = 2.0 * B * shard_pairs / sec / 1e12
        return dict({"bound": "valu", "achieved": achieved, "peak": peak, "unit": "T f64 add/s", "frac": achieved / peak,
                     "algorithmic": "2*B binary64 additions per pair (SURVEY 8d), B=%d, %d pairs per launch" % (B, shard_pairs)},
                    **common)
    achieved = 2.0 * B * shard_pairs / sec / 1e12
    return dict({"bound": "valu", "achieved": achieved, "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s",
                 "frac": achieved / VALU_PEAK_TLANEOPS,
                 "algorithmic": "2*B lane-ops per pair (SURVEY 8d), B=%d, %d pairs per launch" % (B, shard_pairs)},
                **common)


def traffic_of(name, world, info, weighted):
    """HBM/fabric bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/traffic.json: value, the counter file it came from, the commit it was taken at).
    A constant of the profiled build, not a measurement of this run -- hence the source."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(path))
        e = t.get("%s_n%d_k%d%s" % (name, world, int(info.kernel), "" if weighted else "_unweighted"))
        if e:
            return {"traffic": e["bytes"], "traffic_source": "%s @ %s (2*FETCH_SIZE + WRITE_SIZE, separate --pmc pass)"
                                                             % (e["source"], e["commit"])}
    except Exception:
        pass
    return None


def under_a_profiler():
    """True when this process already runs under rocprofv3 / rocprof (tools/profile_round.sh, tools/pmc.sh, or whoever
    runs the bench that way): no second profiler is started from inside one."""
    pre = os.environ.get("LD_PRELOAD", "") + os.environ.get("HSA_TOOLS_LIB", "") + os.environ.get("ROCP_TOOL_LIBRARIES", "")
    return "rocprof" in pre or any(k.startswith("ROCPROF") for k in os.environ)


def live_traffic(args, kernel, timeout_s=150.0, kernels=None):
    """Fabric bytes per launch of `kernel` (the primary record's dominant kernel) measured ON THIS BOX, now:
    two child runs of this script under `rocprofv3 --pmc` -- FETCH_SIZE and WRITE_SIZE in separate passes, counters
    only besides the kernel trace that names the dispatches, the program directly behind `--` -- on the primary
    workload with 3 timed launches, the per-launch averages combined as MI355X_MICROARCH.md prescribes for gfx950:
    (2 * FETCH_SIZE + WRITE_SIZE) KiB (wide streaming reads are tallied at half their bytes; Infinity-Cache hits are
    counted, so this is traffic beyond L2, not DRAM traffic alone).  Called last, when every timing of the line
    is done; any failure (no rocprofv3, a refused counter, a timeout) returns None and the line falls back on the
    committed constants of profiles/traffic.json, saying so."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "no rocprofv3 on this box"
    child = [sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--precision", args.precision,
             "--lengths", args.lengths, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-secondary",
             "--no-live-traffic"] + (["--unweighted"] if args.unweighted else [])
    got = {}
    t0 = time.perf_counter()
    with tempfile.TemporaryDirectory(prefix="ff_pmc_", dir="/tmp") as d:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            left = timeout_s - (time.perf_counter() - t0)
            if left < 10:
                return None, "time budget spent before the %s pass" % counter
            out = os.path.join(d, counter)
            try:
                r = subprocess.run([prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--"] + child,
                                   cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE,
                                   stderr=subprocess.PIPE, timeout=left)
            except subprocess.TimeoutExpired:
                return None, "the %s pass did not finish in %.0f s" % (counter, left)
            except OSError as e:
                return None, "rocprofv3 did not start: %s" % e
            if r.returncode != 0:
                return None, "the %s pass ended with code %d: %s" % (counter, r.returncode,
                                                                     r.stderr.decode(errors="replace").strip().splitlines()[-1:])
            per_kernel = {k: [] for k in (kernels or [kernel])}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") != counter:
                        continue
                    for k in per_kernel:
                        if k in row.get("Kernel_Name", ""):
                            per_kernel[k].append(float(row["Counter_Value"]))
            if not per_kernel[kernel]:
                return None, "no %s rows for %s in the counter file" % (counter, kernel)
            # (per launch of the pass: the dominant kernel's average plus its companions')
            got[counter] = (sum(sum(v) / len(v) for v in per_kernel.values() if v), len(per_kernel[kernel]))
    fetch, nf = got["FETCH_SIZE"]
    write, nw = got["WRITE_SIZE"]
    return {"traffic": (2.0 * fetch + write) * 1024.0,
            "traffic_source": "live on this box: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes of "
                              "`bench.py --steps 3`, averages over %d and %d launches of %s), (2*FETCH_SIZE + WRITE_SIZE) KiB; "
                              "%.0f s" % (nf, nw, kernel, time.perf_counter() - t0),
            "fetch_size_kib": fetch, "write_size_kib": write}, None


def with_traffic(roofline, traffic):
    """The hbm.measured_* figures of a roofline record, for `traffic` bytes per launch (see roofline_of)."""
    sec = max(roofline["kernel_ms"], 1e-9) * 1e-3
    gbps = traffic["traffic"] / sec / 1e9
    roofline["hbm"].update({"measured_GBps": gbps, "measured_frac_of_8000": gbps / HBM_PEAK_GBPS,
                            "measured_frac_of_6290": gbps / HBM_COPY_GBPS,
                            "traffic_ratio": traffic["traffic"] / roofline["hbm"]["algorithmic_bytes"]})
    roofline.update(traffic)
    return roofline


def measure(ctx, cfg, nodes, weighted, precision, steps, warmup, event_every=1):
    """Stages the problem on this rank, runs `warmup` untimed and `steps` timed steps, returns
    rank 0's entry (None on other ranks).  event_every: the HIP-event pair that times the dominant kernel
    brackets every launch (1) or every event_every-th launch of the timed region -- for steps of a few
    microseconds, where two event records per launch cost as much as the launch itself (C2: 6.4 us per step
    without them, about 10 with)."""
    import frackyfrac_amd as ff
    from frackyfrac_amd.distributed import ShardedRun

    torch, dist = ctx.torch, ctx.dist
    n_samples, B = nodes.n_samples, nodes.n_branches
    P = ff.num_pairs(n_samples)
    t0 = time.perf_counter()
    run = ShardedRun(nodes, weighted, ctx.rank, ctx.world, precision=precision, device=ctx.local_rank)
    torch.cuda.synchronize()
    t_stage = time.perf_counter() - t0
    info = run.plan.info
    if ctx.world > 1 and ctx.rank == 1 and run.ipc_gbps is not None:
        log("rank 1: ipc slice copy into the root's buffer %.1f GB/s (all peers at once)" % run.ipc_gbps)
    if ctx.rank == 0:
        log("workload %s: N=%d leaves=%d B=%d nnz=%d pairs=%d | prep %.2fs stage(H2D+quantise) %.3fs | "
            "precision=%s scale=2^%d kernel=%d tiles=%d items=%d wave_slots=%d%s" %
            (cfg["name"], n_samples, cfg["n_leaves"], B, len(nodes.branch_id), P, cfg["prep_s"], t_stage,
             {1: "fixed32", 2: "exact64"}[info.precision], info.scale_log2, info.kernel, info.n_tiles, info.n_items,
             info.n_wave_slots,
             (" | gather: %s%s" % (run.transport, (" (ipc not used: %s)" % run.transport_note) if run.transport_note else ""))
             if ctx.world > 1 else ""))
    for _ in range(warmup):
        run.step()
    barrier(ctx)
    run.timing_collect()
    t0 = time.perf_counter()
    res = None
    for k in range(steps):
        res = run.step(timed=(k % event_every == 0))
    barrier(ctx)
    elapsed = time.perf_counter() - t0
    kernel_ms_total, rare_ms_total, launches = run.timing_collect_parts()
    per_rank = None
    if ctx.world > 1:
        # what a post-mortem of the first multi-GPU run needs, from every rank: its own clock around the
        # timed steps, its kernel's HIP-event time, how long its last slice took from "kernels done" to
        # "in the root's array", what it computed and where
        mine = {"rank": ctx.rank, "device": torch.cuda.get_device_name(ctx.local_rank), "local_rank": ctx.local_rank,
                "pairs": int(run.n_slots), "elapsed_ms_per_step": elapsed / steps * 1e3,
                "kernel_ms": kernel_ms_total / max(launches, 1), "launches": launches,
                "exposed_gather_ms_last_step": run.exposed_gather_ms(),
                "transport": run.transport, "transport_note": run.transport_note,
                "ipc_probe_GBps": None if run.ipc_gbps in (None, float("inf")) else run.ipc_gbps}
        per_rank = [None] * ctx.world
        dist.all_gather_object(per_rank, mine)
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if ctx.rehearse else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    run.check_precision()  # queue overflow / audit failure on ANY rank raises on every rank
    entry = None
    if ctx.rank == 0:
        kernel_ms_events = kernel_ms_total / max(launches, 1)
        # A kernel cannot take longer than the step that contains it: around a kernel of a few microseconds the two
        # event records add 2-3 us of their own (C2: 8.8 us between the events, 6.4 us under rocprofv3, 6.9 us per
        # back-to-back step), so the step time bounds the figure from above there; for millisecond kernels the
        # events' figure is the smaller one and stands.
        kernel_ms = min(kernel_ms_events, elapsed / steps * 1e3)
        # cheap sanity on the result of the last step (not a parity test: tests/ does that);
        # every slot, so a slice that never arrived from its rank cannot go unnoticed
        lo, hi = float(res.min().item()), float(res.max().item())
        assert not bool(torch.isnan(res).any().item()), "NaN in the gathered result"
        assert 0.0 <= lo and hi <= 1.0, "distance outside [0, 1]"
        n_audit, bad, worst = run.plan.audit()
        entry = {"value": P / (elapsed / steps), "unit": "pairs/s", "steps": steps, "warmup": warmup,
                 "ms_per_step": elapsed / steps * 1e3,
                 "dtype": {0: "u32", 3: "u32", 1: "f64", 2: "i8", 4: "i8", 5: "f64", 6: "f64", 7: "f64"}[int(info.kernel)],
                 "config": {"workload": "%s: %d samples x %d-leaf Yule tree (B=%d branches), %s UniFrac, "
                                        "leaf density %g, seed 0x%X%s" %
                                        (cfg["name"], n_samples, cfg["n_leaves"], B,
                                         "weighted" if weighted else "unweighted", cfg["density"], cfg["seed"],
                                         ", branch lengths log-normal (sigma 1.5)" if cfg.get("lengths") == "lognormal" else ""),
                            "pairs": P, "precision": {1: "fixed32", 2: "exact64"}[info.precision],
                            "parallelism": "pair-tile row shards x%d, gather to rank 0 (%s)" % (ctx.world, run.transport)},
                 "roofline": roofline_of(info, B, n_samples, run.n_slots, kernel_ms, launches, weighted,
                                         # (the counters in profiles/traffic.json are the generator-length runs', except
                                         # the exact unweighted kernel's, whose traffic does not depend on the lengths)
                                         None if cfg.get("lengths") == "lognormal" and int(info.kernel) != 5
                                         else traffic_of(cfg["name"], ctx.world, info, weighted),
                                         rare_ms=rare_ms_total / max(launches, 1))}
        entry["_active_fraction"] = float(info.active_fraction)
        entry["roofline"]["rows_staged"] = int(info.n_rows)   # branches some sample reaches (compaction) of B
        entry["roofline"]["timed_every"] = event_every  # the event pair brackets every n-th launch of the timed region
        entry["roofline"]["kernel_ms_between_events"] = kernel_ms_events
        if n_audit:
            uni, found, chk, headroom = run.plan.audit_detail()
            entry["audit"] = {"pairs": n_audit, "failed": bad, "worst_rel_err": worst, "uniform_sample": uni,
                              "risk_pairs_found": found, "risk_pairs_checked": chk,
                              "min_headroom": None if math.isinf(headroom) else headroom}
        if ctx.world > 1:
            # which transport this line did NOT exercise across GPUs (neither has ever crossed an xGMI link before the
            # first real N > 1 run; a rehearsal on one GPU runs the send / receive fallback over gloo, never over RCCL,
            # which refuses two ranks on one device)
            untested = ("nccl (send/recv fallback): RCCL refuses two ranks on one device, the rehearsal ran it over gloo at most"
                        if ctx.rehearse else ("nccl (send/recv fallback)" if run.transport == "ipc" else "ipc (copy engines into the root's mapped array)"))
            entry["gather"] = {"transport": run.transport, "fallback_reason": run.transport_note or None,
                               "untested_transport": untested,
                               "backend": dist.get_backend(), "world_size": dist.get_world_size(),
                               "chunks": run.chunks, "ipc_probe_GBps_rank1": per_rank[1]["ipc_probe_GBps"],
                               "exposed_ms_last_step_max": max(r["exposed_gather_ms_last_step"] for r in per_rank)}
            entry["ranks"] = per_rank
    run.close()
    return entry


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults: about a second of GPU time in the timed region, so that an outside sampler of GPU activity sees it)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="C3", help="C2|C3|C4|C5 or SAMPLESxLEAVES[@DENSITY] (e.g. 2048x5000, 8192x50000@0.01)")
    ap.add_argument("--precision", default="fixed32", choices=["auto", "fixed32", "exact64"])
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="N > 1: weak = samples grow as sqrt(N) (per-GPU pairs fixed), strong = the workload's own size; "
                         "not given (and the default workload): BASELINE configs[3] -- C4, 16,384 samples, pair tiles over "
                         "the N GPUs, strong by construction -- as the primary line and the weak-scaled C3 beside it")
    ap.add_argument("--unweighted", action="store_true")
    ap.add_argument("--lengths", default="generator", choices=["generator", "lognormal"],
                    help="branch lengths: the generator's multiples of 1/1024, or log-normal (sigma 1.5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="only the primary line")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from profiles/traffic.json only (default at N = 1: two child runs under "
                         "rocprofv3 --pmc measure it on this box once the timings are done)")
    ap.add_argument("--secondary-steps", type=int, default=5)
    ap.add_argument("--cpu-budget", type=float, default=18.0)
    ap.add_argument("--end-to-end", action="store_true", help="(kept for old command lines: end to end is in the default line)")
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="skip `end_to_end` (default at N = 1 on the default workload: ff_unifrac_dists through host buffers, "
                         "and the frcfrc executable on C3 and C4 as files)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks that all use GPU 0 with gloo as control plane (RCCL refuses two ranks on "
                         "one device): exercises the sharded code path on a one-GPU box; not a measurement")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    import frackyfrac_amd as ff

    ctx = Ctx()
    ctx.torch, ctx.dist = torch, dist
    ctx.world = world = int(os.environ.get("WORLD_SIZE", "1"))
    ctx.rank = rank = int(os.environ.get("RANK", "0"))
    ctx.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ctx.rehearse = args.rehearse_on_one_gpu
    args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    if ctx.rehearse:
        ctx.local_rank = 0
    torch.cuda.set_device(ctx.local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if ctx.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", ctx.local_rank))

    # ---- primary: BASELINE's metric on its configuration --------------------------------
    from frackyfrac_amd import synth

    # N > 1 with nothing asked for: the line leads with a BASELINE config -- configs[3], C4 at its stated size, its pair
    # tiles sharded over the N GPUs (strong scaling by construction; the N = 1 point of the same curve is C3, whose pairs
    # cost the same: same tree) -- and carries the weak-scaled C3 (samples = 4096 sqrt(N): every GPU keeps C3's pair
    # count, the regime in which the gather is easiest to hide) beside it as `weak_scaling`.
    lead_c4 = (world > 1 and args.scaling is None and args.workload == "C3" and not args.unweighted and
               args.precision == "fixed32" and args.lengths == "generator")
    scaling = "strong" if lead_c4 else (args.scaling or "weak")
    workload = "C4" if lead_c4 else args.workload
    n_override = None
    if world > 1 and scaling == "weak" and workload in synth.CONFIGS:
        n_override = int(round(synth.CONFIGS[workload]["n_samples"] * math.sqrt(world) / 32.0)) * 32
    elif world > 1 and scaling == "weak":
        n_override = int(round(int(workload.lower().split("x")[0]) * math.sqrt(world) / 32.0)) * 32
    cfg, nodes = make_problem(ctx, workload, n_override)
    weighted = cfg["weighted"] and not args.unweighted
    if args.lengths == "lognormal":
        nodes = with_lognormal_lengths(cfg, nodes)
        cfg["lengths"] = "lognormal"
    # (steps of a few microseconds -- the unweighted matrix-core kernels -- carry the event pair on every 8th launch)
    primary = measure(ctx, cfg, nodes, weighted, args.precision, args.steps, args.warmup,
                      event_every=1 if weighted or args.precision == "exact64" else 8)
    weak_entry = None
    strong_base = None
    if lead_c4:
        # the honest base of the strong curve: the SAME problem (C4) on ONE GPU of this node, this build, this run --
        # rank 0 alone reduces the whole triangle while the others wait at the barrier below (C3's pairs cost 1.6 %
        # more each than C4's on one GPU, so value(N) / (N * value(1)) with the N = 1 line's C3 would flatter the curve)
        if rank == 0:
            solo = Ctx()
            solo.torch, solo.dist, solo.world, solo.rank, solo.local_rank, solo.rehearse = torch, dist, 1, 0, ctx.local_rank, ctx.rehearse
            k1 = max(1, min(args.secondary_steps, args.steps))
            e1 = measure(solo, cfg, nodes, True, "fixed32", k1, 1)
            strong_base = {"workload": e1["config"]["workload"], "n_gpus": 1, "value": e1["value"], "unit": "pairs/s",
                           "ms_per_step": e1["ms_per_step"], "steps": k1, "kernel": e1["roofline"]["kernel"],
                           "frac": e1["roofline"]["frac"],
                           "note": "the line's own problem on one GPU of this node (rank 0, the other ranks idle): "
                                   "efficiency of the strong curve = value / (n_gpus * strong_base.value)"}
        barrier(ctx)
        del nodes
        n_weak = int(round(synth.CONFIGS["C3"]["n_samples"] * math.sqrt(world) / 32.0)) * 32
        cfg_w, nodes = make_problem(ctx, "C3", n_weak)
        weak_entry = measure(ctx, cfg_w, nodes, True, "fixed32", args.steps, args.warmup)
        if rank == 0:
            weak_entry["scaling"] = "weak"
            weak_entry.pop("_active_fraction", None)

    e2e = None
    want_e2e = rank == 0 and world == 1 and not args.no_end_to_end and not ctx.rehearse
    if want_e2e:
        # Host-buffer entry point (ff_unifrac_dists): upload of the flat nodes over PCIe,
        # staging, the pair kernels and the download of the distances.  Never `value`.
        P = ff.num_pairs(nodes.n_samples)
        out_host = np.empty(P, dtype=np.float64)
        ff.unifrac_dists(nodes, weighted, precision=args.precision, device=ctx.local_rank, out=out_host)  # warm
        t0 = time.perf_counter()
        ff.unifrac_dists(nodes, weighted, precision=args.precision, device=ctx.local_rank, out=out_host)
        e2e = time.perf_counter() - t0
        log("end-to-end through host buffers (H2D %d MB + stage + kernels + D2H %d MB): %.1f ms = %.3g pairs/s" %
            ((len(nodes.branch_id) * 12) >> 20, (P * 8) >> 20, e2e * 1e3, P / e2e))
        # the same as TEXT (ff_unifrac_text_stream: what a host that prints the distances calls): + the formatter on the
        # device, D2H of the text instead of the doubles; the callback only counts (a writer's time is not the engine's)
        import ctypes
        from frackyfrac_amd import _lib as L
        from frackyfrac_amd.api import _opts
        seen = [0]

        def count(_user, _text, n):
            seen[0] += int(n)
            return 1

        cb = L.TEXT_FN(count)
        pr, op, eb = nodes.problem(), _opts(weighted, args.precision, ctx.local_rank), L.errbuf()
        e2e_text = None
        for _ in range(2):  # (second call: warm)
            seen[0] = 0
            t0 = time.perf_counter()
            L.check(L.lib().ff_unifrac_text_stream(ctypes.byref(pr), ctypes.byref(op), 0, cb, None, eb, L.ERRLEN), eb)
            e2e_text = time.perf_counter() - t0
        e2e_text_bytes = seen[0]
        log("the same as text (formatted on the device, %d MB down): %.1f ms" % (e2e_text_bytes >> 20, e2e_text * 1e3))

    out = None
    if rank == 0:
        primary.pop("_active_fraction", None)
        out = {"metric": "sample-pairs/sec (lower triangle), weighted UniFrac 4096 samples x 10k-leaf tree"
                         if cfg["name"] == "C3" and weighted else "sample-pairs/sec (lower triangle)",
               "value": primary["value"], "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": primary["ms_per_step"], "higher_is_better": True,
               "scaling": scaling if world > 1 else "weak", "vs_baseline": None, "dtype": primary["dtype"],
               "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU, not a measurement)" if ctx.rehearse else ""),
               "config": primary["config"], "roofline": primary["roofline"]}
        for k in ("audit", "gather", "ranks"):
            if k in primary:
                out[k] = primary[k]
        if strong_base is not None:
            out["strong_base"] = strong_base
        if weak_entry is not None:
            out["weak_scaling"] = weak_entry
            out["scaling_note"] = ("N > 1: value = BASELINE configs[3] (C4, 16,384 samples) over the N GPUs, total work fixed; the "
                                   "N = 1 point is BASELINE's headline C3 (4,096 samples, same tree: the same cost per pair); "
                                   "weak_scaling = C3 grown to 4096*sqrt(N) samples, per-GPU pairs fixed")
        if e2e is not None:
            # PCIe-inclusive, informational -- never `value`
            out["end_to_end"] = {"host_buffers_ms": e2e * 1e3,
                                 "host_buffers_note": "ff_unifrac_dists on host arrays: H2D of the flat nodes (%d MB) + staging + "
                                                      "kernels + D2H of the distances (%d MB), second call in this process"
                                                      % ((len(nodes.branch_id) * 12) >> 20, (ff.num_pairs(nodes.n_samples) * 8) >> 20),
                                 "text_stream_ms": e2e_text * 1e3, "text_stream_bytes": e2e_text_bytes,
                                 "text_stream_note": "ff_unifrac_text_stream on the same host arrays: + the formatter on the device, the "
                                                     "distances come down as the lines the reference prints; the callback only counts"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(nodes, weighted, args.cpu_budget)

    # ---- secondary: the other claimed numbers, same clock ------------------------------------
    if (not args.no_secondary and args.workload == "C3" and not args.unweighted and args.precision == "fixed32" and
            args.lengths == "generator" and (world == 1 or lead_c4)):
        sec = []
        k, w = max(1, min(args.secondary_steps, args.steps)), 1
        if world == 1:
            sec.append(measure(ctx, cfg, nodes, True, "exact64", k, w))       # the reference-width figure
            sec.append(measure(ctx, cfg, nodes, False, "fixed32", max(k, args.steps), w, event_every=8))  # int8 matrix cores
            # the same with branch lengths as a real phylogeny has them -- not short binary fractions but spread over
            # orders of magnitude (log-normal, sigma 1.5; the root's stays 0).  What the engine does BY ITSELF with them
            # (precision auto): EXACT64 on pair_exact_unw_kernel, the reference's bits.  On request (fixed32): the
            # integer lengths take the 31-bit budget and the matrix-core sweep multiplies graded digit planes (DESIGN
            # 4.2), within 1e-6.
            nodes_ln = with_lognormal_lengths(cfg, nodes)
            ln = measure(ctx, dict(cfg, lengths="lognormal"), nodes_ln, False, "fixed32", max(k, args.steps), w, event_every=8)
            lx = measure(ctx, dict(cfg, lengths="lognormal"), nodes_ln, False, "auto", k, w)
            del nodes, nodes_ln
            for wl, wtd, steps, every in (("C2", False, max(k, args.steps), 8), ("C4", True, k, 1), ("C5", True, k, 1)):
                c2, n2 = make_problem(ctx, wl)
                sec.append(measure(ctx, c2, n2, wtd, "fixed32", steps, w, event_every=every))
                del n2
            sec.append(ln)  # (last: the entries before it keep the places they had in earlier rounds' lines)
            sec.append(lx)
            if rank == 0:
                # the reference's width next to the headline's: binary64 in the reference's own order of operations
                # (EXACT64), same inputs, same clock -- in the primary record, not only among the secondary ones
                out["reference_width"] = {kk: sec[0][kk] for kk in ("dtype", "value", "unit", "ms_per_step", "steps", "roofline")}
                out["reference_width"]["note"] = ("the headline is 32-bit fixed point (within 1e-6 of the reference, refined + audited); "
                                                  "this is the same workload bit for bit with the reference (FF_PRECISION_EXACT64)")
        else:
            del nodes
            c2, n2 = make_problem(ctx, "C5")    # BASELINE configs[4] at its stated size over the N GPUs
            sec.append(measure(ctx, c2, n2, True, "fixed32", k, w))
            del n2
        if rank == 0:
            for e in sec:
                e.pop("_active_fraction", None)
            out["secondary"] = sec
    if (rank == 0 and world == 1 and not args.no_secondary and args.workload == "C3" and not args.unweighted and
            args.precision == "fixed32" and args.lengths == "generator"):
        # Where real tables live: BASELINE configs[4]'s tree and sample count at 1 % and 0.2 % leaf density
        # instead of its 5-10 %.  The engine compacts the branches no sample reaches, reduces the rows few samples reach
        # over the pairs that both have them (pair_low_kernel: 97-99 % of the rows here) and keeps the rest
        # in the matrix (pair_sad_kernel; `parts`); the roofline stays priced on the UNCOMPACTED
        # 2*B lane-ops per pair, so skipping shows as a fraction above the dense kernel's -- and the
        # reference's merge walk costs O(flat nodes), not O(B), there: its rate on the SAME table stands beside it.
        sparse = []
        for wl in SPARSE_REGIME:
            try:
                c2, n2 = make_problem(ctx, wl)
                e = measure(ctx, c2, n2, True, "fixed32", max(1, min(args.secondary_steps, args.steps)), 1)
                e["active_fraction"] = e.pop("_active_fraction", None)
                e["flat_nodes_per_sample"] = len(n2.branch_id) / float(n2.n_samples)
                if not args.no_cpu_baseline:
                    e["cpu_baseline"] = cpu_baseline(n2, True, min(args.cpu_budget, 6.0))
                    e["gpu_over_cpu_all_cores"] = e["value"] / e["cpu_baseline"]["value"]
                sparse.append(e)
                del n2
            except Exception as ex:  # (the line must come out)
                sparse.append({"workload": wl, "error": "%s: %s" % (type(ex).__name__, ex)})
        out["sparse_regime"] = sparse
    if (want_e2e and args.workload == "C3" and not args.unweighted and args.precision == "fixed32" and
            args.lengths == "generator" and not under_a_profiler()):
        # the command a user runs, on files: C3 and BASELINE configs[3]'s size on this one GPU
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        e2e_runs = []
        for wl in ("C3", "C4"):
            try:
                e2e_runs.append(frcfrc_end_to_end(wl, host_cores()))
                for r in e2e_runs[-1]["runs"]:
                    log("frcfrc %s %s: wall %.2f s, phases %s" % (wl, r["flags"], r["wall_s"], r.get("seconds")))
            except Exception as e:  # (the line must come out)
                e2e_runs.append({"workload": wl, "error": "%s: %s" % (type(e).__name__, e)})
        out.setdefault("end_to_end", {})["frcfrc"] = e2e_runs
    if rank == 0 and world == 1 and not args.no_live_traffic and not ctx.rehearse:
        # last, with every timing of the line done: the primary kernel's traffic beyond L2 as this box's counters see it
        if under_a_profiler():
            out["roofline"]["traffic_note"] = "run under a profiler: no live counter passes"
        else:
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            t, why = None, None
            try:
                t, why = live_traffic(args, out["roofline"]["kernel"], kernels=out["roofline"].get("kernels"))
            except Exception as e:  # (the line must come out whatever happens to the counter passes)
                why = "%s: %s" % (type(e).__name__, e)
            if t:
                committed = out["roofline"].get("traffic")
                with_traffic(out["roofline"], t)
                if committed:
                    out["roofline"]["traffic_committed"] = committed  # (profiles/traffic.json's figure, for comparison)
            else:
                out["roofline"]["traffic_note"] = "live counter passes failed (%s): the figure is the committed one" % why
                log("live traffic: %s" % why)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
