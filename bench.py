#!/usr/bin/env python3
"""bench.py -- throughput of the UniFrac pair reduction (the hot path) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3] [--precision fixed32]

A "step" is one pass of the hot path over one batch of synthetic input: the pair
kernels over this rank's row shard of the staged matrix (already resident in HBM)
plus, for N > 1, the gather of the result slices to rank 0 over RCCL.  N = 1 runs
BASELINE.json's headline configuration C3 (weighted UniFrac, 4096 samples x
10k-leaf tree).  For N > 1 the run is launched by torch.distributed.run, one rank
per GPU, and scales WEAKLY: the sample count grows as 4096*sqrt(N) so that every
GPU keeps C3's pair count; `value` is all ranks' pairs / max-over-ranks time.

Prints ONE JSON line on rank 0 (see the task contract); `roofline` describes the
dominant kernel (pair_sad_kernel), timed with HIP events around every launch of
the timed region (ff_plan_run_timed / ff_plan_timing_collect).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# this pool's driver only supports dmabuf IPC (cross-process device memory: RCCL, the "ipc" gather)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402

# MI355X ceilings (/opt/skills/guides/MI355X_MICROARCH.md, chip table)
HBM_PEAK_GBPS = 8000.0            # HBM3E spec
VALU_PEAK_TLANEOPS = 78.65        # 157.3 TFLOP/s FP32 vector / 2 flops per lane-op
MFMA_I8_PEAK_TOPS = 5000.0        # int8 MFMA issues at 2x the dense bf16 rate (~2.5 PFLOP/s)


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


def cpu_baseline(nodes, weighted, budget_s=15.0):
    """The oracle's merge walk (C restatement of frcfrc/unifrac.go:144-228) on the
    host cores, over a bounded prefix of the pairs in IterPairs order."""
    from oracle import oracle as O

    cores = host_cores()
    onodes = np.zeros(len(nodes.branch_id), dtype=O.FLATNODE)
    onodes["id"] = nodes.branch_id
    onodes["abnd"] = nodes.abnd
    n = nodes.n_samples
    P = n * (n - 1) // 2
    probe = min(P, 4000 * cores)
    t0 = time.perf_counter()
    O.unifrac_dists(nodes.indptr, onodes, nodes.branch_len, weighted, nthreads=cores, pair_begin=0, pair_end=probe)
    dt = max(time.perf_counter() - t0, 1e-6)
    # rows get longer as the prefix grows (later rows hold more pairs of the same cost),
    # so cost per pair is flat: size the sample for the budget from the probe rate
    count = int(min(P, max(probe, probe / dt * budget_s)))
    t0 = time.perf_counter()
    O.unifrac_dists(nodes.indptr, onodes, nodes.branch_len, weighted, nthreads=cores, pair_begin=0, pair_end=count)
    dt = time.perf_counter() - t0
    return {"value": count / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "first %d of %d pairs in IterPairs order, %.1f s, oracle/unifrac_oracle.c merge walk "
                      "(C restatement of the reference algorithm, %d threads)" % (count, P, dt, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C3", help="C2|C3|C4|C5 or SAMPLESxLEAVES (e.g. 2048x5000)")
    ap.add_argument("--precision", default="fixed32", choices=["auto", "fixed32", "exact64"])
    ap.add_argument("--unweighted", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--end-to-end", action="store_true", help="also time ff_unifrac_dists through host buffers")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks that all use GPU 0 with gloo as control plane (RCCL refuses two ranks on "
                         "one device): exercises the sharded code path on a one-GPU box; not a measurement")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import frackyfrac_amd as ff
    from frackyfrac_amd import synth
    from frackyfrac_amd.distributed import ShardedRun

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run "
                             "--nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))

    # ---- workload -----------------------------------------------------------
    if args.workload in synth.CONFIGS:
        cfg = dict(synth.CONFIGS[args.workload])
        name = args.workload
    else:
        ns, nl = args.workload.lower().split("x")
        cfg = dict(n_samples=int(ns), n_leaves=int(nl), density=0.10, weighted=True, seed=synth.SEED_BASE + 77)
        name = "custom"
    weighted = cfg["weighted"] and not args.unweighted
    base_samples = cfg["n_samples"]
    n_samples = base_samples if world == 1 else int(round(base_samples * math.sqrt(world) / 32.0)) * 32
    t0 = time.perf_counter()
    tree, ptr, idx, val = synth.make(n_samples, cfg["n_leaves"], cfg["density"], cfg["seed"])
    T = ff.parse_newick(tree.newick())
    nodes = ff.flatten_leaf_csr(T, ptr, idx, val)        # stage A on the host
    t_prep = time.perf_counter() - t0
    B = nodes.n_branches
    P = ff.num_pairs(n_samples)
    t0 = time.perf_counter()
    run = ShardedRun(nodes, weighted, rank, world, precision=args.precision, device=local_rank)
    torch.cuda.synchronize()
    t_stage = time.perf_counter() - t0
    info = run.plan.info
    if world > 1 and rank == 1 and run.ipc_gbps is not None:
        log("rank 1: ipc slice copy into the root's buffer %.1f GB/s (all peers at once)" % run.ipc_gbps)
    if rank == 0 and world > 1:
        log("gather transport: %s%s" % (run.transport, (" (ipc not used: %s)" % run.transport_note) if run.transport_note else ""))
    if rank == 0:
        log("workload %s: N=%d leaves=%d B=%d nnz=%d pairs=%d | prep %.2fs stage(H2D+quantise) %.3fs | "
            "precision=%s scale=2^%d tiles=%d items=%d wave_slots=%d" %
            (name, n_samples, cfg["n_leaves"], B, len(nodes.branch_id), P, t_prep, t_stage,
             {1: "fixed32", 2: "exact64"}[info.precision], info.scale_log2, info.n_tiles, info.n_items,
             info.n_wave_slots))

    def barrier():
        # every rank first drains its own streams (with the "ipc" transport a peer's slice
        # reaches the root from the PEER's copy stream), then all meet
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run.step()
    barrier()
    run.timing_collect()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = run.step(timed=True)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms_total, launches = run.timing_collect()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    e2e = None
    if rank == 0 and world == 1 and args.end_to_end:
        # Host-buffer entry point (ff_unifrac_dists): upload of the flat nodes over PCIe,
        # staging, the pair kernels and the download of the distances.  Never `value`.
        out_host = np.empty(P, dtype=np.float64)
        ff.unifrac_dists(nodes, weighted, precision=args.precision, device=local_rank, out=out_host)  # warm
        t0 = time.perf_counter()
        ff.unifrac_dists(nodes, weighted, precision=args.precision, device=local_rank, out=out_host)
        e2e = time.perf_counter() - t0
        log("end-to-end through host buffers (H2D %d MB + stage + kernels + D2H %d MB): %.1f ms = %.3g pairs/s" %
            ((len(nodes.branch_id) * 12) >> 20, (P * 8) >> 20, e2e * 1e3, P / e2e))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = P / (elapsed / args.steps)
        # ---- roofline of the dominant kernel (this rank's launch) -------------
        shard_pairs = run.n_slots
        kernel_ms = kernel_ms_total / max(launches, 1)
        elem_bytes = 4 if info.precision == 1 else 8
        # per pair 2*B lane-ops (subtract + |x|-accumulate per branch) and
        # 8 + (elem*N*B + 4*B + 4*N)/P bytes (one f64 result + the pair's share of one
        # compulsory read of the staged matrix, lengths and row sums)
        from frackyfrac_amd._lib import KERNEL_NAMES
        alg_bytes = 8.0 * shard_pairs + elem_bytes * float(n_samples) * B + 4.0 * B + 4.0 * n_samples
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and info.kernel in (0, 3) and weighted:
            try:
                traffic = json.load(open(tpath)).get("%s_n%d" % (name, world))
            except Exception:
                traffic = None
        hbm = {"bound": "hbm", "achieved": alg_bytes / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS,
               "unit": "GB/s", "frac": alg_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
               "algorithmic_bytes": alg_bytes}
        if info.kernel == 2:
            # unweighted on the matrix cores: one multiply-add per branch and pair is the
            # algorithmic work (the base-128 digit passes are the implementation's)
            flops = 2.0 * B * shard_pairs
            achieved = flops / (kernel_ms * 1e-3) / 1e12
            roofline = {"bound": "mfma", "achieved": achieved, "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s",
                        "frac": achieved / MFMA_I8_PEAK_TOPS, "traffic": traffic, "kernel": KERNEL_NAMES[2],
                        "kernel_ms": kernel_ms, "launches": launches, "digits": int(info.n_digits),
                        "algorithmic": "2*B int8 MAC-ops per pair, B=%d, %d pairs per launch" % (B, shard_pairs),
                        "hbm": hbm}
        elif info.kernel == 1:
            # EXACT64: the reference's roundings need six unfused binary64 operations per branch and
            # pair (numer: sub, mul by |.|, add; denom: add, mul, add; unifrac.go:191-192); the FP64
            # vector rate is half the FP32 one (MI355X: 78.6 TFLOP/s FP64 vector = 39.3e12 ops/s)
            ops = 6.0 * B * shard_pairs
            achieved_t = ops / (kernel_ms * 1e-3) / 1e12
            roofline = {"bound": "valu", "achieved": achieved_t, "peak": VALU_PEAK_TLANEOPS / 2, "unit": "T f64 op/s",
                        "frac": achieved_t / (VALU_PEAK_TLANEOPS / 2), "traffic": traffic,
                        "kernel": KERNEL_NAMES[1], "kernel_ms": kernel_ms, "launches": launches,
                        "algorithmic": "6*B unfused binary64 ops per pair, B=%d, %d pairs per launch" % (B, shard_pairs),
                        "hbm": hbm}
        else:
            laneops = 2.0 * B * shard_pairs
            achieved_tl = laneops / (kernel_ms * 1e-3) / 1e12
            roofline = {"bound": "valu", "achieved": achieved_tl, "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s",
                        "frac": achieved_tl / VALU_PEAK_TLANEOPS, "traffic": traffic,
                        "kernel": KERNEL_NAMES[int(info.kernel)], "kernel_ms": kernel_ms, "launches": launches,
                        "algorithmic": "2*B lane-ops per pair (SURVEY 8d), B=%d, %d pairs per launch" % (B, shard_pairs),
                        "hbm": hbm}
        out = {"metric": "sample-pairs/sec (lower triangle), weighted UniFrac 4096 samples x 10k-leaf tree"
                         if name == "C3" and weighted else "sample-pairs/sec (lower triangle)",
               "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "u32" if info.precision == 1 else "f64",
               "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU, not a measurement)" if args.rehearse_on_one_gpu else ""),
               "config": {"workload": "%s: %d samples x %d-leaf Yule tree (B=%d branches), %s UniFrac, "
                                      "leaf density %.2f, seed 0x%X" %
                                      (name, n_samples, cfg["n_leaves"], B, "weighted" if weighted else "unweighted",
                                       cfg["density"], cfg["seed"]),
                          "pairs": P, "precision": {1: "fixed32", 2: "exact64"}[info.precision],
                          "parallelism": "pair-tile row shards x%d, gather to rank 0 (%s)" % (world, run.transport)},
               "roofline": roofline}
        if e2e is not None:
            out["host_buffers_ms"] = e2e * 1e3  # PCIe-inclusive, informational
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(nodes, weighted, args.cpu_budget)
        # cheap sanity on the result of the last step (not a parity test: tests/ does that)
        # (every slot, so a slice that never arrived from its rank cannot go unnoticed)
        lo, hi = float(res.min().item()), float(res.max().item())
        assert not bool(torch.isnan(res).any().item()), "NaN in the gathered result"
        assert 0.0 <= lo and hi <= 1.0000001, "distance outside [0, 1]"
        print(json.dumps(out), flush=True)
    run.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
