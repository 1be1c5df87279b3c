"""bench.py's launcher, without a GPU: `python bench.py --gpus N` (no torch.distributed.run around
it, the way the driver starts the N = 1 run) starts the N ranks itself and relays their verdict."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launch_command_is_torch_distributed_run_with_one_rank_per_gpu():
    import bench

    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "5", "--warmup", "2"], 29517)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29517"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]


def test_self_launch_relays_the_ranks_exit_code_and_keeps_the_parent_off_the_gpu():
    """Here there is no GPU: every rank ends with "bench.py needs a GPU" and the parent must
    pass a non-zero code on -- after having started the ranks, i.e. without the round-1
    SystemExit("must be launched with torch.distributed.run").  FF_BENCH_TRACE_IMPORTS makes
    the parent report whether it ever imported torch."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env["FF_BENCH_TRACE_IMPORTS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0
    assert "launching 2 ranks" in r.stderr
    assert "must be launched with" not in r.stderr
    assert "bench.py needs a GPU" in r.stderr          # said by the ranks
    assert "parent imported torch: False" in r.stderr


def _fake_rocprofv3(tmp_path, body):
    """A stand-in for rocprofv3 on PATH: parses `--pmc NAME` and `-d DIR` and runs `body` (python source with NAME and
    DIR in scope) instead of profiling anything."""
    d = tmp_path / "bin"
    d.mkdir()
    exe = d / "rocprofv3"
    exe.write_text("#!%s\nimport os, sys\na = sys.argv\nNAME = a[a.index('--pmc') + 1]\nDIR = a[a.index('-d') + 1]\n%s\n"
                   % (sys.executable, body))
    exe.chmod(0o755)
    return str(d)


def test_live_traffic_reads_the_counter_files_and_falls_back_with_a_reason(tmp_path, monkeypatch):
    """bench.py's live_traffic without a GPU: the two counter passes (FETCH_SIZE, WRITE_SIZE) are child runs of rocprofv3;
    here a stand-in writes the counter file a pass would leave.  Per-launch averages of the named kernel's rows only,
    (2 * FETCH_SIZE + WRITE_SIZE) KiB; a pass that fails, finds no rows or cannot start gives None and the reason."""
    import argparse
    import bench

    args = argparse.Namespace(workload="C3", precision="fixed32", lengths="generator", unweighted=False)
    ok = '''
os.makedirs(os.path.join(DIR, "host"), exist_ok=True)
v = {"FETCH_SIZE": (1000.0, 3000.0), "WRITE_SIZE": (10.0, 30.0)}[NAME]
with open(os.path.join(DIR, "host", "77_counter_collection.csv"), "w") as f:
    f.write("Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\\n")
    f.write("1,void pair_sad_kernel12(unsigned int const*),%s,%r\\n" % (NAME, v[0]))
    f.write("2,void pair_sad_kernel12(unsigned int const*),%s,%r\\n" % (NAME, v[1]))
    f.write("3,finish_fixed32_kernel(),%s,999999.0\\n" % NAME)
    f.write("4,void pair_sad_kernel12(unsigned int const*),SOMETHING_ELSE,5.0\\n")
'''
    monkeypatch.setenv("PATH", _fake_rocprofv3(tmp_path, ok) + os.pathsep + os.environ["PATH"])
    t, why = bench.live_traffic(args, "pair_sad_kernel12")
    assert why is None and t["fetch_size_kib"] == 2000.0 and t["write_size_kib"] == 20.0
    assert t["traffic"] == (2 * 2000.0 + 20.0) * 1024.0 and t["traffic_source"].startswith("live on this box")
    # merged into a roofline record: the measured rate and the ratio to the algorithmic bytes
    rl = {"kernel_ms": 2.0, "hbm": {"algorithmic_bytes": 1024.0 * 1005.0}}
    bench.with_traffic(rl, t)
    assert abs(rl["hbm"]["traffic_ratio"] - 4.0) < 1e-12 and abs(rl["hbm"]["measured_GBps"] - t["traffic"] / 2e-3 / 1e9) < 1e-12
    # no rows for the kernel asked for
    t, why = bench.live_traffic(args, "pair_exact_unw_kernel")
    assert t is None and "no FETCH_SIZE rows" in why
    # a pass that fails
    (tmp_path / "bin" / "rocprofv3").unlink()
    (tmp_path / "bin").rmdir()
    monkeypatch.setenv("PATH", _fake_rocprofv3(tmp_path, "sys.stderr.write('counter refused\\\\n'); sys.exit(3)") + os.pathsep + os.environ["PATH"])
    t, why = bench.live_traffic(args, "pair_sad_kernel12")
    assert t is None and "code 3" in why and "counter refused" in why
    # inside a profiler nothing is started at all
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    assert bench.under_a_profiler()
    monkeypatch.delenv("LD_PRELOAD")
    assert not bench.under_a_profiler()


def test_committed_traffic_keys_name_a_kernel_and_an_existing_counter_file():
    """profiles/traffic.json is looked up by bench.py as <workload>_n<gpus>_k<ff_kernel>[_unweighted] (traffic_of): every
    key has that shape, its counter file is committed, and the sparse-regime workloads are there under the kernel that
    runs them since the rare-row split (pair_sad_kernel + pair_low_kernel: ff_kernel 0), not the sparse-aware one."""
    import json
    import re

    import bench

    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    keys = [k for k in t if not k.startswith("_")]
    assert keys
    for k in keys:
        assert re.fullmatch(r".+_n\d+_k\d+(_unweighted)?", k), k
        assert t[k]["bytes"] > 0 and os.path.exists(os.path.join(ROOT, t[k]["source"])), k
    for wl in bench.SPARSE_REGIME:
        assert "%s_n1_k0" % wl in t, wl
