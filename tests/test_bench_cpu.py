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
