"""bindings/go/unifrac_gpu.go cannot be compiled here (no Go toolchain), so its call sequence
-- ff_options_default -> ff_plan_create -> ff_plan_set_shard -> ff_plan_run_host per shard ->
ff_plan_destroy, replacing unifracDists (frcfrc/unifrac.go:209-228) -- is exercised by a C
program that makes exactly those calls (tests/harness/go_shim_sequence.c), compiled with gcc
against the public header alone, as cgo would."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, read_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "frackyfrac_amd", "lib")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("shim") / "go_shim_sequence")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), "-o", exe,
                    os.path.join(ROOT, "tests", "harness", "go_shim_sequence.c"), "-L", LIBDIR, "-lfrackyfrac_amd",
                    "-Wl,-rpath," + LIBDIR], check=True)
    return exe


def test_the_go_file_makes_the_calls_the_harness_makes():
    """Every C.ff_* call of the shim appears in the harness (and the harness's compute calls in the
    shim): the two cannot drift apart unnoticed."""
    go = open(os.path.join(ROOT, "bindings", "go", "unifrac_gpu.go")).read()
    c = open(os.path.join(ROOT, "tests", "harness", "go_shim_sequence.c")).read()
    go_calls = set(re.findall(r"C\.(ff_[a-z0-9_]+)\(", go))
    c_calls = set(re.findall(r"\b(ff_[a-z0-9_]+)\(", c))
    assert go_calls == {"ff_options_default", "ff_plan_create", "ff_num_pairs", "ff_plan_destroy", "ff_plan_set_shard",
                        "ff_plan_info_get", "ff_plan_run_host"}
    assert go_calls <= c_calls
    assert len(go.splitlines()) <= 110


def test_harness_builds_against_the_header_alone_and_fails_loudly_without_a_gpu(harness):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the GPU test below runs the sequence for real")
    r = subprocess.run([harness, GOLDEN + "/wtd.tree", GOLDEN + "/wtd.dense", "dense", "1"], capture_output=True, text=True)
    assert r.returncode == 2 and r.stdout == ""
    assert r.stderr.startswith("ERROR: no HIP device available") and "no CPU path" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name,weighted", [("uwtd1", 0), ("uwtd2", 0), ("wtd", 1)])
@pytest.mark.parametrize("kind", ["dense", "sparse"])
@pytest.mark.parametrize("shards", [None, 1, 3])
def test_shim_sequence_reproduces_the_reference_goldens(harness, name, weighted, kind, shards):
    args = [harness, GOLDEN + "/" + name + ".tree", GOLDEN + "/" + name + "." + kind, kind, str(weighted)]
    r = subprocess.run(args + ([str(shards)] if shards else []), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == read_golden(name + ".want")


@pytest.mark.gpu
def test_shim_sequence_on_a_synthetic_table_in_shards_and_after_a_precision_fallback(harness, tmp_path):
    from frackyfrac_amd import synth
    from oracle import oracle as O

    tree, ptr, idx, val = synth.make(200, 700, 0.15, 99)
    # replicates: the first 40 samples are one sample, so FIXED32 queues 780 pairs at distance 0
    k = int(ptr[1])
    ptr2 = np.concatenate([[0], np.cumsum([k] * 40 + list(np.diff(ptr)[40:]))]).astype(np.int64)
    idx2 = np.concatenate([np.tile(idx[:k], 40), idx[ptr[40]:]])
    val2 = np.concatenate([np.tile(val[:k], 40), val[ptr[40]:]])
    (tmp_path / "t.tree").write_text(tree.newick())
    (tmp_path / "t.sparse").write_text(synth.sparse_text(tree, ptr2, idx2, val2))
    ft = O.FlatTree(tree.names, tree.branch_len, tree.size, tree.parent)
    ip, on = O.flatten_samples(ft, ptr2, idx2, val2, 0)
    want = O.unifrac_dists(ip, on, ft.dist, True)
    for env_prec, shards in ((None, "5"), ("1", "4"), ("2", "1")):
        env = dict(os.environ)
        if env_prec:
            env["FF_SHIM_PRECISION"] = env_prec
        r = subprocess.run([harness, str(tmp_path / "t.tree"), str(tmp_path / "t.sparse"), "sparse", "1", shards],
                           capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        got = np.array([float(x) for x in r.stdout.split()])
        assert got.shape == want.shape
        zero = want == 0
        assert np.all(got[zero] == 0)
        assert np.max(np.abs(got[~zero] - want[~zero]) / want[~zero]) <= (0 if env_prec == "2" else 1e-6)
