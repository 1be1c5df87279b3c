"""The frcfrc command line: flag handling, validation messages and exit codes of
frcfrc/frcfrc.go:70-88 and common/common.go:13-18.  Only paths that end before the
device is touched run here (CPU box)."""
import subprocess

import pytest

from conftest import GOLDEN
from frackyfrac_amd import _lib as L


def run(*args, stdin=None):
    return subprocess.run([L.FRCFRC_PATH, *args], input=stdin, capture_output=True, text=True)


def test_no_args_prints_usage_exit_0():
    r = run()
    assert r.returncode == 0 and r.stdout == ""
    assert r.stderr.startswith("FrackyFrac calculates UniFrac on the given abundance table.\n"
                               "Outputs one distance per line in the order (1,2),(1,3),(2,3)...(1,n)...(n-1,n).\n\nParams:\n")
    for flag in ("-i string", "-o string", "-t string", "-p int", "  -w\t", "  -s\t", "  -l\t"):
        assert flag in r.stderr


@pytest.mark.parametrize("args,msg", [
    (["-w"], "please provide a tree file with -t"),
    (["-t", "x.tree", "-p", "0"], "bad number of threads: 0"),
    (["-t", "x.tree", "-p=-3"], "bad number of threads: -3"),
    (["-t", "x.tree", "-l"], "-l can only be used with weighted unifrac"),
    (["--t=x.tree", "--l=true", "-w=false"], "-l can only be used with weighted unifrac"),
    (["-t", "x.tree", "-gpus", "0"], "bad number of GPUs: 0"),
])
def test_argument_errors(args, msg):
    r = run(*args)
    assert r.returncode == 2
    assert r.stderr == "ERROR: %s\n" % msg
    assert r.stdout == ""


def test_flag_package_errors():
    r = run("-zz")
    assert r.returncode == 2 and r.stderr.startswith("flag provided but not defined: -zz\n")
    r = run("-t")
    assert r.returncode == 2 and r.stderr.startswith("flag needs an argument: -t\n")
    r = run("-p", "abc", "-t", "x")
    assert r.returncode == 2 and r.stderr.startswith('invalid value "abc" for flag -p: parse error\n')
    r = run("-h")
    assert r.returncode == 0 and "Params:" in r.stderr


def test_input_errors_exit_2():
    r = run("-t", "/nonexistent/x.tree")
    assert r.returncode == 2 and r.stderr.startswith("Reading tree\nERROR: open /nonexistent/x.tree:")
    r = run("-t", GOLDEN + "/wtd.tree", "-i", "/nonexistent/in")
    assert r.returncode == 2 and "Loading abundances\nERROR: open /nonexistent/in:" in r.stderr
    # malformed table on stdin (default input): message of parser.go:62
    r = run("-t", GOLDEN + "/wtd.tree", stdin="s1 s2\n1\n")
    assert r.returncode == 2 and r.stderr.endswith("ERROR: has 1 values, expected 2\n")
    # species not in the tree: unifrac.go:85-88
    r = run("-t", GOLDEN + "/wtd.tree", "-s", stdin="s1:1 qq:2\n")
    assert r.returncode == 2
    assert r.stderr.endswith('Validating\nERROR: sample #1 has value 2 for species "qq" which is not in the tree\n')
    # a tree file with no tree: frcfrc.go:113
    r = run("-t", "/dev/null")
    assert r.returncode == 2 and r.stderr.endswith("ERROR: no tree in the given file\n")
