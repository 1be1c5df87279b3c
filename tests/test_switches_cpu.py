"""The engine's switches: one list (frackyfrac_amd/switches.py), held against the sources, the tests and INTEGRATION.md."""
import glob
import os
import re

from frackyfrac_amd import switches as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LISTED = {n for n, *_ in S.SWITCHES}


def _read(path):
    with open(path, errors="replace") as f:
        return f.read()


def test_every_switch_the_sources_read_is_listed_and_every_listed_one_is_read():
    product = (glob.glob(os.path.join(ROOT, "frackyfrac_amd", "csrc", "*.[ch]*")) + glob.glob(os.path.join(ROOT, "frackyfrac_amd", "*.py")) +
               [os.path.join(ROOT, "bench.py")] + glob.glob(os.path.join(ROOT, "tests", "harness", "*.c")))
    product = [p for p in product if not p.endswith("switches.py")]
    read = set()
    for p in product:
        read |= set(re.findall(r'"(FF_[A-Z0-9_]+)"', _read(p)))
    read -= {"FF_OK"}
    assert read - LISTED == set(), "read by the sources but not in frackyfrac_amd/switches.py: %s" % sorted(read - LISTED)
    assert LISTED - read == set(), "listed but read nowhere: %s" % sorted(LISTED - read)


def test_every_tuning_switch_is_exercised_by_a_test():
    tests = "".join(_read(p) for p in glob.glob(os.path.join(ROOT, "tests", "*.py")) if not p.endswith("test_switches_cpu.py"))
    tools = "".join(_read(p) for p in glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "*", "*.py")))
    missing = [n for n, d, k, w in S.SWITCHES if k == "tuning" and n not in tests]
    assert missing == [], "tuning switches no test sets: %s" % missing
    assert all(n in tests or n in tools for n, d, k, w in S.SWITCHES if k == "diagnostic")


def test_the_table_in_integration_md_is_the_generated_one():
    doc = _read(os.path.join(ROOT, "INTEGRATION.md"))
    a, b = doc.index(S.BEGIN), doc.index(S.END) + len(S.END)
    assert doc[a:b] == S.markdown(), "INTEGRATION.md section 3 is stale: paste the output of `python -m frackyfrac_amd.switches`"
