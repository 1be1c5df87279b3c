"""The C ABI: the shared library loads, exports every symbol include/*.h declares, and
the ctypes table binds exactly that set.  No compute calls."""
import os
import re
import subprocess

from frackyfrac_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "frackyfrac_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(ff_[a-z0-9_]+)\s*\(", text))


def test_library_exports_every_declared_symbol():
    declared = header_functions()
    assert len(declared) >= 30
    out = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    missing = declared - exported
    assert not missing, missing


def test_ctypes_table_matches_header():
    assert set(L.SIGNATURES) == header_functions()
    lib = L.lib()
    for name in L.SIGNATURES:
        assert getattr(lib, name) is not None
    assert lib.ff_version().decode().startswith("frackyfrac_amd")
    assert lib.ff_num_pairs(4096) == 8386560


def test_struct_layouts(tmp_path):
    import ctypes
    assert ctypes.sizeof(L.ff_problem) == 48
    assert ctypes.sizeof(L.ff_options) == 32
    assert ctypes.sizeof(L.ff_plan_info) == 16 + 11 * 8 + 16 + 8 + 8 + 8 + 8 + 4 * 8 + 8 + 8 + 8   # (+ the audit's four fields, + active_fraction, + rare_rows, + rare_updates)
    # the same from the header itself, as a C compiler lays it out (sizes and the offsets of the last fields)
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "frackyfrac_amd.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu", sizeof(ff_problem), sizeof(ff_options), '
                   'sizeof(ff_plan_info), offsetof(ff_plan_info, n_rows), offsetof(ff_plan_info, planes_per_sweep), '
                   'offsetof(ff_plan_info, rows_three_planes)); printf(" %zu %zu", offsetof(ff_plan_info, audit_checked), '
                   'offsetof(ff_plan_info, audit_min_headroom)); return 0; }\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    I = L.ff_plan_info
    assert got == [ctypes.sizeof(L.ff_problem), ctypes.sizeof(L.ff_options), ctypes.sizeof(I), I.n_rows.offset,
                   I.planes_per_sweep.offset, I.rows_three_planes.offset, I.audit_checked.offset, I.audit_min_headroom.offset]


def test_frcfrc_binary_links_the_library():
    out = subprocess.run(["ldd", L.FRCFRC_PATH], capture_output=True, text=True).stdout
    assert "libfrackyfrac_amd.so" in out and "not found" not in out


def test_lib_path_switch_selects_another_build(tmp_path):
    """FF_LIB_PATH: the Python layer loads the library from there (the diagnostic build of tools/*_stamps.py)."""
    import shutil
    import sys

    other = tmp_path / "libother.so"
    shutil.copy(L.LIB_PATH, other)
    code = "import frackyfrac_amd as ff, frackyfrac_amd._lib as L; L.lib(); print(L.LIB_PATH)"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT,
                       env=dict(os.environ, FF_LIB_PATH=str(other)))
    assert r.returncode == 0 and r.stdout.strip() == str(other), r.stderr
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT,
                       env=dict(os.environ, FF_LIB_PATH=str(tmp_path / "missing.so")))
    assert r.returncode != 0 and "there is no fallback path" in r.stderr
