"""Host-side C++ under AddressSanitizer + UBSan (CPU build only: GPU sanitizers are not
available on this pool).  `make asan` builds the host sources with a device stub and a
self-test driver that feeds them golden, malformed and random inputs."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_selftest_under_asan_ubsan():
    csrc = os.path.join(ROOT, "frackyfrac_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    exe = os.path.join(ROOT, "frackyfrac_amd", "lib", "host_selftest_asan")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "host selftest ok" in r.stdout
