"""SURVEY.md 5 (race / memory-error detection): the reference has none; here the HOST side of the library - argument
validation, the plan / batch bookkeeping, the host twin of the solve, the launch wrappers up to the HIP call - is compiled without
device code and with AddressSanitizer + UndefinedBehaviorSanitizer (`make -C csrc host-asan`) and tests/test_host_logic.py runs
against that build in a child interpreter with the sanitizer runtime preloaded.  (GPU sanitizers are not available on this pool.)"""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "hyperspectral_super-resolution_amd", "csrc")


@pytest.mark.skipif(os.environ.get("HSR_ASAN_CHILD") == "1", reason="already inside the sanitizer run")
def test_host_logic_under_address_and_ub_sanitizers():
    r = subprocess.run(["make", "-C", CSRC, "host-asan", "-j4"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rt = [ln.split("=", 1)[1].strip() for ln in r.stdout.splitlines() if ln.startswith("ASAN_RT=")][-1]
    lib = os.path.join(ROOT, "hyperspectral_super-resolution_amd", "lib", "libhsr_host_asan.so")
    assert os.path.isfile(rt) and os.path.isfile(lib)
    env = dict(os.environ, HSR_LIBRARY=lib, LD_PRELOAD=rt, HSR_ASAN_CHILD="1",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",       # (CPython itself never frees everything)
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    t = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_host_logic.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    tail = (t.stdout + t.stderr)[-3000:]
    assert t.returncode == 0, tail
    assert "passed" in t.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    # the sanitized build really was the library under test
    probe = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from s2_emit import _native as n; n.load(); print(n.library_path())"
                            % os.path.join(ROOT, "hyperspectral_super-resolution_amd")], capture_output=True, text=True, timeout=120, env=env)
    assert probe.returncode == 0 and probe.stdout.strip().endswith("libhsr_host_asan.so"), probe.stdout + probe.stderr
