"""The host-only C++ of the library (mtm_host.cpp: NMS, 1-D peaks, hit sorting, template constants; mtm_group.cpp:
worker threads and the generation-counter hand-over of the single-process multi-GPU group) under AddressSanitizer +
UndefinedBehaviorSanitizer and under ThreadSanitizer: tests/native/sanitize_host.cpp drives them with a fake per-device
context (no GPU needed) through a few thousand randomised jobs.  SURVEY section 5 / round-2 review: "no sanitizer run"."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CSRC = os.path.join(ROOT, "multitemplatematching-python_amd", "csrc")


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_host_units_under_sanitizers(tmp_path, san):
    cxx = shutil.which("g++") or shutil.which("clang++")
    if not cxx:
        pytest.skip("no host C++ compiler")
    exe = str(tmp_path / "sanitize_host")
    cmd = [cxx, "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=" + san, "-pthread",
           os.path.join(ROOT, "tests", "native", "sanitize_host.cpp"), os.path.join(CSRC, "mtm_host.cpp"),
           os.path.join(CSRC, "mtm_group.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "sanitize" in (r.stderr or "").lower() and "cannot find" in r.stderr:
        pytest.skip("sanitizer runtime not installed: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "sanitize_host: ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
