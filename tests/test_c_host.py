"""The drop-in boundary is a C ABI: a plain C99 program (examples/c_host.c) compiles against
include/mtm_hip.h with -pedantic, links libmtm_hip.so and runs the whole path without Python."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
LIBDIR = os.path.join(ROOT, "multitemplatematching-python_amd", "MTM")


def _build_c_host(tmp_path):
    import build as mtm_build
    mtm_build.build()
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    exe = str(tmp_path / "c_host")
    cmd = [gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "c_host.c"), "-o", exe, "-L" + LIBDIR, "-lmtm_hip", "-Wl,-rpath," + LIBDIR]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_c_host_compiles_and_refuses_without_gpu(tmp_path):
    exe = _build_c_host(tmp_path)
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present: covered by the gpu-marked test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3 and "no GPU visible" in r.stderr        # loud refusal, no CPU fallback


@pytest.mark.gpu
def test_c_host_finds_planted_patch(tmp_path):
    exe = _build_c_host(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "(x=200, y=100, w=24, h=24) score 1.000000" in r.stdout
