"""
Builds libmtm_hip.so (gfx950 only) in-tree with hipcc.  Used by __graft_entry__.build(), by the
tests and by hand:  python multitemplatematching-python_amd/build.py [--force]

hipcc cross-compiles without a GPU.  The built .so is git-ignored but travels to the GPU box with
the gpurun snapshot.
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "MTM", "libmtm_hip.so")
STAMP = LIB + ".stamp"
SOURCES = ["mtm_hip.hip", "mtm_host.cpp", "mtm_group.cpp"]
DEPS = SOURCES + ["mtm_device.hip.h", "mtm_mfma.hip.h", "mtm_templates.hip.h", "mtm_bf16.hip.h", "mtm_mfma_step_asm.inc", "mtm_kernels.h", "mtm_internal.h",
                  os.path.join("..", "..", "include", "mtm_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-fvisibility=default"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build libmtm_hip.so for gfx950)")


def _extra_flags():
    return os.environ.get("MTM_EXTRA_FLAGS", "").split()     # experiments only (e.g. -DMTM_PROBE_NO_A)


def _digest():
    h = hashlib.sha256()
    for f in DEPS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p):
            with open(p, "rb") as fh:
                h.update(fh.read())
    h.update(" ".join(FLAGS + _extra_flags()).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as f:
            if f.read().strip() == dig:
                return LIB
    cmd = [_hipcc()] + FLAGS + _extra_flags() + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB, "-ldl", "-pthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    with open(STAMP, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
