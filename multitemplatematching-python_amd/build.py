"""
Builds libmtm_hip.so (gfx950 only) in-tree with hipcc.  Used by __graft_entry__.build(), by the
tests and by hand:  python multitemplatematching-python_amd/build.py [--force]

hipcc cross-compiles without a GPU.  The built .so is git-ignored but travels to the GPU box with
the gpurun snapshot.

The library is several translation units (context / placement / launches / search API / hit exchange, each with the
kernels only it launches; the ~300 instantiations of the MFMA score kernel alone are five more): every unit is compiled to an object of its own, in parallel, and only the units whose inputs changed
are recompiled (objects and their stamps live in csrc/build/, git-ignored).
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# MTM_BUILD_TAG: an experiment build next to the product one (own objects, libmtm_hip_<tag>.so; run it with MTM_LIB_PATH)
_TAG = os.environ.get("MTM_BUILD_TAG", "")
OBJ = os.path.join(CSRC, "build" + ("_" + _TAG if _TAG else ""))
LIB = os.path.join(HERE, "MTM", "libmtm_hip%s.so" % ("_" + _TAG if _TAG else ""))
STAMP = LIB + ".stamp"
SOURCES = ["mtm_context.hip", "mtm_placement.hip", "mtm_launch.hip", "mtm_api.hip", "mtm_comm.hip", "mtm_mfma_plain.hip", "mtm_mfma_rm.hip", "mtm_mfma_ext.hip", "mtm_mfma_kp.hip", "mtm_mfma_rows.hip", "mtm_bf16.hip",
           "mtm_host.cpp", "mtm_group.cpp"]
HEADERS = ["mtm_ctx.h", "mtm_k_image.hip.h", "mtm_k_stats.hip.h", "mtm_k_score.hip.h", "mtm_k_peaks.hip.h", "mtm_score_params.h",
           "mtm_templates_params.h", "mtm_device_util.hip.h", "mtm_mfma.hip.h", "mtm_mfma_params.h", "mtm_templates.hip.h",
           "mtm_bf16.hip.h", "mtm_bf16_params.h", "mtm_refine.hip.h", "mtm_mfma_step_asm.inc", "mtm_kernels.h", "mtm_internal.h",
           "mtm_k_nms.hip.h", "mtm_nms_core.h",
           os.path.join("..", "..", "include", "mtm_hip.h")]
# -save-temps=obj: the device assembly of every unit stays next to its object (csrc/build/*-gfx950.s) - what
# tools/spill_exec_scan.py and tests/test_abi_cpu.py::test_no_spill_ahead_of_an_exec_restore read (this compiler
# can place register spills ahead of a join block's exec restore; the build is checked for it)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-save-temps=obj",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-fvisibility=default", "-Wno-unused-command-line-argument"]
LINK_FLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-ldl", "-pthread"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build libmtm_hip.so for gfx950)")


def _extra_flags():
    return os.environ.get("MTM_EXTRA_FLAGS", "").split()     # experiments only (e.g. -DMTM_PROBE_NO_A)


def _file_digest(h, path):
    if os.path.exists(path):
        with open(path, "rb") as fh:
            h.update(fh.read())


def _unit_digest(src):
    """Everything an object depends on: its source, every header, the flags."""
    h = hashlib.sha256()
    _file_digest(h, os.path.join(CSRC, src))
    for f in HEADERS:
        _file_digest(h, os.path.join(CSRC, f))
    h.update(" ".join(FLAGS + _extra_flags()).encode())
    return h.hexdigest()


def _digest():
    h = hashlib.sha256()
    for src in SOURCES:
        h.update(_unit_digest(src).encode())
    h.update(" ".join(LINK_FLAGS).encode())
    return h.hexdigest()


def _compile(src, force, verbose):
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    stamp = obj + ".stamp"
    dig = _unit_digest(src)
    if not force and os.path.exists(obj) and os.path.exists(stamp):
        with open(stamp) as f:
            if f.read().strip() == dig:
                return obj
    cmd = [_hipcc()] + FLAGS + _extra_flags() + ["-c", os.path.join(CSRC, src), "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    # of -save-temps' by-products only the device assembly is kept (preprocessed sources and bitcode are ~10 MB per unit)
    stem = os.path.splitext(src)[0]
    for f in os.listdir(OBJ):
        if f.startswith(stem + "-") or f.startswith(stem + ".hip-"):
            if not f.endswith("-gfx950.s"):
                os.remove(os.path.join(OBJ, f))
    with open(stamp, "w") as f:
        f.write(dig)
    return obj


def build(force=False, verbose=False):
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as f:
            if f.read().strip() == dig:
                return LIB
    os.makedirs(OBJ, exist_ok=True)
    jobs = max(1, min(len(SOURCES), int(os.environ.get("MTM_BUILD_JOBS", os.cpu_count() or 1))))
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        objs = list(pool.map(lambda s: _compile(s, force, verbose), SOURCES))
    cmd = [_hipcc()] + objs + ["-o", LIB] + LINK_FLAGS
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    with open(STAMP, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
