"""ctypes wrapper of oracle/libmtm_cpu.so (C++ CPU port of the reference pipeline; TEST INFRASTRUCTURE - see
oracle/cpu/mtm_cpu.cpp).  Used by bench.py's cpu_baseline leg and by tests/test_cpu_baseline_cpu.py."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HIT = np.dtype([("templ_idx", "<i4"), ("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4"), ("score", "<f4")])
_lib = None


def load():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "libmtm_cpu.so")
        if not os.path.exists(path):
            import build_oracle
            build_oracle.build()
        _lib = ctypes.CDLL(path)
        _lib.mtm_cpu_find_matches.restype = ctypes.c_int
        _lib.mtm_cpu_find_matches.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                              ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    return _lib


def find_matches(listTemplates, image, method=5, score_threshold=0.5, n_threads=None):
    """Pre-NMS hits [(label, (x, y, w, h), score)] of uint8 single-channel templates over a uint8 image, one task per
    template on n_threads threads (default round(cpu_count / 2), the reference's pool size), and the timing
    breakdown (shared precomputation, per-template phase, total) in seconds."""
    lib = load()
    image = np.ascontiguousarray(image, dtype=np.uint8)
    ts = [np.ascontiguousarray(t[1], dtype=np.uint8) for t in listTemplates]
    n = len(ts)
    ptrs = (ctypes.c_void_p * max(n, 1))(*[t.ctypes.data for t in ts])
    th = np.asarray([t.shape[0] for t in ts], dtype=np.int32)
    tw = np.asarray([t.shape[1] for t in ts], dtype=np.int32)
    if n_threads is None:
        n_threads = max(1, round((os.cpu_count() or 1) * 0.5))
    cap = 1 << 20
    out = np.empty(cap, dtype=HIT)
    n_out = ctypes.c_int64(0)
    secs = np.zeros(3)
    rc = lib.mtm_cpu_find_matches(image.ctypes.data, image.shape[0], image.shape[1], image.strides[0], ptrs, th.ctypes.data,
                                  tw.ctypes.data, n, int(method), float(score_threshold), int(n_threads), out.ctypes.data, cap,
                                  ctypes.byref(n_out), secs.ctypes.data)
    if rc != 0:
        raise RuntimeError("mtm_cpu_find_matches failed (%d)" % rc)
    hits = [(listTemplates[int(r["templ_idx"])][0], (int(r["x"]), int(r["y"]), int(r["w"]), int(r["h"])), np.float32(r["score"]))
            for r in out[:n_out.value]]
    return hits, {"shared_s": float(secs[0]), "templates_s": float(secs[1]), "total_s": float(secs[2]), "threads": int(n_threads)}
