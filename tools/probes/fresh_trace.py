#!/usr/bin/env python3
"""Host time stamps (MTM_HOST_TRACE=1) of the headline call with fresh template bytes in every call against the same call
with unchanged templates: where the ~0.1 ms between `fresh_templates` and `value` of bench.py goes.   fresh_trace.py [calls]"""
import os, sys, time
os.environ["MTM_HOST_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import synth
import MTM
from MTM import _lib

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 40
img, units, _ = synth.make_workload(seed=3, image_hw=(2160, 3840), n_base=32, templ=64, noisy_per_unit=3)
variants = []
for i in range(4):
    lt = []
    for (name, t, *rest) in units:
        t2 = t.copy(); t2[0, 0] ^= (i + 1)
        lt.append((name, t2) + tuple(rest))
    variants.append(lt)
for mode in ("fresh", "unchanged"):
    ctx = _lib.Context(0)
    _lib._default_ctx = ctx
    for i in range(6):
        MTM.matchTemplates(variants[i % 4] if mode == "fresh" else units, img, method=5, score_threshold=0.5, maxOverlap=0.25)
    ctx.reset_trace() if hasattr(ctx, "reset_trace") else None
    st = []
    for i in range(calls):
        lt = variants[i % 4] if mode == "fresh" else units
        t0 = time.perf_counter()
        MTM.matchTemplates(lt, img, method=5, score_threshold=0.5, maxOverlap=0.25)
        st.append(time.perf_counter() - t0)
    sys.stderr.write("== %s: median %.4f ms per call\n" % (mode, float(np.median(st)) * 1e3))
    sys.stderr.flush()
    _lib._default_ctx = None
    del ctx                     # the trace is printed when the context is destroyed
    import gc; gc.collect()
