#!/usr/bin/env python3
"""Photograph-like images (smooth score maps: many pixels above the threshold): per-call time, candidate counts and
the mode the library picked, against forced map mode.  GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools", "probes"))
import numpy as np
from MTM import _lib
import MTM


import synth
nat = synth.smooth_u8(11, (2160, 3840))
lt = synth.cut_templates(5, nat, 32, 64)
ctx = _lib.default_context()
if os.environ.get("DENSE_HIT_CAP"):            # e.g. 1048576: a candidate list that holds every pixel above the threshold
    ctx.set_option(_lib.OPT_HIT_CAPACITY, int(os.environ["DENSE_HIT_CAP"]))
THRS = [float(v) for v in sys.argv[1:]] or [0.5, 0.7, 0.9]
for thr in THRS:
    for honly in (1, 0):
        ctx.set_option(_lib.OPT_HITS_ONLY, honly)
        for _ in range(20):
            MTM.matchTemplates(lt, nat, score_threshold=thr, maxOverlap=0.25)
        ts, ks, modes, nh = [], [], [], []
        for _ in range(40):
            t = time.perf_counter(); h = MTM.matchTemplates(lt, nat, score_threshold=thr, maxOverlap=0.25); ts.append(time.perf_counter() - t)
            tm = ctx.timing(); ks.append(tm["ncc_kernel_ms"]); modes.append(tm["hits_only"]); nh.append(tm["n_hits"])
        print("thr %.1f  HITS_ONLY=%d: call median %.3f ms (min %.3f max %.3f), ncc kernel %.3f ms, hits-only calls %d/40, raw hits %d, final %d" % (
            thr, honly, np.median(ts) * 1e3, min(ts) * 1e3, max(ts) * 1e3, np.median(ks), sum(modes), int(np.median(nh)), len(h)), flush=True)
        print("   timing of the last call:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in ctx.timing().items()}, flush=True)
ctx.set_option(_lib.OPT_HITS_ONLY, 1)

# where the host time of a dense call goes
raw = MTM._raw_matches(lt, nat, 5, float("inf"), 0.5)
def med(f, n=20):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t)
    return np.median(ts) * 1e3, r
t_raw, raw = med(lambda: MTM._raw_matches(lt, nat, 5, float("inf"), 0.5))
t_nms, kept = med(lambda: MTM._nms_raw(raw, 0.5, False, float("inf"), 0.25))
t_list, out = med(lambda: MTM._to_hit_list(kept, lt, 0, 0))
print("dense call pieces: _raw_matches %.3f ms (%d raw hits; gpu total %.3f ms), NMS %.3f ms (%d kept), hit list %.3f ms" % (
    t_raw, len(raw), ctx.timing()["total_ms"], t_nms, len(kept), t_list))
