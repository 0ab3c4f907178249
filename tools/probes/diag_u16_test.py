"""Diagnostic 'test' to run BEHIND a prefix of the GPU suite in the same process (shares the default context's state):
the uint16 scenario of test_uint16_many_templates on the default context - where exactly do the maps differ?  -s to see it."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def test_diag_u16():
    import MTM
    from MTM import _lib
    import mtm_oracle as O
    rng = np.random.default_rng(4242)
    H, W = 150, 333
    img = rng.integers(0, 65536, (H, W), dtype=np.uint16)
    img[70:100, 40:120] = 777
    lt = []
    for i in range(37):
        y, x = int(rng.integers(0, H - 20)), int(rng.integers(0, W - 70))
        t = img[y:y + 20, x:x + 70].copy()
        if i % 3 == 0:
            t = np.clip(t.astype(np.int64) + rng.integers(-2000, 2000, t.shape), 0, 65535).astype(np.uint16)
        lt.append(("w%d" % i, t))
    for i in range(18):
        y, x = int(rng.integers(0, H - 70)), int(rng.integers(0, W - 12))
        lt.append(("t%d" % i, img[y:y + 70, x:x + 12].copy()))
    f32 = img.astype(np.float32)
    units = [(t, None) for _, t in lt]
    ctx = _lib.default_context()
    exp = O.find_matches([(n, t.astype(np.float32)) for n, t in lt], f32, method=5, score_threshold=0.6)
    e = {(h[0], tuple(h[1])) for h in exp}
    for rep in range(3):
        got = MTM.findMatches(lt, img, method=5, score_threshold=0.6)
        g = {(h[0], tuple(h[1])): float(h[2]) for h in got}
        tm = ctx.timing()
        print("\nDIAG call %d: %d hits (oracle %d), extra %s, kernel_used %s hits_only %s" % (
            rep, len(g), len(e), [(k, g[k]) for k in sorted(g.keys() - e)][:5], tm["kernel_used"], tm["hits_only"]), flush=True)
    ctx.set_option(_lib.OPT_HITS_ONLY, 0)
    try:
        for rep in range(2):
            ctx.search(units, img, 5, _lib.PEAKS_LOCAL, 0.6)
            for idx in (0, 12, 20, 36, 37, 50):
                t = lt[idx][1]
                shape = (H - t.shape[0] + 1, W - t.shape[1] + 1)
                m = ctx.last_score_map(idx, shape)
                o = O.match_template(f32, t.astype(np.float32), 5)
                bad = np.argwhere(~(np.abs(m - o) <= 1e-5))
                msg = "DIAG maps rep %d template %d %s: %d of %d pixels differ" % (rep, idx, t.shape, len(bad), m.size)
                if len(bad):
                    ys, xs = bad[:, 0], bad[:, 1]
                    msg += "; rows %d..%d cols %d..%d; distinct rows %s; distinct cols %s; first %s -> got %s exp %s" % (
                        ys.min(), ys.max(), xs.min(), xs.max(), sorted(set(ys.tolist()))[:12], sorted(set(xs.tolist()))[:12],
                        bad[:4].tolist(), [float(m[tuple(b)]) for b in bad[:4]], [float(o[tuple(b)]) for b in bad[:4]])
                print(msg, flush=True)
    finally:
        ctx.set_option(_lib.OPT_HITS_ONLY, 1)
