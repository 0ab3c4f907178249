"""Diagnostic 'test' to run BEHIND a prefix of the GPU suite in the same process (the state that exposed round 5's uint16
bug: scratch memory full of other launches' data): uint8 scenarios whose LAST row segment is only partly inside the map,
for every kernel family whose instantiations spill registers - whole maps against the oracle.  -s to see it."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def test_diag_u8():
    from MTM import _lib
    import mtm_oracle as O
    rng = np.random.default_rng(777)
    H, W = 150, 333
    ctx = _lib.default_context()
    ctx.set_option(_lib.OPT_HITS_ONLY, 0)
    total_bad = 0
    try:
        for name, chans, method, shapes, n_t, masked in (
                ("gray m5, 37 + 18 templates (two-row + plain tilings)", 1, 5, [(20, 70), (70, 12)], [37, 18], False),
                ("gray m5, 5 templates (row-multiplexed)", 1, 5, [(24, 40)], [5], False),
                ("gray m1 (SQDIFF_NORMED)", 1, 1, [(20, 70)], [20], False),
                ("gray m3", 1, 3, [(20, 70)], [37], False),
                ("gray m0 / m2 / m4 raw sums", 1, 4, [(20, 70)], [20], False),
                ("RGB m5, 20 templates", 3, 5, [(20, 40)], [20], False),
                ("RGB m5, 5 templates (row-multiplexed)", 3, 5, [(20, 40)], [5], False),
                ("RGB m1", 3, 1, [(20, 40)], [20], False),
                ("RGB m3", 3, 3, [(20, 40)], [20], False),
                ("4 channels m5 (generic epilogue)", 4, 5, [(20, 40)], [20], False),
                ("2 channels m3 (generic epilogue)", 2, 3, [(20, 40)], [20], False),
                ("gray masked m3, 20 templates", 1, 3, [(24, 32)], [20], True),
                ("gray masked m3, 3 templates", 1, 3, [(24, 32)], [3], True)):
            img = rng.integers(0, 256, (H, W) + ((chans,) if chans > 1 else ()), dtype=np.uint8)
            units = []
            for (h, w), n in zip(shapes, n_t):
                yy, xx = np.mgrid[0:h, 0:w]
                disc = ((((yy - h / 2 + 0.5) / (h / 2)) ** 2 + ((xx - w / 2 + 0.5) / (w / 2)) ** 2) <= 1.0).astype(np.uint8) * 255
                for i in range(n):
                    y, x = int(rng.integers(0, H - h)), int(rng.integers(0, W - w))
                    t = np.ascontiguousarray(img[y:y + h, x:x + w])
                    units.append((t, disc if masked else None))
            ctx.search(units, img, method, _lib.PEAKS_LOCAL, 0.5 if method != 4 else 1e9)
            n = len(units)
            bad_c = 0
            worst = None
            for idx in sorted(set([0, 3, 4, 7, 8, 12, 15, 16, 19, 20, 31, 36, 37, 41, 50, n - 1]) & set(range(n))):
                t, m = units[idx]
                shape = (H - t.shape[0] + 1, W - t.shape[1] + 1)
                got = ctx.last_score_map(idx, shape)
                exp = O.match_template(img, t, method, mask=m)
                tol = 1e-5 if method not in (0, 2, 4) else 1e-5 * max(1.0, float(np.abs(exp).max()))
                d = ~(np.abs(got - exp) <= tol) & ~(np.isnan(got) & np.isnan(exp))
                if d.any():
                    bad_c += int(d.sum())
                    if worst is None:
                        b = np.argwhere(d)
                        worst = (idx, int(d.sum()), sorted(set(b[:, 1].tolist()))[:8], sorted(set(b[:, 0].tolist()))[:4])
            total_bad += bad_c
            print("\nDIAG8 %s: kernel_used %s, %d wrong pixels in the sampled maps %s" % (name, ctx.timing()["kernel_used"], bad_c, worst or ""), flush=True)
    finally:
        ctx.set_option(_lib.OPT_HITS_ONLY, 1)
    print("\nDIAG8 total wrong pixels: %d" % total_bad, flush=True)
