#!/usr/bin/env python3
"""Differential fuzzing of the HIP path against the oracle (GPU box): random image sizes / dtypes / channels, template
counts, sizes (up to slab-sized), masks, methods, thresholds, N_object, search boxes.  Hit lists must agree; a
difference is reported as BENIGN when every unmatched hit sits within 2e-5 of the threshold or has a partner one
pixel away with the same score (plateau ties under float noise), REAL otherwise.
Usage: fuzz_parity.py [first_seed] [n_cases]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import warnings
import numpy as np
import MTM, synth
import mtm_oracle as O

first = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].lstrip("-").isdigit() else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 100


def make_case(seed):
    rng = np.random.default_rng(50000 + seed)
    big = seed % 7 == 0                                       # >= 1 Mpx: banded uploads
    H, W = (int(rng.integers(900, 1200)), int(rng.integers(1150, 1400))) if big else (int(rng.integers(30, 420)), int(rng.integers(30, 640)))
    dtype = str(rng.choice(["uint8"] * 5 + ["uint16", "float32"]))
    if os.environ.get("FUZZ_DTYPE"):                          # e.g. FUZZ_DTYPE=float32: every case of that dtype
        dtype = os.environ["FUZZ_DTYPE"]
    chans = int(rng.choice([1, 1, 1, 3])) if dtype == "uint8" else 1
    kind = int(rng.integers(0, 3))
    if kind == 0:
        img = rng.integers(0, 256, (H, W) + ((chans,) if chans > 1 else ()), dtype=np.uint8)
    elif kind == 1:
        img = synth.smooth_u8(seed, (H, W), scales=(3, 9, 27), noise=0.1)
        if chans > 1:
            img = np.stack([np.roll(img, 3 * c, axis=1) for c in range(chans)], axis=2)
    else:
        img = rng.integers(0, 256, (H, W) + ((chans,) if chans > 1 else ()), dtype=np.uint8)
        img[H // 4:H // 2, W // 4:W // 2] = 77                # flat region
    img = np.ascontiguousarray(img)
    if dtype == "uint16":
        img = img.astype(np.uint16) * int(rng.integers(1, 257)) + int(rng.integers(0, 200))
    elif dtype == "float32":
        img = img.astype(np.float32) * np.float32(rng.uniform(0.01, 30)) + np.float32(rng.uniform(-50, 500))
    method = int(rng.choice([0, 1, 2, 3, 4, 5, 5, 5, 3, 1]))
    n_t = int(rng.integers(1, 6 if big else 45))
    lt = []
    for i in range(n_t):
        hmax, wmax = min(90, H), min(120, W)
        if not big and rng.random() < 0.08:
            wmax = min(330, W)                                # slab-sized
        h, w = int(rng.integers(2, hmax + 1)), int(rng.integers(2, wmax + 1))
        if i % 3 == 0 and lt:
            h, w = lt[-1][1].shape[:2]
        y, x = int(rng.integers(0, H - h + 1)), int(rng.integers(0, W - w + 1))
        t = img[y:y + h, x:x + w].copy()
        if i % 2:
            amp = max(1.0, float(np.ptp(img)) * 0.12)
            t = np.clip(t.astype(np.float64) + rng.uniform(-amp, amp, t.shape), 0 if dtype != "float32" else -1e9,
                        {"uint8": 255, "uint16": 65535, "float32": 1e9}[dtype]).astype(dtype)
        tup = ("t%d" % i, np.ascontiguousarray(t))
        if method in (0, 3) and chans == 1 and rng.random() < 0.25:
            m = (rng.random(t.shape[:2]) > 0.3)
            m[0, 0] = True
            tup = tup + (((m * 255).astype(np.uint8)) if dtype == "uint8" else m.astype(dtype),)
        lt.append(tup)
    if method in (1,):
        thr = float(rng.choice([0.05, 0.2, 0.4]))
    elif method in (3, 5):
        thr = float(rng.choice([0.3, 0.5, 0.8, 0.95]))
    else:
        thr = None                                            # unnormalised: N_object = 1 only
    n_obj = 1 if thr is None or rng.random() < 0.2 else float("inf")
    box = None
    if rng.random() < 0.2:
        bw, bh = int(rng.integers(max(w for _, w in [(0, t[1].shape[1]) for t in lt]), W + 1)), int(rng.integers(max(t[1].shape[0] for t in lt), H + 1))
        box = (int(rng.integers(0, W - bw + 1)), int(rng.integers(0, H - bh + 1)), bw, bh)
    return img, lt, method, thr, n_obj, box


def as_oracle(img, lt):
    f = (lambda a: a) if img.dtype == np.uint8 else (lambda a: a.astype(np.float32))
    return f(img), [(t[0],) + tuple(f(a) for a in t[1:]) for t in lt]


def classify(got, exp, thr, tol):
    """'' if the lists agree, else BENIGN / REAL with a short reason."""
    g = {(h[0], tuple(h[1])): float(h[2]) for h in got}
    e = {(h[0], tuple(h[1])): float(h[2]) for h in exp}
    bad = [k for k in g.keys() & e.keys() if abs(g[k] - e[k]) > tol * max(1.0, abs(e[k]))]
    if bad:
        k = bad[0]
        return "REAL score %s %r vs %r" % (k, g[k], e[k])
    only = [(k, g[k], e) for k in g.keys() - e.keys()] + [(k, e[k], g) for k in e.keys() - g.keys()]
    for k, v, other in only:
        near_thr = thr is not None and abs(v - thr) <= 2e-5
        (name, (x, y, w, h)) = k
        partner = any((name, (x + dx, y + dy, w, h)) in other and abs(other[(name, (x + dx, y + dy, w, h))] - v) <= tol * max(1.0, abs(v))
                      for dx in (-1, 0, 1) for dy in (-1, 0, 1) if dx or dy)
        if not (near_thr or partner):
            return "REAL unmatched %s score %r (got %d exp %d)" % (k, v, len(got), len(exp))
    return "BENIGN %d unmatched" % len(only) if only else ""


def main():
    global exact_ctx, vs_exact_cases, vs_exact_diffs
    warnings.simplefilter("ignore")
    run()


# FUZZ_VS_EXACT=1: float32 cases are ALSO run on a context with MTM_OPT_F32_MFMA = 0 (the float64 kernel) and the raw hit
# records of the default route (bf16 screen + exact re-scoring) must equal its records byte for byte - the oracle's
# float64 FFT is no judge of plateau ties (an exact-zero plateau of the FMA chain is 1e-10 noise there).
exact_ctx = None
vs_exact_cases = vs_exact_diffs = 0


def vs_exact(lt, img, kw):
    """'' or a description of the first difference between the default float32 route and the float64 kernel"""
    image, _, _ = MTM._validate_search(lt, img, kw["N_object"], kw["searchBox"])
    args = (lt, image, kw["method"], kw["N_object"], kw.get("score_threshold", 0.5))
    a = MTM._raw_matches(*args).copy()
    from MTM import _lib
    route = _lib.default_context().timing()["f32_route"]
    b = MTM._raw_matches(*args, context=exact_ctx).copy()
    if a.tobytes() == b.tobytes():
        return ""
    if len(a) != len(b):
        return "route %d: %d records vs %d of the float64 kernel" % (route, len(a), len(b))
    k = next(i for i in range(len(a)) if a[i].tobytes() != b[i].tobytes())
    return "route %d: record %d %r vs %r" % (route, k, a[k], b[k])


def run():
    global exact_ctx, vs_exact_cases, vs_exact_diffs
    if os.environ.get("FUZZ_VS_EXACT"):
        from MTM import _lib
        exact_ctx = _lib.Context(0)
        exact_ctx.set_option(_lib.OPT_F32_MFMA, 0)
    real = benign = 0
    t0 = time.time()
    for seed in range(first, first + count):
        img, lt, method, thr, n_obj, box = make_case(seed)
        if os.environ.get("FUZZ_DTYPES") and str(img.dtype) not in os.environ["FUZZ_DTYPES"].split(","):
            continue
        oimg, olt = as_oracle(img, lt)
        kw = dict(method=method, N_object=n_obj, searchBox=box)
        if thr is not None:
            kw["score_threshold"] = thr
        try:
            got = MTM.findMatches(lt, img, **kw)
            exp = O.find_matches(olt, oimg, **kw)
        except Exception as ex:                                    # noqa: BLE001
            try:
                O.find_matches(olt, oimg, **kw)
                verdict = "REAL exception %r" % (ex,)
            except Exception as ex2:                               # both refuse: same class of error?
                verdict = "" if type(ex2).__name__ == type(ex).__name__ or isinstance(ex, ValueError) else "REAL exception %r vs %r" % (ex, ex2)
            got = exp = []
        else:
            tol = 1e-4 if img.dtype == np.float32 else 2e-5
            if method in (0, 2, 4):
                # raw sums: relative to the magnitude of the sums they are differences of (the oracle's float64 FFT is itself
                # only that accurate: an exact copy gives SQDIFF ~1e-3 there, 0 in the integer paths)
                tpl = np.asarray(olt[0][1], np.float64)
                scale = max([abs(float(h[2])) for h in exp] + [1.0, float((tpl * tpl).sum())])
                g2 = [(h[0], h[1], float(h[2]) / scale) for h in got]
                e2 = [(h[0], h[1], float(h[2]) / scale) for h in exp]
                verdict = classify(g2, e2, None, 1e-5)
            else:
                verdict = classify(got, exp, thr, tol)
        if exact_ctx is not None and img.dtype == np.float32 and not (verdict.startswith("REAL exception")):
            try:
                d = vs_exact(lt, img, kw)
            except Exception as ex:                                # noqa: BLE001 - both routes refuse the same inputs
                d = ""
            vs_exact_cases += 1
            if d:
                vs_exact_diffs += 1
                print("seed %d: DIFFERS FROM THE FLOAT64 KERNEL %s" % (seed, d), flush=True)
        if verdict.startswith("REAL score") or verdict.startswith("REAL unmatched"):
            # the same call once more: a list that changes between two identical calls is a race or a read of memory the
            # library never wrote, not an arithmetic difference - say so, with the templates concerned
            try:
                again = MTM.findMatches(lt, img, **kw)
                ga = {(h[0], tuple(h[1])): float(h[2]) for h in again}
                g1 = {(h[0], tuple(h[1])): float(h[2]) for h in got}
                if ga != g1:
                    names = sorted({k[0] for k in set(ga.items()) ^ set(g1.items())})
                    from MTM import _lib as _l
                    verdict += " | NOT REPRODUCIBLE: the same call again differs in %d records (templates %s); timing %s" % (
                        len(set(ga.items()) ^ set(g1.items())), names[:8],
                        {k: v for k, v in _l.default_context().timing().items() if k in ("kernel_used", "hits_only", "f32_route", "ncc_launches")})
            except Exception as ex:                                # noqa: BLE001
                verdict += " | rerun raised %r" % (ex,)
        if verdict.startswith("REAL"):
            real += 1
        elif verdict:
            benign += 1
        if verdict:
            print("seed %d: %s | img %s %s, %d templates, method %d, thr %s, N_object %s, box %s" % (
                seed, verdict, img.shape, img.dtype, len(lt), method, thr, n_obj, box), flush=True)
    print("fuzz: %d cases from seed %d in %.0f s: %d REAL, %d BENIGN" % (count, first, time.time() - t0, real, benign))
    if exact_ctx is not None:
        print("float32 default route vs float64 kernel: %d cases, %d with different records" % (vs_exact_cases, vs_exact_diffs))
    sys.exit(1 if real or vs_exact_diffs else 0)


if __name__ == "__main__":
    main()
