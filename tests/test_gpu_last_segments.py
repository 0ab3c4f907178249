"""
Whole maps on partly filled LAST row segments, for every epilogue family of ncc_mfma_kernel (-m gpu).

A wave of the score kernel owns 256 consecutive outputs of a row; in the last segment of a map only a few of its 64 lanes
are inside the map, but ALL of them move accumulators through the transposition buffer on behalf of other lanes' pixels.
That is where round 5's uint16 kernel returned wrong scores depending on what earlier launches had left in scratch memory
(profiles/r05_flake/diag.txt, mechanism: profiles/r06_flake/README.md): pixels 4..6 of every 16-pixel block, for the
templates of transposition stages 1..3.  These are round 5's diagnostics (tools/probes/diag_u8_test.py, diag_u16_test.py)
as tests of the driver's suite: image width 333, so that the last segments hold 8 .. 66 outputs; whole maps against the oracle
in both normalisation modes, hit lists of the hits-only route against map mode; and the memory the kernels must not depend on
is poisoned ahead of every call (conftest's fixture does it ahead of every test, these tests repeat it with both patterns).
"""
import os

import numpy as np
import pytest

import mtm_oracle as O

pytestmark = pytest.mark.gpu

H, W = 150, 333


@pytest.fixture(scope="module")
def lib():
    import build as mtm_build
    mtm_build.build()
    from MTM import _lib
    assert _lib.load().mtm_device_count() >= 1
    return _lib


def _disc(h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    return ((((yy - h / 2 + 0.5) / (h / 2)) ** 2 + ((xx - w / 2 + 0.5) / (w / 2)) ** 2) <= 1.0).astype(np.uint8) * 255


U8_FAMILIES = [
    # name, channels, method, [(h, w)], [templates per shape], masked
    ("two_row_and_plain_m5", 1, 5, [(20, 70), (70, 12)], [37, 18], False),
    ("two_row_64wide_m5", 1, 5, [(24, 64)], [37], False),
    ("row_multiplexed_m5", 1, 5, [(24, 40)], [5], False),
    ("sqdiff_normed_m1", 1, 1, [(20, 70)], [20], False),
    ("ccorr_normed_m3", 1, 3, [(20, 70)], [37], False),
    ("raw_sums_m4", 1, 4, [(20, 70)], [20], False),
    ("raw_sums_m0", 1, 0, [(20, 70)], [20], False),
    ("rgb_m5", 3, 5, [(20, 40)], [20], False),
    ("rgb_row_multiplexed_m5", 3, 5, [(20, 40)], [5], False),
    ("rgb_m1", 3, 1, [(20, 40)], [20], False),
    ("rgb_m3", 3, 3, [(20, 40)], [20], False),
    ("generic_4ch_m5", 4, 5, [(20, 40)], [20], False),
    ("generic_2ch_m3", 2, 3, [(20, 40)], [20], False),
    ("masked_m3_plain", 1, 3, [(24, 32)], [20], True),
    ("masked_m3_row_multiplexed", 1, 3, [(24, 32)], [3], True),
    ("masked_m1", 1, 1, [(24, 32)], [20], True),
]


def _family_units(rng, img, shapes, counts, masked):
    units = []
    for (h, w), n in zip(shapes, counts):
        disc = _disc(h, w)
        for i in range(n):
            y, x = int(rng.integers(0, H - h)), int(rng.integers(0, W - w))
            t = np.ascontiguousarray(img[y:y + h, x:x + w])
            if i % 3 == 1:
                t = np.clip(t.astype(np.int32) + rng.integers(-30, 31, t.shape), 0, 255).astype(np.uint8)
            units.append((t, disc if masked else None))
    return units


@pytest.mark.parametrize("exact", [1, 0], ids=["ieee_division", "reciprocal"])
@pytest.mark.parametrize("family", U8_FAMILIES, ids=[f[0] for f in U8_FAMILIES])
def test_uint8_whole_maps_on_partly_filled_last_segments(lib, family, exact):
    name, chans, method, shapes, counts, masked = family
    rng = np.random.default_rng(777)
    img = rng.integers(0, 256, (H, W) + ((chans,) if chans > 1 else ()), dtype=np.uint8)
    units = _family_units(rng, img, shapes, counts, masked)
    n = len(units)
    ctx = lib.Context(0)
    try:
        ctx.set_option(lib.OPT_KERNEL, 3)
        ctx.set_option(lib.OPT_EXACT_DIV, exact)
        ctx.set_option(lib.OPT_HITS_ONLY, 0)
        normed = method in (1, 3, 5)
        thr = 0.5 if normed else (-1.0 if method == 0 else 1e12)       # raw sums: nothing listed (the maps are what is checked)
        picks = sorted(set([0, 3, 4, 7, 8, 12, 15, 16, 19, 20, 31, 36, 37, 41, 50, n - 1]) & set(range(n)))
        for pattern in (0xFF, 0x7F):
            ctx.debug_poison(pattern, 7)
            ctx.search(units, img, method, lib.PEAKS_LOCAL, thr)
            assert ctx.timing()["kernel_used"] == 3, name
            for idx in picks:
                t, m = units[idx]
                shape = (H - t.shape[0] + 1, W - t.shape[1] + 1)
                got = ctx.last_score_map(idx, shape)
                exp = O.match_template(img, t, method, mask=m)
                both_nan = np.isnan(got) & np.isnan(exp)
                if exact and not masked:
                    same = (got == exp) | both_nan                       # IEEE division: bit-identical to the oracle
                else:
                    tol = 1e-6 * np.maximum(1.0, np.abs(exp))
                    same = (np.abs(got.astype(np.float64) - exp) <= tol) | both_nan
                bad = np.argwhere(~same)
                assert len(bad) == 0, "%s pattern %#x template %d %s: %d wrong pixels, columns %s rows %s" % (
                    name, pattern, idx, t.shape, len(bad), sorted(set(bad[:, 1].tolist()))[:12], sorted(set(bad[:, 0].tolist()))[:6])
        if normed and not masked:
            # the hits-only route (screens, candidate list) must list what map mode lists
            ref = ctx.search(units, img, method, lib.PEAKS_LOCAL, thr).copy()
            ctx.set_option(lib.OPT_HITS_ONLY, 1)
            for pattern in (0xFF, 0x7F):
                ctx.debug_poison(pattern, 7)
                got = ctx.search(units, img, method, lib.PEAKS_LOCAL, thr).copy()
                assert got.tobytes() == ref.tobytes(), (name, pattern, len(got), len(ref))
            assert len(ref) >= n - 2          # (nearly) every template finds itself
    finally:
        del ctx


@pytest.mark.parametrize("exact", [1, 0], ids=["ieee_division", "reciprocal"])
@pytest.mark.parametrize("method", [5, 3, 1])
def test_uint16_whole_maps_on_partly_filled_last_segments(lib, method, exact):
    """The configuration that exposed the defect (test_uint16_many_templates: 37 templates 20x70 + 18 of 70x12, packed K),
    plus a 64-wide class (the non-packed instantiations - the ones tools/spill_exec_scan.py flags in round 5's unpatched
    sources)."""
    rng = np.random.default_rng(4242)
    img = rng.integers(0, 65536, (H, W), dtype=np.uint16)
    img[70:100, 40:120] = 777
    units = []
    for (h, w), cnt in (((20, 70), 37), ((70, 12), 18), ((24, 64), 21)):
        for i in range(cnt):
            y, x = int(rng.integers(0, H - h)), int(rng.integers(0, W - w))
            t = img[y:y + h, x:x + w].copy()
            if i % 3 == 0:
                t = np.clip(t.astype(np.int64) + rng.integers(-2000, 2000, t.shape), 0, 65535).astype(np.uint16)
            units.append((t, None))
    f32 = img.astype(np.float32)
    n = len(units)
    ctx = lib.Context(0)
    try:
        ctx.set_option(lib.OPT_EXACT_DIV, exact)
        ctx.set_option(lib.OPT_HITS_ONLY, 0)
        thr = 0.6 if method != 1 else 0.3
        for pattern in (0xFF, 0x7F):
            ctx.debug_poison(pattern, 7)
            ref = ctx.search(units, img, method, lib.PEAKS_LOCAL, thr).copy()
            if not os.environ.get("MTM_KERNEL"):               # (tools/alt_modes.sh forces the VALU kernels through it)
                assert ctx.timing()["kernel_used"] == 4        # MTM_KERNEL_MFMA16
            for idx in (0, 4, 12, 13, 20, 36, 37, 41, 50, 54, 55, 59, 67, n - 1):
                t = units[idx][0]
                shape = (H - t.shape[0] + 1, W - t.shape[1] + 1)
                got = ctx.last_score_map(idx, shape)
                # (corr="direct": sums of integer-valued float64 products below 2^53 are exact - the oracle's FFT route for
                # float32 input, which is what the reference turns uint16 into, carries ~1e-13 of noise: one last float32 bit
                # in half a million pixels)
                exp = O.match_template(f32, t.astype(np.float32), method, corr="direct")
                if exact:
                    same = got == exp
                else:
                    same = np.abs(got.astype(np.float64) - exp) <= 1e-6 * np.maximum(1.0, np.abs(exp))
                bad = np.argwhere(~same)
                assert len(bad) == 0, "pattern %#x template %d %s: %d wrong pixels, columns %s rows %s" % (
                    pattern, idx, t.shape, len(bad), sorted(set(bad[:, 1].tolist()))[:12], sorted(set(bad[:, 0].tolist()))[:6])
        ctx.set_option(lib.OPT_HITS_ONLY, 1)
        for pattern in (0xFF, 0x7F):
            ctx.debug_poison(pattern, 7)
            got = ctx.search(units, img, method, lib.PEAKS_LOCAL, thr).copy()
            assert got.tobytes() == ref.tobytes(), (pattern, len(got), len(ref))
    finally:
        del ctx


def test_poison_reaches_what_it_claims(lib):
    """mtm_debug_poison is only worth something if the pattern really lands: the context's per-call work buffers are read
    back (the score-map arena through mtm_last_score_map is refused after a poison - maps_valid is cleared - so this reads a
    map, poisons, and checks that the NEXT call's maps are complete and right again)."""
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (H, W), dtype=np.uint8)
    t = img[30:54, 100:164].copy()
    ctx = lib.Context(0)
    try:
        ctx.set_option(lib.OPT_HITS_ONLY, 0)
        ctx.search([(t, None)], img, 5, lib.PEAKS_LOCAL, 0.9)
        shape = (H - 24 + 1, W - 64 + 1)
        a = ctx.last_score_map(0, shape)
        ctx.debug_poison(0xFF, 7)
        with pytest.raises(lib.MtmError):
            ctx.last_score_map(0, shape)                   # the arena holds the pattern now, not maps
        ctx.search([(t, None)], img, 5, lib.PEAKS_LOCAL, 0.9)
        b = ctx.last_score_map(0, shape)
        assert np.array_equal(a, b) and not np.isnan(b).any()
        assert np.array_equal(b, O.match_template(img, t, 5))
    finally:
        del ctx


def test_quotient_without_division(lib):
    """The IEEE-division epilogues of the single-channel uint8 score kernel take (float)(num / t) - the value OpenCV's
    common_matchTemplate stores - from a reciprocal product; quotients next to a float32 rounding boundary go
    through the division (and, in the function's general form, non-zero ones below 2^-120: the epilogues' operands cannot
    produce those, and their instantiation leaves that test out - the kernel checks both forms) (csrc/mtm_device_util.hip.h: quotient_as_float).  The function itself against the division, on
    operand triples shaped like the epilogue's (square roots of integer energies, integer-valued and fractional numerators),
    half of them constructed to straddle a rounding boundary by a few ulp(double): no result may differ in any bit, the
    distance between the two float64 quotients must stay inside the bound the source states (6 ulp; the margin is 32), and
    the adversarial half must really have reached the fall-back."""
    ctx = lib.Context(0)
    try:
        total = 0
        for seed in (1, 2, 3, 4, 5, 6, 7, 8):
            r = ctx.debug_quotient_check(1 << 28, seed)
            assert r["mismatches"] == 0, r
            assert r["max_ulp_distance"] <= 6, r
            # adversarial cases (half of all) land within 4 + 6 ulp of a boundary or are tiny: all of them take the division;
            # of the random half 1.2e-7 do
            assert 0.49 * r["cases"] < r["took_division"] < 0.51 * r["cases"], r
            total += r["cases"]
        assert total >= 1 << 31
    finally:
        del ctx


@pytest.mark.parametrize("method", [5, 3, 1])
@pytest.mark.parametrize("shape", [(20, 70), (24, 64), (24, 40)], ids=["two_row", "64wide", "row_multiplexed"])
def test_flat_windows_and_saturated_quotients_whole_maps(lib, shape, method):
    """The IEEE-division epilogues keep three things off their common path (round 6): the division (quotients next to a float32
    rounding boundary take it), and the rules' constants for |num| >= t - flat windows (t == 0: black or constant regions,
    whole waves of them and single lanes at their borders), exact copies (|num| == t up to rounding: +-1) and constant
    templates.  Whole maps, bit for bit against the oracle, on an image that has all of them."""
    h, w = shape
    rng = np.random.default_rng(90 + method)
    img = rng.integers(0, 256, (H, W), dtype=np.uint8)
    img[10:80, 20:200] = 0                                   # a black region wider than a wave's 256 outputs is not possible at
    img[90:140, 150:330] = 131                               # W = 333; these cover whole 64-lane runs and their borders
    img[100:104, 160:170] = 132                              # (almost flat windows: tiny t, quotients of every size)
    units = []
    n_units = 5 if shape == (24, 40) else 20
    for i in range(n_units):
        y, x = int(rng.integers(0, H - h)), int(rng.integers(0, W - w))
        t = np.ascontiguousarray(img[y:y + h, x:x + w])      # exact copies: some of them flat, some straddling a border
        if i % 4 == 1:
            t = np.clip(t.astype(np.int32) + rng.integers(-30, 31, t.shape), 0, 255).astype(np.uint8)
        if i == 2:
            t = np.full((h, w), 200, np.uint8)               # a constant template
        if i == 3:
            t = 255 - t                                      # anti-correlated with its source: -1
        units.append((t, None))
    ctx = lib.Context(0)
    try:
        ctx.set_option(lib.OPT_KERNEL, 3)
        ctx.set_option(lib.OPT_EXACT_DIV, 1)
        ctx.set_option(lib.OPT_HITS_ONLY, 0)
        ctx.search(units, img, method, lib.PEAKS_LOCAL, 0.5)
        assert ctx.timing()["kernel_used"] == 3
        for idx, (t, _) in enumerate(units):
            got = ctx.last_score_map(idx, (H - h + 1, W - w + 1))
            exp = O.match_template(img, t, method)
            bad = np.argwhere(~((got == exp) | (np.isnan(got) & np.isnan(exp))))
            assert len(bad) == 0, (shape, method, idx, len(bad), bad[:5].tolist(), got[tuple(bad[0])], exp[tuple(bad[0])])
        # and the lists of the hits-only route are map mode's
        ref = ctx.search(units, img, method, lib.PEAKS_LOCAL, 0.5).copy()
        ctx.set_option(lib.OPT_HITS_ONLY, 1)
        got = ctx.search(units, img, method, lib.PEAKS_LOCAL, 0.5).copy()
        assert got.tobytes() == ref.tobytes(), (len(got), len(ref))
    finally:
        del ctx


@pytest.mark.parametrize("method", [3, 1])
@pytest.mark.parametrize("shape,n_units", [((24, 32), 20), ((24, 32), 3)], ids=["plain", "row_multiplexed"])
def test_masked_maps_without_square_root_and_division_equal_the_valu_kernel(lib, shape, n_units, method):
    """Masked normalised methods, IEEE mode (round 6): the matrix-core epilogue takes (float)(num / sqrt(tms c2)) from a
    product of reciprocals and sends quotients next to a float32 rounding boundary, tiny ones and non-finite ones (c2 == 0:
    OpenCV's inf and 0 / 0) through the reference sequence.  Its maps must be the VALU kernel's - which computes that
    sequence at every output - bit for bit, NaNs at the same pixels, on an image with regions that are zero under the mask."""
    h, w = shape
    rng = np.random.default_rng(600 + method)
    img = rng.integers(0, 256, (H, W), dtype=np.uint8)
    img[20:90, 30:180] = 0                                   # c2 == 0 over whole windows: NaN (0 / 0)
    img[100:140, 200:330] = 7                                # constant: quotients of one size over many outputs
    disc = _disc(h, w)
    units = []
    for i in range(n_units):
        y, x = int(rng.integers(0, H - h)), int(rng.integers(0, W - w))
        t = np.ascontiguousarray(img[y:y + h, x:x + w])
        if i % 3 == 1:
            t = np.clip(t.astype(np.int32) + rng.integers(-30, 31, t.shape), 0, 255).astype(np.uint8)
        units.append((t, disc))
    maps = {}
    for kernel in (3, 2):                                    # matrix cores, v_dot4 VALU kernel
        ctx = lib.Context(0)
        try:
            ctx.set_option(lib.OPT_KERNEL, kernel)
            ctx.set_option(lib.OPT_EXACT_DIV, 1)
            ctx.set_option(lib.OPT_HITS_ONLY, 0)
            ctx.debug_poison(0xFF, 7)
            ctx.search(units, img, method, lib.PEAKS_LOCAL, 0.8 if method == 3 else 0.2)
            used = ctx.timing()["kernel_used"]
            assert (used == 3) == (kernel == 3), used        # (masked classes of the VALU route report the float64 kernel's code)
            maps[kernel] = [ctx.last_score_map(i, (H - h + 1, W - w + 1)).copy() for i in range(n_units)]
        finally:
            del ctx
    n_nan = 0
    for i in range(n_units):
        a, b = maps[3][i], maps[2][i]
        nan_a, nan_b = np.isnan(a), np.isnan(b)
        assert np.array_equal(nan_a, nan_b), (i, int(nan_a.sum()), int(nan_b.sum()))
        assert np.array_equal(a.view(np.uint32)[~nan_a], b.view(np.uint32)[~nan_a]), (i, np.argwhere(a.view(np.uint32) != b.view(np.uint32))[:5])
        exp = O.match_template(img, units[i][0], method, mask=disc)
        ok = np.isnan(exp) == nan_a
        assert ok.all(), i
        fin = np.isfinite(exp)
        assert np.array_equal(np.isinf(a), np.isinf(exp)) and np.array_equal(a[np.isinf(a)] > 0, exp[np.isinf(exp)] > 0), i     # x / 0
        if fin.any():                                        # (a template cut from the zero region: tms == 0, x / 0 everywhere)
            assert np.abs(a[fin].astype(np.float64) - exp[fin]).max() <= 1e-6 * max(1.0, float(np.abs(exp[fin]).max()))
        n_nan += int(nan_a.sum())
    assert n_nan > 0                                         # the zero region really produced 0 / 0
