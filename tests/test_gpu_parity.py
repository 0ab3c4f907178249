"""
GPU parity tests (run with -m gpu on an MI355X): every test drives the HIP kernels through the
C ABI (ctypes) and checks them against the CPU oracle on the same inputs, against the committed
golden fixtures, and - at BASELINE.json sizes - through size-independent properties.

Tolerance for score maps: |ours - oracle| <= 1e-4 * max(1, |oracle|) (north_star).  The uint8
path is exact integer arithmetic with a float64 epilogue in the oracle's operation order, so it is
in fact held to 1e-6 here.
"""
import os
import sys

import numpy as np
import pytest

import mtm_oracle as O
import synth
from helpers import (assert_hits_equal, canon, coin_templates, hits_json, load_coins, load_golden)

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

G = load_golden()
REF = G["reference_run"]
KERNELS = {"naive": 1, "dot4": 2, "mfma": 3, "auto": 0}


@pytest.fixture(scope="module")
def mtm():
    import build as mtm_build
    mtm_build.build()
    import MTM
    assert MTM._lib.load().mtm_device_count() >= 1, "no GPU visible to libmtm_hip"
    return MTM


@pytest.fixture(scope="module")
def ctx(mtm):
    return mtm._lib.default_context()


@pytest.fixture(scope="module")
def coins():
    return load_coins()


def hits_of(oracle_hits):
    """Oracle hits in the package's output form (tuples, np.float32 scores)."""
    return [(h[0], tuple(int(v) for v in h[1]), np.float32(h[2])) for h in oracle_hits]


def set_kernel(ctx, name):
    """"auto" = what a fresh context starts with: the library's choice - or the kernel tools/alt_modes.sh forces through
    MTM_KERNEL for this run of the suite (the tests' `finally: set_kernel(ctx, "auto")` must put THAT back)."""
    if name == "auto":
        from conftest import fresh_options
        ctx.set_option(1, fresh_options()[1])
    else:
        ctx.set_option(1, KERNELS[name])


def restore_hits_only(ctx):
    """What a fresh context starts with (1 - or the 0 of tools/alt_modes.sh's MTM_HITS_ONLY=0 run); setting the option also
    clears the dense-map back-off."""
    from conftest import fresh_options
    from MTM import _lib
    ctx.set_option(_lib.OPT_HITS_ONLY, fresh_options()[_lib.OPT_HITS_ONLY])


def map_close(got, exp, tol=1e-4):
    assert got.shape == exp.shape and got.dtype == np.float32
    err = np.abs(got.astype(np.float64) - exp.astype(np.float64))
    bound = tol * np.maximum(1.0, np.abs(exp.astype(np.float64)))
    bad = err > bound
    assert not bad.any(), "max err %.3g at %s" % (err.max(), np.unravel_index(np.argmax(err), err.shape))
    return float(err.max())


def set_exact(ctx, on):
    """MTM_OPT_EXACT_DIV: IEEE division in the MFMA epilogue (bit-identical to the other kernels; the library's default
    since round 5), 0 = the correctly rounded reciprocals of rounds 1-4."""
    ctx.set_option(5, 1 if on else 0)


def restore_exact(ctx):
    """Back to what a fresh context starts with (conftest.fresh_options: the shipped default, 1, unless an environment
    switch of tools/alt_modes.sh says otherwise).  Round 5's suite restored `False` here and so ran everything behind
    test_guards in the reciprocal mode - the mode that no longer ships as the default."""
    from conftest import fresh_options
    ctx.set_option(5, fresh_options()[5])


def restore_options(ctx):
    from conftest import fresh_options
    for opt, value in fresh_options().items():
        ctx.set_option(opt, value)


def ulp_close(a, b, tol=1.2e-7):
    """float32 maps equal up to the last bit of values <= 1 (the default MFMA epilogue multiplies by
    correctly rounded reciprocals instead of dividing)."""
    assert a.shape == b.shape
    err = np.abs(a.astype(np.float64) - b.astype(np.float64))
    assert float(err.max()) <= tol * max(1.0, float(np.abs(b).max())), float(err.max())


def hits_close(a, b, tol=2e-7):
    assert [(h[0], h[1]) for h in a] == [(h[0], h[1]) for h in b]
    assert all(abs(float(x[2]) - float(y[2])) <= tol * max(1.0, abs(float(y[2]))) for x, y in zip(a, b))


def otsu_mask(small):
    return ((small > G["otsu_threshold"]) * 255).astype(np.uint8)


def default_routes():
    """False while tools/alt_modes.sh runs the suite with an environment switch that moves work to another kernel or
    mode: assertions about WHICH route a call took only hold for the default routes (results are asserted always)."""
    return not any(os.environ.get(k) for k in ("MTM_KERNEL", "MTM_HITS_ONLY", "MTM_F32_MFMA", "MTM_TEMPL_ON_DEVICE", "MTM_ROW_MUX",
                                               "MTM_FUSE_STATS"))


# ------------------------------------------------------------------------------------------------
# score maps
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kernel", ["naive", "dot4", "mfma"])
@pytest.mark.parametrize("method", [0, 1, 2, 3, 4, 5])
def test_score_map_coins_u8(mtm, ctx, coins, method, kernel):
    set_kernel(ctx, kernel)
    try:
        for t in coin_templates(coins):
            got = mtm.computeScoreMap(t, coins, method)
            exp = O.compute_score_map(t, coins, method)
            map_close(got, exp, tol=1e-6)
    finally:
        set_kernel(ctx, "auto")


@pytest.mark.parametrize("method", [0, 3])
def test_score_map_masked(mtm, coins, method):
    small, _ = coin_templates(coins)
    mask = otsu_mask(small)
    got = mtm.computeScoreMap(small, coins, method, mask=mask)
    exp = O.compute_score_map(small, coins, method, mask=mask)
    map_close(got, exp, tol=1e-6)
    # float32 image + float32 weights mask
    imf = coins.astype(np.float32) / 255
    wm = np.linspace(0, 1, small.size, dtype=np.float32).reshape(small.shape)
    got = mtm.computeScoreMap(imf[37:75, 80:121], imf, method, mask=wm)
    exp = O.compute_score_map(imf[37:75, 80:121], imf, method, mask=wm)
    map_close(got, exp, tol=1e-5)


@pytest.mark.parametrize("method", [1, 3, 5])
def test_score_map_float32_uint16_rgb(mtm, ctx, coins, method):
    img16 = coins.astype(np.uint16) * 257
    map_close(mtm.computeScoreMap(img16[37:75, 80:121], img16, method),
              O.compute_score_map(img16[37:75, 80:121], img16, method), tol=1e-5)
    imf = coins.astype(np.float32) / 255.0
    map_close(mtm.computeScoreMap(imf[14:73, 302:367], imf, method),
              O.compute_score_map(imf[14:73, 302:367], imf, method), tol=5e-5)        # bf16-piece kernel (MTM_F32_MFMA)
    rgb = np.stack([coins, np.roll(coins, 3, axis=1), 255 - coins], axis=2)
    t = np.ascontiguousarray(rgb[37:75, 80:121])
    for kernel in ("naive", "dot4", "mfma"):
        set_kernel(ctx, kernel)
        try:
            map_close(mtm.computeScoreMap(t, rgb, method), O.compute_score_map(t, rgb, method), tol=1e-6)
        finally:
            set_kernel(ctx, "auto")


def test_score_map_golden_fixture(mtm, coins):
    """HIP path against the committed fixture (reference run), not just against the live oracle."""
    small, big = coin_templates(coins)
    from helpers import GOLDEN_DIR
    sub = np.load(GOLDEN_DIR + "/coins_maps_sub3.npz")
    for name, t in (("small", small), ("big", big)):
        for m in (1, 2, 3, 4, 5):
            got = mtm.computeScoreMap(t, coins, m)
            np.testing.assert_allclose(got[::3, ::3], sub["%s_m%d" % (name, m)], rtol=1e-5, atol=1e-5)
    got = mtm.computeScoreMap(small, coins, 3, mask=otsu_mask(small))
    np.testing.assert_allclose(got[::3, ::3], sub["small_m3_mask"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("shape,tshape", [((200, 300), (100, 70)), ((150, 333), (33, 129)), ((400, 400), (300, 260)),
                                          ((97, 131), (97, 10)), ((97, 131), (5, 131)), ((64, 64), (64, 64)),
                                          ((70, 90), (1, 1)), ((300, 500), (130, 3))])
def test_score_map_shapes(mtm, ctx, shape, tshape):
    """chunked templates (> 64), widths not multiple of 4, uint64 accumulation (300x260), 1-D and
    1x1 maps, template == image."""
    img = synth.rand_u8(11, 1, shape)
    y0, x0 = (shape[0] - tshape[0]) // 2, (shape[1] - tshape[1]) // 3
    t = np.ascontiguousarray(img[y0:y0 + tshape[0], x0:x0 + tshape[1]])
    for method in (1, 3, 5):
        exp = O.compute_score_map(t, img, method)
        for kernel in ("naive", "dot4", "mfma"):
            set_kernel(ctx, kernel)
            try:
                map_close(mtm.computeScoreMap(t, img, method), exp, tol=1e-6)
            finally:
                set_kernel(ctx, "auto")


def test_guards(mtm):
    img = np.full((40, 50), 7, dtype=np.uint8)
    img[20:, :] = 9
    t = np.full((8, 8), 7, dtype=np.uint8)
    assert np.all(mtm.computeScoreMap(t, img, 5) == 1.0)       # constant template
    t2 = img[16:24, 10:18].copy()
    ctx = mtm._lib.default_context()
    for m in (1, 3, 5):
        exp = O.compute_score_map(t2, img, m)
        assert np.array_equal(mtm.computeScoreMap(t2, img, m), exp), m         # the default (IEEE division): bit-identical
        set_exact(ctx, False)                                                  # the reciprocal epilogue of rounds 1-4
        try:
            got = mtm.computeScoreMap(t2, img, m)
        finally:
            restore_exact(ctx)
        ulp_close(got, exp)
        special = (exp == 0.0) | (np.abs(exp) == 1.0)               # flat windows, saturation branches
        assert np.array_equal(got[special], exp[special]), m
    black = np.zeros((30, 30), np.uint8)
    assert np.array_equal(mtm.computeScoreMap(black[:5, :5], black, 1), O.compute_score_map(black[:5, :5], black, 1))


def test_dot4_variants_agree(mtm, ctx):
    img, units, _ = synth.make_workload(seed=21, image_hw=(400, 900), n_base=7, templ=48)
    ctx.set_image(img)
    ctx.set_templates([(u[1], None) for u in units], 5)
    shape = (400 - 48 + 1, 900 - 48 + 1)
    set_kernel(ctx, "naive")
    base = [ctx.score_map(i, shape) for i in range(len(units))]
    set_kernel(ctx, "dot4")
    try:
        for v in range(7):
            ctx.set_option(4, v)
            # batched path (all templates in one launch) through find_matches, then single maps
            hits = ctx.find_matches(0, 0.5)
            assert len(hits) == 4 * len(units), (v, len(hits))
            for i in range(len(units)):
                assert np.array_equal(ctx.score_map(i, shape), base[i]), (v, i)
        set_kernel(ctx, "mfma")
        set_exact(ctx, False)
        hits = ctx.find_matches(0, 0.5)
        assert len(hits) == 4 * len(units) and ctx.timing()["kernel_used"] == 3
        for i in range(len(units)):
            ulp_close(ctx.score_map(i, shape), base[i])
        set_exact(ctx, True)
        for i in range(len(units)):
            assert np.array_equal(ctx.score_map(i, shape), base[i]), ("mfma exact", i)
    finally:
        ctx.set_option(4, 0)
        restore_exact(ctx)
        set_kernel(ctx, "auto")


@pytest.mark.parametrize("n_templ,side,method", [(37, 24, 5), (16, 40, 3), (17, 33, 1), (70, 16, 5)])
def test_mfma_template_groups(mtm, ctx, n_templ, side, method):
    """MFMA kernel: 16-template groups (MB = 1 and 2), partial last group, several groups, sizes
    that are not multiples of 16/64; batched hits and single maps against the naive kernel."""
    img = synth.rand_u8(31, 0, (260, 610))
    rng = np.random.default_rng(n_templ)
    units = []
    for i in range(n_templ):
        y, x = int(rng.integers(0, 260 - side)), int(rng.integers(0, 610 - side))
        units.append(("t%d" % i, np.ascontiguousarray(img[y:y + side, x:x + side])))
    thr = {5: 0.6, 3: 0.95, 1: 0.2}[method]
    res = {}
    for kernel in ("naive", "mfma", "dot4", "mfma_exact"):
        set_kernel(ctx, kernel.split("_")[0])
        set_exact(ctx, kernel.endswith("exact"))
        try:
            res[kernel] = mtm.findMatches(units, img, method=method, score_threshold=thr)
            if kernel != "naive":
                for i in (0, n_templ // 2, n_templ - 1):
                    set_kernel(ctx, "naive")
                    a = mtm.computeScoreMap(units[i][1], img, method)
                    set_kernel(ctx, kernel.split("_")[0])
                    b = mtm.computeScoreMap(units[i][1], img, method)
                    if kernel == "mfma":
                        ulp_close(b, a)
                    else:
                        assert np.array_equal(a, b), (kernel, i)
        finally:
            restore_exact(ctx)
            set_kernel(ctx, "auto")
    assert len(res["naive"]) >= n_templ
    assert res["mfma_exact"] == res["naive"] and res["dot4"] == res["naive"]
    hits_close(res["mfma"], res["naive"])
    exp = O.find_matches(units[:3], img, method=method, score_threshold=thr)
    assert_hits_equal([h for h in res["mfma"] if h[0] in ("t0", "t1", "t2")], hits_json(exp), tol=1e-6)


# ------------------------------------------------------------------------------------------------
# full API against the golden fixtures
# ------------------------------------------------------------------------------------------------
def test_notebook_goldens(mtm, coins):
    small, _ = coin_templates(coins)
    assert_hits_equal(mtm.matchTemplates([("small", small)], coins, score_threshold=0.5, method=5, maxOverlap=0),
                      G["notebook_G1"]["hits"], tol=1e-4)
    assert_hits_equal(mtm.matchTemplates([("testMask", small)], coins, method=3, score_threshold=0.8, maxOverlap=0),
                      G["notebook_G2"]["hits"], tol=1e-4)
    assert_hits_equal(mtm.matchTemplates([("testMask", small, otsu_mask(small))], coins, method=3, score_threshold=0.8, maxOverlap=0),
                      G["notebook_G3"]["hits"], tol=1e-4)


CALLS = {
    "G1": lambda im, s, b: ([("small", s)], dict(score_threshold=0.5, method=5, maxOverlap=0)),
    "testpy": lambda im, s, b: ([("small", s), ("big", b)], dict(score_threshold=0.3, method=5, maxOverlap=0)),
    "tut1_two": lambda im, s, b: ([("small", s), ("large", b)], dict(score_threshold=0.4, method=5, maxOverlap=0)),
    "sqdiff_normed": lambda im, s, b: ([("small", s), ("big", b)], dict(method=1, score_threshold=0.2, maxOverlap=0)),
    "overlap025": lambda im, s, b: ([("small", s), ("big", b)], dict(score_threshold=0.3, method=5, maxOverlap=0.25)),
    "nobj3": lambda im, s, b: ([("small", s), ("big", b)], dict(score_threshold=0.3, method=5, maxOverlap=0.25, N_object=3)),
    "nobj1": lambda im, s, b: ([("small", s)], dict(method=5, N_object=1)),
    "nobj1_sqdiff": lambda im, s, b: ([("big", b)], dict(method=1, N_object=1)),
    "nobj0": lambda im, s, b: ([("small", s), ("big", b)], dict(score_threshold=0.3, method=5, N_object=0)),
    "searchbox_exact": lambda im, s, b: ([("big", b)], dict(searchBox=(302, 14) + b.shape[::-1])),
    "searchbox": lambda im, s, b: ([("small", s)], dict(score_threshold=0.5, maxOverlap=0, searchBox=(10, 20, 300, 200))),
    "full_image": lambda im, s, b: ([("all", im)], dict()),
}


@pytest.mark.parametrize("name", sorted(CALLS))
def test_reference_run_match_templates(mtm, coins, name):
    small, big = coin_templates(coins)
    templates, kw = CALLS[name](coins, small, big)
    hits = mtm.matchTemplates(templates, coins, **kw)
    if len(templates) > 1:
        assert_hits_equal(canon(hits), canon([(h[0], tuple(h[1]), h[2]) for h in REF[name]]), tol=1e-5)
    else:
        assert_hits_equal(hits, REF[name], tol=1e-5)
    assert all(isinstance(h[2], np.float32) and isinstance(h[1][0], int) for h in hits)


def test_reference_run_misc(mtm, coins):
    small, big = coin_templates(coins)
    tall, wide = coins[:, 100:141], coins[50:90, :]
    assert_hits_equal(canon(mtm.findMatches([("tall", tall)], coins, score_threshold=0.5)), REF["tall"], tol=1e-5)
    assert_hits_equal(canon(mtm.findMatches([("wide", wide)], coins, score_threshold=0.5)), REF["wide"], tol=1e-5)
    img16 = coins.astype(np.uint16) * 257
    assert_hits_equal(mtm.matchTemplates([("small", img16[37:75, 80:121])], img16, score_threshold=0.5, method=5, maxOverlap=0),
                      REF["uint16"], tol=1e-5)
    imgf = coins.astype(np.float32) / 255.0
    assert_hits_equal(mtm.matchTemplates([("small", imgf[37:75, 80:121])], imgf, method=3, score_threshold=0.95, maxOverlap=0.1),
                      REF["float32_m3"], tol=1e-5)
    assert_hits_equal(canon(mtm.findMatches([("small", small), ("big", big)], coins, score_threshold=0.3)),
                      REF["find_pre_nms"], tol=1e-5)
    rgb = np.stack([coins, np.roll(coins, 3, axis=1), 255 - coins], axis=2)
    assert_hits_equal(mtm.matchTemplates([("small", np.ascontiguousarray(rgb[37:75, 80:121]))], rgb, score_threshold=0.5, method=5, maxOverlap=0),
                      REF["rgb"], tol=1e-5)


def test_method0_raises_after_compute(mtm, coins):
    small, _ = coin_templates(coins)
    with pytest.raises(ValueError, match="The method TM_SQDIFF is not supported"):
        mtm.matchTemplates([("small", small)], coins, method=0)
    # findMatches with method 0 works (local minima of the raw difference)
    got = mtm.findMatches([("small", small)], coins, method=0, score_threshold=1e6)
    exp = O.find_matches([("small", small)], coins, method=0, score_threshold=1e6)
    assert_hits_equal(got, hits_json(exp), tol=1e-5)


def test_mask_warnings(mtm, coins):
    small, _ = coin_templates(coins)
    mask = otsu_mask(small)
    with pytest.warns(UserWarning, match="not supporting the use of Mask"):
        a = mtm.matchTemplates([("s", small, mask)], coins, method=5, maxOverlap=0)
    assert_hits_equal(a, [["s"] + h[1:] for h in REF["G1"]], tol=1e-5)
    with pytest.warns(UserWarning, match="same dimension or bit depth"):
        mtm.computeScoreMap(small, coins, 3, mask=mask[:-1])
    with pytest.warns(UserWarning, match="not compatible with use of mask"):
        mtm.computeScoreMap(small, coins, 5, mask=mask)


@pytest.mark.parametrize("name", sorted(G["synthetic"]))
def test_synthetic_reference_runs(mtm, name):
    case = G["synthetic"][name]
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in case["kwargs"].items()}
    img, units, plants = synth.make_workload(**kw)
    pre = mtm.findMatches(units, img, method=case["method"], score_threshold=case["score_threshold"])
    assert_hits_equal(canon(pre), case["pre_nms"], tol=1e-5)
    post = mtm.matchTemplates(units, img, method=case["method"], score_threshold=case["score_threshold"], maxOverlap=0.25)
    assert_hits_equal(canon(post), case["post_nms"], tol=1e-5)


# ------------------------------------------------------------------------------------------------
# peak extraction details
# ------------------------------------------------------------------------------------------------
def test_peaks_vs_oracle_low_threshold(mtm, ctx, coins):
    """Thousands of peaks, plateaus (saturated +-1 scores), both border rules, minima and maxima,
    and the growing device hit buffer."""
    small, big = coin_templates(coins)
    lt = [("small", small), ("big", big)]
    for border, code in (("constant", 0), ("nearest", 1)):
        ctx.set_option(2, code)
        ctx.set_option(3, 64)          # tiny device hit buffer: forces the grow-and-rerun path
        try:
            for method, thr in ((5, -0.5), (5, 0.05), (3, 0.6), (1, 0.9), (1, 0.05)):
                got = mtm.findMatches(lt, coins, method=method, score_threshold=thr)
                exp = O.find_matches(lt, coins, method=method, score_threshold=thr, border=border)
                assert len(got) == len(exp) and (len(got) > 64 or thr == 0.05)
                assert_hits_equal(canon(got), canon(exp), tol=1e-5)
        finally:
            ctx.set_option(2, 1)          # back to the default (edge replication)
            ctx.set_option(3, 1 << 18)


def test_peaks_trivial_and_plateau(mtm):
    img = np.zeros((60, 80), np.uint8)
    img[20:30, 30:40] = 200
    t = np.zeros((10, 10), np.uint8)
    t[:, :] = 200
    t[0, 0] = 0
    for method, thr in ((3, 0.5), (5, 0.5), (1, 0.5)):
        got = mtm.findMatches([("t", t)], img, method=method, score_threshold=thr)
        exp = O.find_matches([("t", t)], img, method=method, score_threshold=thr)
        assert_hits_equal(canon(got), canon(exp), tol=1e-6)
    # constant template + method 5 -> all-ones map -> "trivial image": no peaks at all
    assert mtm.findMatches([("c", np.full((5, 5), 9, np.uint8))], img, method=5, score_threshold=0.5) == []


def test_global_extremum_first_occurrence(mtm):
    img = np.zeros((50, 70), np.uint8)
    for (y, x) in ((7, 9), (7, 40), (30, 9)):
        img[y:y + 6, x:x + 6] = synth.rand_u8(5, 0, (6, 6))
    t = img[7:13, 9:15].copy()
    for method in (1, 3, 5):
        got = mtm.findMatches([("t", t)], img, method=method, N_object=1)
        exp = O.find_matches([("t", t)], img, method=method, N_object=1)
        assert_hits_equal(got, hits_json(exp), tol=1e-6)
        assert got[0][1][:2] == (9, 7)      # three exact copies: the first in row-major order wins
    blank = np.full((20, 20), 3, np.uint8)
    got = mtm.findMatches([("b", blank[:4, :4])], blank, method=3, N_object=1)
    assert got[0][1] == (0, 0, 4, 4)


# ------------------------------------------------------------------------------------------------
# BASELINE.json sizes: oracle on cfg2, properties on cfg3-sized input
# ------------------------------------------------------------------------------------------------
def test_cfg2_full_size_against_oracle(mtm):
    img, units, plants = synth.make_config("cfg2")
    got = mtm.matchTemplates(units, img, score_threshold=0.5)
    exp = O.match_templates(units, img, score_threshold=0.5)
    assert_hits_equal(canon(got), canon(exp), tol=1e-5)
    assert {(p[0], p[1]) for p in plants} == {(h[0], h[1]) for h in got}
    m = mtm.computeScoreMap(units[3][1], img, 5)
    map_close(m, O.compute_score_map(units[3][1], img, 5), tol=1e-6)


def test_cfg3_size_properties(mtm, ctx):
    img, units, plants = synth.make_config("cfg3")
    hits = mtm.matchTemplates(units, img, score_threshold=0.5)
    found = {(h[0], h[1]): float(h[2]) for h in hits}
    assert set(found) == {(p[0], p[1]) for p in plants}          # every plant, nothing else
    for p in plants:
        if p[2] == 0:
            assert found[(p[0], p[1])] == 1.0                     # exact copies score exactly 1
        else:
            assert 0.6 < found[(p[0], p[1])] < 0.99
    assert hits == mtm.matchTemplates(units, img, score_threshold=0.5)     # idempotent, deterministic
    sc = [h[2] for h in hits]
    assert all(a >= b for a, b in zip(sc, sc[1:]))                # sorted by descending score
    # rot90 symmetry: matching the rotated image with the rotated templates gives rotated boxes
    img_r = np.ascontiguousarray(np.rot90(img))
    units_r = [(u[0], np.ascontiguousarray(np.rot90(u[1]))) for u in units[:16]]
    a = mtm.matchTemplates(units[:16], img, score_threshold=0.5)
    b = mtm.matchTemplates(units_r, img_r, score_threshold=0.5)
    W = img.shape[1]
    rot = sorted((h[0], (h[1][1], W - h[1][0] - h[1][2], h[1][3], h[1][2]), round(float(h[2]), 5)) for h in a)
    assert rot == sorted((h[0], h[1], round(float(h[2]), 5)) for h in b)
    t = ctx.timing()
    assert t["kernel_used"] in (2, 3) and t["ncc_launches"] >= 1
    # the MFMA and dot4 kernels produce the same hits at full size
    res = {}
    for kernel in ("dot4", "mfma"):
        set_kernel(ctx, kernel)
        try:
            res[kernel] = mtm.matchTemplates(units, img, score_threshold=0.5)
        finally:
            set_kernel(ctx, "auto")
    assert res["mfma"] == hits
    hits_close(sorted(res["dot4"], key=lambda h: (h[0], h[1])), sorted(hits, key=lambda h: (h[0], h[1])))


def _batched_maps(mtm, ctx, units, img, method, thr, picks):
    """One production-style batched search in map mode (MTM_OPT_HITS_ONLY = 0): returns its hit records and
    the complete score maps of the templates in `picks`, copied out of the map arena as the batched launch
    wrote them (mtm_last_score_map) - not recomputed one template at a time."""
    from MTM import _lib
    with ctx.lock:
        ctx.set_option(_lib.OPT_HITS_ONLY, 0)
        try:
            ctx.set_image(img)
            ctx.set_templates([(u[1], u[2] if len(u) >= 3 else None) for u in units], method)
            raw = ctx.find_matches(_lib.PEAKS_LOCAL, thr).copy()
            assert ctx.timing()["hits_only"] == 0
            maps = {}
            for i in picks:
                t = units[i][1]
                maps[i] = ctx.last_score_map(i, (img.shape[0] - t.shape[0] + 1, img.shape[1] - t.shape[1] + 1))
        finally:
            restore_hits_only(ctx)
    return raw, maps


def test_cfg3_full_size_maps_against_oracle(mtm, ctx):
    """BASELINE configs[2] at full size (3840x2160 x 128 units): complete score maps out of the batched
    128-unit launch against the oracle, pixel by pixel."""
    img, units, plants = synth.make_config("cfg3")
    picks = [5, 77, 126]                               # a 0-degree, a 90-degree and a 180-degree unit
    raw, maps = _batched_maps(mtm, ctx, units, img, 5, 0.5, picks)
    cache = {}
    for i in picks:
        exp = O.match_template(img, units[i][1], 5, cache=cache)
        map_close(maps[i], exp, tol=1e-6)
        assert np.array_equal(maps[i], mtm.computeScoreMap(units[i][1], img, 5))      # single-template launch: same bits
    # the map-mode hit records are those of the default (hits-only) call
    hits = mtm.findMatches(units, img, score_threshold=0.5)
    assert len(raw) == len(hits) == len(plants)
    assert sorted((units[int(r["templ_idx"])][0], (int(r["x"]), int(r["y"]), int(r["w"]), int(r["h"])), float(r["score"])) for r in raw) == \
        sorted((h[0], h[1], float(h[2])) for h in hits)


def test_cfg3_every_map_matrix_cores_equal_valu_kernel():
    """BASELINE configs[2] at full size, ALL 128 score maps (1.06 G outputs) of the batched production launch on the int8
    matrix cores against the same batched call on the independent VALU kernel (v_dot4_u32_u8, north_star's kernel):
    exact-division mode bit for bit; default mode (correctly rounded reciprocals) at most one float32 ulp apart on a
    vanishing share of the pixels.  The oracle checks three of these maps pixel by pixel (the test above); this one
    leaves no map of the launch unchecked."""
    from MTM import _lib
    if not default_routes():
        pytest.skip("compares the default matrix-core route with the VALU kernel")
    img, units, _ = synth.make_config("cfg3")
    tl = [(u[1], None) for u in units]
    shape = (img.shape[0] - 64 + 1, img.shape[1] - 64 + 1)
    mf, d4 = _lib.Context(0), _lib.Context(0)
    try:
        d4.set_option(_lib.OPT_KERNEL, 2)
        for c_ in (mf, d4):
            c_.set_option(_lib.OPT_HITS_ONLY, 0)
            c_.set_image(img)
        d4.set_option(_lib.OPT_EXACT_DIV, 1)
        d4.set_templates(tl, 5)
        ref_hits = d4.find_matches(_lib.PEAKS_LOCAL, 0.5).copy()
        assert d4.timing()["kernel_used"] == 2
        for exact in (1, 0):
            mf.set_option(_lib.OPT_EXACT_DIV, exact)
            mf.set_templates(tl, 5)
            hits = mf.find_matches(_lib.PEAKS_LOCAL, 0.5).copy()
            assert mf.timing()["kernel_used"] == 3 and mf.timing()["hits_only"] == 0
            assert np.array_equal(hits, ref_hits) or (exact == 0 and len(hits) == len(ref_hits))
            n_diff = 0
            for i in range(len(tl)):
                a, b = mf.last_score_map(i, shape), d4.last_score_map(i, shape)
                if exact:
                    assert np.array_equal(a, b), i
                else:
                    ne = a != b
                    if ne.any():
                        n_diff += int(ne.sum())
                        ulp = np.abs(a[ne].view(np.int32).astype(np.int64) - b[ne].view(np.int32).astype(np.int64))
                        assert ulp.max() <= 1, (i, int(ulp.max()))
            if not exact:
                assert n_diff <= 1e-6 * len(tl) * shape[0] * shape[1], n_diff
    finally:
        mf.close()
        d4.close()


def test_cfg5_full_size_maps_against_oracle(mtm, ctx):
    """BASELINE configs[4] at full size (7680x4320, 80 masked units at 5 scales, TM_CCORR_NORMED): two complete
    score maps per size class - 32, 56, 80, 104 and 128 pixels, disc masks - out of the batched launch against
    the oracle's matchTemplateMask, plus the hit list properties."""
    img, units, plants = synth.make_config("cfg5")
    sides = sorted({u[1].shape[0] for u in units})
    assert sides == [32, 56, 80, 104, 128]
    picks = []
    for sd in sides:
        idx = [i for i, u in enumerate(units) if u[1].shape[0] == sd]
        picks += [idx[0], idx[-1]]
    raw, maps = _batched_maps(mtm, ctx, units, img, 3, 0.9, picks)
    if default_routes():
        assert ctx.timing()["kernel_used"] == 3        # every class on the int8 matrix cores (sum I^2 M included)
    cache = {}
    for i in picks:
        exp = O.match_template(img, units[i][1], 3, mask=units[i][2], cache=cache)
        assert np.isfinite(exp).all()
        map_close(maps[i], exp, tol=1e-6)
    hits = mtm.matchTemplates(units, img, method=3, score_threshold=0.9, maxOverlap=0.25)
    found = {(h[0], h[1]): float(h[2]) for h in hits}
    planted = {(p[0], p[1]): p[2] for p in plants}
    assert set(planted) <= set(found)                   # every plant; masked CCORR_NORMED also fires near plants
    for k, amp in planted.items():
        if amp == 0:
            assert abs(found[k] - 1.0) <= 1e-6          # exact copies under the mask
    pre = mtm.findMatches(units, img, method=3, score_threshold=0.9)
    assert len(pre) == len(raw)


def test_cfg4_full_list_and_eight_way_shards(mtm):
    """BASELINE configs[3]: 3840x2160 x 256 units.  The whole list on one GPU, the 32-unit shard rank 0 of an
    8-GPU job holds (LPT partition), and the sharded pipeline end to end - every rank's search done by the HIP
    path, the exchange replaced by an in-process gather - against the unsharded call."""
    from MTM import _lib, _raw_matches
    from MTM.distributed import matchTemplates_sharded, shard_units, unit_cost
    img, units, plants = synth.make_config("cfg4")
    assert len(units) == 256
    full = mtm.matchTemplates(units, img, score_threshold=0.5)
    found = {(h[0], h[1]): float(h[2]) for h in full}
    assert set(found) == {(p[0], p[1]) for p in plants}          # every plant, nothing else
    for p in plants:
        assert (found[(p[0], p[1])] == 1.0) if p[2] == 0 else (0.6 < found[(p[0], p[1])] < 0.99)
    world = 8
    shards = shard_units([unit_cost(u[1], img.shape) for u in units], world)
    assert sorted(i for s in shards for i in s) == list(range(256)) and all(len(s) == 32 for s in shards)
    # one GPU's share of the job
    sub = [units[i] for i in shards[0]]
    part = mtm.matchTemplates(sub, img, score_threshold=0.5)
    labels = {u[0] for u in sub}
    assert part == [h for h in full if h[0] in labels]           # plants never overlap: the global NMS drops nothing
    # all eight ranks, one after the other on this GPU
    raws = []
    for r in range(world):
        raw = _raw_matches([units[i] for i in shards[r]], img, 5, float("inf"), 0.5).copy()
        raw["templ_idx"] = np.asarray(shards[r], dtype=np.int32)[raw["templ_idx"]]
        raws.append(raw)
    gathered = np.concatenate(raws)

    class Gather:                                                  # the exchange: every rank receives every rank's hits
        world_size = world

        def __init__(self, rank):
            self.rank = rank

        def allgather(self, hits):
            assert np.array_equal(hits, raws[self.rank])
            return gathered

    for r in (0, 3, 7):
        def local(sub_list, image, r=r):
            assert [u[0] for u in sub_list] == [units[i][0] for i in shards[r]]
            return _raw_matches(sub_list, image, 5, float("inf"), 0.5)
        assert matchTemplates_sharded(units, img, Gather(r), score_threshold=0.5, find_local=local) == full


# ------------------------------------------------------------------------------------------------
# one call for image + search (mtm_find_matches_image): banded upload under the score kernel
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("bands", ["0.16,0.44,0.72,1", "0.5,1", "0.03,0.06,0.5,0.51,1", "1"])
def test_fused_image_call_equals_two_calls(mtm, bands, monkeypatch):
    """mtm_find_matches_image == mtm_set_image + mtm_find_matches, whatever the band layout: plain and
    row-multiplexed classes (band boundaries fall inside 8R-row blocks), hits-only and map mode, both peak modes,
    single-channel uint16 (byte-plane kernel), and the layouts that upload in one piece (several size classes, RGB,
    masks)."""
    from MTM import _lib
    monkeypatch.setenv("MTM_UPLOAD_BANDS", bands)
    monkeypatch.setenv("MTM_BAND_MIN_FILL", "0")      # band even these small images (by default a band must fill the chip)
    fused, plain = _lib.Context(0), _lib.Context(0)
    try:
        img, units, plants = synth.make_workload(seed=21, image_hw=(1100, 1200), n_base=20, templ=32, noisy_per_unit=2)
        rgb_img, rgb_units, _ = synth.make_workload(seed=22, image_hw=(700, 1600), n_base=3, templ=24, channels=3)
        msk_img, msk_units, _ = synth.make_workload(seed=23, image_hw=(1050, 1100), n_base=2, templ=32, scales=(24, 40), masked=True)
        img16 = img.astype(np.uint16) * 200 + 7
        imgf = img.astype(np.float32) * np.float32(0.37) + np.float32(3.0)
        cases = [
            (img, [(u[1], None) for u in units], 5, 0.5),                      # 20 templates: plain MFMA class, banded
            (img, [(u[1], None) for u in units[:5]], 5, 0.5),                  # row-multiplexed (nt = 8, R = 2)
            (img, [(units[0][1], None)], 3, 0.7),                              # row-multiplexed, one template (R = 16: 128-row blocks)
            (img, [(u[1], None) for u in units[:3]] + [(np.ascontiguousarray(units[4][1][:20, :28]), None)], 1, 0.3),   # two classes
            (rgb_img, [(u[1], None) for u in rgb_units], 5, 0.5),
            (msk_img, [(u[1], u[2]) for u in msk_units], 3, 0.9),
            (img16, [(u[1].astype(np.uint16) * 200 + 7, None) for u in units[:4]], 5, 0.5),   # uint16 kernel, banded as well
            (img16, [(u[1].astype(np.uint16) * 200 + 7, None) for u in units], 3, 0.8),       # two groups of 16 templates
            # several classes, the heaviest of them banded, the others (a masked one among them) on the complete image
            (img, [(u[1], None) for u in units[:6]] + [(np.ascontiguousarray(units[8][1][:24, :28]), None)] +
                  [(np.ascontiguousarray(units[7][1][:28, :30]), (units[7][1][:28, :30] > 90).astype(np.uint8))], 3, 0.8),
            # round 6: single-channel float32 images in two bands (bf16 kernel, one launch per band; the statistics of a band
            # from the rows that have arrived) - the one-product screen, the three-product maps, and a raw-sum method whose
            # maps send the call to the float64 kernel behind the banding decision
            (imgf, [(u[1].astype(np.float32) * np.float32(0.37) + np.float32(3.0), None) for u in units], 5, 0.5),
            (imgf, [(u[1].astype(np.float32) * np.float32(0.37) + np.float32(3.0), None) for u in units[:4]], 3, 0.97),
            (imgf, [(u[1].astype(np.float32) * np.float32(0.37) + np.float32(3.0), None) for u in units[:4]], 1, 0.05),
            (imgf, [(u[1].astype(np.float32) * np.float32(0.37) + np.float32(3.0), None) for u in units[:4]], 4, 1e5),
        ]
        for honly in (1, 0):
            fused.set_option(_lib.OPT_HITS_ONLY, honly)
            plain.set_option(_lib.OPT_HITS_ONLY, honly)
            for mode in (_lib.PEAKS_LOCAL, _lib.PEAKS_GLOBAL):
                for im, tl, method, thr in cases:
                    a = fused.search(tl, im, method, mode, thr)
                    plain.set_image(im)
                    plain.set_templates(tl, method)
                    b = plain.find_matches(mode, thr)
                    assert len(a) == len(b) and len(a) >= len(tl), (bands, honly, mode, method, len(a), len(b))
                    assert np.array_equal(a, b), (bands, honly, mode, method)
                    if honly == 0 and mode == _lib.PEAKS_LOCAL:                # the maps of the banded launches, bit for bit
                        shape = (im.shape[0] - tl[-1][0].shape[0] + 1, im.shape[1] - tl[-1][0].shape[1] + 1)
                        assert np.array_equal(fused.last_score_map(len(tl) - 1, shape), plain.last_score_map(len(tl) - 1, shape))
        # the banded call really ran in bands, and measured the clock it ran at
        fused.set_option(_lib.OPT_HITS_ONLY, 1)
        fused.search(cases[0][1], img, 5, _lib.PEAKS_LOCAL, 0.5)
        t = fused.timing()
        if not any(os.environ.get(k) for k in ("MTM_FUSE_STATS", "MTM_KERNEL")):   # (tools/alt_modes.sh)
            assert (t["ncc_launches"] == 1) if bands == "1" else (2 <= t["ncc_launches"] <= len(bands.split(",")))
            assert 300.0 < t["sclk_mhz"] < 3500.0
        if default_routes():        # the float32 call too: two launches of the bf16 kernel, the one-product screen in both
            fused.search(cases[-4][1], imgf, 5, _lib.PEAKS_LOCAL, 0.5)
            t = fused.timing()
            assert t["kernel_used"] == 5 and t["ncc_launches"] == (1 if bands == "1" else 2) and t["f32_pieces"] == 1, t
        # a different image through the same fused context: nothing stale survives
        img2 = np.ascontiguousarray(img[::-1, ::-1])
        a = fused.search(cases[0][1], img2, 5, _lib.PEAKS_LOCAL, 0.5)
        plain.set_image(img2)
        plain.set_templates(cases[0][1], 5)
        assert np.array_equal(a, plain.find_matches(_lib.PEAKS_LOCAL, 0.5))
        # the image of a banded call stays usable by everything else: the float64 / naive kernels and the generic
        # statistics read the float32 plane, which the banded upload leaves out and the library rebuilds on demand
        t0 = cases[0][1][0][0]
        shape = (img2.shape[0] - t0.shape[0] + 1, img2.shape[1] - t0.shape[1] + 1)
        for kern in (1, 0):                                # naive kernel; automatic choice (matrix cores)
            for c_ in (fused, plain):
                c_.set_option(_lib.OPT_KERNEL, kern)
                c_.set_templates(cases[0][1][:3], 5)
            m_f, m_p = fused.score_map(0, shape), plain.score_map(0, shape)
            assert np.array_equal(m_f, m_p), kern
        msk = [(t0, (t0 > 100).astype(np.uint8))]
        for c_ in (fused, plain):
            c_.set_templates(msk, 5)                       # masked + method 5 is not a matrix-core case: float64 kernel
        fused.search(cases[0][1], img2, 5, _lib.PEAKS_LOCAL, 0.5)      # (banded upload again)
        fused.set_templates(msk, 3)
        plain.set_templates(msk, 3)
        fused.set_option(_lib.OPT_KERNEL, 1)
        plain.set_option(_lib.OPT_KERNEL, 1)
        assert np.array_equal(fused.score_map(0, shape), plain.score_map(0, shape))
    finally:
        fused.close()
        plain.close()


def test_banded_call_with_float64_classes_on_two_lanes(mtm, monkeypatch):
    """A banded upload leaves the float32 plane out; two more classes of the call run the float64 kernel (masked
    templates wider than the matrix-core tiling takes), which reads that plane, on two different lanes.  The plane is
    built once ahead of both lanes (round 3 built it lazily on whichever lane asked first - the other lane could read
    it before it was written).  Fused call == set_image + find_matches, repeatedly."""
    from MTM import _lib
    monkeypatch.setenv("MTM_BAND_MIN_FILL", "0")
    monkeypatch.setenv("MTM_UPLOAD_BANDS", "0.3,1")
    fused, plain = _lib.Context(0), _lib.Context(0)
    try:
        img, units, _ = synth.make_workload(seed=77, image_hw=(1100, 1200), n_base=6, templ=32, noisy_per_unit=2)
        wide1 = np.ascontiguousarray(img[300:308, 100:360])
        wide2 = np.ascontiguousarray(img[700:710, 400:664])
        tl = [(u[1], None) for u in units] + [(wide1, (wide1 > 60).astype(np.uint8)), (wide2, (wide2 > 90).astype(np.uint8))]
        plain.set_image(img)
        plain.set_templates(tl, 3)
        exp = plain.find_matches(_lib.PEAKS_LOCAL, 0.9)
        assert len(exp) >= len(tl)
        for rep in range(6):
            im = img if rep % 2 == 0 else np.ascontiguousarray(img[::-1])
            got = fused.search(tl, im, 3, _lib.PEAKS_LOCAL, 0.9)
            if rep % 2 == 0:
                assert np.array_equal(got, exp), rep
            else:
                plain.set_image(im)
                assert np.array_equal(got, plain.find_matches(_lib.PEAKS_LOCAL, 0.9)), rep
                plain.set_image(img)
    finally:
        fused.close()
        plain.close()


@pytest.mark.parametrize("env", ["MTM_KERNEL=dot4", "MTM_ROW_MUX=0", "MTM_HITS_ONLY=0", "MTM_MASKSQ_FUSED=0", "MTM_MFMA_R2=0",
                                 "MTM_SCREEN_L1=0", "MTM_CLASS_LANES=1", "MTM_FUSE_STATS=0", "MTM_TEMPL_ON_DEVICE=0",
                                 # round 5: what the fixed-cost work replaced, where the replaced route still serves other calls
                                 "MTM_FUSE_LAYOUT=0", "MTM_CAND_PINNED=0", "MTM_SEG_SKIP=0", "MTM_UPLOAD_BANDS=1", "MTM_EXACT_DIV=0",
                                 "MTM_EXACT_DIV=2"])
def test_production_reachable_routes_give_the_default_lists(mtm, env, monkeypatch):
    """Every switch that selects a kernel or route a production call can also reach by itself (the VALU kernel for shapes
    the matrix-core path does not take, plain instead of row-multiplexed tiling, maps in memory, the unfused sum I^2 M
    pass, host-packed operands, the two-pass statistics, ...) returns the hit lists of the default route: same boxes in the
    same order, scores to 1e-6.  Round 6: these are ALL the environment switches of the library that select a compute
    route (20 MTM_* variables are left in the native code, 39 in round 5: the others are tuning / diagnostics - bands,
    lanes, spin, time-out, trace, band fill - or have a test of their own); tools/alt_modes.sh runs the whole suite
    under each, this is the driver's share."""
    from MTM import _lib
    img, units, _ = synth.make_workload(seed=41, image_hw=(700, 1500), n_base=20, templ=32, noisy_per_unit=2)
    msk_img, msk_units, _ = synth.make_workload(seed=43, image_hw=(640, 900), n_base=2, templ=32, scales=(24, 40), masked=True)
    cases = [(img, [(u[1], None) for u in units], 5, 0.5),                        # > 16 templates: two-row tiling
             (img, [(u[1], None) for u in units[:5]], 3, 0.7),                    # row-multiplexed, nt = 8
             (img, [(units[0][1], None)], 5, 0.5),                                # one template: R = 16, the edge steps
             (img, [(u[1], None) for u in units[:3]] + [(np.ascontiguousarray(units[4][1][:20, :28]), None)], 1, 0.3),
             (msk_img, [(u[1], u[2]) for u in msk_units], 3, 0.9)]                # masked classes: the sum I^2 M pass
    ref_ctx = _lib.Context(0)
    k, v = env.split("=")
    monkeypatch.setenv(k, v)
    alt = _lib.Context(0)
    try:
        for mode in (_lib.PEAKS_LOCAL, _lib.PEAKS_GLOBAL):
            for im, tl, method, thr in cases:
                a = alt.search(tl, im, method, mode, thr)
                b = ref_ctx.search(tl, im, method, mode, thr)
                assert len(a) == len(b) >= len(tl), (env, method, mode, len(a), len(b))
                for f in ("templ_idx", "x", "y", "w", "h"):
                    assert np.array_equal(a[f], b[f]), (env, method, mode, f)
                assert np.abs(a["score"] - b["score"]).max() <= 1e-6, (env, method, mode)
    finally:
        alt.close()
        ref_ctx.close()


@pytest.mark.parametrize("seed", [78, 180, 244])
def test_float32_exact_zero_plateaus_name_the_exact_answer(mtm, seed, monkeypatch):
    """The three float32 fuzz cases of round 3 where the library and the oracle disagreed (tools/fuzz_parity.py,
    FUZZ_DTYPE=float32; profiles/r03_r03fz/fuzz_f32.txt), settled against exact arithmetic instead of asserted.  All three
    are difference scores (TM_SQDIFF / TM_SQDIFF_NORMED) of templates cut from a flat region: wherever the window equals
    the template value for value, sum (I - T)^2 is exactly 0, cv2's expression max(S2 - 2 corr + T2, 0) is 0 in exact
    arithmetic and so is the normalised score - a plateau of exact zeros.  The float64 kernel's FMA chain returns exactly
    0.0 there; the oracle's float64 summation order leaves ~1e-10 noise on it, so its peak finder sees fewer plateau
    pixels and its first-minimum lands on whichever zero its noise favours.  Checked here: (a) every hit only one of the two
    lists has sits on a window that IS an exact copy (or exact copy under the mask), and the library's score there is
    exactly 0.0; (b) with N_object == 1 the library returns cv2.minMaxLoc's answer for the exact map - the FIRST exact copy
    in row-major order (reference MTM/__init__.py:226-230)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    monkeypatch.setenv("FUZZ_DTYPE", "float32")
    import fuzz_parity as F
    img, lt, method, thr, n_obj, box = F.make_case(seed)
    assert img.dtype == np.float32 and method in (0, 1)
    oimg, olt = F.as_oracle(img, lt)
    kw = dict(method=method, N_object=n_obj, searchBox=box)
    if thr is not None:
        kw["score_threshold"] = thr
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = mtm.findMatches(lt, img, **kw)
        exp = O.find_matches(olt, oimg, **kw)
    templ = {t[0]: t for t in lt}

    def exact_copy(label, bx):
        x, y, w, h = bx
        t = templ[label]
        win = img[y:y + h, x:x + w]
        d = win.astype(np.float64) - t[1].astype(np.float64)          # float32 values: the difference is exact in float64
        if len(t) >= 3 and method == 0:
            d = d * (np.asarray(t[2], np.float64) != 0)
        return not d.any()

    g = {(h[0], tuple(h[1])): float(h[2]) for h in got}
    e = {(h[0], tuple(h[1])): float(h[2]) for h in exp}
    disputed = sorted(g.keys() ^ e.keys())
    assert disputed, "the case no longer differs from the oracle: drop it from this test"
    for k in disputed:
        assert exact_copy(*k), (seed, k)
        if k in g:
            assert g[k] == 0.0, (seed, k, g[k])
    for k, v in g.items():                       # and nowhere else does the library call something an exact zero
        if v == 0.0:
            assert exact_copy(*k), (seed, k)
    if n_obj == 1:
        x0, y0 = (box[0], box[1]) if box else (0, 0)
        H, W = (box[3], box[2]) if box else img.shape[:2]
        for label, bx, score in got:
            if score != 0.0:
                continue
            x, y, w, h = bx
            first = None                          # first exact copy in row-major order of the searched region
            for yy in range(y0, y + 1):
                xs = range(x0, (x if yy == y else x0 + W - w) + 1)
                hit = next((xx for xx in xs if exact_copy(label, (xx, yy, w, h))), None)
                if hit is not None:
                    first = (hit, yy)
                    break
            assert first == (x, y), (seed, label, bx, first)


def test_group_rccl_exchange_in_process(mtm):
    """The exchange north_star names for one process: ncclCommInitAll over the group's devices, one all-gather
    per device inside ncclGroupStart / End, rank 0's gathered list merged.  On a one-GPU box that is a communicator of one
    rank - the code path is the N-rank one.  A device listed twice cannot form an RCCL communicator: the group says so
    and keeps the host merge.  Either exchange returns the single-context list."""
    from MTM import _lib
    img, units, plants = synth.make_workload(seed=33, image_hw=(900, 1300), n_base=9, templ=32, noisy_per_unit=2)
    tl = [(u[1], None) for u in units]
    ctx = _lib.Context(0)
    g1, g2 = _lib.Group([0]), _lib.Group([0, 0])
    try:
        ref = ctx.search(tl, img, 5, _lib.PEAKS_LOCAL, 0.5)
        assert len(ref) >= len(tl)
        assert g1.comm_ranks() == 0 and g1.exchange_used() == "host"
        with pytest.raises(_lib.MtmError):
            g1.set_exchange("rccl")                          # no communicators yet
        assert g1.comm_init() == 1 and g1.comm_ranks() == 1
        for kind in ("rccl", "host", "rccl"):
            g1.set_exchange(kind)
            assert np.array_equal(g1.search(tl, img, 5, _lib.PEAKS_LOCAL, 0.5), ref), kind
            assert g1.exchange_used() == kind
        # more hits than one 512-record slot: the slot follows the longest list
        many = g1.search(tl, img, 5, _lib.PEAKS_LOCAL, 0.05)
        assert len(many) > 600 and np.array_equal(many, ctx.search(tl, img, 5, _lib.PEAKS_LOCAL, 0.05))
        assert g1.exchange_used() == "rccl"
        # one rank per device: an aliased group keeps the host merge
        assert g2.comm_init(strict=False) == 0 and g2.comm_ranks() == 0
        with pytest.raises(_lib.MtmError, match="listed twice"):
            g2.comm_init()
        assert np.array_equal(g2.search(tl, img, 5, _lib.PEAKS_LOCAL, 0.5), ref) and g2.exchange_used() == "host"
    finally:
        g1.close()
        g2.close()
        ctx.close()


def test_device_group_equals_single_context(mtm, coins, monkeypatch):
    """Several contexts in one process (mtm_group; here all on the one GPU of the box): LPT shards, concurrent
    upload + search per context, host merge - the hit list of the single-context call, for every shard count."""
    from MTM import _lib
    from MTM.distributed import shard_units, unit_cost
    img, units, plants = synth.make_workload(seed=31, image_hw=(900, 1300), n_base=11, templ=32, noisy_per_unit=2)
    units.append(("wide", np.ascontiguousarray(img[100:140, 200:330])))        # unequal costs
    units.append(("tiny", np.ascontiguousarray(img[500:512, 700:716])))
    ref = mtm.matchTemplates(units, img, score_threshold=0.5)
    ref1 = mtm.findMatches(units, img, N_object=1)
    for devs in ([0, 0], [0, 0, 0], [0] * 8, [0] * 16):
        assert mtm.matchTemplates(units, img, score_threshold=0.5, devices=devs) == ref
        assert mtm.findMatches(units, img, N_object=1, devices=devs) == ref1
        g = _lib.engine_for(devs)
        assert isinstance(g, _lib.Group) and len(g) == len(devs)
        dev = g.shards([(u[1], None) for u in units], img.shape, 5)
        exp = shard_units([unit_cost(u[1], img.shape) for u in units], len(devs))
        assert [sorted(np.flatnonzero(dev == d).tolist()) for d in range(len(devs))] == exp
    # masks, a difference score, a search box and an error from a worker
    small, _ = coin_templates(coins)
    lt = [("a", small, otsu_mask(small)), ("b", np.ascontiguousarray(small[:30, :30]))]
    assert mtm.matchTemplates(lt, coins, method=3, score_threshold=0.8, devices=[0, 0]) == \
        mtm.matchTemplates(lt, coins, method=3, score_threshold=0.8)
    assert mtm.matchTemplates(lt[1:], coins, method=1, score_threshold=0.3, searchBox=(20, 10, 300, 250), devices="0,0,0") == \
        mtm.matchTemplates(lt[1:], coins, method=1, score_threshold=0.3, searchBox=(20, 10, 300, 250))
    with pytest.raises(_lib.MtmError, match="device 0"):
        _lib.engine_for([0, 0]).search([(np.zeros((8, 8, 2), np.uint8), None), (small, None)], coins, 5, 0, 0.5)
    # MTM_DEVICES selects the engine of a plain call
    monkeypatch.setenv("MTM_DEVICES", "0,0")
    assert isinstance(_lib.engine_for(None), _lib.Group)
    assert mtm.matchTemplates(units, img, score_threshold=0.5) == ref
    monkeypatch.setenv("MTM_DEVICES", "all")
    assert mtm.matchTemplates(units, img, score_threshold=0.5) == ref


# ------------------------------------------------------------------------------------------------
# float32 images on the bf16 matrix cores (two bfloat16 pieces per value, centred operands)
# ------------------------------------------------------------------------------------------------
def test_float32_on_bf16_matrix_cores(mtm, ctx, coins, monkeypatch):
    """Non-uint8 input is matched in float32 (reference MTM/__init__.py:71-74).  Unmasked float32 classes run on the
    bf16 matrix cores; the maps must stay within 5e-5 of the float64 oracle (north_star allows 1e-4) whatever the
    brightness offset, scale or gradient of the image - the centring of both operands is what makes 16 significant
    bits enough - and the hit lists must be those of the exact float64 kernel (MTM_F32_MFMA=0)."""
    if os.environ.get("MTM_F32_MFMA") or os.environ.get("MTM_KERNEL", "auto") != "auto":
        pytest.skip("the float32 route is forced by the environment")
    from MTM import _lib
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:303, 0:384]
    variants = {
        "unit range": coins.astype(np.float32) / 255.0,
        "large offset": coins.astype(np.float32) * 0.02 + 1000.0,                       # contrast 5 on a level of 1000
        "illumination gradient": coins.astype(np.float32) + (3.0 * yy + 2.0 * xx).astype(np.float32),
        "tiny scale": coins.astype(np.float32) * 1e-6,
        "signed noise": rng.standard_normal((303, 384)).astype(np.float32) * 40.0,
    }
    worst = 0.0
    for name, im in variants.items():
        small, big = np.ascontiguousarray(im[37:75, 80:121]), np.ascontiguousarray(im[14:73, 302:367])
        tall = np.ascontiguousarray(im[10:150, 200:240])                                 # 140 rows: several LDS chunks
        wide = np.ascontiguousarray(im[100:130, 20:250])                                 # 230 columns: 8 tap blocks
        for t in (small, big, tall, wide):
            for method in (5, 3, 1):
                got = mtm.computeScoreMap(t, im, method)
                assert ctx.timing()["kernel_used"] == (5 if not os.environ.get("MTM_F32_MFMA") else ctx.timing()["kernel_used"])
                exp = O.compute_score_map(t, im, method)
                worst = max(worst, map_close(got, exp, tol=5e-5))
    # many templates (two MFMA groups), RGB, raw methods, hit lists against the exact float64 kernel
    im = variants["signed noise"] + variants["unit range"] * 20.0
    lt = [("t%d" % k, np.ascontiguousarray(im[7 * k:7 * k + 28, 11 * k:11 * k + 36])) for k in range(21)]
    rgbf = np.stack([im, np.roll(im, 5, axis=1), 300.0 - im], axis=2).astype(np.float32)
    lt_rgb = [("c%d" % k, np.ascontiguousarray(rgbf[20 * k:20 * k + 30, 30 * k:30 * k + 40])) for k in range(4)]
    monkeypatch.setenv("MTM_F32_MFMA", "0")
    exact = _lib.Context(0)
    monkeypatch.delenv("MTM_F32_MFMA")
    try:
        for tl, img, method, thr in ((lt, im, 5, 0.5), (lt, im, 1, 0.3), (lt_rgb, rgbf, 5, 0.5), (lt_rgb, rgbf, 1, 0.3)):
            a = mtm.findMatches(tl, img, method=method, score_threshold=thr)
            raw = exact.search([(t[1], None) for t in tl], img, method, _lib.PEAKS_LOCAL, thr)
            assert exact.timing()["kernel_used"] == 0
            b = mtm._to_hit_list(raw, tl, 0, 0)
            assert len(a) == len(b) >= len(tl)
            assert_hits_equal(hits_json(a), hits_json(b), tol=0.0, ordered=True)        # refined: the float64 kernel's list
        for method in (0, 2, 4):     # raw sums stay on the float64 kernel (they can cancel to ~0: no relative bound)
            got = mtm.computeScoreMap(lt[3][1], im, method)
            assert ctx.timing()["kernel_used"] == 0
            exp = O.compute_score_map(lt[3][1], im, method)
            assert np.abs(got.astype(np.float64) - exp).max() <= 1e-6 * np.abs(exp).max()
    finally:
        exact.close()
    print("bf16-piece kernel: worst |score - oracle| = %.2e" % worst)


@pytest.mark.gpu
def test_float32_hit_lists_are_the_float64_kernels(coins):
    """The default float32 route (bf16 matrix cores as a screen + exact float64 re-scoring, MTM_OPT_F32_MFMA = 1) returns
    the hit records of the float64 kernel (MTM_OPT_F32_MFMA = 0) - same pixels, same order, same float32 scores bit for
    bit - on images whose score maps are full of plateaus and near-ties, where a 1e-6 perturbation decides which pixel
    "equals its 3x3 maximum": a smooth photograph-like image (TM_CCORR_NORMED of all-positive data is ~0.9 everywhere),
    a three-row template, a piecewise-constant image (exact ties).  Every way the refinement can run is walked by
    shrinking the hit capacity: kernel candidates (hits-only and with the maps in memory), the map scan after the
    candidate list overflowed, the float64 kernel after the scan's list overflowed too; N_object == 1 likewise.
    Reference: MTM/__init__.py:71-74 (float32 cast), :45 (peak_local_max), :226 (minMaxLoc)."""
    if not default_routes():
        pytest.skip("asserts the default float32 routes (bf16 screen + exact re-scoring)")
    from MTM import _lib
    fast, exact = _lib.Context(0), _lib.Context(0)
    exact.set_option(_lib.OPT_F32_MFMA, 0)
    rng = np.random.default_rng(11)
    smooth = synth.smooth_u8(3, (300, 420)).astype(np.float32) * 0.37 + 12.5
    blocks = np.kron(rng.integers(0, 6, (38, 53)).astype(np.float32), np.ones((8, 8), np.float32)) * 17.25 + 3.0
    cf = coins.astype(np.float32) * 0.5 + np.linspace(0, 40, coins.shape[1], dtype=np.float32)[None, :]
    cases = []
    for name, im in (("smooth", smooth), ("blocks", blocks), ("coins", cf)):
        lt = [np.ascontiguousarray(im[20:52, 30:70]), np.ascontiguousarray(im[100:103, 200:260]),      # 3 rows: near-flat maps
              np.ascontiguousarray(im[150:190, 40:64])] + \
             [np.ascontiguousarray(im[10 * k:10 * k + 24, 16 * k:16 * k + 24]) for k in range(18)]   # a 21-template class
        cases.append((name, im, lt))
    routes, pieces = set(), set()
    try:
        for name, im, lt in cases:
            templs = [(t, None) for t in lt]
            for method, thr in ((3, 0.5), (3, 0.98), (5, 0.3), (5, 0.8), (1, 0.1)):
                ref = exact.search(templs, im, method, _lib.PEAKS_LOCAL, thr)
                assert exact.timing()["kernel_used"] == 0 and exact.timing()["f32_route"] == 0
                for cap in (1 << 18, 20000, 1500):
                    for honly in (1, 0):
                        fast.set_option(_lib.OPT_HIT_CAPACITY, cap)
                        fast.set_option(_lib.OPT_HITS_ONLY, honly)
                        got = fast.search(templs, im, method, _lib.PEAKS_LOCAL, thr)
                        tm = fast.timing()
                        routes.add(tm["f32_route"])
                        pieces.add((tm["f32_route"], honly, tm["f32_pieces"]))
                        assert tm["f32_route"] in (1, 2, 3), tm
                        assert len(got) == len(ref), (name, method, thr, cap, honly, tm["f32_route"], len(got), len(ref))
                        for f in ("templ_idx", "x", "y", "w", "h"):
                            assert np.array_equal(got[f], ref[f]), (name, method, thr, cap, honly, tm["f32_route"], f)
                        assert np.array_equal(got["score"].view(np.uint32), ref["score"].view(np.uint32)), \
                            (name, method, thr, cap, honly, tm["f32_route"])
            # N_object == 1: the global extremum per template (first occurrence on exact ties)
            # (round 4: the raw-sum methods 0 / 2 / 4 too - their refined extremum lists by rigorous error bounds, an
            # exact copy is TM_SQDIFF 0 and the templates of the piecewise-constant image have exact ties)
            for method in (5, 3, 1, 0, 2, 4):
                ref = exact.search(templs, im, method, _lib.PEAKS_GLOBAL, 0.0)
                for cap in (1 << 18, 64):
                    fast.set_option(_lib.OPT_HIT_CAPACITY, cap)
                    fast.set_option(_lib.OPT_HITS_ONLY, 1)
                    got = fast.search(templs, im, method, _lib.PEAKS_GLOBAL, 0.0)
                    routes.add(fast.timing()["f32_route"])
                    if cap == 1 << 18 and name != "blocks":
                        assert fast.timing()["f32_route"] == 1 and fast.timing()["kernel_used"] == 5, (name, method, fast.timing())
                    assert len(got) == len(ref) == len(lt)
                    for f in ("templ_idx", "x", "y"):
                        assert np.array_equal(got[f], ref[f]), (name, method, cap, f, fast.timing()["f32_route"])
                    assert np.array_equal(got["score"].view(np.uint32), ref["score"].view(np.uint32)), (name, method, cap)
        assert routes == {1, 2, 3}, routes          # every way of refining has been exercised
        # round 6: the hits-only candidate route starts with the one-product screen (and only it: maps in memory are the
        # three-product kernel's); overflowing lists went through the three-product launch on their way to the scan
        assert (1, 1, 1) in pieces and not any(hon == 0 and pc == 1 for _, hon, pc in pieces), sorted(pieces)
        assert any(pc == 3 and hon == 1 for _, hon, pc in pieces), sorted(pieces)
        # raw sums with thresholds: listed by the bound of the sum and re-scored (round 5; route 3 while a back-off lasts -
        # test_float32_raw_sums_with_thresholds_equal_the_float64_kernels); their maps stay on the float64 kernel
        name, im, lt = cases[0]
        fast.set_option(_lib.OPT_HIT_CAPACITY, 1 << 18)
        got = fast.search([(t, None) for t in lt[:4]], im, 4, _lib.PEAKS_LOCAL, 1e6)
        assert fast.timing()["f32_route"] in (1, 3)
        ref = exact.search([(t, None) for t in lt[:4]], im, 4, _lib.PEAKS_LOCAL, 1e6)
        assert got.tobytes() == ref.tobytes()
        shape = (im.shape[0] - lt[0].shape[0] + 1, im.shape[1] - lt[0].shape[1] + 1)
        fast.set_templates([(lt[0], None)], 0)
        exact.set_templates([(lt[0], None)], 0)
        fast.set_image(im)
        exact.set_image(im)
        assert np.array_equal(fast.score_map(0, shape), exact.score_map(0, shape))
        # MTM_OPT_F32_MFMA = 2: the bf16 scores as they are (no re-scoring), still within tolerance of the exact ones
        fast.set_option(_lib.OPT_HIT_CAPACITY, 1 << 18)
        fast.set_option(_lib.OPT_F32_MFMA, 2)
        name, im, lt = cases[2]
        got = fast.search([(t, None) for t in lt], im, 5, _lib.PEAKS_LOCAL, 0.8)
        assert fast.timing()["f32_route"] == 0 and fast.timing()["kernel_used"] == 5 and len(got) >= len(lt)
    finally:
        fast.close()
        exact.close()


# ------------------------------------------------------------------------------------------------
# templates on the device: views of device-resident sources, device-side packing, on-device augmentation
# ------------------------------------------------------------------------------------------------
def test_device_resident_templates_equal_host_packing(mtm, coins, monkeypatch):
    """uint8 template sets live on the device (views + pack_units_kernel) by default; MTM_TEMPL_ON_DEVICE=0 keeps
    the host packers.  Same hit records and the same score maps, bit for bit, on every route a class can take:
    plain / row-multiplexed MFMA packs, masked (T*M packs + the mask pack of the sum I^2 M pass), RGB, several
    size classes, and the kernels with host-packed operands (dot4, float64 / naive: pixels fetched back)."""
    from MTM import _lib
    monkeypatch.setenv("MTM_TEMPL_ON_DEVICE", "0")
    host = _lib.Context(0)
    monkeypatch.setenv("MTM_TEMPL_ON_DEVICE", "1")
    dev = _lib.Context(0)
    try:
        img, units, _ = synth.make_workload(seed=41, image_hw=(420, 700), n_base=19, templ=24, noisy_per_unit=1)
        rgb_img, rgb_units, _ = synth.make_workload(seed=42, image_hw=(300, 520), n_base=5, templ=20, channels=3)
        msk_img, msk_units, _ = synth.make_workload(seed=43, image_hw=(400, 640), n_base=3, templ=32, scales=(24, 40), masked=True)
        small, big = coin_templates(coins)
        strided = coins[10:90:2, 20:140:3]                                     # non-contiguous rows
        cases = [
            (img, [(u[1], None) for u in units], 5),
            (img, [(u[1], None) for u in units[:7]], 3),
            (img, [(units[0][1], None), (np.ascontiguousarray(units[1][1][:17, :23]), None), (units[2][1], None)], 1),
            (rgb_img, [(u[1], None) for u in rgb_units], 5),
            (rgb_img, [(u[1], None) for u in rgb_units] * 4, 4),
            (msk_img, [(u[1], u[2]) for u in msk_units], 3),
            (msk_img, [(u[1], u[2]) for u in msk_units], 0),
            (coins, [(small, otsu_mask(small)), (big, None), (strided, None)], 3),
            (coins, [(np.full((9, 9), 7, np.uint8), None), (small, None)], 5),   # constant template: all-ones map
        ]
        for kernel in ("auto", "dot4", "naive"):
            for cx in (host, dev):
                cx.set_option(_lib.OPT_KERNEL, KERNELS[kernel])
                cx.set_option(_lib.OPT_HITS_ONLY, 0)
            for im, tl, method in cases:
                thr = {0: 1e7, 1: 0.35, 3: 0.9, 4: 1e5, 5: 0.5}[method]
                a = dev.search(tl, im, method, _lib.PEAKS_LOCAL, thr)
                b = host.search(tl, im, method, _lib.PEAKS_LOCAL, thr)
                assert np.array_equal(a, b), (kernel, method, len(a), len(b))
                for i in (0, len(tl) - 1):
                    shape = (im.shape[0] - tl[i][0].shape[0] + 1, im.shape[1] - tl[i][0].shape[1] + 1)
                    ma, mb = dev.last_score_map(i, shape), host.last_score_map(i, shape)
                    assert np.array_equal(ma, mb, equal_nan=True), (kernel, method, i)
    finally:
        host.close()
        dev.close()


def test_augmentation_on_device(mtm, ctx, coins):
    """MTM.augment.matchTemplatesAugmented(bases, variants, image) == matchTemplates(expand(bases, variants), image):
    rotations, mirrors, exact area resizes and integer downscales built as device views, never on the host; the
    device resize equals the numpy one byte for byte (checked through constant-free score maps)."""
    A = mtm.augment
    img, base_units, _ = synth.make_workload(seed=51, image_hw=(500, 760), n_base=5, templ=32, rotations=1, noisy_per_unit=1)
    bases = [(u[0].split("_")[0], u[1]) for u in base_units]
    # plant rotated / mirrored / resized copies so that every variant has something to find
    spec_all = A.variants(angles=(0, 90, 180, 270), flip_lr=True, flip_ud=True)
    for k, (name, t) in enumerate(A.expand(bases[:2], spec_all)):
        y, x = 40 + 48 * (k // 12), 30 + 58 * (k % 12)
        img[y:y + t.shape[0], x:x + t.shape[1]] = t
    specs = [
        A.variants(angles=(0, 90, 180, 270)),
        spec_all,
        A.variants(sizes=(20, (24, 40), 48), angles=(0, 90)),
        A.variants(factors=(1, 2, 3), flip_lr=True),
    ]
    for row, spec in enumerate(specs[2:]):
        for k, (name, t) in enumerate(A.expand(bases[2:3], spec)):
            y, x = 200 + 100 * row, 30 + 60 * k
            img[y:y + t.shape[0], x:x + t.shape[1]] = t
    for spec in specs:
        for method, thr in ((5, 0.6), (1, 0.3), (3, 0.93)):
            got = A.matchTemplatesAugmented(bases, spec, img, method=method, score_threshold=thr)
            exp = mtm.matchTemplates(A.expand(bases, spec), img, method=method, score_threshold=thr)
            assert got == exp and len(got) > 0, (len(spec), method, len(got), len(exp))
        assert A.matchTemplatesAugmented(bases, spec, img, N_object=1) == mtm.matchTemplates(A.expand(bases, spec), img, N_object=1)
        sb = (16, 8, 600, 400)
        assert A.matchTemplatesAugmented(bases, spec, img, score_threshold=0.6, searchBox=sb) == \
            mtm.matchTemplates(A.expand(bases, spec), img, score_threshold=0.6, searchBox=sb)
    # masks travel with their template (method 3), RGB bases, and the fallback kernels (pixels fetched back)
    small, big = coin_templates(coins)
    mb = [("s", small, otsu_mask(small)), ("b", np.ascontiguousarray(big[:50, :50]), otsu_mask(np.ascontiguousarray(big[:50, :50])))]
    spec = A.variants(angles=(0, 90), sizes=(28, 36), flip_lr=True)
    assert A.matchTemplatesAugmented(mb, spec, coins, method=3, score_threshold=0.85) == \
        mtm.matchTemplates(A.expand(mb, spec), coins, method=3, score_threshold=0.85)
    with pytest.warns(UserWarning, match="not supporting the use of Mask"):
        a = A.matchTemplatesAugmented(mb, spec, coins, method=5, score_threshold=0.4)
    assert a == mtm.matchTemplates(A.expand([t[:2] for t in mb], spec), coins, method=5, score_threshold=0.4)
    rgb = np.stack([coins, np.roll(coins, 3, axis=1), 255 - coins], axis=2)
    rb = [("c", np.ascontiguousarray(rgb[37:75, 80:121]))]
    spec = A.variants(angles=(0, 90, 180, 270), sizes=((30, 34),))
    assert A.matchTemplatesAugmented(rb, spec, rgb, score_threshold=0.4) == mtm.matchTemplates(A.expand(rb, spec), rgb, score_threshold=0.4)
    for kernel in ("dot4", "naive"):
        set_kernel(ctx, kernel)
        try:
            spec = A.variants(angles=(0, 270), sizes=(24,))
            assert A.matchTemplatesAugmented(bases[:2], spec, img, score_threshold=0.6) == \
                mtm.matchTemplates(A.expand(bases[:2], spec), img, score_threshold=0.6)
        finally:
            set_kernel(ctx, "auto")
    # other pixel types: expanded on the host, same answer
    f_img, f_bases = img.astype(np.float32), [(n, t.astype(np.float32)) for n, t in bases[:2]]
    spec = A.variants(angles=(0, 180))
    assert A.matchTemplatesAugmented(f_bases, spec, f_img, score_threshold=0.6) == \
        mtm.matchTemplates(A.expand(f_bases, spec), f_img, score_threshold=0.6)


def test_cfg3_and_cfg5_from_bases_only(mtm):
    """BASELINE configs[2] and configs[4] submitted the way they are described - 32 bases x 4 rotations, 16 bases
    x 5 scales (masked) - as bases + an augmentation spec: hit lists identical to the host-augmented lists."""
    A = mtm.augment
    img, units, plants = synth.make_config("cfg3")
    bases = [(u[0][:-2], u[1]) for u in units[::4]]                     # labels "<i>_0" -> "<i>"
    spec = A.variants(angles=(0, 90, 180, 270))
    assert [u[0] for u in A.expand(bases, spec)] == [u[0] for u in units]
    got = A.matchTemplatesAugmented(bases, spec, img, score_threshold=0.5)
    assert got == mtm.matchTemplates(units, img, score_threshold=0.5)
    assert {(h[0], h[1]) for h in got} == {(p[0], p[1]) for p in plants}
    img, units, plants = synth.make_config("cfg5")
    sides = (32, 56, 80, 104, 128)
    kw = synth.CONFIGS["cfg5"]
    bases = [(str(b), synth.rand_u8(kw["seed"], 1000 + b, (64, 64)), synth._disc_mask(64)) for b in range(kw["n_base"])]
    assert len(bases) == 16 and len(units) == 80
    spec = A.variants(sizes=sides)
    host_units = A.expand(bases, spec)
    for hu, u in zip(host_units, units):
        assert np.array_equal(hu[1], u[1])                              # the templates are synth's own resizes
    got = A.matchTemplatesAugmented(bases, spec, img, method=3, score_threshold=0.9)
    exp = mtm.matchTemplates(host_units, img, method=3, score_threshold=0.9)
    assert got == exp and len(got) >= len(plants)


def test_large_templates_as_slabs_on_mfma(mtm, ctx):
    """Templates beyond the int32 accumulator / LDS tile limits of the MFMA kernel (w > 256 or w*h*C > 131071; the
    reference's own benchmark matches a 414 x 400 template, tutorials/Benchmark.ipynb:203) are cut into slabs, every
    slab runs on the matrix cores in raw mode and slab_combine_kernel adds them up - exact integers: the maps equal
    the dot4 kernel's bit for bit and the oracle's to 1e-6, for one template (row-multiplexed raw launches), a few,
    more than 16 (plain raw launches) and RGB."""
    from MTM import _lib
    img = synth.rand_u8(61, 0, (900, 1100))
    big = np.ascontiguousarray(img[100:514, 300:700]).copy()              # 414 x 400, an exact copy in the image
    noisy = np.clip(big.astype(np.int32) + (synth.rand_u8(61, 9, big.shape).astype(np.int32) % 61) - 30, 0, 255).astype(np.uint8)
    wide = np.ascontiguousarray(img[600:640, 200:900])                    # 40 x 700: wider than three tiles
    cases = [
        ([("big", big)], img),
        ([("big", big), ("big180", np.ascontiguousarray(np.rot90(big, 2))), ("noisy", noisy)], img),
        ([("wide", wide)], img),
        ([("t%d" % k, np.ascontiguousarray(img[20 * k:20 * k + 300, 10 * k:10 * k + 450])) for k in range(18)], img),
    ]
    rgb = synth.rand_u8(62, 0, (500, 640, 3))
    cases.append(([("rgb", np.ascontiguousarray(rgb[50:290, 100:330]))], rgb))        # 240 x 230 x 3 = 165,600 taps
    for lt, im in cases:
        for method, thr in ((5, 0.5), (3, 0.9), (1, 0.3)):
            res = {}
            for kernel in ("auto", "dot4"):
                set_kernel(ctx, kernel)
                try:
                    res[kernel] = (mtm.findMatches(lt, im, method=method, score_threshold=thr),
                                   mtm.computeScoreMap(lt[-1][1], im, method), ctx.timing()["kernel_used"])
                finally:
                    set_kernel(ctx, "auto")
            if not os.environ.get("MTM_SLAB_MFMA") and default_routes():
                assert res["auto"][2] == 3 and res["dot4"][2] == 2             # matrix cores vs VALU fallback
            assert res["auto"][0] == res["dot4"][0] and len(res["auto"][0]) >= 1, (len(lt), method)
            assert np.array_equal(res["auto"][1], res["dot4"][1])
        map_close(res["auto"][1], O.compute_score_map(lt[-1][1], im, 1), tol=1e-6)
        exp = O.match_templates(lt, im, method=5, score_threshold=0.5)
        assert_hits_equal(canon(mtm.matchTemplates(lt, im, method=5, score_threshold=0.5)), canon(exp), tol=1e-5)
        assert mtm.findMatches(lt, im, N_object=1) == [h for h in mtm.findMatches(lt, im, N_object=1, devices=[0, 0])]
        # N_object == 1: the extremum comes out of slab_combine_kernel (no maps, no extremum_kernel) - the record of the
        # maps + extremum_kernel route and the oracle's, maxima and minima
        for method in (5, 1, 2):
            fused = mtm.findMatches(lt, im, method=method, N_object=1)
            tm = ctx.timing()
            assert (tm["hits_only"] == 1 and tm["kernel_used"] == 3) or not default_routes(), tm
            ctx.set_option(_lib.OPT_HITS_ONLY, 0)
            try:
                via_maps = mtm.findMatches(lt, im, method=method, N_object=1)
                assert ctx.timing()["hits_only"] == 0
            finally:
                restore_hits_only(ctx)
            assert fused == via_maps and len(fused) == len(lt), method
            if method != 2:
                exp1 = O.find_matches(lt, im, method=method, N_object=1)
                assert_hits_equal(hits_json(fused), hits_json(exp1), tol=1e-5, ordered=False)
    # maps materialised == hits only
    ctx.set_option(_lib.OPT_HITS_ONLY, 0)
    try:
        a = mtm.findMatches(cases[1][0], img, score_threshold=0.4)
    finally:
        restore_hits_only(ctx)
    assert a == mtm.findMatches(cases[1][0], img, score_threshold=0.4)


BORDER_CALLS = {
    "sqdiff_normed": lambda im: ([("small", im[37:75, 80:121]), ("big", im[14:73, 302:367])],
                                 dict(method=1, score_threshold=0.2, maxOverlap=0)),
    "corner_m1": lambda im: ([("corner", im[0:38, 0:41]), ("edge", im[120:158, 343:384])],
                             dict(method=1, score_threshold=0.25, maxOverlap=0.1)),
    "corner_m5_negthr": lambda im: ([("corner", im[0:38, 0:41])], dict(method=5, score_threshold=-0.2, maxOverlap=0.0)),
}


@pytest.mark.parametrize("border", ["constant", "nearest"])
@pytest.mark.parametrize("name", sorted(BORDER_CALLS))
def test_border_rules_reference_runs(mtm, ctx, coins, name, border):
    """Objects touching the image border under a difference score: scikit-image <= 0.18 (zero padding, real 0.18.3
    fixtures) never reports them, >= 0.19 (edge replication, the default here) does."""
    templates, kw = BORDER_CALLS[name](coins)
    ctx.set_option(2, {"constant": 0, "nearest": 1}[border])
    try:
        hits = mtm.matchTemplates(templates, coins, **kw)
        pre = mtm.findMatches(templates, coins, method=kw["method"], score_threshold=kw["score_threshold"])
    finally:
        ctx.set_option(2, 1)
    assert_hits_equal(canon(hits), canon([(h[0], tuple(h[1]), h[2]) for h in REF["%s@%s" % (name, border)]]), tol=1e-5)
    if name == "corner_m1":
        assert_hits_equal(canon(pre), REF["corner_m1_pre@" + border], tol=1e-5)
        assert (("corner", (0, 0, 41, 38)) in {(h[0], h[1]) for h in hits}) == (border == "nearest")


# ------------------------------------------------------------------------------------------------
# RCCL hit exchange (single rank: exercises dlopen(librccl), unique id, comm init, both all-gathers)
# ------------------------------------------------------------------------------------------------
def test_rccl_allgather_single_rank(mtm):
    from MTM import _lib
    from MTM.distributed import _stdout_to_stderr
    ctx = _lib.Context(0)
    try:
        with _stdout_to_stderr():
            ctx.comm_init(_lib.comm_unique_id(), 1, 0)
        hits = np.zeros(5, dtype=_lib.HIT_DTYPE)
        hits["templ_idx"] = np.arange(5)
        hits["x"] = 7
        hits["score"] = np.linspace(0, 1, 5, dtype=np.float32)
        out, counts = ctx.allgather_hits(hits)
        assert list(counts) == [5] and np.array_equal(out, hits)
        out, counts = ctx.allgather_hits(hits[:0])
        assert list(counts) == [0] and len(out) == 0
        big = np.zeros(5000, dtype=_lib.HIT_DTYPE)          # more than one fixed slot: second round
        big["x"] = np.arange(5000)
        out, counts = ctx.allgather_hits(big)
        assert list(counts) == [5000] and np.array_equal(out, big)
    finally:
        ctx.close()


def test_sharded_api_single_rank(mtm, coins):
    """matchTemplates_sharded with the RCCL exchange at world size 1 == matchTemplates."""
    from MTM import _lib
    from MTM.distributed import HitExchange, matchTemplates_sharded
    small, big = coin_templates(coins)
    lt = [("small", small), ("big", big)]
    ex = HitExchange("rccl", 0, 1, context=_lib.default_context())
    a = matchTemplates_sharded(lt, coins, ex, score_threshold=0.3, maxOverlap=0.25)
    b = mtm.matchTemplates(lt, coins, score_threshold=0.3, maxOverlap=0.25)
    assert a == b


# ------------------------------------------------------------------------------------------------
# masked uint8 templates on the integer path (MFMA for sum I*(T*M), dot4 for sum I^2*M)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("method", [0, 1, 2, 3])
def test_masked_integer_path(mtm, ctx, method):
    img = synth.rand_u8(77, 0, (230, 410))
    rng = np.random.default_rng(5)
    disc = synth._disc_mask(37)
    other = ((rng.integers(0, 3, (37, 37)) > 0) * 255).astype(np.uint8)
    templs = []
    for i in range(19):                      # two classes of the same size: the masks differ
        y, x = int(rng.integers(0, 230 - 37)), int(rng.integers(0, 410 - 37))
        templs.append((np.ascontiguousarray(img[y:y + 37, x:x + 37]), disc if i % 3 else other))
    shape = (230 - 37 + 1, 410 - 37 + 1)
    res = {}
    for kernel, exact in (("naive", False), ("mfma", True), ("mfma", False)):
        set_kernel(ctx, kernel)
        set_exact(ctx, exact)
        try:
            ctx.set_image(img)
            ctx.set_templates(templs, method)
            res[(kernel, exact)] = [ctx.score_map(i, shape) for i in (0, 1, 7, 18)]
            if kernel == "mfma":
                hits = ctx.find_matches(0, 0.97 if method in (1, 3) else 1e9)
                assert ctx.timing()["kernel_used"] == 3       # the integer path really ran
        finally:
            restore_exact(ctx)
            set_kernel(ctx, "auto")
    for k, i in enumerate((0, 1, 7, 18)):
        exp = O.match_template(img, templs[i][0], method, mask=templs[i][1])
        map_close(res[("naive", False)][k], exp, tol=1e-6)
        assert np.array_equal(res[("mfma", True)][k], res[("naive", False)][k]), (method, i)
        ulp_close(res[("mfma", False)][k], res[("naive", False)][k], tol=2.4e-7)


def test_masked_api_uses_integer_path(mtm, ctx, coins):
    small, _ = coin_templates(coins)
    mask = otsu_mask(small)
    hits = mtm.matchTemplates([("testMask", small, mask)], coins, method=3, score_threshold=0.8, maxOverlap=0)
    assert ctx.timing()["kernel_used"] == 3 or not default_routes()
    assert_hits_equal(hits, G["notebook_G3"]["hits"], tol=1e-4)


def test_template_matcher_stream(mtm):
    # the matchers below own fresh contexts (kernel choice = the environment's); the calls they are compared with go
    # through the shared default context, which earlier tests leave on "auto": align the two when MTM_KERNEL is set
    set_kernel(mtm._lib.default_context(), os.environ.get("MTM_KERNEL", "auto").lower())
    try:
        _template_matcher_stream(mtm)
    finally:
        set_kernel(mtm._lib.default_context(), "auto")


def _template_matcher_stream(mtm):
    img, units, _ = synth.make_workload(seed=8, image_hw=(300, 520), n_base=5, templ=32)
    matcher = mtm.TemplateMatcher(units, score_threshold=0.5)
    for k in range(3):
        im = np.ascontiguousarray(np.roll(img, 17 * k, axis=1))
        assert matcher.match(im) == mtm.matchTemplates(units, im, score_threshold=0.5)
    small = np.ascontiguousarray(img[:200, :400])                 # a different image size re-places the maps
    assert matcher.match(small) == mtm.matchTemplates(units, small, score_threshold=0.5)
    # double-buffered stream (mtm_find_matches_next): sizes change mid-stream, strided views, N_object=1
    stream = [np.ascontiguousarray(np.roll(img, 31 * k, axis=0)) for k in range(5)]
    stream.insert(2, small)
    stream.append(img[5:290, 3:500])                               # non-contiguous rows
    assert list(matcher.match_stream(stream)) == [mtm.matchTemplates(units, im, score_threshold=0.5) for im in stream]
    assert matcher.match(img) == mtm.matchTemplates(units, img, score_threshold=0.5)      # set_image after a stream
    one = mtm.TemplateMatcher(units[:3], N_object=1)
    assert list(one.match_stream(stream[:3])) == [mtm.matchTemplates(units[:3], im, N_object=1) for im in stream[:3]]
    f32 = [im.astype(np.float32) for im in stream[:3]]
    mf = mtm.TemplateMatcher([(n, t.astype(np.float32)) for n, t in units[:3]], score_threshold=0.5)
    assert list(mf.match_stream(f32)) == [mtm.matchTemplates([(n, t.astype(np.float32)) for n, t in units[:3]], im, score_threshold=0.5) for im in f32]


# ------------------------------------------------------------------------------------------------
# assorted edge cases through the public API
# ------------------------------------------------------------------------------------------------
def test_edge_cases_api(mtm, coins):
    # (fresh matcher contexts vs the shared default context: see test_template_matcher_stream)
    set_kernel(mtm._lib.default_context(), os.environ.get("MTM_KERNEL", "auto").lower())
    try:
        _edge_cases_api(mtm, coins)
    finally:
        set_kernel(mtm._lib.default_context(), "auto")


def _edge_cases_api(mtm, coins):
    small, big = coin_templates(coins)
    assert mtm.matchTemplates([], coins) == []
    assert mtm.findMatches([], coins) == []
    # unnormalised methods (raw thresholds), maxima (2, 4)
    for method, thr in ((2, 2.0e7), (4, 1.5e6)):
        got = mtm.matchTemplates([("small", small)], coins, method=method, score_threshold=thr, maxOverlap=0.3)
        exp = O.match_templates([("small", small)], coins, method=method, score_threshold=thr, maxOverlap=0.3)
        assert len(got) > 0
        assert_hits_equal(got, hits_json(exp), tol=1e-5)
    # N_object == 1 with a mask, and with several templates (global extremum per template, then best)
    mask = otsu_mask(small)
    got = mtm.matchTemplates([("m", small, mask)], coins, method=3, N_object=1)
    exp = O.match_templates([("m", small, mask)], coins, method=3, N_object=1)
    assert_hits_equal(got, hits_json(exp), tol=1e-5)
    got = mtm.matchTemplates([("small", small), ("big", big)], coins, method=1, N_object=1)
    assert got[0][2] == 0.0 and got[0][0] in ("small", "big")
    # float32 images through the resident-template API
    imf = coins.astype(np.float32) / 255
    tm = mtm.TemplateMatcher([("s", imf[37:75, 80:121])], method=5, score_threshold=0.5, maxOverlap=0)
    assert tm.match(imf) == mtm.matchTemplates([("s", imf[37:75, 80:121])], imf, method=5, score_threshold=0.5, maxOverlap=0)
    # non-contiguous inputs (column-strided view) are copied by the binding
    view = coins[:, ::2]
    t = np.ascontiguousarray(view[40:70, 30:60])
    got = mtm.matchTemplates([("v", t)], view, score_threshold=0.6)
    exp = O.match_templates([("v", t)], np.ascontiguousarray(view), score_threshold=0.6)
    assert_hits_equal(got, hits_json(exp), tol=1e-5)


# ------------------------------------------------------------------------------------------------
# hits-only mode (MTM_OPT_HITS_ONLY): no score maps in memory, peaks from the candidate list alone
# ------------------------------------------------------------------------------------------------
def test_hits_only_equals_map_mode(mtm, ctx, coins):
    small, big = coin_templates(coins)
    mask = otsu_mask(small)
    rng = np.random.default_rng(11)
    noise = rng.integers(0, 256, (300, 500), dtype=np.uint8)
    flat = noise.copy()
    flat[100:220, 150:400] = 77                     # flat windows: guarded denominators
    cases = [
        ([("small", small), ("big", big)], coins, 5, 0.3),
        ([("small", small), ("big", big)], coins, 5, -0.2),     # negative threshold, saturated scores
        ([("small", small), ("big", big)], coins, 3, 0.7),
        ([("small", small), ("big", big)], coins, 1, 0.4),      # minima
        ([("small", small), ("big", big)], coins, 1, 1.5),      # every pixel below the threshold
        ([("small", small), ("big", big)], coins, 2, 2.0e7),    # raw scores
        ([("small", small), ("big", big)], coins, 4, 1.0e6),
        ([("m", small, mask)], coins, 3, 0.8),                  # masked integer path
        ([("n%d" % i, noise[10 * i:10 * i + 24, 7 * i:7 * i + 24].copy()) for i in range(20)], noise, 5, 0.4),
        ([("f%d" % i, flat[90 + 9 * i:90 + 9 * i + 20, 140 + 5 * i:140 + 5 * i + 30].copy()) for i in range(6)], flat, 5, 0.2),
        ([("f%d" % i, flat[90 + 9 * i:90 + 9 * i + 20, 140 + 5 * i:140 + 5 * i + 30].copy()) for i in range(6)], flat, 3, 0.9),
        ([("f%d" % i, flat[90 + 9 * i:90 + 9 * i + 20, 140 + 5 * i:140 + 5 * i + 30].copy()) for i in range(6)], flat, 1, 0.1),
        ([("c", np.full((12, 12), 50, np.uint8)), ("small", small)], coins, 5, 0.5),   # all-ones map + normal one
    ]
    set_kernel(ctx, "mfma")
    try:
        for exact in (0, 1):
            set_exact(ctx, exact)
            for border in (0, 1):
                ctx.set_option(2, border)
                for lt, img, method, thr in cases:
                    res = []
                    for honly in (0, 1):
                        ctx.set_option(6, honly)
                        res.append(mtm.findMatches(lt, img, method=method, score_threshold=thr))
                    assert len(res[0]) == len(res[1]), (method, thr, exact, border, len(res[0]), len(res[1]))
                    assert canon(res[0]) == canon(res[1]), (method, thr, exact, border)
        # hits-only results against the oracle directly (dense: thousands of candidates)
        restore_hits_only(ctx)
        ctx.set_option(2, 1)
        lt = [("small", small), ("big", big)]
        for exact in (1, 0):                    # the shipped default (IEEE division) and the reciprocal epilogue
            set_exact(ctx, exact)
            for method, thr in ((5, 0.05), (3, 0.6), (1, 0.9)):
                got = mtm.findMatches(lt, coins, method=method, score_threshold=thr)
                exp = O.find_matches(lt, coins, method=method, score_threshold=thr)
                assert len(got) == len(exp) and len(got) > 50, (exact, method)
                assert_hits_equal(canon(got), canon(exp), tol=1e-5)
        restore_exact(ctx)
        # the timing record says which mode ran
        restore_hits_only(ctx)
        mtm.findMatches(lt, coins)
    finally:
        set_kernel(ctx, "auto")
        restore_exact(ctx)
        ctx.set_option(2, 1)
        restore_hits_only(ctx)


# ------------------------------------------------------------------------------------------------
# device-side area downscale (mtm_set_image_downscaled) and the augmentation helpers
# ------------------------------------------------------------------------------------------------
def test_device_downscale_and_augmentation(mtm, ctx, coins):
    A = mtm.augment
    rng = np.random.default_rng(21)
    rgb = rng.integers(0, 256, (203, 317, 3), dtype=np.uint8)
    cases = [(coins, np.uint8), (coins, np.float32), (rgb, np.uint8), (rgb, np.float32)]
    with ctx.lock:
        for img0, dt in cases:
            img = img0.astype(dt)
            for f in (2, 3, 4, 7):
                small = O.downscale_area(img, f)
                t = np.ascontiguousarray(small[3:3 + 20, 5:5 + 31])
                ctx.set_image(img, downscale=f)
                ctx.set_templates([(t, None)], 5)
                got = ctx.score_map(0, (small.shape[0] - 19, small.shape[1] - 30))
                map_close(got, O.match_template(small, t, 5), tol=1e-5)
                # the exact copy scores 1 at its origin only if the device image equals the oracle's
                assert abs(float(got[3, 5]) - 1.0) < (1e-5 if dt == np.float32 else 1e-6), (dt, f, float(got[3, 5]))   # f32: bf16-piece kernel
    small, big = coin_templates(coins)
    lt = [("small", small), ("big", big)]
    for f in (2, 3):
        got = A.matchTemplatesDownscaled(lt, coins, f, score_threshold=0.4, maxOverlap=0.3)
        exp = A.upscale_hits(O.match_templates([(n, O.downscale_area(t, f)) for n, t in lt], O.downscale_area(coins, f),
                                               score_threshold=0.4, maxOverlap=0.3), f)
        assert_hits_equal(got, hits_json(exp), tol=1e-5)
    # rotations: every rotated copy of a planted template is found where np.rot90 puts it
    img, units, plants = synth.make_workload(seed=12, image_hw=(400, 640), n_base=3, templ=32, rotations=4)
    got = mtm.matchTemplates(units, img, score_threshold=0.5)
    exp = O.match_templates(units, img, score_threshold=0.5)
    assert_hits_equal(canon(got), canon(exp), tol=1e-5)


# ------------------------------------------------------------------------------------------------
# seeded random configurations: shapes, template counts, methods, masks, thresholds
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(16))
def test_random_configs_against_oracle(mtm, seed):
    rng = np.random.default_rng(1000 + seed)
    H, W = int(rng.integers(40, 260)), int(rng.integers(40, 420))
    chans = int(rng.choice([1, 1, 1, 3]))
    shape = (H, W) if chans == 1 else (H, W, chans)
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    if seed % 4 == 0:                                   # smooth image: dense maps, plateaus after quantisation
        base = rng.random((H // 8 + 2, W // 8 + 2) + shape[2:])
        img = (np.kron(base, np.ones((8, 8) + (1,) * (len(shape) - 2)))[:H, :W] * 255).astype(np.uint8)
    method = int(rng.choice([1, 3, 5, 5]))
    n_t = int(rng.integers(1, 40))
    lt = []
    for i in range(n_t):
        h, w = int(rng.integers(3, min(70, H))), int(rng.integers(3, min(90, W)))
        if i % 3 == 0 and lt:                           # repeat a size: several templates in one class
            h, w = lt[-1][1].shape[:2]
        y, x = int(rng.integers(0, H - h + 1)), int(rng.integers(0, W - w + 1))
        t = img[y:y + h, x:x + w].copy()
        if i % 2:
            t = np.clip(t.astype(np.int32) + rng.integers(-30, 31, t.shape), 0, 255).astype(np.uint8)
        tup = ("t%d" % i, t)
        if method == 3 and chans == 1 and i % 4 == 1:
            m = (rng.random(t.shape) > 0.3).astype(np.uint8) * 255
            m[0, 0] = 255
            tup = tup + (m,)
        lt.append(tup)
    thr = float(rng.choice([0.2, 0.5, 0.8])) if method != 1 else float(rng.choice([0.1, 0.3]))
    got = mtm.findMatches(lt, img, method=method, score_threshold=thr)
    exp = O.find_matches(lt, img, method=method, score_threshold=thr)
    assert len(got) == len(exp), (len(got), len(exp))
    assert_hits_equal(canon(got), canon(exp), tol=2e-5)
    got = mtm.matchTemplates(lt, img, method=method, score_threshold=thr, maxOverlap=0.3, N_object=int(rng.choice([1, 3, 50])))
    assert len(got) <= 50


# ------------------------------------------------------------------------------------------------
# uint16 pixels: exact integer matching through byte planes on the int8 MFMA kernel (MTM_U16)
# ------------------------------------------------------------------------------------------------
def _as_f32(lt):
    return [(tup[0],) + tuple(a.astype(np.float32) for a in tup[1:]) for tup in lt]


def test_uint16_exact_path(mtm, ctx, coins):
    rng = np.random.default_rng(77)
    img = rng.integers(0, 65536, (180, 300), dtype=np.uint16)
    img[40:90, 100:220] = 31000                         # flat region: guarded denominators
    img[:8, :8] = 65535                                 # saturated corner: largest byte products
    lt = [("a", img[20:52, 30:70].copy()), ("b", img[60:92, 150:190].copy()), ("c", img[100:133, 200:251].copy()),
          ("flat", np.full((9, 9), 500, np.uint16))]
    noisy = np.clip(lt[0][1].astype(np.int64) + rng.integers(-3000, 3000, lt[0][1].shape), 0, 65535).astype(np.uint16)
    lt.append(("noisy", noisy))
    f32img = img.astype(np.float32)
    # score maps, every method, against the oracle on the float32 cast (what the reference feeds cv2)
    for method in range(6):
        for name, t in lt[:3]:
            got = mtm.computeScoreMap(t, img, method)
            assert ctx.timing()["kernel_used"] == 4 or not default_routes(), "uint16 pair must take the byte-plane MFMA path"
            exp = O.match_template(f32img, t.astype(np.float32), method)
            if method in (0, 2, 4):
                # raw sums of ~1e13: the oracle's float64 FFT carries ~1e-3 of absolute noise there (the
                # integer path is the exact one), which only shows where SQDIFF cancels to ~0
                assert np.abs(got.astype(np.float64) - exp).max() <= 1e-7 * np.abs(exp).max()
            else:
                map_close(got, exp, tol=1e-6)
    # the constant template: all-ones map for method 5, guarded elsewhere
    map_close(mtm.computeScoreMap(lt[3][1], img, 5), O.match_template(f32img, lt[3][1].astype(np.float32), 5), tol=1e-6)
    # hit lists (several classes, several members per class after rot180)
    lt2 = lt + [(n + "_r", np.ascontiguousarray(t[::-1, ::-1])) for n, t in lt[:3]]
    for method, thr in ((5, 0.5), (3, 0.9), (1, 0.2)):
        got = mtm.findMatches(lt2, img, method=method, score_threshold=thr)
        exp = O.find_matches(_as_f32(lt2), f32img, method=method, score_threshold=thr)
        assert len(got) == len(exp) and len(got) >= 3
        # exact copies score exactly 0 (method 1) here and ~1e-15 in the oracle's FFT: compare order-free
        assert_hits_equal(hits_json(got), hits_json(exp), tol=1e-6, ordered=False)
    got = mtm.matchTemplates(lt2, img, N_object=1)
    exp = O.match_templates(_as_f32(lt2), f32img, N_object=1)
    assert_hits_equal(got, hits_json(exp), tol=1e-6)
    # same numbers as the float32 route (the float64 kernel) - the policy change is invisible
    a = mtm.computeScoreMap(lt[0][1], img, 5)
    b = mtm.computeScoreMap(lt[0][1].astype(np.float32), f32img, 5)
    assert ctx.timing()["kernel_used"] in (0, 5)        # float64 kernel, or the bf16-piece kernel (~1e-5, MTM_F32_MFMA)
    map_close(a, b, tol=2e-5)
    # mixed dtypes and masks keep the reference's float32 policy
    m8 = mtm.computeScoreMap(lt[0][1].astype(np.uint8), img, 5)
    map_close(m8, O.match_template(f32img, lt[0][1].astype(np.uint8).astype(np.float32), 5), tol=1e-5)
    mask = (rng.random(lt[0][1].shape) > 0.4).astype(np.uint16)
    got = mtm.computeScoreMap(lt[0][1], img, 3, mask=mask)
    exp = O.match_template(f32img, lt[0][1].astype(np.float32), 3, mask=mask.astype(np.float32))
    map_close(got, exp, tol=1e-5)
    # RGB uint16: float64 kernel from the uint16 upload
    rgb = rng.integers(0, 65536, (90, 120, 3), dtype=np.uint16)
    t3 = rgb[10:30, 20:50].copy()
    map_close(mtm.computeScoreMap(t3, rgb, 5), O.match_template(rgb.astype(np.float32), t3.astype(np.float32), 5), tol=1e-5)
    # big template (two 64-row chunks, three 64-tap blocks) and the resident-template stream
    big = rng.integers(0, 65536, (300, 420), dtype=np.uint16)
    tb = big[50:50 + 130, 60:60 + 150].copy()
    map_close(mtm.computeScoreMap(tb, big, 5), O.match_template(big.astype(np.float32), tb.astype(np.float32), 5), tol=1e-6)
    tm = mtm.TemplateMatcher(lt2, score_threshold=0.5)
    frames = [img, np.ascontiguousarray(img[::-1]), img]
    assert list(tm.match_stream(frames)) == [mtm.matchTemplates(lt2, f, score_threshold=0.5) for f in frames]
    # downscaled matching in 16 bit
    got = mtm.augment.matchTemplatesDownscaled(lt[:3], img, 2, score_threshold=0.5)
    exp = mtm.augment.upscale_hits(O.match_templates(_as_f32([(n, O.downscale_area(t, 2)) for n, t in lt[:3]]),
                                                     O.downscale_area(img, 2).astype(np.float32), score_threshold=0.5), 2)
    assert_hits_equal(got, hits_json(exp), tol=1e-6)


def test_uint16_fused_window_statistics_bit_for_bit(mtm, monkeypatch):
    """stats_u16_kernel (window sums from the two byte planes, uint32 / uint64 prefix scans in LDS) against the two-pass
    float64 statistics it replaces: every method's score map bit for bit, on random pixels, on a saturated image (the
    largest window sums: 255 x 255 windows of 65535, one below the uint32 limit of the kernel) and on strips wider than
    one work-group."""
    from MTM import _lib
    rng = np.random.default_rng(4242)
    wide = rng.integers(0, 65536, (70, 2300), dtype=np.uint16)
    sat = np.full((300, 340), 65535, np.uint16)
    sat[100:140, 50:90] = rng.integers(0, 65536, (40, 40), dtype=np.uint16)
    cases = [(wide, [wide[5:37, 40:100].copy(), wide[20:52, 1500:1560].copy()]),
             (sat, [np.ascontiguousarray(sat[20:275, 30:285])]),
             (rng.integers(0, 65536, (130, 190), dtype=np.uint16), [rng.integers(0, 65536, (17, 23), dtype=np.uint16)])]
    monkeypatch.setenv("MTM_FUSE_STATS", "0")
    two_pass = _lib.Context()
    monkeypatch.delenv("MTM_FUSE_STATS")
    fused = _lib.Context()
    try:
        for img, ts in cases:
            for method in range(6):
                maps = []
                for c_ in (fused, two_pass):
                    c_.set_image(img)
                    c_.set_templates([(t, None) for t in ts], method)
                    shape = (img.shape[0] - ts[0].shape[0] + 1, img.shape[1] - ts[0].shape[1] + 1)
                    maps.append([c_.score_map(i, shape) for i in range(len(ts))])
                    assert c_.timing()["kernel_used"] == 4 or not default_routes()
                for a, b in zip(*maps):
                    assert np.array_equal(a, b, equal_nan=True), (img.shape, method)
    finally:
        fused.close()
        two_pass.close()


def test_uint16_hits_only_screen_changes_nothing(monkeypatch):
    """The uint16 kernel's hits-only screen (float32 bound of the three partial sums per lane against the block ranges
    of the statistics) only skips work: hit records with and without it are identical - bright and dim images, high and
    low contrast (where the bound is too loose to skip anything), thresholds near the scores of planted copies, local
    maxima and the fused global extremum."""
    from MTM import _lib
    monkeypatch.setenv("MTM_SCREEN_L1", "0")
    plain = _lib.Context()
    monkeypatch.delenv("MTM_SCREEN_L1")
    screened = _lib.Context()
    rng = np.random.default_rng(20260927)
    try:
        for c_ in (plain, screened):
            c_.set_option(_lib.OPT_HITS_ONLY, 1)
        n_rec = 0
        for case, (mean, sigma) in enumerate([(32896, 9000), (1200, 300), (60000, 2500), (40000, 12), (500, 3)]):
            base = rng.normal(mean, sigma, (150, 330))
            base[40:80, 200:260] += sigma * np.linspace(-2, 2, 60)[None, :]          # structure
            img = np.clip(np.rint(base), 0, 65535).astype(np.uint16)
            ts = [img[10:34, 20:60].copy(), img[50:74, 210:250].copy(), img[100:124, 100:140].copy()]
            noisy = np.clip(ts[0].astype(np.int64) + np.rint(rng.normal(0, sigma * 0.8, ts[0].shape)).astype(np.int64), 0, 65535)
            ts.append(noisy.astype(np.uint16))
            ts += [np.ascontiguousarray(t[::-1]) for t in ts[:2]]
            tl = [(t, None) for t in ts]
            for method in (3, 5):
                for mode, thr in ((_lib.PEAKS_LOCAL, 0.5), (_lib.PEAKS_LOCAL, 0.9), (_lib.PEAKS_LOCAL, 0.05), (_lib.PEAKS_GLOBAL, 0.0)):
                    a = screened.search(tl, img, method, mode, thr)
                    b = plain.search(tl, img, method, mode, thr)
                    assert (screened.timing()["kernel_used"] == 4 and screened.timing()["hits_only"] == 1) or not default_routes()
                    assert np.array_equal(a, b), (case, method, mode, thr, len(a), len(b))
                    n_rec += len(a)
        assert n_rec > 100
    finally:
        plain.close()
        screened.close()


def test_uint16_many_templates(mtm):
    """uint16 classes of more than 16 templates: several work-item groups ([T_hi | T_lo] of 16 templates each), a
    partly filled last group, two size classes; every map (one-template launches and last_score_map of a whole-set
    call) and the hit lists against the oracle on the float32 cast; hits-only == maps."""
    from MTM import _lib
    rng = np.random.default_rng(4242)
    H, W = 150, 333
    img = rng.integers(0, 65536, (H, W), dtype=np.uint16)
    img[70:100, 40:120] = 777
    lt = []
    for i in range(37):
        y, x = int(rng.integers(0, H - 20)), int(rng.integers(0, W - 70))
        t = img[y:y + 20, x:x + 70].copy()                     # two 64-tap blocks
        if i % 3 == 0:
            t = np.clip(t.astype(np.int64) + rng.integers(-2000, 2000, t.shape), 0, 65535).astype(np.uint16)
        lt.append(("w%d" % i, t))
    for i in range(18):
        y, x = int(rng.integers(0, H - 70)), int(rng.integers(0, W - 12))
        lt.append(("t%d" % i, img[y:y + 70, x:x + 12].copy()))  # two 64-row chunks
    f32img = img.astype(np.float32)
    c = _lib.Context(0)
    try:
        for method in (5, 1, 3):
            c.set_image(img)
            c.set_templates([(t, None) for _, t in lt], method)
            for idx in (0, 15, 16, 31, 36, 37, 54):
                t = lt[idx][1]
                got = c.score_map(idx, (H - t.shape[0] + 1, W - t.shape[1] + 1))
                assert c.timing()["kernel_used"] == 4 or os.environ.get("MTM_KERNEL")
                map_close(got, O.match_template(f32img, t.astype(np.float32), method), tol=1e-6)
            thr = {5: 0.6, 1: 0.15, 3: 0.95}[method]
            c.set_option(_lib.OPT_HITS_ONLY, 0)
            hm = c.find_matches(_lib.PEAKS_LOCAL, thr)
            for idx in (3, 20, 36, 40):
                t = lt[idx][1]
                map_close(c.last_score_map(idx, (H - t.shape[0] + 1, W - t.shape[1] + 1)),
                          O.match_template(f32img, t.astype(np.float32), method), tol=1e-6)
            c.set_option(_lib.OPT_HITS_ONLY, 1)
            hh = c.find_matches(_lib.PEAKS_LOCAL, thr)
            assert len(hm) >= 40 and sorted(map(tuple, hm.tolist())) == sorted(map(tuple, hh.tolist()))
        got = mtm.findMatches(lt, img, method=5, score_threshold=0.6)
        exp = O.find_matches(_as_f32(lt), f32img, method=5, score_threshold=0.6)
        assert_hits_equal(hits_json(got), hits_json(exp), tol=1e-6, ordered=False)
    finally:
        c.close()


# ------------------------------------------------------------------------------------------------
# row-multiplexed MFMA mode (classes of <= 16 templates: A rows = templates x output rows)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_templ", [1, 2, 3, 5, 8, 9, 16])
def test_row_multiplexed_mode(mtm, n_templ):
    rng = np.random.default_rng(300 + n_templ)
    H, W = 157, 531                                        # not multiples of the 8R-row / 256-column work items
    img = rng.integers(0, 256, (H, W), dtype=np.uint8)
    img[60:100, 200:330] = 93                              # flat windows
    shapes = [(24, 24), (70, 33), (9, 130), (12, 250)]     # one chunk, two 64-row chunks, 3 / 4 64-tap blocks (fewer rows per group)
    for (h, w) in shapes:
        lt = []
        for i in range(n_templ):
            y, x = int(rng.integers(0, H - h + 1)), int(rng.integers(0, W - w + 1))
            t = img[y:y + h, x:x + w].copy()
            if i % 2:
                t = np.clip(t.astype(np.int32) + rng.integers(-40, 41, t.shape), 0, 255).astype(np.uint8)
            lt.append(("t%d" % i, t))
        ctx = mtm._lib.Context(0)
        try:
            ctx.set_option(1, 3)                           # MFMA
            ctx.set_image(img)
            for method in (1, 3, 5, 0, 2, 4):
                ctx.set_templates([(t, None) for _, t in lt], method)
                for exact in (1, 0):
                    ctx.set_option(5, exact)
                    for li in sorted({0, n_templ // 2, n_templ - 1}):
                        got = ctx.score_map(li, (H - h + 1, W - w + 1))
                        exp = O.match_template(img, lt[li][1], method)
                        if exact and method in (1, 3, 5):
                            assert np.array_equal(got, exp), (n_templ, (h, w), method, li, float(np.abs(got - exp).max()))
                        elif method in (1, 3, 5):
                            ulp_close(got, exp)
                        else:
                            assert np.array_equal(got, exp)
                if method in (1, 3, 5):
                    thr = 0.3 if method == 1 else 0.6
                    res = []
                    for honly in (0, 1):
                        ctx.set_option(6, honly)
                        res.append(ctx.find_matches(0, thr).copy())
                        assert ctx.timing()["kernel_used"] == 3
                    assert res[0].tobytes() == res[1].tobytes()
                    exp = O.find_matches(lt, img, method=method, score_threshold=thr)
                    assert len(res[1]) == len(exp)
                    got = [(lt[int(r["templ_idx"])][0], (int(r["x"]), int(r["y"]), int(r["w"]), int(r["h"])), r["score"]) for r in res[1]]
                    assert_hits_equal(canon(got), canon(exp), tol=1e-6)
        finally:
            del ctx


# ------------------------------------------------------------------------------------------------
# RGB classes of <= 16 templates: the row-multiplexed kernel with three channel packs
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("n_templ", [1, 3, 8, 16])
def test_rgb_row_multiplexed(mtm, n_templ):
    rng = np.random.default_rng(700 + n_templ)
    H, W = 141, 397
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    img[60:100, 200:330] = (93, 12, 240)                   # flat windows in every channel
    ctx = mtm._lib.Context(0)
    try:
        ctx.set_option(1, 3)                               # MFMA
        ctx.set_image(img)
        for (h, w) in [(24, 24), (70, 33), (9, 130)]:      # one chunk, two 64-row chunks, three 64-tap blocks
            lt = []
            for i in range(n_templ):
                y, x = int(rng.integers(0, H - h + 1)), int(rng.integers(0, W - w + 1))
                t = img[y:y + h, x:x + w].copy()
                if i % 2:
                    t = np.clip(t.astype(np.int32) + rng.integers(-40, 41, t.shape), 0, 255).astype(np.uint8)
                lt.append(("t%d" % i, t))
            for method in (5, 3, 1, 0, 2, 4):
                ctx.set_templates([(t, None) for _, t in lt], method)
                for exact in (1, 0):
                    ctx.set_option(5, exact)
                    for li in sorted({0, n_templ // 2, n_templ - 1}):
                        got = ctx.score_map(li, (H - h + 1, W - w + 1))
                        assert ctx.timing()["kernel_used"] == 3
                        exp = O.match_template(img, lt[li][1], method)
                        if exact or method in (0, 2, 4):
                            assert np.array_equal(got, exp), (n_templ, (h, w), method, li, float(np.abs(got - exp).max()))
                        else:
                            ulp_close(got, exp)
                if method in (1, 3, 5):
                    thr = 0.3 if method == 1 else 0.6
                    res = []
                    for honly in (0, 1):
                        ctx.set_option(6, honly)
                        res.append(ctx.find_matches(0, thr).copy())
                    restore_hits_only(ctx)
                    assert res[0].tobytes() == res[1].tobytes()
                    exp = O.find_matches(lt, img, method=method, score_threshold=thr)
                    assert len(res[1]) == len(exp)
                    got = [(lt[int(r["templ_idx"])][0], (int(r["x"]), int(r["y"]), int(r["w"]), int(r["h"])), r["score"]) for r in res[1]]
                    assert_hits_equal(canon(got), canon(exp), tol=1e-6)
                    one = ctx.find_matches(1, 0.5)         # N_object == 1 (maps + extremum_kernel for RGB)
                    exp1 = O.find_matches(lt, img, method=method, N_object=1)
                    got1 = [(lt[int(r["templ_idx"])][0], (int(r["x"]), int(r["y"]), int(r["w"]), int(r["h"])), r["score"]) for r in one]
                    assert_hits_equal(got1, hits_json(exp1), tol=1e-6)
    finally:
        del ctx


# ------------------------------------------------------------------------------------------------
# N_object == 1: cv2.minMaxLoc fused into the MFMA epilogue (no score maps, running best per template
# as the threshold) == score maps + extremum_kernel == oracle, ties included
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("n_templ,row_mux", [(20, "1"), (37, "1"), (5, "0"), (16, "0"), (5, "1"), (12, "1"), (16, "1")])
def test_fused_global_extremum(mtm, n_templ, row_mux):
    rng = np.random.default_rng(900 + n_templ)
    H, W = 157, 531
    img = rng.integers(0, 256, (H, W), dtype=np.uint8)
    img[60:100, 200:330] = 93                              # flat windows
    tile = rng.integers(0, 256, (30, 40), dtype=np.uint8)  # exact copies: ties at score 1 / distance 0
    for (y, x) in ((3, 470), (110, 12), (110, 300), (20, 100)):
        img[y:y + 30, x:x + 40] = tile
    old = os.environ.get("MTM_ROW_MUX")
    os.environ["MTM_ROW_MUX"] = row_mux                    # "0": classes of <= 16 templates stay on the plain kernel, "1": row-multiplexed
    try:
        ctx = mtm._lib.Context(0)
    finally:
        if old is None:
            os.environ.pop("MTM_ROW_MUX", None)
        else:
            os.environ["MTM_ROW_MUX"] = old
    try:
        ctx.set_option(1, 3)                               # MFMA
        ctx.set_image(img)
        for (h, w) in [(24, 24), (70, 33), (9, 130)]:
            lt = []
            for i in range(n_templ):
                if i % 5 == 0 and h <= 30 and w <= 40:
                    t = tile[:h, :w].copy()                # several exact occurrences
                else:
                    y, x = int(rng.integers(0, H - h + 1)), int(rng.integers(0, W - w + 1))
                    t = img[y:y + h, x:x + w].copy()
                    if i % 2:
                        t = np.clip(t.astype(np.int32) + rng.integers(-40, 41, t.shape), 0, 255).astype(np.uint8)
                if i == 3:
                    t = np.full((h, w), 77, np.uint8)      # constant template
                lt.append(("t%d" % i, t))
            for method in (5, 3, 1, 0, 2, 4):
                ctx.set_templates([(t, None) for _, t in lt], method)
                for exact in (0, 1):
                    ctx.set_option(5, exact)
                    res = []
                    for honly in (1, 0):
                        ctx.set_option(6, honly)
                        res.append(ctx.find_matches(1, 0.5).copy())
                        tm = ctx.timing()
                        fused = honly if os.environ.get("1") != "0" else 0
                        assert tm["kernel_used"] == 3 and tm["hits_only"] == fused, tm
                    assert res[0].tobytes() == res[1].tobytes(), (n_templ, (h, w), method, exact)
                restore_hits_only(ctx)
                r = res[0]
                assert len(r) == n_templ and list(r["templ_idx"]) == list(range(n_templ))
                exp = O.find_matches(lt, img, method=method, N_object=1)
                got = [(lt[int(q["templ_idx"])][0], (int(q["x"]), int(q["y"]), int(q["w"]), int(q["h"])), q["score"]) for q in r]
                if method in (1, 3, 5):
                    assert_hits_equal(got, hits_json(exp), tol=1e-6)
                else:                                      # unnormalised: exact integers in both
                    assert [g[1] for g in got] == [tuple(e[1]) for e in exp]
                    assert np.allclose([g[2] for g in got], [e[2] for e in exp], rtol=1e-6)
    finally:
        del ctx


# ------------------------------------------------------------------------------------------------
# mtm_find_matches_async / _wait == mtm_find_matches (the same call split at its first synchronisation)
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_fused_global_extremum_masked(mtm, ctx, coins):
    """N_object == 1 on masked classes (binary uint8 masks, methods 0..3): the per-template extremum comes out of the
    score kernel's epilogue (no maps, no extremum_kernel) - same record as the maps + extremum_kernel route and as the
    oracle, for row-multiplexed (<= 16 templates) and plain (> 16) classes."""
    from MTM import _lib
    small, big = coin_templates(coins)
    m_small, m_big = otsu_mask(small), ((big > 120) * 255).astype(np.uint8)
    few = [("s", small, m_small), ("b", big, m_big), ("s2", np.ascontiguousarray(small[::-1]), m_small)]
    many = [("t%d" % k, np.ascontiguousarray(coins[5 * k:5 * k + 30, 7 * k:7 * k + 34]), ((coins[5 * k:5 * k + 30, 7 * k:7 * k + 34] > 90) * 255).astype(np.uint8))
            for k in range(5)]
    many = many + [("u%d" % k, np.ascontiguousarray(coins[40 + 3 * k:70 + 3 * k, 100 + 4 * k:134 + 4 * k]), many[0][2]) for k in range(19)]   # 20 share a mask
    for lt in (few, many):
        for method in (0, 1, 2, 3):
            res = []
            for honly in (1, 0):
                ctx.set_option(_lib.OPT_HITS_ONLY, honly)
                try:
                    res.append(mtm.findMatches(lt, coins, method=method, N_object=1))
                    res.append(ctx.timing()["hits_only"])
                finally:
                    restore_hits_only(ctx)
            assert res[0] == res[2] and len(res[0]) == len(lt), (method, len(lt))
            if not any(os.environ.get(k) for k in ("MTM_EXACT_DIV", "MTM_KERNEL", "MTM_HITS_ONLY")):
                assert res[1] == 1 and res[3] == 0            # fused route really ran
            exp = O.find_matches(lt, coins, method=method, N_object=1)
            assert [(h[0], h[1]) for h in res[0]] == [(h[0], h[1]) for h in exp]
            assert_hits_equal(res[0], hits_json(exp), tol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["uint16", "float32"])
def test_fused_global_extremum_uint16_float32(mtm, dtype):
    """N_object == 1 on uint16 (byte-plane passes) and float32 (bf16-piece kernel) classes: the extremum comes out of
    the score kernel's epilogue - the same record as the maps + extremum_kernel route, and the oracle's; several
    work-item groups, exact copies (ties: the first index wins), a constant template, all methods."""
    from MTM import _lib
    rng = np.random.default_rng(515)
    H, W = 140, 300
    top = 65536 if dtype == "uint16" else 256
    img = rng.integers(0, top, (H, W)).astype(dtype)
    if dtype == "float32":
        img = img * np.float32(3.25) + np.float32(100.0)
    tile = img[10:34, 20:60].copy()
    for (y, x) in ((90, 200), (50, 130)):
        img[y:y + 24, x:x + 40] = tile                        # exact copies
    lt = [("tile", tile)]
    for i in range(20):
        y, x = int(rng.integers(0, H - 24)), int(rng.integers(0, W - 40))
        t = img[y:y + 24, x:x + 40].copy()
        if i % 2:
            t = np.clip(t.astype(np.float64) + rng.integers(-top // 10, top // 10, t.shape), 0, None).astype(dtype)
        lt.append(("t%d" % i, t))
    lt.append(("const", np.full((24, 40), 77, dtype)))
    lt += [("n%d" % i, img[5 * i:5 * i + 9, 11 * i:11 * i + 70].copy()) for i in range(3)]     # a second size class
    f32img = img.astype(np.float32)
    c = _lib.Context(0)
    try:
        c.set_image(img)
        for method in (5, 3, 1, 0, 2, 4):
            c.set_templates([(t, None) for _, t in lt], method)
            res = []
            for honly in (1, 0):
                c.set_option(_lib.OPT_HITS_ONLY, honly)
                res.append(c.find_matches(_lib.PEAKS_GLOBAL, 0.5).copy())
                tm = c.timing()
                if not any(os.environ.get(k) for k in ("MTM_KERNEL", "MTM_HITS_ONLY", "MTM_F32_MFMA")):
                    fused_here = True      # (round 4: float32 raw sums too - refined extremum by rigorous error bounds)
                    # float32 with the maps in memory: no fused extremum to refine - the float64 kernel decides (route 3)
                    on_mfma = fused_here and (dtype == "uint16" or honly == 1)
                    assert tm["kernel_used"] == (4 if dtype == "uint16" else 5 if on_mfma else 0), tm
                    assert tm["hits_only"] == (honly if on_mfma else 0), tm
            c.set_option(_lib.OPT_HITS_ONLY, 1)
            assert res[0].tobytes() == res[1].tobytes(), (dtype, method)
            r = res[0]
            assert len(r) == len(lt) and list(r["templ_idx"]) == list(range(len(lt)))
            exp = O.find_matches([(n, t.astype(np.float32)) for n, t in lt], f32img, method=method, N_object=1)
            got = [(lt[int(q["templ_idx"])][0], (int(q["x"]), int(q["y"]), int(q["w"]), int(q["h"])), q["score"]) for q in r]
            tol = 1e-6 if dtype == "uint16" else 5e-5
            if method in (1, 3, 5):
                # near-ties between different positions may resolve differently within the tolerance: compare scores,
                # and positions wherever the oracle's best is isolated
                for g, e in zip(got, exp):
                    assert g[0] == e[0] and abs(float(g[2]) - float(e[2])) <= tol, (method, g, e)
                assert [g[1] for g in got[:1]] == [tuple(e[1]) for e in exp[:1]]      # the tile: first of three exact copies
            else:
                # raw sums of ~1e11..1e13: the oracle's float64 FFT carries absolute noise there (the uint16 integer path is
                # the exact one), which only shows where SQDIFF cancels to ~0
                ev = np.array([e[2] for e in exp], np.float64)
                gv = np.array([g[2] for g in got], np.float64)
                rel = 1e-5 if dtype == "float32" else 1e-6     # float32: ~1e-6 of the sums that cancel in SQDIFF
                bad = np.abs(gv - ev) > rel * np.abs(ev) + rel * np.abs(ev).max()
                assert not bad.any(), (method, [(lt[i][0], gv[i], ev[i]) for i in np.flatnonzero(bad)])
    finally:
        c.close()


@pytest.mark.gpu
def test_same_template_objects_call_after_call(mtm, coins):
    """The Python layer memoises the marshalling of the template list on the identity of its arrays; the pixels are
    still compared by the library in every call: a template changed IN PLACE, or given another shape in place, is a
    new template."""
    small, big = coin_templates(coins)
    a, b = small.copy(), np.ascontiguousarray(big[:40, :60])
    lt = [("a", a), ("b", b)]
    first = mtm.matchTemplates(lt, coins, score_threshold=0.5)
    assert mtm.matchTemplates(lt, coins, score_threshold=0.5) == first == hits_of(O.match_templates(lt, coins, score_threshold=0.5))
    a[...] = coins[150:150 + a.shape[0], 200:200 + a.shape[1]]          # same object, other pixels
    second = mtm.matchTemplates(lt, coins, score_threshold=0.5)
    assert second == hits_of(O.match_templates(lt, coins, score_threshold=0.5)) and second != first
    b.shape = (60, 40)                                                  # same object, same bytes, other geometry
    third = mtm.matchTemplates(lt, coins, score_threshold=0.5)
    assert third == hits_of(O.match_templates(lt, coins, score_threshold=0.5))
    assert all(h[1][2:] == (40, 60) for h in third if h[0] == "b")


@pytest.mark.gpu
def test_copied_template_views_are_reread(mtm, coins):
    """A template the binding has to copy (np.rot90 view, reversed columns: rows without contiguous pixels) is marshalled
    again in every call - the records of the first call would point at a private copy of the first call's pixels, and an
    in-place edit of the caller's array would go unnoticed (the reference re-reads its templates on every call)."""
    small, _ = coin_templates(coins)
    base = np.ascontiguousarray(small)
    for view in (np.rot90(base), base[:, ::-1], base.T):
        assert not view.flags.c_contiguous
        lt = [("v", view)]
        first = mtm.matchTemplates(lt, coins, score_threshold=0.5)
        assert first == hits_of(O.match_templates([("v", np.ascontiguousarray(view))], coins, score_threshold=0.5))
        keep = base.copy()
        base[...] = coins[150:150 + base.shape[0], 200:200 + base.shape[1]]     # same view object, other pixels
        second = mtm.matchTemplates(lt, coins, score_threshold=0.5)
        assert second == hits_of(O.match_templates([("v", np.ascontiguousarray(view))], coins, score_threshold=0.5))
        assert second != first
        base[...] = keep


@pytest.mark.gpu
def test_pinned_image_arrays(mtm, coins):
    """MTM.pinned_empty: numpy arrays in page-locked memory behave like any other array (same hits, views, dtypes);
    the block is released with its last view."""
    small, big = coin_templates(coins)
    lt = [("s", small), ("b", big)]
    ref = mtm.matchTemplates(lt, coins, score_threshold=0.5)
    p = mtm.pinned_empty(coins.shape, coins.dtype)
    assert p.shape == coins.shape and p.dtype == coins.dtype and p.flags.c_contiguous and p.flags.writeable
    p[...] = coins
    assert mtm.matchTemplates(lt, p, score_threshold=0.5) == ref
    assert mtm.matchTemplates(lt, p[:, ::-1][:, ::-1], score_threshold=0.5) == ref           # a view of it
    big_img = np.tile(coins, (5, 6))                      # banded upload path (>= 1 Mpx)
    pb = mtm.pinned_empty(big_img.shape)
    pb[...] = big_img
    assert mtm.matchTemplates(lt, pb, score_threshold=0.8) == mtm.matchTemplates(lt, big_img, score_threshold=0.8)
    p16 = mtm.pinned_empty((40, 50, 3), np.uint16)
    assert p16.shape == (40, 50, 3) and p16.dtype == np.uint16 and p16.nbytes == 40 * 50 * 3 * 2
    assert mtm.pinned_empty(0).size == 0
    del p, pb, p16


@pytest.mark.gpu
def test_find_matches_async(mtm, coins):
    lib = mtm._lib
    small, big = coin_templates(coins)
    ctx = lib.Context(0)
    try:
        ctx.set_image(coins)
        with pytest.raises(lib.MtmError):
            ctx.find_matches_wait()                        # nothing in flight
        for method, mode, thr in ((5, 0, 0.5), (1, 0, 0.3), (3, 1, 0.5), (5, 0, 0.99)):
            ctx.set_templates([(small, None), (big, None)], method)
            ref = ctx.find_matches(mode, thr).copy()
            for _ in range(3):
                ctx.find_matches_async(mode, thr)
                got = ctx.find_matches_wait()
                assert got.tobytes() == ref.tobytes()
                assert ctx.timing()["n_hits"] == len(ref)
        ctx.find_matches_async(0, 0.5)
        with pytest.raises(lib.MtmError):
            ctx.find_matches_async(0, 0.5)                 # one call in flight per context
        with pytest.raises(lib.MtmError, match="in flight"):
            ctx.set_option(6, 1)                           # ... and the context belongs to it
        with pytest.raises(lib.MtmError, match="in flight"):
            ctx.set_image(coins)
        with pytest.raises(lib.MtmError, match="in flight"):
            ctx.find_matches(0, 0.5)
        ctx.find_matches_wait()
        with pytest.raises(lib.MtmError):
            ctx.find_matches_wait()                        # already collected
        # a context without image and templates
        empty = lib.Context(0)
        try:
            empty.find_matches_async(0, 0.5)               # no image, no templates: refused or empty
            assert len(empty.find_matches_wait()) == 0
        except lib.MtmError as e:
            assert str(e)
        with pytest.raises(lib.MtmError):
            empty.find_matches_wait()                      # a refused call is not in flight
        # low threshold: more hits than the first fetch holds (overflow inside the split call)
        ctx.set_templates([(small, None)], 5)
        ref = ctx.find_matches(0, -1.0).copy()
        ctx.find_matches_async(0, -1.0)
        assert ctx.find_matches_wait().tobytes() == ref.tobytes()
        ctx.find_matches_async(0, 0.5)                     # destroyed with a call in flight: must not hang or crash
    finally:
        del ctx


# ------------------------------------------------------------------------------------------------
# state machine of a context: random sequences of uploads, template sets, options and queries
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(4))
def test_context_state_sequences(mtm, seed):
    """Everything a context caches (planes of two image slots, squares of the image, packs, the identical-
    templates shortcut, hits-only back-off, the image staged by mtm_find_matches_next) must follow the
    inputs: after any sequence of calls the answers are those of a fresh computation."""
    rng = np.random.default_rng(4000 + seed)
    ctx = mtm._lib.Context(0)
    images, templs, method, staged = None, None, 5, None

    def new_image(dtype):
        H, W = int(rng.integers(60, 200)), int(rng.integers(80, 300))
        top = 65536 if dtype == np.uint16 else 256
        return rng.integers(0, top, (H, W)).astype(dtype)

    def new_templates(img, masked):
        out = []
        for i in range(int(rng.integers(1, 7))):
            h, w = int(rng.integers(4, 40)), int(rng.integers(4, 50))
            if i and rng.random() < 0.5:
                h, w = out[-1][0].shape
            y, x = int(rng.integers(0, img.shape[0] - h + 1)), int(rng.integers(0, img.shape[1] - w + 1))
            t = img[y:y + h, x:x + w].copy()
            m = None
            if masked and img.dtype == np.uint8:
                m = (rng.random(t.shape) > 0.3).astype(np.uint8) * 255
                m[0, 0] = 255
            out.append((t, m))
        return out

    def as_cv2(a):
        return a.astype(np.float32) if a is not None and a.dtype == np.uint16 else a

    def check(cur_img):
        thr = 0.6 if method != 1 else 0.25
        lt = [("t%d" % i, as_cv2(t)) + ((as_cv2(m),) if m is not None else ()) for i, (t, m) in enumerate(templs)]
        exp = O.find_matches(lt, as_cv2(cur_img), method=method, score_threshold=thr)
        return thr, lt, exp

    img = new_image(np.uint8)
    ctx.set_image(img)
    templs = new_templates(img, False)
    ctx.set_templates(templs, method)
    for step in range(30):
        op = int(rng.integers(0, 8))
        if op == 0:                                    # new image, maybe another size / dtype
            dt = np.uint16 if rng.random() < 0.25 else np.uint8
            if dt != img.dtype:                        # the pixel policy keeps image and templates alike
                img = new_image(dt)
                templs = new_templates(img, False)
                ctx.set_image(img)
                ctx.set_templates(templs, method)
            else:
                img = new_image(dt)
                if any(t.shape[0] > img.shape[0] or t.shape[1] > img.shape[1] for t, _ in templs):
                    templs = new_templates(img, False)
                    ctx.set_templates(templs, method)
                ctx.set_image(img)
        elif op == 1:                                  # new templates (sometimes masked, sometimes identical)
            if rng.random() < 0.3:
                ctx.set_templates(templs, method)      # identical: the shortcut
            else:
                masked = rng.random() < 0.4
                method = int(rng.choice([0, 3])) if masked else int(rng.choice([1, 3, 5]))
                templs = new_templates(img, masked)
                ctx.set_templates(templs, method)
        elif op == 2:                                  # options
            ctx.set_option(6, int(rng.integers(0, 2)))
            ctx.set_option(5, int(rng.integers(0, 2)))
        elif op == 3:                                  # tiny hit buffer: overflow / back-off paths
            ctx.set_option(3, int(rng.choice([8, 1 << 18])))
        elif op == 4 and method != 0:                  # stream step: current result + stage another image of the same kind
            nxt = new_image(img.dtype)
            if all(t.shape[0] <= nxt.shape[0] and t.shape[1] <= nxt.shape[1] for t, _ in templs):
                thr, lt, exp = check(img)
                got = ctx.find_matches(0, thr, next_image=nxt)
                assert len(got) == len(exp), ("next", step, len(got), len(exp))
                img = nxt
        elif op == 5:                                  # one score map
            li = int(rng.integers(0, len(templs)))
            t, m = templs[li]
            got = ctx.score_map(li, (img.shape[0] - t.shape[0] + 1, img.shape[1] - t.shape[1] + 1))
            exp = O.match_template(as_cv2(img), as_cv2(t), method, mask=as_cv2(m))
            ok = np.isfinite(exp)
            assert np.abs(got[ok].astype(np.float64) - exp[ok]).max() <= 1e-5 * max(1.0, float(np.abs(exp[ok]).max())), ("map", step)
        elif method != 0:                              # local or global search
            thr, lt, exp = check(img)
            got = ctx.find_matches(0, thr)
            assert len(got) == len(exp), ("find", step, method, len(got), len(exp))
            names = [(lt[int(r["templ_idx"])][0], (int(r["x"]), int(r["y"]), int(r["w"]), int(r["h"])), r["score"]) for r in got]
            assert_hits_equal(hits_json(names), hits_json(exp), tol=1e-5, ordered=False)


def test_dense_backoff_route_against_the_oracle(mtm):
    """Dense maps, the route taken while the back-off lasts: with a candidate capacity below the number of pixels above
    the threshold the first call overflows (hits-only), the following ones go to memory with segment flags (or the full
    peak pass) and must return the oracle's hits - at two capacities, for maxima and minima (TM_SQDIFF_NORMED), plateaus
    included (flat patches), peaks in the last column / row.  (Round 4 also listed row maxima as candidates on this
    route, MTM_DENSE_ROWMAX: slower than what it replaced, removed in round 5.)"""
    from MTM import _lib
    dense = synth.smooth_u8(7, (300, 520), scales=(3, 9, 27), noise=0.1)
    dense[40:70, 100:180] = 90                               # flat patches: plateaus of equal scores
    dense[200:240, 300:420] = 90
    rng = np.random.default_rng(9)
    lt = []
    for i in range(20):
        y, x = int(rng.integers(0, 300 - 24)), int(rng.integers(0, 520 - 32))
        lt.append(("t%d" % i, dense[y:y + 24, x:x + 32].copy()))
    lt.append(("right", dense[100:124, 488:520].copy()))     # exact copies at the last column / the last row of their maps:
    lt.append(("bottom", dense[276:300, 200:232].copy()))    # peaks whose right / lower neighbours do not exist
    for method, thr in ((5, 0.3), (1, 0.05)):
        n_above = n_row = 0
        for _, t in lt:
            m = O.match_template(dense, t, method).astype(np.float32)
            q = -m if method == 1 else m
            above = q > np.float32(-thr if method == 1 else thr)
            left = np.pad(q[:, :-1], ((0, 0), (1, 0)), constant_values=-np.inf)
            right = np.pad(q[:, 1:], ((0, 0), (0, 1)), constant_values=-np.inf)
            n_above += int(above.sum())
            n_row += int((above & ~(left > q) & ~(right > q)).sum())
        assert n_row * 2 < n_above, (n_row, n_above)
        exp = hits_json(O.find_matches(lt, dense, method=method, score_threshold=thr))
        assert len(exp) > 500
        # the kernel's row test only sees the 256 pixels of a wave: a few more candidates than the count above
        for cap in ((n_above + 2 * n_row) // 3, max(64, n_row // 4)):
            c = _lib.Context(0)
            try:
                c.set_option(_lib.OPT_HIT_CAPACITY, cap)
                c.set_templates([(t, None) for _, t in lt], method)
                for k in range(4):
                    raw = c.find_matches_image(dense, _lib.PEAKS_LOCAL, thr)
                    assert c.timing()["hits_only"] in (0, 2)      # (2: the flagged-segment peak pass of the back-off calls)
                    got = [(lt[int(r["templ_idx"])][0], (int(r["x"]), int(r["y"]), int(r["w"]), int(r["h"])), r["score"]) for r in raw]
                    assert len(got) == len(exp), (method, cap, k, len(got), len(exp))
                    assert_hits_equal(hits_json(got), exp, tol=1e-6, ordered=False)
            finally:
                c.close()


def test_sparse_maps_route_equals_the_full_maps_route(mtm):
    """Calls under the dense-map back-off write only the row segments in which something passes the threshold and scan
    those (sparse maps, timing hits_only == 2).  Against a context with MTM_SPARSE_MAPS=0 (full maps + full peak pass):
    the same records bit for bit - maxima and minima methods, both border rules, RGB, two size classes, a threshold
    nothing passes, a negative threshold (every segment flagged), a constant image (skimage: no peaks in a map every
    pixel of which equals its local maximum) and plateaus at segment borders."""
    from MTM import _lib
    rng = np.random.default_rng(77)
    dense = synth.smooth_u8(17, (260, 700), scales=(3, 9, 27), noise=0.1)
    dense_rgb = np.stack([dense, np.roll(dense, 5, 0), np.roll(dense, 7, 1)], axis=2)
    flat = np.full((260, 700), 90, np.uint8)
    blocky = np.kron(rng.integers(0, 256, (26, 70), dtype=np.uint8), np.ones((10, 10), np.uint8)).astype(np.uint8)   # plateaus
    def templates(img, sizes, n):
        out = []
        for i in range(n):
            h, w = sizes[i % len(sizes)]
            y, x = int(rng.integers(0, img.shape[0] - h)), int(rng.integers(0, img.shape[1] - w))
            out.append((np.ascontiguousarray(img[y:y + h, x:x + w]), None))
        return out
    yy, xx = np.mgrid[0:24, 0:32]
    disc = (((yy - 11.5) / 12.0) ** 2 + ((xx - 15.5) / 16.0) ** 2 <= 1.0).astype(np.uint8) * 255
    cases = [
        ("ccoeff_normed", dense, templates(dense, [(24, 32)], 20), 5, 0.3),
        ("two classes", dense, templates(dense, [(24, 32), (17, 40)], 40), 5, 0.35),
        ("row-multiplexed tiling (5 templates)", dense, templates(dense, [(24, 32)], 5), 5, 0.3),
        ("one template + a class of 20", dense, templates(dense, [(30, 30)], 1) + templates(dense, [(24, 32)], 20), 1, 0.6),
        ("sqdiff_normed (minima)", dense, templates(dense, [(24, 32)], 20), 1, 0.6),
        ("ccorr_normed", dense, templates(dense, [(24, 32)], 20), 3, 0.97),
        ("rgb", dense_rgb, templates(dense_rgb, [(24, 32)], 20), 5, 0.3),
        ("masked ccorr_normed (20 templates, one disc mask)", dense, [(t, disc) for t, _ in templates(dense, [(24, 32)], 20)], 3, 0.97),
        ("masked sqdiff_normed (3 templates)", dense, [(t, disc) for t, _ in templates(dense, [(24, 32)], 3)], 1, 0.5),
        ("nothing passes", dense, templates(dense, [(24, 32)], 20), 5, 1.5),
        ("negative threshold", dense, templates(dense, [(24, 32)], 20), 5, -0.5),
        ("constant image", flat, templates(dense, [(24, 32)], 20), 5, -0.5),
        ("plateaus", blocky, templates(blocky, [(20, 30)], 20), 5, 0.2),
    ]
    def make(sparse):
        old = os.environ.get("MTM_SPARSE_MAPS")
        os.environ["MTM_SPARSE_MAPS"] = "1" if sparse else "0"
        try:
            c = _lib.Context(0)
        finally:
            if old is None:
                del os.environ["MTM_SPARSE_MAPS"]
            else:
                os.environ["MTM_SPARSE_MAPS"] = old
        c.set_option(_lib.OPT_HIT_CAPACITY, 1024)
        return c
    ca, cb = make(True), make(False)
    try:
        forced = any(os.environ.get(v) for v in ("MTM_KERNEL", "MTM_HITS_ONLY", "MTM_ROW_MUX", "MTM_SPARSE_MAPS"))
        routes = []
        for border in (_lib.BORDER_CONSTANT, _lib.BORDER_NEAREST):
            for c in (ca, cb):
                c.set_option(_lib.OPT_PEAK_BORDER, border)
            for name, img, tl, method, thr in cases:
                # the dense image first: its candidates overflow the 1024-record list and start the back-off
                warm_img = dense_rgb if img.ndim == 3 else dense
                warm = templates(warm_img, [(24, 32)], 20)
                # (from a cleared back-off state - MTM_OPT_HITS_ONLY clears it -: the next 16 calls run under it)
                for c in (ca, cb):
                    if not forced:
                        c.set_option(_lib.OPT_HITS_ONLY, 1)
                    c.set_option(_lib.OPT_HIT_CAPACITY, 1024)     # (a call with more peaks than that has grown it)
                    c.search(warm, warm_img, 5, _lib.PEAKS_LOCAL, 0.3)
                ra = ca.search(tl, img, method, _lib.PEAKS_LOCAL, thr)
                route = ca.timing()["hits_only"]
                rb = cb.search(tl, img, method, _lib.PEAKS_LOCAL, thr)
                assert cb.timing()["hits_only"] != 2
                routes.append((name, route))
                assert len(ra) == len(rb), (name, border, len(ra), len(rb))
                assert ra.tobytes() == rb.tobytes(), (name, border)
                if name in ("nothing passes", "constant image"):
                    assert len(ra) == 0, name
                if name in ("ccoeff_normed", "plateaus") and border == _lib.BORDER_CONSTANT:
                    lt = [("t%d" % i, t) for i, (t, _) in enumerate(tl)]
                    exp = hits_json(O.find_matches(lt, img, method=method, score_threshold=thr))
                    got = [(lt[int(r["templ_idx"])][0], (int(r["x"]), int(r["y"]), int(r["w"]), int(r["h"])), r["score"]) for r in ra]
                    assert_hits_equal(hits_json(got), exp, tol=1e-6, ordered=False)
        if not forced:          # every case ran under the back-off, on the flagged-segment route - or, where every pixel
            # is a "peak" (the per-segment lists are bounded), on the full scan that takes over from it
            assert [x for x in routes if x[1] != 2 and x[0] not in ("negative threshold", "constant image")] == []
            assert any(r == 2 for _, r in routes)
    finally:
        ca.close()
        cb.close()


def test_fused_nms_call_equals_find_then_nms(mtm, monkeypatch):
    """mtm_find_matches_image_nms (what MTM.matchTemplates calls on the 8-bit path) returns the hits mtm_nms selects from the
    list mtm_find_matches_image returns, in the same order, bit for bit - on the host route (few hits), and on the device
    route (dense image under the back-off, the peaks suppressed on the GPU: forced down to short lists here), for maxima
    and minima methods, several overlap limits, a finite N_object, ties in the score (exact copies score 1.0) and two
    template sizes (grid cell = the larger box)."""
    from MTM import _lib
    rng = np.random.default_rng(5)
    dense = synth.smooth_u8(23, (300, 640), scales=(3, 9, 27), noise=0.1)
    sparse = rng.integers(0, 256, dense.shape, dtype=np.uint8)
    lt = []
    for i in range(24):
        h, w = ((24, 32), (30, 20))[i % 2]
        y, x = int(rng.integers(0, 300 - h)), int(rng.integers(0, 640 - w))
        lt.append((dense[y:y + h, x:x + w].copy(), None))
        for _ in range(3):                                           # exact copies: equal scores of 1.0 in both images
            yy, xx = int(rng.integers(0, 300 - h)), int(rng.integers(0, 640 - w))
            sparse[yy:yy + h, xx:xx + w] = lt[-1][0]
    monkeypatch.setenv("MTM_NMS_DEVICE_MIN", "32")
    c = _lib.Context(0)
    ref = _lib.Context(0)
    try:
        routes = set()
        for method, thr in ((5, 0.3), (1, 0.55), (3, 0.95)):
            for img in (sparse, dense, dense, dense):
                for c_ in (c, ref):
                    c_.set_option(_lib.OPT_HIT_CAPACITY, 2048 if img is dense else 1 << 18)
                for overlap, n_obj in ((0.25, -1), (0.0, -1), (0.6, 7), (1.0, -1)):
                    raw = ref.search(lt, img, method, _lib.PEAKS_LOCAL, thr)
                    idx = _lib.nms_hits(raw, thr, overlap, ascending=(method == 1))
                    exp = raw[idx] if n_obj < 0 else raw[idx][:n_obj]
                    got = c.search_nms(lt, img, method, thr, overlap, n_obj)
                    tm = c.timing()
                    routes.add(tm["hits_only"])
                    assert tm["n_hits"] == len(raw), (method, overlap, tm["n_hits"], len(raw))
                    assert len(got) == len(exp), (method, overlap, n_obj, len(got), len(exp), len(raw))
                    assert got.tobytes() == exp.tobytes(), (method, overlap, n_obj)
        if not any(os.environ.get(v) for v in ("MTM_KERNEL", "MTM_HITS_ONLY", "MTM_SPARSE_MAPS", "MTM_NMS_DEVICE_MIN")):
            assert 2 in routes and 1 in routes, routes      # the device's suppression ran (flagged-segment route) and the host's
    finally:
        c.close()
        ref.close()


def test_dense_maps_candidate_overflow(mtm):
    """Smooth images at a low threshold: far more pixels above the threshold than the candidate list holds.  The
    overflowing hits-only launch leaves early, the call is repeated with the maps in memory and the full peak pass;
    the following calls go there directly (back-off), later ones retry.  Every call returns the oracle's hits - also
    when the image turns sparse again, and with the wave-aggregated candidate append of the score kernel in between."""
    from MTM import _lib
    dense = synth.smooth_u8(7, (300, 520), scales=(3, 9, 27), noise=0.1)
    rng = np.random.default_rng(8)
    sparse = rng.integers(0, 256, dense.shape, dtype=np.uint8)
    lt = []
    for i in range(20):                                   # two work-item groups
        y, x = int(rng.integers(0, 300 - 24)), int(rng.integers(0, 520 - 32))
        lt.append(("t%d" % i, dense[y:y + 24, x:x + 32].copy()))
        sparse[y:y + 24, x:x + 32] = lt[-1][1]
    exp_dense = hits_json(O.find_matches(lt, dense, method=5, score_threshold=0.3))
    exp_sparse = hits_json(O.find_matches(lt, sparse, method=5, score_threshold=0.3))
    assert len(exp_dense) > 3000 and 20 <= len(exp_sparse) < 200
    c = _lib.Context(0)
    try:
        c.set_option(_lib.OPT_HIT_CAPACITY, 2048)             # candidate list of 2048 records: the dense image overflows it
        c.set_templates([(t, None) for _, t in lt], 5)
        modes = []
        for k, (im, exp) in enumerate([(dense, exp_dense)] * 3 + [(sparse, exp_sparse)] * 2 + [(dense, exp_dense)] * 20 +
                                      [(sparse, exp_sparse)] * 3):
            raw = c.find_matches_image(im, _lib.PEAKS_LOCAL, 0.3)
            modes.append(c.timing()["hits_only"])
            got = [(lt[int(r["templ_idx"])][0], (int(r["x"]), int(r["y"]), int(r["w"]), int(r["h"])), r["score"]) for r in raw]
            assert len(got) == len(exp), (k, len(got), len(exp))
            assert_hits_equal(hits_json(got), exp, tol=1e-6, ordered=False)
            # the records' order: template, descending score, row-major position (thousands: the radix sort)
            keys = [(int(r["templ_idx"]), -float(r["score"]), int(r["y"]), int(r["x"])) for r in raw]
            assert keys == sorted(keys), k
        if not any(os.environ.get(v) for v in ("MTM_KERNEL", "MTM_HITS_ONLY")):
            # overflow -> maps; then the back-off period, on sparse maps (round 4) unless they are switched off
            assert modes[0] == 0 and modes[1] == (0 if os.environ.get("MTM_SPARSE_MAPS") == "0" else 2)
            for _ in range(80):                               # sparse again: hits-only is back once the period ran out
                raw = c.find_matches_image(sparse, _lib.PEAKS_LOCAL, 0.3)
                assert len(raw) == len(exp_sparse)
                if c.timing()["hits_only"] == 1:
                    break
            assert c.timing()["hits_only"] == 1
        # a list that holds everything: same hits from the hits-only route (device hash verification)
        c.set_option(_lib.OPT_HIT_CAPACITY, 1 << 18)
        c.set_option(_lib.OPT_HITS_ONLY, 1)                   # (also clears the back-off)
        raw = c.find_matches_image(dense, _lib.PEAKS_LOCAL, 0.3)
        got = [(lt[int(r["templ_idx"])][0], (int(r["x"]), int(r["y"]), int(r["w"]), int(r["h"])), r["score"]) for r in raw]
        assert_hits_equal(hits_json(got), exp_dense, tol=1e-6, ordered=False)
        if not any(os.environ.get(v) for v in ("MTM_KERNEL", "MTM_HITS_ONLY")):
            assert c.timing()["hits_only"] == 1
    finally:
        c.close()


@pytest.mark.parametrize("kind", ["uint16", "float32", "slabs"])
def test_dense_maps_other_kernels(mtm, kind):
    """The same dense regime on the kernels with their own candidate append (uint16 finishing pass, float32 kernel,
    slab combination): a candidate list that overflows, one that holds everything, and map mode return the same
    records; uint16 / slabs also equal the oracle's."""
    from MTM import _lib
    dense = synth.smooth_u8(17, (260, 480), scales=(3, 9, 27), noise=0.1)
    if kind == "slabs":
        lt = [("wide", dense[30:50, 40:40 + 300].copy()), ("wide2", dense[200:220, 100:100 + 300].copy())]   # w > 256
        img = dense
    else:
        lt = synth.cut_templates(3, dense, 20, 28)
        img = dense.astype(np.uint16) * 201 + 7 if kind == "uint16" else dense.astype(np.float32) * np.float32(1.5) + np.float32(3)
        lt = [(n, (t.astype(np.uint16) * 201 + 7) if kind == "uint16" else t.astype(np.float32) * np.float32(1.5) + np.float32(3))
              for n, t in lt]
    c = _lib.Context(0)
    try:
        c.set_templates([(t, None) for _, t in lt], 5)
        res = {}
        small = 64 if kind == "slabs" else 1024
        for name, cap, honly in (("maps", 1 << 18, 0), ("list_holds", 1 << 18, 1), ("overflow", small, 1), ("overflow_again", small, 1)):
            c.set_option(_lib.OPT_HIT_CAPACITY, cap)
            c.set_option(_lib.OPT_HITS_ONLY, honly)
            res[name] = c.find_matches_image(img, _lib.PEAKS_LOCAL, 0.3).copy()
        n = len(res["maps"])
        assert n > (15 if kind == "slabs" else 2000), n
        for name in ("list_holds", "overflow", "overflow_again"):
            assert res[name].tobytes() == res["maps"].tobytes(), (kind, name, len(res[name]), n)
        if kind != "float32":
            exp = O.find_matches([(a, b.astype(np.float32)) for a, b in lt], img.astype(np.float32), method=5, score_threshold=0.3)
            got = [(lt[int(r["templ_idx"])][0], (int(r["x"]), int(r["y"]), int(r["w"]), int(r["h"])), r["score"]) for r in res["maps"]]
            assert_hits_equal(hits_json(got), hits_json(exp), tol=1e-6, ordered=False)
    finally:
        c.close()


# ------------------------------------------------------------------------------------------------
# RGB uint8: the compile-time-method epilogue with per-channel window sums (CH = 3)
# ------------------------------------------------------------------------------------------------
def test_rgb_lean_path(mtm, ctx):
    rng = np.random.default_rng(909)
    img = rng.integers(0, 256, (170, 290, 3), dtype=np.uint8)
    img[50:90, 100:180] = (40, 90, 200)                    # flat windows in every channel
    lt = []
    for i in range(20):                                     # 20 templates of one size: two MFMA groups
        y, x = int(rng.integers(0, 170 - 28)), int(rng.integers(0, 290 - 36))
        t = img[y:y + 28, x:x + 36].copy()
        if i % 2:
            t = np.clip(t.astype(np.int32) + rng.integers(-25, 26, t.shape), 0, 255).astype(np.uint8)
        lt.append(("t%d" % i, t))
    lt.append(("big", img[10:10 + 70, 20:20 + 130].copy()))  # chunked rows, three 64-tap blocks
    set_kernel(ctx, "mfma")
    try:
        for method in range(6):
            for exact in (1, 0):
                set_exact(ctx, exact)
                for name, t in (lt[0], lt[7], lt[-1]):
                    got = mtm.computeScoreMap(t, img, method)
                    assert ctx.timing()["kernel_used"] == 3
                    exp = O.match_template(img, t, method)
                    if exact or method in (0, 2, 4):
                        assert np.array_equal(got, exp), (method, exact, name, float(np.abs(got - exp).max()))
                    else:
                        ulp_close(got, exp)
        for exact, (method, thr) in [(e, mt) for e in (1, 0) for mt in ((5, 0.5), (3, 0.85), (1, 0.3))]:
            set_exact(ctx, exact)
            res = []
            for honly in (0, 1):
                ctx.set_option(6, honly)
                res.append(mtm.findMatches(lt, img, method=method, score_threshold=thr))
                if os.environ.get("1") != "0":      # hits-only needs the fused candidates
                    assert ctx.timing()["hits_only"] == honly
            assert canon(res[0]) == canon(res[1])
            exp = O.find_matches(lt, img, method=method, score_threshold=thr)
            assert len(res[1]) == len(exp) and len(exp) >= 10
            assert_hits_equal(canon(res[1]), canon(exp), tol=1e-6)
            # N_object == 1: fused global extremum (21 RGB templates: plain and row-multiplexed classes) == maps
            one = []
            for honly in (0, 1):
                ctx.set_option(6, honly)
                one.append(mtm.findMatches(lt, img, method=method, N_object=1))
            assert one[0] == one[1]
            assert_hits_equal(one[1], hits_json(O.find_matches(lt, img, method=method, N_object=1)), tol=1e-6)
    finally:
        set_kernel(ctx, "auto")
        restore_exact(ctx)
        restore_hits_only(ctx)


def test_bench_gpus_flag_runs_the_device_group_without_a_launcher():
    """`python bench.py --gpus 2` with WORLD_SIZE unset: one process, two contexts behind mtm_group (both on this box's
    one GPU: BENCH_GROUP_ALIAS), 64 units sharded 32 / 32, one JSON line with n_gpus 2, every planted template found."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["BENCH_GROUP_ALIAS"] = "1"
    env["BENCH_PREWARM_S"] = "0.05"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["units"] == 64 and out["config"]["processes"] == 1
    assert out["multi_gpu"]["units_by_device"] == [32, 32] and len(out["multi_gpu"]["kernel_ms_per_step_by_device"]) == 2
    assert out["planted_found"] and out["hits"] >= 64 and out["value"] > 0


def test_4k_photograph_default_mode_hit_lists(mtm):
    """A 4K photograph-like image (smooth score maps: thousands of peaks per template, plateau-free but full of
    near-ties) in the library's DEFAULT mode - reciprocal normalisation (not MTM_OPT_EXACT_DIV), hits-only first, map
    mode after the candidate list overflowed - against the oracle: the same hit boxes in the same order for four
    templates, before and after the NMS.  A last-bit difference of the default normalisation (<= 1 float32 ulp on
    ~1e-8 of the pixels) could in principle make or break a 3x3 equality: this is the check that it does not on the
    regime where such equalities are densest.  (Round-2 review: no default-mode list check on smooth images at 4K.)"""
    import MTM
    img = synth.smooth_u8(11, (2160, 3840))
    lt = synth.cut_templates(5, img, 4, 64)
    exp = O.find_matches(lt, img, method=5, score_threshold=0.5)
    assert len(exp) > 1000
    for attempt in range(3):           # hits-only (overflows), map mode, back-off period: every call the same list
        got = MTM.findMatches(lt, img, method=5, score_threshold=0.5)
        assert len(got) == len(exp), (attempt, len(got), len(exp))
        assert_hits_equal(hits_json(got), hits_json(exp), tol=1e-6, ordered=False)
    gm = MTM.matchTemplates(lt, img, method=5, score_threshold=0.5, maxOverlap=0.25)
    em = O.match_templates(lt, img, method=5, score_threshold=0.5, maxOverlap=0.25)
    assert len(gm) == len(em) > 100
    assert_hits_equal(hits_json(gm), hits_json(em), tol=1e-6, ordered=False)
    # higher threshold: sparse again (hits-only route with the per-lane screen), same parity
    got = MTM.findMatches(lt, img, method=5, score_threshold=0.9)
    exp9 = O.find_matches(lt, img, method=5, score_threshold=0.9)
    assert len(got) == len(exp9) >= 4
    assert_hits_equal(hits_json(got), hits_json(exp9), tol=1e-6, ordered=False)
    restore_hits_only(MTM._lib.default_context())      # clears the back-off for the tests that follow


# ---- round 5: the float32 route's listing decisions rest on a per-output error bound, not on empirical margins ----------
def _bf16_bound_map(img, templ, method, pieces=3):
    """numpy restatement of Bf16Params::rig's bound M(x, y) for a single-channel float32 image and one template:
    rig_eps * sqrt(sum (I - mu)^2) / sq * (escale * sqrt(sum (T - mean)^2) / templ_norm), with mu the constant the
    bf16 kernel's work item (128 columns x 4 rows of outputs) subtracts: the mean of an 8 x 8 sample grid over its patch."""
    I = img.astype(np.float64)
    T = templ.astype(np.float64)
    rows, cols = I.shape
    h, w = T.shape
    oh, ow = rows - h + 1, cols - w + 1
    A = float(h * w)
    nkb = (w + 31) // 32
    lds_cols = 128 + 32 * nkb
    eps = 3.0518e-5 + 2.0 * 3.0 * h * nkb * 5.97e-8
    if pieces == 1:         # the one-product screen (bf16_rig_eps(.., np = 1), the accumulation term not doubled)
        eps = 2.0 ** -7 * (1.0 + 2.0 ** -9) + 2.0 * 1.0 * h * nkb * 5.97e-8
    c1 = np.zeros((rows + 1, cols + 1))
    c1[1:, 1:] = I.cumsum(0).cumsum(1)
    c2 = np.zeros((rows + 1, cols + 1))
    c2[1:, 1:] = (I * I).cumsum(0).cumsum(1)
    box = lambda c: c[h:, w:] - c[:-h, w:] - c[h:, :-w] + c[:-h, :-w]     # noqa: E731
    S1, S2 = box(c1), box(c2)
    mu = np.zeros((oh, ow))
    for y0 in range(0, oh, 4):
        sr = np.minimum(y0 + (np.arange(8) * (h + 2)) // 7, rows - 1)
        for x0 in range(0, ow, 128):
            sc = np.minimum(x0 + (np.arange(8) * (lds_cols - 1)) // 7, cols - 1)
            mu[y0:y0 + 4, x0:x0 + 128] = np.float32(img[np.ix_(sr, sc)].astype(np.float32).mean())
    s2c = np.maximum(S2 - 2.0 * mu * S1 + A * mu * mu, 0.0) * 1.000001 + 1e-12 * (np.abs(S2) + A * mu * mu)
    t2c = ((T - T.mean()) ** 2).sum()
    if method == 5:
        diff2 = np.maximum(S2 - S1 * S1 / A, 0.0)
        tn = np.sqrt(t2c)
        esc = 1.0
    else:
        diff2 = S2
        tn = np.sqrt((T * T).sum())
        esc = 2.0 if method == 1 else 1.0
    flat = diff2 <= np.minimum(0.5, 10.0 * np.finfo(np.float32).eps * S2)
    sq = np.where(flat, 0.0, np.sqrt(diff2))
    with np.errstate(divide="ignore", invalid="ignore"):
        M = eps * np.sqrt(s2c) / sq * (esc * np.sqrt(t2c) / tn) + 3e-7
    return np.where(sq > 0.0, M, 0.0), sq > 0.0


def _step_image(seed, shape=(300, 420), lo=0.0, hi=1.0, noise=2e-3):
    """A 0 -> 1 brightness step down the middle with low-contrast texture on both sides: windows a few columns away
    from the step are nearly flat while their work item's tile constant sits half way up the step."""
    rng = np.random.default_rng(seed)
    im = np.full(shape, lo, np.float32)
    im[:, shape[1] // 2:] = hi
    im += rng.normal(0.0, noise, shape).astype(np.float32)
    im[40:90, 20:90] += rng.normal(0.0, 0.2, (50, 70)).astype(np.float32)     # one textured patch on the dark side
    return im


@pytest.mark.gpu
@pytest.mark.parametrize("pieces", [3, 1])
def test_float32_error_bound_holds(pieces):
    """|bf16 kernel score - float64 kernel score| <= M at every output, M the bound the refined routes list by
    (Bf16Params::rig): on the adversarial geometry (low-contrast windows beside a brightness step, templates cut across
    the step) and on random data, for the three normalised methods.  Also reports how tight the bound is.
    pieces = 1: the one-product screen of the hits-only routes (round 6; MTM_OPT_F32_MFMA = 4 publishes its raw scores for
    this test alone) against ITS bound, 2^-7 (1 + 2^-9) of the norms' product + the accumulation.
    Reference: MTM/__init__.py:71-74 (everything not uint8 is matched as float32)."""
    if os.environ.get("MTM_KERNEL", "auto") != "auto":
        pytest.skip("measures the bf16 kernel; the environment forces another one")
    from MTM import _lib
    fast, exact = _lib.Context(0), _lib.Context(0)
    fast.set_option(_lib.OPT_F32_MFMA, 2 if pieces == 3 else 4)           # the bf16 scores as they are
    exact.set_option(_lib.OPT_F32_MFMA, 0)
    rng = np.random.default_rng(5)
    step = _step_image(1)
    rnd = rng.normal(10.0, 3.0, (260, 400)).astype(np.float32)
    ramp = (np.linspace(0, 500, 400, dtype=np.float32)[None, :] + rng.normal(0, 0.05, (260, 400)).astype(np.float32))
    worst = 0.0
    try:
        for name, im in (("step", step), ("random", rnd), ("ramp", ramp)):
            cx = im.shape[1] // 2
            templs = [np.ascontiguousarray(im[100:132, cx - 20:cx + 20]),     # across the step
                      np.ascontiguousarray(im[10:34, 30:54]), np.ascontiguousarray(im[150:214, cx + 40:cx + 104])]
            for method in (5, 3, 1):
                for t in templs:
                    shape = (im.shape[0] - t.shape[0] + 1, im.shape[1] - t.shape[1] + 1)
                    maps = []
                    for ctx in (fast, exact):
                        ctx.set_image(im)
                        ctx.set_templates([(t, None)], method)
                        maps.append(ctx.score_map(0, shape).astype(np.float64))
                    assert (fast.timing()["kernel_used"] == 5 or not default_routes()) and exact.timing()["kernel_used"] == 0
                    assert fast.timing()["f32_pieces"] == pieces
                    M, live = _bf16_bound_map(im, t, method, pieces)
                    unsat = live & (np.abs(maps[1]) < 1.0) & (np.abs(maps[0]) < 1.0)
                    d = np.abs(maps[0] - maps[1])
                    assert (d[unsat] <= M[unsat]).all(), (name, method, t.shape, float((d[unsat] / M[unsat]).max()))
                    if unsat.any():
                        worst = max(worst, float((d[unsat] / M[unsat]).max()))
        print("float32 bound, %d piece product(s): worst |error| / bound = %.3f" % (pieces, worst))
        assert worst <= 1.0
    finally:
        fast.close()
        exact.close()


@pytest.mark.gpu
def test_float32_adversarial_lists_equal_the_float64_kernels():
    """The case the empirical margins of rounds 3-4 did not cover: flat / low-contrast windows adjacent to a brightness
    step (their tile constant is far from their own mean, so the bf16 pieces carry the brightness, not the contrast),
    templates cut across the step, and thresholds placed within 1e-5 of true peak scores.  The default float32 route must
    return the float64 kernel's records - same pixels, order and float32 scores - by local extrema (hits-only and with maps
    in memory) and by global extremum.  Reference: MTM/__init__.py:71-74, :45, :226."""
    if not default_routes():
        pytest.skip("asserts the default float32 routes")
    from MTM import _lib
    fast, exact = _lib.Context(0), _lib.Context(0)
    exact.set_option(_lib.OPT_F32_MFMA, 0)
    routes = set()
    try:
        for seed, (lo, hi, noise) in enumerate(((0.0, 1.0, 2e-3), (0.0, 255.0, 0.5), (100.0, 101.0, 1e-2), (0.0, 1.0, 2e-5))):
            im = _step_image(seed + 3, lo=lo, hi=hi, noise=noise)
            cx = im.shape[1] // 2
            lt = [np.ascontiguousarray(im[100:132, cx - 20:cx + 20]), np.ascontiguousarray(im[200:232, cx - 8:cx + 32]),
                  np.ascontiguousarray(im[45:77, 30:70]), np.ascontiguousarray(im[20:52, cx + 60:cx + 100]),
                  np.ascontiguousarray(im[150:182, 10:50])]
            templs = [(t, None) for t in lt]
            for method in (5, 3, 1):
                base_thr = {5: 0.3, 3: 0.9, 1: 0.2}[method]
                ref0 = exact.search(templs, im, method, _lib.PEAKS_LOCAL, base_thr)
                # thresholds a hair on either side of true peak scores (and the base threshold itself)
                thrs = [base_thr]
                for s in np.unique(ref0["score"])[:: max(1, len(np.unique(ref0["score"])) // 6)][:6]:
                    if 0.0 < float(s) < 1.0:
                        thrs += [float(s) - 1e-5, float(s) + 1e-5, float(np.nextafter(np.float32(s), np.float32(0)))]
                for thr in thrs:
                    ref = exact.search(templs, im, method, _lib.PEAKS_LOCAL, thr)
                    for honly in (1, 0):
                        fast.set_option(_lib.OPT_HITS_ONLY, honly)
                        got = fast.search(templs, im, method, _lib.PEAKS_LOCAL, thr)
                        tm = fast.timing()
                        routes.add(tm["f32_route"])
                        assert tm["f32_route"] in (1, 2, 3), tm
                        assert len(got) == len(ref), (seed, method, thr, honly, tm["f32_route"], len(got), len(ref))
                        for f in ("templ_idx", "x", "y", "w", "h"):
                            assert np.array_equal(got[f], ref[f]), (seed, method, thr, honly, tm["f32_route"], f)
                        assert np.array_equal(got["score"].view(np.uint32), ref["score"].view(np.uint32)), (seed, method, thr, honly)
                fast.set_option(_lib.OPT_HITS_ONLY, 1)
                ref = exact.search(templs, im, method, _lib.PEAKS_GLOBAL, 0.0)
                got = fast.search(templs, im, method, _lib.PEAKS_GLOBAL, 0.0)
                routes.add(fast.timing()["f32_route"])
                for f in ("templ_idx", "x", "y"):
                    assert np.array_equal(got[f], ref[f]), (seed, method, f, fast.timing()["f32_route"])
                assert np.array_equal(got["score"].view(np.uint32), ref["score"].view(np.uint32)), (seed, method)
        assert 1 in routes, routes
    finally:
        fast.close()
        exact.close()


@pytest.mark.gpu
def test_float32_one_product_screen_and_its_fallback():
    """Round 6: the hits-only refined routes screen with ONE bfloat16 piece product first (mtm_timing.f32_pieces = 1; bound
    2^-7 of the norms' product instead of 2^-15) - the records must stay the float64 kernel's.  On noise, with a threshold
    two standard deviations up the score distribution, the one-product screen lists a multiple of what the three-product
    screen lists: walking the list capacity down finds capacities the first overflows and the second fits - the launch is
    then repeated with three products (f32_pieces = 3, still route 1), and the calls after it start there until the
    back-off ends.  MTM_OPT_F32_MFMA = 3 never takes the tier.  Reference: MTM/__init__.py:71-74, :45-52."""
    if not default_routes():
        pytest.skip("asserts the default float32 routes")
    from MTM import _lib
    fast, three, exact = _lib.Context(0), _lib.Context(0), _lib.Context(0)
    exact.set_option(_lib.OPT_F32_MFMA, 0)
    three.set_option(_lib.OPT_F32_MFMA, 3)
    rng = np.random.default_rng(77)
    im = rng.normal(50.0, 9.0, (260, 400)).astype(np.float32)
    lt = [(np.ascontiguousarray(im[y:y + 24, x:x + 24]), None) for y, x in ((10, 20), (120, 300), (200, 77))]
    seen = set()
    try:
        for method, thr in ((5, 0.085), (3, 0.9712), (1, 0.0576)):
            ref = exact.search(lt, im, method, _lib.PEAKS_LOCAL, thr).copy()
            assert len(ref) >= 3
            for cap in (1 << 18, 40000, 20000, 10000, 5000, 2500):
                for ctx in (fast, three):
                    ctx.set_option(_lib.OPT_HIT_CAPACITY, cap)
                    ctx.set_option(_lib.OPT_HITS_ONLY, 1)          # (also ends a back-off)
                    got = ctx.search(lt, im, method, _lib.PEAKS_LOCAL, thr)
                    tm = ctx.timing()
                    assert got.tobytes() == ref.tobytes(), (method, cap, tm["f32_route"], tm["f32_pieces"], len(got), len(ref))
                    if ctx is three:
                        assert tm["f32_pieces"] in (0, 3), tm
                    else:
                        seen.add((method, cap, tm["f32_route"], tm["f32_pieces"]))
                        if tm["f32_route"] == 1 and tm["f32_pieces"] == 3:
                            # the screen overflowed in this call: the next call starts with three products (back-off) ...
                            again = fast.search(lt, im, method, _lib.PEAKS_LOCAL, thr)
                            assert again.tobytes() == ref.tobytes() and fast.timing()["f32_pieces"] == 3
        print("one-product tier:", sorted(seen))
        assert any(r == 1 and pc == 1 for _, _, r, pc in seen), sorted(seen)            # the tier ran and sufficed
        assert any(r == 1 and pc == 3 for _, _, r, pc in seen), sorted(seen)            # ... overflowed, three products sufficed
        # N_object == 1 by bounds: the screen's wider bounds list more outputs around each template's best - same records
        for method in (5, 3, 1, 0, 2, 4):
            ref = exact.search(lt, im, method, _lib.PEAKS_GLOBAL, 0.0).copy()
            for cap in (1 << 18, 256):
                fast.set_option(_lib.OPT_HIT_CAPACITY, cap)
                fast.set_option(_lib.OPT_HITS_ONLY, 1)
                got = fast.search(lt, im, method, _lib.PEAKS_GLOBAL, 0.0)
                tm = fast.timing()
                assert got.tobytes() == ref.tobytes(), (method, cap, tm["f32_route"], tm["f32_pieces"])
                if cap == 1 << 18:
                    assert tm["f32_route"] == 1 and tm["f32_pieces"] == 1, (method, tm)
    finally:
        fast.close()
        three.close()
        exact.close()


@pytest.mark.gpu
def test_float32_raw_sums_with_thresholds_equal_the_float64_kernels():
    """Raw-sum methods (TM_SQDIFF 0, TM_CCORR 2, TM_CCOEFF 4) on float32 images with a threshold, local extrema: the kernel
    lists every output whose upper bound (score + E, E = eps sqrt(sum (I - mu)^2 sum (T - mean)^2), twice that for TM_SQDIFF)
    passes the threshold and the float64 re-scoring decides - the records must be the float64 kernel's, bit for bit, for
    thresholds on either side of true peak scores; a threshold that lists more than the list holds ends on the float64
    kernel (route 3) with the same records.  Reference: MTM/__init__.py:71-74 (cv2.matchTemplate), :45-52."""
    if not default_routes():
        pytest.skip("asserts the default float32 routes")
    from MTM import _lib
    fast, exact = _lib.Context(0), _lib.Context(0)
    exact.set_option(_lib.OPT_F32_MFMA, 0)
    routes = set()
    try:
        for seed, (lo, hi, noise) in enumerate(((0.0, 1.0, 2e-3), (0.0, 255.0, 0.5), (100.0, 101.0, 1e-2))):
            im = _step_image(seed + 11, lo=lo, hi=hi, noise=noise)
            cx = im.shape[1] // 2
            lt = [np.ascontiguousarray(im[100:132, cx - 20:cx + 20]), np.ascontiguousarray(im[45:77, 30:70]),
                  np.ascontiguousarray(im[20:52, cx + 60:cx + 100])]
            templs = [(t, None) for t in lt]
            for method in (0, 2, 4):
                # a threshold every window passes: peaks of the whole map by the float64 kernel, then thresholds between them
                everything = -3.0e38 if method != 0 else 3.0e38
                ref0 = exact.search(templs, im, method, _lib.PEAKS_LOCAL, everything)
                sc = np.sort(np.unique(ref0["score"].astype(np.float64)))
                if method != 0:
                    sc = sc[::-1]
                picks = [float(s) for s in sc[:: max(1, len(sc) // 5)][:5]]
                thrs = []
                for s in picks:
                    d = max(abs(s), 1.0) * 2e-6
                    thrs += [s - d, s + d, float(np.nextafter(np.float32(s), np.float32(0)))]
                for thr in thrs:
                    ref = exact.search(templs, im, method, _lib.PEAKS_LOCAL, thr)
                    got = fast.search(templs, im, method, _lib.PEAKS_LOCAL, thr)
                    tm = fast.timing()
                    routes.add(tm["f32_route"])
                    assert tm["f32_route"] in (1, 3), tm
                    assert len(got) == len(ref), (seed, method, thr, tm["f32_route"], len(got), len(ref))
                    for f in ("templ_idx", "x", "y", "w", "h"):
                        assert np.array_equal(got[f], ref[f]), (seed, method, thr, tm["f32_route"], f)
                    assert np.array_equal(got["score"].view(np.uint32), ref["score"].view(np.uint32)), (seed, method, thr)
        assert 1 in routes, routes
    finally:
        fast.close()
        exact.close()


@pytest.mark.gpu
def test_sharded_step_in_one_native_call(mtm):
    """mtm_find_matches_image_sharded_nms (round 5): this rank's shard searched, its hits renumbered to the caller's list,
    exchanged, merged and suppressed in one native call.  On a one-GPU box: without a communicator and with a communicator
    of one rank (the RCCL all-gather end to end) a "shard" that is the whole list in a permuted order returns exactly what
    MTM.matchTemplates returns for the list in its own order - maxima and minima methods, finite N_object, masks; a rank
    without units takes part with an empty list.  Reference: MTM/__init__.py:173-177, :289-296, MTM/NMS.py:53-84."""
    from MTM import _lib
    from MTM.distributed import _u8_units
    img, units, _ = synth.make_workload(seed=61, image_hw=(420, 640), n_base=7, templ=32, rotations=2, noisy_per_unit=2)
    perm = [5, 0, 13, 2, 9, 1, 7, 3, 11, 4, 6, 8, 10, 12]           # local template i is list position perm[i]
    assert sorted(perm) == list(range(len(units)))
    sub = [units[g] for g in perm]
    ctx = _lib.Context(0)
    try:
        for with_comm in (False, True):
            if with_comm:
                ctx.comm_init(_lib.comm_unique_id(), 1, 0)
            for method, thr, ov, nobj in ((5, 0.5, 0.25, -1), (5, 0.5, 0.0, 5), (1, 0.3, 0.4, -1), (3, 0.8, 0.25, -1)):
                exp = mtm.matchTemplates(units, img, method=method, score_threshold=thr, maxOverlap=ov,
                                         N_object=float("inf") if nobj < 0 else nobj)
                raw = ctx.search_sharded_nms(_u8_units(sub, img, method), img, method, thr, ov, nobj, perm)
                got = mtm._to_hit_list(raw, units, 0, 0)
                assert len(got) == len(exp) > 0, (with_comm, method, len(got), len(exp))
                assert [(h[0], h[1]) for h in got] == [(h[0], h[1]) for h in exp], (with_comm, method)
                assert all(np.float32(a[2]) == np.float32(b[2]) for a, b in zip(got, exp)), (with_comm, method)
        # masked units through the same call
        mimg, munits, _ = synth.make_workload(seed=62, image_hw=(400, 520), n_base=2, templ=32, scales=(24, 40), masked=True)
        order = list(range(len(munits)))[::-1]
        exp = mtm.matchTemplates(munits, mimg, method=3, score_threshold=0.9, maxOverlap=0.25)
        raw = ctx.search_sharded_nms(_u8_units([munits[g] for g in order], mimg, 3), mimg, 3, 0.9, 0.25, -1, order)
        got = mtm._to_hit_list(raw, munits, 0, 0)
        assert [(h[0], h[1]) for h in got] == [(h[0], h[1]) for h in exp] and len(got) > 0
        # a rank without units: nothing to search, an empty contribution to the exchange
        assert len(ctx.search_sharded_nms([], None, 5, 0.5, 0.25, -1, [])) == 0
    finally:
        ctx.close()


# ---- round 6: float32 templates WITH masks on the bf16 matrix cores (mtm_maskf32.hip.h) -----------------------------------
def _mf32_disc(h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    return ((((yy - h / 2 + 0.5) / (h / 2)) ** 2 + ((xx - w / 2 + 0.5) / (w / 2)) ** 2) <= 1.0).astype(np.float32)


def _mf32_case(rng, H, W, h, w, n, kind, scale):
    img = (rng.random((H, W)).astype(np.float32) * scale).astype(np.float32)
    if kind == "step":                      # low-contrast windows beside a brightness step: where the bound has to be wide
        img[:, W // 2:] += np.float32(scale)
        img[:, :W // 2] *= np.float32(0.01)
    units = []
    for i in range(n):
        y, x = int(rng.integers(0, H - h)), int(rng.integers(0, W - w))
        t = img[y:y + h, x:x + w].copy()
        if i % 3 == 1:
            t = (t + rng.normal(0, 0.05 * scale, t.shape)).astype(np.float32)
        m = _mf32_disc(h, w) if i % 2 == 0 else rng.random((h, w)).astype(np.float32)      # binary and weight masks
        units.append((t, m))
    return img, units


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [(150, 333, 24, 40, 5, "noise", 1.0), (150, 333, 20, 70, 20, "noise", 255.0),
                                  (300, 420, 33, 17, 7, "step", 1.0), (260, 610, 64, 64, 37, "noise", 65535.0),
                                  (200, 300, 9, 130, 3, "step", 100.0)], ids=lambda g: "%dx%d_%dx%d_n%d_%s" % g[:6])
def test_float32_masked_templates_equal_the_float64_kernels(geom):
    """Masked float32 templates (float weights or binary masks; everything that is not uint8 / uint8 reaches cv2 as float32:
    MTM/__init__.py:71-88, :212-219) with a threshold: two raw launches of the bf16 kernel + a rigorous bound screen the
    outputs, what could pass is re-scored with the float64 kernel's own chains (f32_route 4).  The records must be the
    float64 kernel's, bit for bit, in order - for both methods masks are allowed with, thresholds far from and within one
    float32 ulp of true peak scores, lists that overflow the screen's capacity (the float64 kernel takes over)."""
    from MTM import _lib
    rng = np.random.default_rng(11)
    H, W, h, w, n, kind, scale = geom
    img, units = _mf32_case(rng, H, W, h, w, n, kind, scale)
    fast, exact = _lib.Context(0), _lib.Context(0)
    exact.set_option(_lib.OPT_F32_MFMA, 0)
    try:
        for method in (3, 0):
            if method == 3:
                thrs = [0.9, 0.5, 0.999]
            else:
                base = float((units[0][0].astype(np.float64) * units[0][1]).var() * h * w)
                thrs = [1e-3 * base + 1e-6, 0.3 * base]
            ref = exact.search(units, img, method, _lib.PEAKS_LOCAL, thrs[0]).copy()
            # thresholds a float32 ulp on either side of true peak scores
            for s in sorted(set(float(v) for v in ref["score"]))[:3]:
                thrs += [float(np.nextafter(np.float32(s), np.float32(-np.inf))), float(np.nextafter(np.float32(s), np.float32(np.inf)))]
            routes = set()
            for thr in thrs:
                for pattern in (0xFF, 0x7F):
                    fast.debug_poison(pattern, 7)
                    a = fast.search(units, img, method, _lib.PEAKS_LOCAL, thr).copy()
                    routes.add(fast.timing()["f32_route"])
                    b = exact.search(units, img, method, _lib.PEAKS_LOCAL, thr).copy()
                    assert exact.timing()["f32_route"] == 0
                    assert a.tobytes() == b.tobytes(), (geom, method, thr, len(a), len(b))
            if default_routes():
                assert 4 in routes, (geom, method, routes)         # the screen really ran (an overflowing list may add route 0 / 3)
        # N_object == 1 (cv2.minMaxLoc): the same screen with the templates' own best lower bound as the threshold - exact
        # extremum, first occurrence among exact ties (the image holds exact copies of some templates twice)
        img2 = img.copy()
        t0 = units[0][0]
        if W - w > 2 * w and H - h > 1:
            img2[1:1 + h, 1:1 + w] = t0
            img2[1:1 + h, W - w - 1:W - 1] = t0
        for method in (3, 0):
            for pattern in (0xFF, 0x7F):
                fast.debug_poison(pattern, 7)
                a = fast.search(units, img2, method, _lib.PEAKS_GLOBAL, 0.0).copy()
                route = fast.timing()["f32_route"]
                b = exact.search(units, img2, method, _lib.PEAKS_GLOBAL, 0.0).copy()
                assert a.tobytes() == b.tobytes() and len(a) == n, (geom, method, len(a), len(b))
                if default_routes():
                    assert route == 4, (geom, method, route)
        # the route that must NOT take the screen: one published score map - and it agrees with the float64 kernel
        for c_ in (fast, exact):
            c_.set_image(img)
            c_.set_templates(units, 3)
        shape = (H - h + 1, W - w + 1)
        assert np.array_equal(fast.score_map(0, shape), exact.score_map(0, shape), equal_nan=True)
        # the maps of a screened call hold placeholders: not published
        fast.search(units, img, 3, _lib.PEAKS_LOCAL, 0.9)
        if fast.timing()["f32_route"] == 4:
            with pytest.raises(_lib.MtmError):
                fast.last_score_map(0, shape)
    finally:
        fast.close()
        exact.close()


@pytest.mark.gpu
def test_float32_masked_through_the_module_api_against_the_oracle(mtm):
    """The same route through MTM.matchTemplates / findMatches - uint16 pixels with a uint16 mask (what the reference casts to
    float32, mask included: MTM/__init__.py:71-74, :81-88), a float32 image with a float32 weight mask, and a call that mixes
    masked and unmasked templates - against the oracle: same boxes, scores to 1e-5."""
    rng = np.random.default_rng(23)
    img16 = rng.integers(0, 65536, (180, 260), dtype=np.uint16)
    m16 = (_mf32_disc(30, 44) * 65535).astype(np.uint16)
    lt16 = [("a%d" % i, img16[y:y + 30, x:x + 44].copy(), m16) for i, (y, x) in enumerate([(10, 20), (100, 150), (60, 200)])]
    got = mtm.findMatches(lt16, img16, method=3, score_threshold=0.9)
    exp = O.find_matches([(n, t.astype(np.float32), m.astype(np.float32)) for n, t, m in lt16], img16.astype(np.float32),
                         method=3, score_threshold=0.9)
    assert len(got) == len(exp) >= 3, (len(got), len(exp))
    assert_hits_equal(canon(got), canon(exp), tol=1e-5)
    imgf = rng.random((200, 310)).astype(np.float32)
    wm = rng.random((25, 37)).astype(np.float32)
    ltf = [("w%d" % i, imgf[y:y + 25, x:x + 37].copy(), wm) for i, (y, x) in enumerate([(5, 9), (120, 200), (77, 33), (150, 260)])]
    ltf.append(("plain", imgf[40:40 + 25, 100:100 + 37].copy()))             # no mask: the call mixes both kinds
    got = mtm.matchTemplates(ltf, imgf, method=3, score_threshold=0.95, maxOverlap=0.2)
    exp = O.match_templates(ltf, imgf, method=3, score_threshold=0.95, maxOverlap=0.2)
    assert len(got) == len(exp) >= 5
    # (five exact copies: every score is 1 to within the tolerance, so the order by score is not defined - MTM_F32_MFMA=2
    # publishes the bf16 kernel's 0.99999994 for the unmasked one; compared by label and box)
    assert_hits_equal(canon(got), canon(exp), tol=1e-5, ordered=False)
    # TM_SQDIFF (minima below the threshold, raw units): noisy copies, so that the scores are sums and not cancellation noise
    lts = [(n, (t + rng.normal(0, 0.05, t.shape)).astype(np.float32), m) for n, t, m in ltf[:4]]
    got = mtm.findMatches(lts, imgf, method=0, score_threshold=3.0)
    exp = O.find_matches(lts, imgf, method=0, score_threshold=3.0)
    assert len(got) == len(exp) >= 4, (len(got), len(exp))
    assert_hits_equal(canon(got), canon(exp), tol=1e-5)
