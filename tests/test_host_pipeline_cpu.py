"""The Python host layer (dtype policy, unit grouping, searchBox offsets, hit construction, NMS
hand-off) exercised on CPU with a stand-in context whose kernels are the oracle.  This checks the
host logic only; the HIP kernels are checked by tests/test_gpu_parity.py on the GPU box."""
import threading

import numpy as np
import pytest

import mtm_oracle as O
from helpers import assert_hits_equal, canon, coin_templates, load_coins, load_golden

REF = load_golden()["reference_run"]


class OracleContext:
    """Implements the _lib.Context surface the host layer uses, on the oracle."""

    def __init__(self, hit_dtype):
        self.lock = threading.RLock()
        self.hit_dtype = hit_dtype

    @staticmethod
    def _as_cv2_sees_it(a):
        # MTM_U16: the library takes uint16 pixels as they are; the reference casts them to float32
        # (exactly) before cv2.matchTemplate (MTM/__init__.py:71-74)
        return a.astype(np.float32) if a is not None and a.dtype == np.uint16 else a

    def set_image(self, image, downscale=1):
        self.image = self._as_cv2_sees_it(O.downscale_area(image, downscale))       # mtm_set_image_downscaled

    def set_templates(self, templates, method):
        self.templates, self.method = [(self._as_cv2_sees_it(t), self._as_cv2_sees_it(m)) for t, m in templates], method

    def score_map(self, idx, shape):
        t, m = self.templates[idx]
        out = O.match_template(self.image, t, self.method, mask=m)
        assert out.shape == tuple(shape)
        return out

    def search(self, templates, image, method, mode, thr):          # the engine interface (Context / Group)
        self.set_templates(templates, method)
        return self.find_matches_image(image, mode, thr)

    def find_matches_image(self, image, mode, thr):                 # mtm_find_matches_image
        self.set_image(image)
        return self._find(mode, thr)

    def find_matches(self, mode, thr, next_image=None):
        try:
            return self._find(mode, thr)
        finally:
            if next_image is not None:      # mtm_find_matches_next: the next image becomes current
                self.image = next_image

    def _find(self, mode, thr):
        rows = []
        for i, (t, m) in enumerate(self.templates):
            cmap = O.match_template(self.image, t, self.method, mask=m)
            if mode == 1:
                _, _, mn, mx = O.min_max_loc(cmap)
                peaks = [mn[::-1]] if self.method in (0, 1) else [mx[::-1]]
            elif self.method in (0, 1):
                peaks = O.find_local_min(cmap, thr)
            else:
                peaks = O.find_local_max(cmap, thr)
            rows += [(i, int(p[1]), int(p[0]), t.shape[1], t.shape[0], cmap[tuple(p)]) for p in peaks]
        return np.array(rows, dtype=self.hit_dtype) if rows else np.zeros(0, dtype=self.hit_dtype)


@pytest.fixture()
def mtm(monkeypatch):
    import build as mtm_build
    mtm_build.build()
    import MTM
    monkeypatch.setattr(MTM._lib, "_default_ctx", OracleContext(MTM._lib.HIT_DTYPE))
    return MTM


def test_pipeline_against_reference_runs(mtm):
    coins = load_coins()
    small, big = coin_templates(coins)
    lt = [("small", small), ("big", big)]
    assert_hits_equal(mtm.matchTemplates([("small", small)], coins, score_threshold=0.5, method=5, maxOverlap=0), REF["G1"], tol=1e-6)
    assert_hits_equal(canon(mtm.matchTemplates(lt, coins, score_threshold=0.3, method=5, maxOverlap=0)),
                      canon([(h[0], tuple(h[1]), h[2]) for h in REF["testpy"]]), tol=1e-6)
    assert_hits_equal(canon(mtm.matchTemplates(lt, coins, method=1, score_threshold=0.2, maxOverlap=0)),
                      canon([(h[0], tuple(h[1]), h[2]) for h in REF["sqdiff_normed"]]), tol=1e-6)
    assert_hits_equal(mtm.matchTemplates([("small", small)], coins, score_threshold=0.5, maxOverlap=0, searchBox=(10, 20, 300, 200)),
                      REF["searchbox"], tol=1e-6)
    assert_hits_equal(mtm.matchTemplates([("small", small)], coins, method=5, N_object=1), REF["nobj1"], tol=1e-6)
    assert mtm.matchTemplates(lt, coins, score_threshold=0.3, method=5, N_object=0) == []
    img16 = coins.astype(np.uint16) * 257
    assert_hits_equal(mtm.matchTemplates([("small", img16[37:75, 80:121])], img16, score_threshold=0.5, method=5, maxOverlap=0),
                      REF["uint16"], tol=1e-6)
    assert_hits_equal(canon(mtm.findMatches(lt, coins, score_threshold=0.3)), REF["find_pre_nms"], tol=1e-6)
    with pytest.raises(ValueError, match="TM_SQDIFF is not supported"):
        mtm.matchTemplates([("small", small)], coins, method=0)


def test_fused_search_and_nms_entry_is_used_and_equivalent(mtm, monkeypatch):
    """matchTemplates hands the non-maxima suppression to the engine where the engine offers `search_nms` (one native call:
    mtm_find_matches_image_nms) - for every N_object but 1, never for TM_SQDIFF, never for the general (non 8-bit) path -
    and returns what the two-step route returns.  The stand-in's search_nms is search + the library's host NMS (mtm_nms)."""
    from MTM import _lib
    calls = []

    class FusedContext(OracleContext):
        def search_nms(self, templates, image, method, thr, max_overlap, n_object=-1):
            calls.append((method, n_object))
            raw = self.search(templates, image, method, 0, thr)
            if len(raw) <= 1:                                   # MTM/NMS.py:53-55
                return raw
            idx = _lib.nms_hits(raw, thr, max_overlap, ascending=(method == 1))
            return raw[idx] if n_object < 0 else raw[idx][:n_object]

    coins = load_coins()
    small, big = coin_templates(coins)
    lt = [("small", small), ("big", big)]
    cases = [dict(score_threshold=0.3, method=5, maxOverlap=0.25), dict(score_threshold=0.3, method=5, maxOverlap=0.0, N_object=3),
             dict(method=1, score_threshold=0.2, maxOverlap=0.1), dict(score_threshold=0.5, maxOverlap=0, searchBox=(10, 20, 300, 200)),
             dict(score_threshold=0.3, method=3, maxOverlap=0.3, N_object=0), dict(score_threshold=0.99, method=5)]
    two_step = [mtm.matchTemplates(lt, coins, **kw) for kw in cases]
    best = mtm.matchTemplates(lt, coins, method=5, N_object=1)
    monkeypatch.setattr(mtm._lib, "_default_ctx", FusedContext(mtm._lib.HIT_DTYPE))
    assert [mtm.matchTemplates(lt, coins, **kw) for kw in cases] == two_step
    assert calls == [(5, -1), (5, 3), (1, -1), (5, -1), (3, 0), (5, -1)]
    assert mtm.matchTemplates(lt, coins, method=5, N_object=1) == best and len(calls) == 6           # the global extremum: search()
    with pytest.raises(ValueError, match="TM_SQDIFF is not supported"):
        mtm.matchTemplates(lt, coins, method=0)
    assert len(calls) == 6
    img_f = coins.astype(np.float32)                                                                # general path: per-dtype groups, two steps
    assert mtm.matchTemplates([("s", small.astype(np.float32))], img_f, score_threshold=0.5) and len(calls) == 6


def test_mixed_dtype_units_are_grouped(mtm):
    """The pixel policy is per template (MTM/__init__.py:71): a float32 template next to a uint8
    one is matched in float32 against the float32 image, the uint8 one stays 8-bit."""
    coins = load_coins()
    small, big = coin_templates(coins)
    lt = [("f", small.astype(np.float32)), ("u", big), ("f2", big.astype(np.uint16))]
    got = mtm.findMatches(lt, coins, score_threshold=0.5)
    exp = O.find_matches(lt, coins, score_threshold=0.5)
    assert_hits_equal(got, [[h[0], list(h[1]), float(h[2])] for h in exp], tol=1e-6)
    assert [h[0] for h in got] == sorted([h[0] for h in got], key=["f", "u", "f2"].index)


def test_score_map_and_types(mtm):
    coins = load_coins()
    small, _ = coin_templates(coins)
    m = mtm.computeScoreMap(small, coins)
    assert m.dtype == np.float32 and m.shape == (266, 344)
    hits = mtm.matchTemplates([("small", small)], coins, maxOverlap=0)
    assert isinstance(hits, list) and isinstance(hits[0], tuple) and isinstance(hits[0][2], np.float32)
    assert all(type(v) is int for v in hits[0][1])


def test_template_matcher_equals_match_templates(mtm):
    """Resident-template API (stream of images) == one matchTemplates call per image."""
    coins = load_coins()
    small, big = coin_templates(coins)
    lt = [("small", small), ("big", big)]
    matcher = mtm.TemplateMatcher(lt, score_threshold=0.3, maxOverlap=0.25, context=mtm._lib.default_context())
    for img in (coins, np.ascontiguousarray(coins[::-1]), coins[20:280, 10:380]):
        assert matcher.match(img) == mtm.matchTemplates(lt, img, score_threshold=0.3, maxOverlap=0.25)
    assert matcher.match(coins, searchBox=(10, 20, 300, 200)) == \
        mtm.matchTemplates(lt, coins, score_threshold=0.3, maxOverlap=0.25, searchBox=(10, 20, 300, 200))
    with pytest.raises(ValueError, match="pixel type"):
        matcher.match(coins.astype(np.float32))
    # stream form: same results, in order, any mix of sizes; empty iterable -> nothing
    imgs = [coins, np.ascontiguousarray(coins[::-1]), coins[20:280, 10:380], coins]
    assert list(matcher.match_stream(imgs)) == [matcher.match(i) for i in imgs]
    assert list(matcher.match_stream(iter(imgs[:1]))) == [matcher.match(coins)]
    assert list(matcher.match_stream([])) == []


def test_augment_helpers_and_downscaled_matching(mtm):
    """MTM.augment: the tutorials' rot90 / flip augmentation, integer-factor INTER_AREA downscale (checked
    against the oracle's restatement) and the downscale -> match -> upscale recipe in one call."""
    A = mtm.augment
    rng = np.random.default_rng(5)
    for shape in [(37, 53), (64, 64, 3), (100, 41)]:
        for dt in (np.uint8, np.float32):
            img = (rng.random(shape) * 255).astype(dt)
            for f in (1, 2, 3, 4, 5, 8):
                assert np.array_equal(A.downscale(img, f), O.downscale_area(img, f)), (shape, dt, f)
    with pytest.raises(ValueError, match="64-bit"):
        A.downscale(np.zeros((8, 8)), 2)
    t = np.arange(12, dtype=np.uint8).reshape(3, 4)
    rots = A.rotations([("t", t)], angles=(0, 90, 180))
    assert [r[0] for r in rots] == ["t_0", "t_90", "t_180"]
    assert np.array_equal(rots[1][1], np.rot90(t)) and rots[1][1].flags.c_contiguous
    fl = A.flips([("t", t, t)])
    assert [r[0] for r in fl] == ["t", "t_lr", "t_ud"] and np.array_equal(fl[1][2], np.fliplr(t))
    assert [r[0] for r in A.scales([("t", np.zeros((8, 8), np.uint8))], (1, 2))] == ["t_d1", "t_d2"]
    assert A.upscale_hits([("a", (1, 2, 3, 4), 0.5)], 4) == [("a", (4, 8, 12, 16), 0.5)]
    # one call == downscale both on the host, match, upscale
    coins = load_coins()
    small, big = coin_templates(coins)
    lt = [("small", small), ("big", big)]
    for f in (2, 3):
        got = A.matchTemplatesDownscaled(lt, coins, f, score_threshold=0.4, maxOverlap=0.3)
        exp = A.upscale_hits(mtm.matchTemplates([(n, A.downscale(t, f)) for n, t in lt], A.downscale(coins, f),
                                                score_threshold=0.4, maxOverlap=0.3), f)
        assert got == exp and len(got) > 3
    with pytest.raises(ValueError, match="larger than image"):
        A.matchTemplatesDownscaled([("all", coins)], coins[:200], 2)


def test_augmentation_spec_and_host_expansion(mtm):
    """MTM.augment.variants / expand (the host reference of mtm_set_templates_augmented): order, labels and pixels of
    the copies; resize_area is the exact area average (identity, integer factors = block means rounded half up)."""
    A = mtm.augment
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (12, 20), dtype=np.uint8)
    m = (a > 100).astype(np.uint8) * 255
    spec = A.variants(angles=(0, 90, 180, 270))
    assert [s for s, _ in spec] == ["_0", "_90", "_180", "_270"]
    ex = A.expand([("t", a, m)], spec)
    assert [e[0] for e in ex] == ["t_0", "t_90", "t_180", "t_270"]
    for k, e in enumerate(ex):
        assert np.array_equal(e[1], np.rot90(a, k)) and np.array_equal(e[2], np.rot90(m, k)) and e[1].flags.c_contiguous
    # the tutorial's own augmentation (rotations helper) is the same list
    assert all(np.array_equal(x[1], y[1]) and x[0] == y[0] for x, y in zip(ex, A.rotations([("t", a, m)])))
    spec = A.variants(flip_lr=True, flip_ud=True, sizes=((6, 10), 8), angles=(0, 90))
    assert [s for s, _ in spec] == ["_s6x10_0", "_s6x10_90", "_s6x10_lr_0", "_s6x10_lr_90", "_s6x10_ud_0", "_s6x10_ud_90",
                                    "_s8x8_0", "_s8x8_90", "_s8x8_lr_0", "_s8x8_lr_90", "_s8x8_ud_0", "_s8x8_ud_90"]
    ex = A.expand([("t", a)], spec)
    small = A.resize_area(a, 6, 10)
    assert np.array_equal(ex[0][1], small) and np.array_equal(ex[3][1], np.rot90(np.fliplr(small)))
    assert np.array_equal(ex[4][1], np.flipud(small)) and ex[7][1].shape == (8, 8)
    # resize_area: identity; integer factors = block means rounded half up; constant images stay constant; RGB per channel
    assert np.array_equal(A.resize_area(a, 12, 20), a)
    blocks = a.reshape(6, 2, 10, 2).astype(np.int64).sum(axis=(1, 3))
    assert np.array_equal(A.resize_area(a, 6, 10), ((2 * blocks + 4) // 8).astype(np.uint8))
    assert np.array_equal(A.resize_area(np.full((7, 9), 201, np.uint8), 13, 4), np.full((13, 4), 201, np.uint8))
    rgb = rng.integers(0, 256, (10, 14, 3), dtype=np.uint8)
    up = A.resize_area(rgb, 15, 21)
    assert up.shape == (15, 21, 3) and all(np.array_equal(up[..., c], A.resize_area(np.ascontiguousarray(rgb[..., c]), 15, 21)) for c in range(3))
    # integer downscale factors go through MTM.augment.downscale (OpenCV's INTER_AREA rounding)
    ex = A.expand([("t", a)], A.variants(factors=(1, 2)))
    assert [e[0] for e in ex] == ["t_d1", "t_d2"] and np.array_equal(ex[0][1], a) and np.array_equal(ex[1][1], A.downscale(a, 2))
    with pytest.raises(ValueError):
        A.variants(angles=(45,))
    with pytest.raises(ValueError):
        A.variants(sizes=(8,), factors=(2,))
    # matchTemplatesAugmented == matchTemplates(expand(...)) through the (oracle-backed) host pipeline for non-uint8 input
    coins = load_coins().astype(np.float32)
    small_t, _ = coin_templates(coins)
    spec = A.variants(angles=(0, 180))
    assert A.matchTemplatesAugmented([("s", small_t)], spec, coins, score_threshold=0.6) == \
        mtm.matchTemplates(A.expand([("s", small_t)], spec), coins, score_threshold=0.6)
