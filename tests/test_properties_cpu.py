"""Property tests (hypothesis) of the host-side pieces that run without a GPU: the C++ NMS behind the
C ABI against the oracle's restatement of cv2.dnn.NMSBoxes, the unit sharding, the INTER_AREA downscale,
and algebraic properties of the oracle itself (the checker has to be right before it checks)."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import mtm_oracle as O        # conftest.py puts oracle/ and the package directory on sys.path

COMMON = dict(deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])


@pytest.fixture(scope="module")
def MTM():
    import build as mtm_build
    mtm_build.build()
    import MTM
    return MTM


boxes = st.tuples(st.integers(-50, 400), st.integers(-50, 400), st.integers(0, 90), st.integers(0, 90))
scores = st.floats(0, 1, width=32, allow_nan=False)


@settings(max_examples=150, **COMMON)
@given(st.lists(st.tuples(boxes, scores), min_size=0, max_size=120), st.floats(0, 1), st.floats(0, 1),
       st.booleans(), st.sampled_from([float("inf"), 0, 1, 2, 7]))
def test_nms_equals_oracle(MTM, items, thr, overlap, ascending, n_object):
    """mtm_nms (grid-accelerated, float32-faithful) makes the decisions of the plain greedy restatement:
    degenerate boxes, duplicates, ties, negative coordinates, every N_object branch of MTM/NMS.py."""
    hits = [("t%d" % (i % 3), b, np.float32(s)) for i, (b, s) in enumerate(items)]
    got = MTM.NMS(hits, thr, ascending, n_object, overlap)
    exp = O.NMS(hits, thr, ascending, n_object, overlap)
    assert [(h[0], tuple(h[1]), float(h[2])) for h in got] == [(h[0], tuple(h[1]), float(h[2])) for h in exp]


def test_nms_trivial_cases_follow_the_reference(MTM):
    """MTM/NMS.py:53-55,61-69: a list of at most one hit comes back as a copy OF ITS TYPE (listHit[:]); N_object == 1
    picks with python's min() / max() - first of equal scores, and a NaN only when it comes first."""
    one = (("a", (1, 2, 3, 4), np.float32(0.1)),)
    got = MTM.NMS(one, 0.5)
    assert type(got) is tuple and got == one                      # unthresholded copy, same sequence type
    assert MTM.NMS([], 0.5) == [] and MTM.NMS((), 0.5) == ()
    nan = float("nan")
    hits = [("a", (0, 0, 5, 5), 0.3), ("b", (9, 9, 5, 5), nan), ("c", (20, 20, 5, 5), 0.8), ("d", (40, 40, 5, 5), 0.8)]
    assert MTM.NMS(hits, 0.5, N_object=1) == [hits[2]] == O.NMS(hits, 0.5, N_object=1)
    assert MTM.NMS(hits, 0.5, sortAscending=True, N_object=1) == [hits[0]]
    first_nan = [hits[1]] + hits[2:]
    got = MTM.NMS(first_nan, 0.5, N_object=1)
    assert len(got) == 1 and got[0][0] == "b"                     # max() never replaces a leading NaN


@settings(max_examples=40, **COMMON)
@given(st.integers(0, 2 ** 31), st.integers(64, 3000))
def test_nms_dense_grid_path(MTM, seed, n):
    """Thousands of boxes (the multi-GPU gather): the O(n) grid path against the O(n^2) oracle."""
    rng = np.random.default_rng(seed)
    side = int(rng.integers(8, 80))
    hits = [("t", (int(x), int(y), side + int(dw), side + int(dh)), np.float32(s))
            for x, y, dw, dh, s in zip(rng.integers(0, 1500, n), rng.integers(0, 900, n), rng.integers(0, 9, n),
                                       rng.integers(0, 9, n), rng.random(n))]
    ov = float(rng.choice([0.0, 0.1, 0.25, 0.5, 0.9]))
    got = MTM.NMS(hits, 0.2, False, float("inf"), ov)
    exp = O.NMS(hits, 0.2, False, float("inf"), ov)
    assert [(tuple(h[1]), float(h[2])) for h in got] == [(tuple(h[1]), float(h[2])) for h in exp]
    # many equal scores (ties keep the input order: the radix sort of long lists is stable), both sort directions,
    # a signed zero among them
    tied = [(h[0], h[1], np.float32(round(float(h[2]) * 20) / 20)) for h in hits]
    tied[0] = (tied[0][0], tied[0][1], np.float32(-0.0))
    for ascending, thr in ((False, 0.2), (True, 0.7)):
        got = MTM.NMS(tied, thr, ascending, float("inf"), ov)
        exp = O.NMS(tied, thr, ascending, float("inf"), ov)
        assert [(tuple(h[1]), float(h[2])) for h in got] == [(tuple(h[1]), float(h[2])) for h in exp]


@settings(max_examples=200, **COMMON)
@given(st.lists(st.floats(0.1, 1e6), min_size=0, max_size=300), st.integers(1, 16))
def test_shard_units_partition(MTM, costs, world):
    from MTM.distributed import shard_units
    parts = shard_units(costs, world)
    assert len(parts) == world
    flat = sorted(i for p in parts for i in p)
    assert flat == list(range(len(costs)))                          # every unit exactly once
    loads = [sum(costs[i] for i in p) for p in parts]
    if costs:
        # LPT guarantee: makespan <= average + largest unit
        assert max(loads) <= sum(costs) / world + max(costs) + 1e-6 * sum(costs)
    assert all(list(p) == sorted(p) for p in parts)                 # deterministic order inside a rank


@settings(max_examples=80, **COMMON)
@given(st.integers(0, 2 ** 31), st.integers(1, 6), st.sampled_from(["uint8", "float32"]), st.sampled_from([1, 3]))
def test_downscale_equals_oracle(MTM, seed, factor, dtype, chans):
    rng = np.random.default_rng(seed)
    shape = (int(rng.integers(factor, 70)), int(rng.integers(factor, 90))) + ((chans,) if chans > 1 else ())
    img = (rng.random(shape) * 255).astype(dtype)
    a, b = MTM.augment.downscale(img, factor), O.downscale_area(img, factor)
    assert a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b)
    if dtype == "uint8" and factor > 1:                              # a mean never leaves the value range
        blocks = img[:a.shape[0] * factor, :a.shape[1] * factor]
        assert a.min() >= blocks.min() and a.max() <= blocks.max()


small_images = st.integers(0, 2 ** 31)


@settings(max_examples=40, **COMMON)
@given(small_images, st.sampled_from([1, 3, 5]))
def test_oracle_invariances(seed, method):
    """Properties the arithmetic of cv2.matchTemplate has by construction; the oracle must have them too."""
    rng = np.random.default_rng(seed)
    H, W = int(rng.integers(12, 40)), int(rng.integers(12, 40))
    h, w = int(rng.integers(2, 9)), int(rng.integers(2, 9))
    img = rng.integers(0, 256, (H, W)).astype(np.uint8)
    y0, x0 = int(rng.integers(0, H - h + 1)), int(rng.integers(0, W - w + 1))
    t = img[y0:y0 + h, x0:x0 + w].copy()
    m = O.match_template(img, t, method)
    assert m.shape == (H - h + 1, W - w + 1) and m.dtype == np.float32
    if t.std() > 0:
        best = 0.0 if method == 1 else 1.0
        assert abs(float(m[y0, x0]) - best) < 1e-6                  # an exact copy is a perfect match
    assert np.all(m <= 1.0 + 1e-6) and np.all(m >= (0.0 if method != 5 else -1.0) - 1e-6)
    # rot90 equivariance: rotating image and template together rotates the map
    m90 = O.match_template(np.ascontiguousarray(np.rot90(img)), np.ascontiguousarray(np.rot90(t)), method)
    assert np.allclose(np.rot90(m), m90, atol=2e-6)
    if method == 5:
        # zero-mean normalised correlation ignores gain and offset of the template (float32 path)
        tf = t.astype(np.float32)
        m1 = O.match_template(img.astype(np.float32), tf, 5)
        m2 = O.match_template(img.astype(np.float32), (0.5 * tf + 7).astype(np.float32), 5)
        if t.std() > 1:
            assert np.allclose(m1, m2, atol=1e-4)


@settings(max_examples=40, **COMMON)
@given(small_images)
def test_oracle_peaks_definition(seed):
    """peak_local_max as the reference uses it: a returned pixel is above the threshold and equals its 3x3
    maximum (zero padded); every such pixel is returned unless the map is constant."""
    rng = np.random.default_rng(seed)
    a = np.round(rng.random((int(rng.integers(3, 14)), int(rng.integers(3, 14)))), 1).astype(np.float32)   # plateaus likely
    thr = float(rng.choice([-0.5, 0.0, 0.3, 0.7]))
    got = {tuple(p) for p in O.find_local_max(a, thr)}
    pad = np.pad(a, 1, constant_values=0)
    exp = set()
    for y in range(a.shape[0]):
        for x in range(a.shape[1]):
            if a[y, x] > thr and a[y, x] == pad[y:y + 3, x:x + 3].max():
                exp.add((y, x))
    all_local = all(a[y, x] == pad[y:y + 3, x:x + 3].max() for y in range(a.shape[0]) for x in range(a.shape[1]))
    assert got == (set() if all_local else exp)


@settings(max_examples=120, **COMMON)
@given(st.lists(st.tuples(st.integers(0, 5), boxes, scores), min_size=0, max_size=80), st.floats(0, 1), st.floats(0, 1),
       st.booleans(), st.sampled_from([float("inf"), 0, 1, 3]))
def test_raw_array_pipeline_equals_list_pipeline(MTM, items, thr, overlap, ascending, n_object):
    """The host layer keeps hits in a structured array until the very end (_nms_raw + _to_hit_list); that must
    be MTM.NMS applied to the equivalent list of tuples (the reference's data flow, MTM/__init__.py:296)."""
    from MTM import _lib, _nms_raw, _to_hit_list
    lt = [("label%d" % i, None) for i in range(6)]
    raw = np.zeros(len(items), dtype=_lib.HIT_DTYPE)
    for k, (t, (x, y, w, h), s) in enumerate(items):
        raw[k] = (t, x, y, w, h, np.float32(s))
    as_list = _to_hit_list(raw, lt, 0, 0)
    assert [h[0] for h in as_list] == ["label%d" % t for t, _, _ in items]
    got = _to_hit_list(_nms_raw(raw, thr, ascending, n_object, overlap), lt, 0, 0)
    exp = MTM.NMS(as_list, thr, ascending, n_object, overlap)
    assert got == exp
    # offsets of a searchBox are added to x and y only
    moved = _to_hit_list(raw, lt, 7, 11)
    assert all(m[1] == (a[1][0] + 7, a[1][1] + 11, a[1][2], a[1][3]) for m, a in zip(moved, as_list))


@pytest.mark.parametrize("ascending", [False, True])
@pytest.mark.parametrize("scores", [[0.2, float("nan"), 0.9, 0.9, float("nan"), 0.1],
                                    [float("nan"), 0.5, 0.9],
                                    [0.4, float("nan"), float("nan")],
                                    [float("nan"), float("nan")]])
def test_single_object_pick_with_nan_scores(MTM, scores, ascending):
    """N_object == 1 on raw arrays: python max() / min() (reference MTM/NMS.py:61-69) never select a NaN score unless it
    comes first (masked TM_CCORR_NORMED yields 0/0 = NaN, as OpenCV does); the array fast path must agree with the
    reference's list semantics, which MTM.NMS.NMS reproduces literally."""
    import math
    from MTM import _lib, _nms_raw, _to_hit_list
    lt = [("t%d" % i, None) for i in range(3)]
    raw = np.zeros(len(scores), dtype=_lib.HIT_DTYPE)
    for k, s in enumerate(scores):
        raw[k] = (k % 3, 10 * k, 5 * k, 8, 8, np.float32(s))
    got = _to_hit_list(_nms_raw(raw, 0.5, ascending, 1, 0.3), lt, 0, 0)
    as_list = _to_hit_list(raw, lt, 0, 0)
    exp = [(min if ascending else max)(as_list, key=lambda h: h[2])]          # the reference's own expression
    assert len(got) == 1 and got[0][:2] == exp[0][:2]
    assert (math.isnan(got[0][2]) and math.isnan(exp[0][2])) or got[0][2] == exp[0][2]
    lib = MTM.NMS(as_list, 0.5, ascending, 1, 0.3)
    assert lib[0][:2] == exp[0][:2]
