"""Host-layer behaviour that needs no GPU: the reference's argument validation, messages and
their order (MTM/__init__.py:129-167, :286-287, :67-68), checked against the messages captured
from the unmodified reference (tests/golden/golden.json "errors")."""
import os

import numpy as np
import pytest

from helpers import coin_templates, load_coins, load_golden

ERR = load_golden()["errors"]


@pytest.fixture(scope="module")
def mtm():
    import build as mtm_build
    mtm_build.build()
    import MTM
    return MTM


def _raises(name, fn):
    exp_type, exp_msg = ERR[name]
    with pytest.raises(Exception) as ei:
        fn()
    assert type(ei.value).__name__ == exp_type and str(ei.value) == exp_msg


def test_error_messages_match_reference(mtm):
    image = load_coins()
    small, big = coin_templates(image)
    lt = [("small", small), ("big", big)]
    _raises("searchbox_small", lambda: mtm.matchTemplates(lt, image, searchBox=(0, 0, 20, 20)))
    _raises("too_large", lambda: mtm.matchTemplates([("tooLarge", np.pad(image, 1))], image))
    _raises("nobj_float", lambda: mtm.matchTemplates([("small", small)], image, N_object=2.5))
    _raises("nobj_npint", lambda: mtm.matchTemplates([("small", small)], image, N_object=np.int64(2)))
    _raises("overlap_range", lambda: mtm.matchTemplates([("small", small)], image, maxOverlap=1.5))
    _raises("not_tuple", lambda: mtm.matchTemplates([small], image))
    _raises("float64", lambda: mtm.matchTemplates([("small", small.astype(np.float64))], image.astype(np.float64)))
    _raises("empty_image_h", lambda: mtm.matchTemplates([("small", small)], image[0:0]))
    _raises("empty_image_w", lambda: mtm.matchTemplates([("small", small)], image[:, 0:0]))
    _raises("empty_templ_h", lambda: mtm.matchTemplates([("e", small[0:0])], image))
    _raises("empty_templ_w", lambda: mtm.matchTemplates([("e", small[:, 0:0])], image))


def test_validation_order(mtm):
    image = load_coins()
    small, _ = coin_templates(image)
    # maxOverlap is checked before anything else (MTM/__init__.py:286), N_object before the image
    with pytest.raises(ValueError, match="Maximal overlap"):
        mtm.matchTemplates([small], image[0:0], maxOverlap=2, N_object=1.5)
    with pytest.raises(TypeError, match="N_object must be an integer"):
        mtm.matchTemplates([small], image[0:0], N_object=1.5)
    with pytest.raises(ValueError, match="Image has a height of 0"):
        mtm.matchTemplates([small], image[0:0])


def test_api_surface(mtm):
    import inspect
    def reference_part(fn):
        """the parameters a reference caller can use; anything the drop-in adds must be keyword-only"""
        ps = list(inspect.signature(fn).parameters.values())
        extra = [q for q in ps if q.kind is inspect.Parameter.KEYWORD_ONLY]
        assert all(q.default is None for q in extra)
        return [q.name for q in ps if q.kind is not inspect.Parameter.KEYWORD_ONLY]

    sig = inspect.signature(mtm.matchTemplates)
    assert reference_part(mtm.matchTemplates) == ["listTemplates", "image", "method", "N_object", "score_threshold", "maxOverlap", "searchBox"]
    assert sig.parameters["method"].default == 5 and sig.parameters["maxOverlap"].default == 0.25
    assert sig.parameters["score_threshold"].default == 0.5 and sig.parameters["N_object"].default == float("inf")
    assert reference_part(mtm.findMatches) == ["listTemplates", "image", "method", "N_object", "score_threshold", "searchBox"]
    sig = inspect.signature(mtm.computeScoreMap)
    assert list(sig.parameters) == ["template", "image", "method", "mask"]
    sig = inspect.signature(mtm.NMS)
    assert list(sig.parameters) == ["listHit", "scoreThreshold", "sortAscending", "N_object", "maxOverlap"]
    assert sig.parameters["maxOverlap"].default == 0.5
    assert mtm.__version__.startswith("2.0.1")


def test_draw_boxes(mtm):
    image = load_coins()
    hits = [("a", (10, 20, 30, 40), 1.0)]
    rgb = mtm.drawBoxesOnRGB(image, hits, boxThickness=1)
    assert rgb.shape == image.shape + (3,) and tuple(rgb[20, 10]) == (255, 255, 0) and tuple(rgb[60, 40]) == (255, 255, 0)
    assert tuple(rgb[30, 25]) == (image[30, 25],) * 3
    gray = mtm.drawBoxesOnGray(image, hits, boxThickness=1)
    assert gray.shape == image.shape and gray[20, 10] == 255 and gray[30, 25] == image[30, 25]


def test_draw_fixture(mtm):
    """drawBoxesOnRGB / drawBoxesOnGray against tests/golden/draw_fixture.json, which make_draw_fixture.py derives from
    the definitions of cv2.rectangle (thickness 1), COLOR_GRAY2RGB and the fixed-point COLOR_RGB2GRAY with code of its
    own (reference MTM/__init__.py:327-341, :375-389): outlines include both corners, boxes running off the canvas are
    clipped, the input image is not modified."""
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "draw_fixture.json")) as f:
        fx = json.load(f)
    gray, rgb = np.array(fx["image_gray"], np.uint8), np.array(fx["image_rgb"], np.uint8)
    hits = [(h[0], tuple(h[1]), h[2]) for h in fx["hits"]]
    g0, r0 = gray.copy(), rgb.copy()
    assert mtm.drawBoxesOnRGB(gray, hits, boxThickness=1).tolist() == fx["drawBoxesOnRGB(gray, thickness=1)"]
    assert mtm.drawBoxesOnRGB(rgb, hits, boxThickness=1, boxColor=(10, 20, 30)).tolist() == \
        fx["drawBoxesOnRGB(rgb, thickness=1, boxColor=(10,20,30))"]
    assert mtm.drawBoxesOnGray(rgb, hits, boxThickness=1).tolist() == fx["drawBoxesOnGray(rgb, thickness=1)"]
    assert mtm.drawBoxesOnGray(gray, hits, boxThickness=1, boxColor=99).tolist() == fx["drawBoxesOnGray(gray, thickness=1, boxColor=99)"]
    assert (gray == g0).all() and (rgb == r0).all()
    px16 = np.array([fx["rgb2gray_uint16"]["pixels"]], np.uint16)
    assert mtm.drawBoxesOnGray(px16, []).tolist() == [fx["rgb2gray_uint16"]["gray"]]


def test_draw_labels_and_gray_conversion(mtm):
    """showLabel draws the template name with its bottom-left corner at the box corner (cv2.putText's anchor), in
    labelColor; drawBoxesOnGray converts RGB with OpenCV's 15-bit fixed-point weights."""
    image = np.zeros((80, 160), np.uint8)
    hits = [("AB_90", (20, 40, 50, 30), 0.9)]
    plain = mtm.drawBoxesOnGray(image, hits, boxThickness=1, boxColor=200)
    lab = mtm.drawBoxesOnGray(image, hits, boxThickness=1, boxColor=200, showLabel=True, labelColor=255, labelScale=0.5)
    text = (lab == 255)
    assert text.any() and not (plain == 255).any()
    ys, xs = np.nonzero(text)
    assert ys.max() == 39 and ys.min() == 40 - 14 and xs.min() == 20          # 7 dots x 2 px tall, anchored at (x, y)
    assert xs.max() - xs.min() + 1 == (6 * 5 - 1) * 2                         # five glyphs, 5 columns + 1 gap each
    rgb = mtm.drawBoxesOnRGB(image, hits, showLabel=True, labelColor=(1, 2, 3))
    assert tuple(rgb[ys[0], xs[0]]) == (1, 2, 3)
    # a label running off the canvas is clipped, not an error
    mtm.drawBoxesOnRGB(image, [("a very long label that does not fit at all", (150, 3, 5, 5), 1.0)], showLabel=True)
    # RGB -> gray
    col = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [12, 200, 77]]], np.uint8)
    g = mtm.drawBoxesOnGray(col, [])
    assert g.dtype == np.uint8 and g.tolist() == [[76, 150, 29, 255, (9798 * 12 + 19235 * 200 + 3735 * 77 + 16384) >> 15]]
    assert mtm.drawBoxesOnGray(col.astype(np.float32), []).dtype == np.float32


def test_template_matcher_refuses_reentry_while_streaming(mtm):
    """A generator of match_stream owns the context until it is finished or closed (the lock is re-entrant, so
    the consuming thread itself must be told)."""
    class Ctx:
        import threading
        lock = threading.RLock()

        def set_templates(self, t, m):
            pass

        def set_image(self, im):
            pass

        def find_matches(self, mode, thr, nxt=None):
            return np.zeros(0, dtype=mtm._lib.HIT_DTYPE)
        find_matches_image = lambda self, im, mode, thr: np.zeros(0, dtype=mtm._lib.HIT_DTYPE)   # noqa: E731

    img = np.zeros((40, 40), np.uint8)
    m = mtm.TemplateMatcher([("t", img[:8, :8])], context=Ctx())
    gen = m.match_stream([img, img, img])
    assert next(gen) == []
    with pytest.raises(RuntimeError, match="match_stream"):
        m.match(img)
    with pytest.raises(RuntimeError, match="match_stream"):
        next(m.match_stream([img]))
    gen.close()
    assert m.match(img) == []                 # released
    assert list(m.match_stream([img, img])) == [[], []]


def test_template_records_memo_only_for_zero_copy_arrays():
    """MTM._lib.Context._records reuses the marshalled template list of the previous call only when every record points
    at the caller's own buffer; arrays that had to be copied (np.rot90 views ...) are marshalled - copied - again, so an
    in-place edit between two calls is seen (round 2 advisor finding)."""
    import numpy as np
    from MTM import _lib
    ctx = object.__new__(_lib.Context)          # no library / GPU needed for the marshalling
    ctx._rec_key, ctx._rec, ctx._rec_keep = None, None, None
    base = np.arange(12 * 9, dtype=np.uint8).reshape(12, 9)
    mask = np.ones((12, 9), np.uint8)
    lt = [(base, None), (base[2:8, 1:7], mask[2:8, 1:7])]       # contiguous pixels per row: zero copy
    r1 = ctx._records(lt)
    assert ctx._rec_key is not None and ctx._records(lt) is r1
    assert int(r1["px"][0]) == base.ctypes.data and int(r1["px"][1]) == base[2:8, 1:7].ctypes.data
    view = np.rot90(base)
    lt2 = [(view, None)]
    r2 = ctx._records(lt2)
    assert ctx._rec_key is None                                   # copied: never reused
    copy1 = ctx._rec_keep[0][0]
    assert copy1 is not view and (copy1 == view).all()
    base[...] = 255
    r3 = ctx._records(lt2)
    assert r3 is not r2
    assert (ctx._rec_keep[0][0] == 255).all()
    assert _lib._zero_copy(lt, [base, lt[1][0], lt[1][1]]) and not _lib._zero_copy(lt2, [copy1])


def test_validated_list_memo_never_hides_a_change(mtm):
    """Round 5: the per-template checks of a list are remembered by the identity of its tuples, the shapes of their arrays
    and the image's shape (a loop over images passes the same tuples again and again).  Anything that could change the
    outcome of the reference's checks (MTM/__init__.py:147-167) must still be seen: another tuple in the list, a list
    that grew, an array reshaped in place, a smaller image, a searchBox."""
    image = load_coins()
    small, big = coin_templates(image)
    lt = [("small", small), ("big", big)]
    v = mtm._validate_search
    assert v(lt, image, float("inf"), None)[1:] == (0, 0)
    assert mtm._list_memo is not None and mtm._list_memo.matches(lt) and mtm._list_memo.matches(list(lt))
    v(lt, image, float("inf"), None)                                  # remembered: nothing to see, nothing raised
    # a template that does not fit any more: same list object, one tuple replaced
    lt[0] = ("tooLarge", np.pad(image, 1))
    _raises("too_large", lambda: v(lt, image, float("inf"), None))
    lt[0] = ("small", small)
    v(lt, image, float("inf"), None)
    # the same tuples against a smaller image / a searchBox
    with pytest.raises(ValueError, match="is larger than image"):
        v(lt, image[:50, :50], float("inf"), None)
    _raises("searchbox_small", lambda: v(lt, image, float("inf"), (0, 0, 20, 20)))
    # not a tuple, appended to a remembered list
    lt.append(small)
    _raises("not_tuple", lambda: v(lt, image, float("inf"), None))
    lt.pop()
    # an array reshaped IN PLACE (same object, same buffer): one row of 1558 pixels now, wider than the image
    t = np.ascontiguousarray(small)
    lt2 = [("tooLarge", t)]
    v(lt2, image, float("inf"), None)
    t.shape = (1, t.size)
    _raises("too_large", lambda: v(lt2, image, float("inf"), None))
    # dtype of a template / shape or dtype of a MASK changed in place: the records built for the old geometry must not be reused
    t8, m8 = np.ascontiguousarray(small), np.full(small.shape, 255, np.uint8)
    lt3 = [("m", t8, m8)]
    v(lt3, image, float("inf"), None)
    assert mtm._list_memo.matches(lt3)
    m8.dtype = np.int8
    assert not mtm._list_memo.matches(lt3)
    m8.dtype = np.uint8
    assert mtm._list_memo.matches(lt3)
    m8.shape = (1, m8.size)
    assert not mtm._list_memo.matches(lt3)
    m8.shape = small.shape
    t8.dtype = np.int8
    assert not mtm._list_memo.matches(lt3)
    # labels follow the list, not the memo of another one
    raw = np.zeros(2, dtype=mtm._lib.HIT_DTYPE)
    raw["templ_idx"] = [0, 1]
    assert [h[0] for h in mtm._to_hit_list(raw, lt, 0, 0)] == ["small", "big"]
    other = [("a", small), ("b", big)]
    assert [h[0] for h in mtm._to_hit_list(raw, other, 0, 0)] == ["a", "b"]
