"""Host-layer behaviour that needs no GPU: the reference's argument validation, messages and
their order (MTM/__init__.py:129-167, :286-287, :67-68), checked against the messages captured
from the unmodified reference (tests/golden/golden.json "errors")."""
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
