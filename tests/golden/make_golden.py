"""
Golden-vector generator.  Run in the BUILD CONTAINER only:

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden.py

It imports the UNMODIFIED reference package from /root/reference (read-only) with
tests/golden/cv2_standin on the path (cv2 is not installable here), so that the reference's own
orchestration (MTM/__init__.py:95-296, MTM/NMS.py:20-84) and the REAL skimage 0.18.3 /
scipy peak finders (MTM/__init__.py:34,40,45) produce the expected outputs; only the cv2
arithmetic comes from oracle/mtm_oracle.py.  Outputs are DATA (inputs + expected results) written
to tests/golden/*.npz / *.json.  No reference source is copied.

Also stores the hit lists printed in the reference's executed notebooks (G1-G3); those values were
produced by real OpenCV 4.7.0 and are what pins the oracle's arithmetic.
"""
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, "cv2_standin"))
sys.path.insert(0, "/root/reference")
sys.path.append(os.path.join(ROOT, "multitemplatematching-python_amd"))     # synth only: `MTM` must be the reference's
sys.dont_write_bytecode = True
warnings.simplefilter("ignore")

import cv2  # the stand-in  # noqa: E402
import MTM  # the unmodified reference  # noqa: E402
import synth  # noqa: E402
from skimage.data import coins  # noqa: E402
from skimage.filters import threshold_otsu  # noqa: E402

assert MTM.__file__.startswith("/root/reference"), MTM.__file__


def hits_json(hits):
    return [[h[0], [int(v) for v in h[1]], float(np.float32(h[2]))] for h in hits]


def canon(hits):
    return sorted(hits_json(hits), key=lambda h: (-h[2], h[0], h[1]))


out = {}
image = coins()
np.savez_compressed(os.path.join(HERE, "coins.npz"), image=image)
small = image[37:37 + 38, 80:80 + 41]
big = image[14:14 + 59, 302:302 + 65]

# ---- hit lists printed in the reference notebooks (real OpenCV 4.7.0) ------------------------
out["notebook_G1"] = {  # tutorials/Tutorial1-Introduction.ipynb cell 11/13
    "call": dict(templates=["small"], method=5, score_threshold=0.5, maxOverlap=0),
    "hits": [["small", [80, 37, 41, 38], 1.0], ["small", [133, 108, 41, 38], 0.8608093],
             ["small", [134, 178, 41, 38], 0.8161434], ["small", [82, 106, 41, 38], 0.80047214],
             ["small", [316, 105, 41, 38], 0.74788666], ["small", [23, 178, 41, 38], 0.74224424],
             ["small", [185, 103, 41, 38], 0.68531644], ["small", [257, 34, 41, 38], 0.6578578],
             ["small", [26, 103, 41, 38], 0.61795336], ["small", [22, 37, 41, 38], 0.60669357],
             ["small", [335, 247, 41, 38], 0.58576703], ["small", [227, 239, 41, 38], 0.5786777],
             ["small", [95, 244, 41, 38], 0.5769806], ["small", [286, 237, 41, 38], 0.54384065],
             ["small", [251, 106, 41, 38], 0.5254569], ["small", [157, 234, 41, 38], 0.5249821],
             ["small", [134, 37, 41, 38], 0.5114375], ["small", [196, 27, 41, 38], 0.50683254]]}
out["notebook_G2"] = {  # tutorials/WithMask.ipynb cell 8
    "call": dict(templates=["small"], method=3, score_threshold=0.8, maxOverlap=0),
    "hits": [["testMask", [80, 37, 41, 38], 1.0000002], ["testMask", [82, 106, 41, 38], 0.9777622],
             ["testMask", [22, 37, 41, 38], 0.9775883], ["testMask", [133, 108, 41, 38], 0.97702277],
             ["testMask", [23, 178, 41, 38], 0.97312117], ["testMask", [341, 247, 41, 38], 0.9723443],
             ["testMask", [134, 36, 41, 38], 0.97204673], ["testMask", [26, 103, 41, 38], 0.9717183],
             ["testMask", [257, 33, 41, 38], 0.9712], ["testMask", [199, 33, 41, 38], 0.97099215],
             ["testMask", [336, 158, 41, 38], 0.96930236], ["testMask", [134, 178, 41, 38], 0.967385],
             ["testMask", [283, 247, 41, 38], 0.96563774], ["testMask", [229, 242, 41, 38], 0.9653908],
             ["testMask", [185, 104, 41, 38], 0.9650693], ["testMask", [251, 106, 41, 38], 0.9625988],
             ["testMask", [316, 105, 41, 38], 0.96015817], ["testMask", [157, 236, 41, 38], 0.95904356],
             ["testMask", [29, 231, 41, 38], 0.9586952], ["testMask", [315, 13, 41, 38], 0.95846957],
             ["testMask", [91, 246, 41, 38], 0.9571323], ["testMask", [82, 180, 41, 38], 0.95397776],
             ["testMask", [193, 169, 41, 38], 0.9513115], ["testMask", [256, 178, 41, 38], 0.94258595],
             ["testMask", [305, 202, 41, 38], 0.8054776]]}
out["notebook_G3"] = {  # tutorials/WithMask.ipynb cell 11 (Otsu mask)
    "call": dict(templates=["small+otsu mask"], method=3, score_threshold=0.8, maxOverlap=0),
    "hits": [["testMask", [80, 37, 41, 38], 1.0], ["testMask", [5, 70, 41, 38], 0.99374694],
             ["testMask", [0, 2, 41, 38], 0.9936011], ["testMask", [41, 141, 41, 38], 0.99231666],
             ["testMask", [137, 28, 41, 38], 0.991843], ["testMask", [230, 0, 41, 38], 0.99184036],
             ["testMask", [211, 69, 41, 38], 0.9910499], ["testMask", [158, 141, 41, 38], 0.9904751],
             ["testMask", [102, 137, 41, 38], 0.9896537], ["testMask", [334, 159, 41, 38], 0.98959434],
             ["testMask", [157, 67, 41, 38], 0.9886728], ["testMask", [335, 248, 41, 38], 0.9880982],
             ["testMask", [225, 239, 41, 38], 0.9880205], ["testMask", [280, 139, 41, 38], 0.9877867],
             ["testMask", [274, 0, 41, 38], 0.9876592], ["testMask", [213, 130, 41, 38], 0.9875014],
             ["testMask", [82, 177, 41, 38], 0.9870325], ["testMask", [285, 246, 41, 38], 0.98693585],
             ["testMask", [289, 201, 41, 38], 0.9841472], ["testMask", [21, 246, 41, 38], 0.98321056],
             ["testMask", [95, 245, 41, 38], 0.9819636], ["testMask", [159, 236, 41, 38], 0.98101896],
             ["testMask", [279, 65, 41, 38], 0.9805516], ["testMask", [189, 0, 41, 38], 0.9799391],
             ["testMask", [323, 19, 41, 38], 0.97912544], ["testMask", [343, 68, 41, 38], 0.97203827],
             ["testMask", [0, 206, 41, 38], 0.95725894], ["testMask", [50, 96, 41, 38], 0.9151733],
             ["testMask", [220, 171, 41, 38], 0.8099606]]}

otsu = int(threshold_otsu(small))
mask = ((small > otsu) * 255).astype(np.uint8)
out["otsu_threshold"] = otsu
out["otsu_mask_count"] = int((mask > 0).sum())

# ---- the same calls through the unmodified reference + real skimage/scipy ----------------------
ref = {}
ref["G1"] = hits_json(MTM.matchTemplates([("small", small)], image, score_threshold=0.5, method=5, maxOverlap=0))
ref["G2"] = hits_json(MTM.matchTemplates([("testMask", small)], image, method=3, score_threshold=0.8, maxOverlap=0))
ref["G3"] = hits_json(MTM.matchTemplates([("testMask", small, mask)], image, method=3, score_threshold=0.8, maxOverlap=0))
# test.py:24
ref["testpy"] = hits_json(MTM.matchTemplates([("small", small), ("big", big)], image, score_threshold=0.3, method=5, maxOverlap=0))
# Tutorial1 cell 22 (two templates, thr 0.4)
ref["tut1_two"] = hits_json(MTM.matchTemplates([("small", small), ("large", big)], image, score_threshold=0.4, method=5, maxOverlap=0))
# method 1 (difference score, minima)
ref["sqdiff_normed"] = hits_json(MTM.matchTemplates([("small", small), ("big", big)], image, method=1, score_threshold=0.2, maxOverlap=0))
# ---- border rule of peak_local_max.  scikit-image 0.18.3 (the only release importable here) runs its 3x3
# maximum filter with mode='constant'; releases >= 0.19 pass mode='nearest' - with zero padding a local MINIMUM
# on the map border (methods 0/1: the map is negated, MTM/__init__.py:53) can never be a peak.  The
# "<name>@nearest" fixtures are the same calls through the same 0.18.3 code with that one argument replaced.
import contextlib  # noqa: E402
import skimage.feature.peak as _pk  # noqa: E402


class _NdiNearest:
    """scipy.ndimage with maximum_filter(..., mode='nearest'), every other attribute untouched."""

    def __init__(self, ndi):
        self._ndi = ndi

    def __getattr__(self, name):
        return getattr(self._ndi, name)

    def maximum_filter(self, *a, **kw):
        kw["mode"] = "nearest"
        return self._ndi.maximum_filter(*a, **kw)


@contextlib.contextmanager
def nearest_border():
    saved = _pk.ndi
    _pk.ndi = _NdiNearest(saved)
    try:
        yield
    finally:
        _pk.ndi = saved


corner = image[0:38, 0:41]          # an object touching the image corner: its best match is map pixel (0, 0)
edge = image[120:158, 343:384]      # ... and one touching the right edge
border_calls = {
    "sqdiff_normed": dict(lt=[("small", small), ("big", big)], kw=dict(method=1, score_threshold=0.2, maxOverlap=0)),
    "corner_m1": dict(lt=[("corner", corner), ("edge", edge)], kw=dict(method=1, score_threshold=0.25, maxOverlap=0.1)),
    "corner_m5_negthr": dict(lt=[("corner", corner)], kw=dict(method=5, score_threshold=-0.2, maxOverlap=0.0)),
}
for name, cdef in border_calls.items():
    ref[name + "@constant"] = hits_json(MTM.matchTemplates(cdef["lt"], image, **cdef["kw"]))
    with nearest_border():
        ref[name + "@nearest"] = hits_json(MTM.matchTemplates(cdef["lt"], image, **cdef["kw"]))
ref["corner_m1_pre@constant"] = canon(MTM.findMatches([("corner", corner), ("edge", edge)], image, method=1, score_threshold=0.25))
with nearest_border():
    ref["corner_m1_pre@nearest"] = canon(MTM.findMatches([("corner", corner), ("edge", edge)], image, method=1, score_threshold=0.25))
# maxOverlap > 0, finite N_object
ref["overlap025"] = hits_json(MTM.matchTemplates([("small", small), ("big", big)], image, score_threshold=0.3, method=5, maxOverlap=0.25))
ref["nobj3"] = hits_json(MTM.matchTemplates([("small", small), ("big", big)], image, score_threshold=0.3, method=5, maxOverlap=0.25, N_object=3))
ref["nobj1"] = hits_json(MTM.matchTemplates([("small", small)], image, method=5, N_object=1))
ref["nobj1_sqdiff"] = hits_json(MTM.matchTemplates([("big", big)], image, method=1, N_object=1))
ref["nobj0"] = hits_json(MTM.matchTemplates([("small", small), ("big", big)], image, score_threshold=0.3, method=5, N_object=0))
# searchBox (test.py:41-42): as large as the search region -> 1x1 map
ref["searchbox_exact"] = hits_json(MTM.matchTemplates([("big", big)], image, searchBox=(302, 14) + big.shape[::-1]))
ref["searchbox"] = hits_json(MTM.matchTemplates([("small", small)], image, score_threshold=0.5, maxOverlap=0, searchBox=(10, 20, 300, 200)))
# 1x1 map: template == image
ref["full_image"] = hits_json(MTM.matchTemplates([("all", image)], image))
# 1-D maps: template as tall / as wide as the image
tall = image[:, 100:141]
wide = image[50:90, :]
ref["tall"] = canon(MTM.findMatches([("tall", tall)], image, score_threshold=0.5))
ref["wide"] = canon(MTM.findMatches([("wide", wide)], image, score_threshold=0.5))
# uint16 -> float32 policy (MTM/__init__.py:71-74)
img16 = image.astype(np.uint16) * 257
ref["uint16"] = hits_json(MTM.matchTemplates([("small", img16[37:75, 80:121])], img16, score_threshold=0.5, method=5, maxOverlap=0))
# float32, CCORR_NORMED and CCOEFF (unnormalised: thresholds are raw)
imgf = image.astype(np.float32) / 255.0
ref["float32_m3"] = hits_json(MTM.matchTemplates([("small", imgf[37:75, 80:121])], imgf, method=3, score_threshold=0.95, maxOverlap=0.1))
# findMatches pre-NMS (canonical order: the reference's cross-template order is thread timing)
ref["find_pre_nms"] = canon(MTM.findMatches([("small", small), ("big", big)], image, score_threshold=0.3))
# NMS demo of MTM/NMS.py:89-94
demo = [("1", (780, 350, 700, 480), 0.8), ("1", (806, 416, 716, 442), 0.6), ("1", (1074, 530, 680, 390), 0.4)]
ref["nms_demo"] = hits_json(MTM.NMS(demo, scoreThreshold=0.3, sortAscending=False, maxOverlap=0.5, N_object=2))
# RGB
rgb = np.stack([image, np.roll(image, 3, axis=1), 255 - image], axis=2)
rgb_small = np.ascontiguousarray(rgb[37:75, 80:121])
ref["rgb"] = hits_json(MTM.matchTemplates([("small", rgb_small)], rgb, score_threshold=0.5, method=5, maxOverlap=0))

# error messages (test.py:39-45 and the validation block MTM/__init__.py:129-167, :286-292)
errors = {}


def err(name, fn):
    try:
        fn()
        errors[name] = None
    except Exception as e:  # noqa: BLE001
        errors[name] = [type(e).__name__, str(e)]


err("searchbox_small", lambda: MTM.matchTemplates([("small", small), ("big", big)], image, searchBox=(0, 0, 20, 20)))
err("too_large", lambda: MTM.matchTemplates([("tooLarge", np.pad(image, 1))], image))
err("nobj_float", lambda: MTM.matchTemplates([("small", small)], image, N_object=2.5))
err("nobj_npint", lambda: MTM.matchTemplates([("small", small)], image, N_object=np.int64(2)))
err("overlap_range", lambda: MTM.matchTemplates([("small", small)], image, maxOverlap=1.5))
err("method0", lambda: MTM.matchTemplates([("small", small)], image, method=0))
err("not_tuple", lambda: MTM.matchTemplates([small], image))
err("float64", lambda: MTM.matchTemplates([("small", small.astype(np.float64))], image.astype(np.float64)))
err("empty_image_h", lambda: MTM.matchTemplates([("small", small)], image[0:0]))
err("empty_image_w", lambda: MTM.matchTemplates([("small", small)], image[:, 0:0]))
err("empty_templ_h", lambda: MTM.matchTemplates([("e", small[0:0])], image))
err("empty_templ_w", lambda: MTM.matchTemplates([("e", small[:, 0:0])], image))
out["errors"] = errors

# ---- score maps (float32) ---------------------------------------------------------------------
maps = {}
for name, t in (("small", small), ("big", big)):
    for m in (1, 2, 3, 4, 5):
        maps["%s_m%d" % (name, m)] = MTM.computeScoreMap(t, image, m)
maps["small_m3_mask"] = MTM.computeScoreMap(small, image, 3, mask=mask)
maps["small_m0_mask"] = MTM.computeScoreMap(small, image, 0, mask=mask)
maps["small_m0"] = MTM.computeScoreMap(small, image, 0)
maps["rgb_m5"] = MTM.computeScoreMap(rgb_small, rgb, 5)
maps["u16_m5"] = MTM.computeScoreMap(img16[37:75, 80:121], img16, 5)
# keep the fixture small: every 3rd row/column of each map plus whole-map float64 checksums
sub = {k: np.ascontiguousarray(v[::3, ::3]) for k, v in maps.items()}
out["map_checksums"] = {k: dict(shape=list(v.shape), sum=float(v.astype(np.float64).sum()),
                               abs_sum=float(np.abs(v.astype(np.float64)).sum()),
                               argmax=int(np.argmax(v)), argmin=int(np.argmin(v))) for k, v in maps.items()}
np.savez_compressed(os.path.join(HERE, "coins_maps_sub3.npz"), mask=mask, **sub)

# ---- synthetic planted-template cases (reduced-size versions of BASELINE configs 2-5) -----------
synth_cases = {
    "cfg2_small": dict(seed=2, image_hw=(360, 640), n_base=4, templ=32),
    "cfg3_small": dict(seed=3, image_hw=(400, 640), n_base=3, templ=32, rotations=4),
    "cfg4_small": dict(seed=4, image_hw=(480, 800), n_base=40, templ=32, noisy_per_unit=1),
    "cfg5_small": dict(seed=5, image_hw=(480, 800), n_base=2, templ=32, scales=(16, 28, 40, 52, 64), masked=True),
    "rgb_small": dict(seed=6, image_hw=(300, 420), n_base=3, templ=24, channels=3),
}
syn = {}
for name, kw in synth_cases.items():
    img, units, plants = synth.make_workload(**kw)
    method = 3 if kw.get("masked") else 5
    thr = 0.9 if kw.get("masked") else 0.5
    syn[name] = dict(
        kwargs={k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()},
        method=method, score_threshold=thr,
        image_sum=int(img.astype(np.int64).sum()),
        pre_nms=canon(MTM.findMatches(units, img, method=method, score_threshold=thr)),
        post_nms=canon(MTM.matchTemplates(units, img, method=method, score_threshold=thr, maxOverlap=0.25)),
        plants=[[p[0], list(p[1]), p[2]] for p in plants])
out["synthetic"] = syn
out["reference_run"] = ref
out["versions"] = dict(MTM=MTM.__version__, numpy=np.__version__,
                       skimage=__import__("skimage").__version__, scipy=__import__("scipy").__version__)

with open(os.path.join(HERE, "golden.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote golden.json:", {k: (len(v) if hasattr(v, "__len__") else v) for k, v in ref.items()})
print("errors:", errors)
