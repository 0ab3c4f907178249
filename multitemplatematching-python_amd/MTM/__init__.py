"""
Multi-Template-Matching on MI355X: drop-in for the reference package's module-level API.

Same functions, arguments, defaults, exceptions and warnings as the reference
(``MTM/__init__.py:56`` computeScoreMap, ``:95`` findMatches, ``:247`` matchTemplates,
``MTM/NMS.py:20`` NMS), but the arithmetic - cv2.matchTemplate, skimage peak_local_max / scipy
find_peaks, cv2.minMaxLoc, cv2.dnn.NMSBoxes - runs in libmtm_hip.so (hand-written HIP kernels
for gfx950 behind the C ABI of include/mtm_hip.h, bound with ctypes).  All templates of a call
are batched into a few launches instead of one thread-pool task per template
(reference ``MTM/__init__.py:172-175``).

No OpenCV, scikit-image, scipy or PyTorch on this path, and no CPU fallback: a missing library or
GPU raises.
"""
import warnings
from operator import is_ as _is
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .NMS import NMS, Hit
from .version import __version__

pinned_empty = _lib.pinned_empty        # numpy arrays in page-locked memory (faster uploads); optional

__all__ = ["NMS", "Hit", "matchTemplates", "findMatches", "computeScoreMap", "TemplateMatcher", "pinned_empty", "drawBoxesOnRGB",
           "drawBoxesOnGray", "TM_SQDIFF", "TM_SQDIFF_NORMED", "TM_CCORR", "TM_CCORR_NORMED",
           "TM_CCOEFF", "TM_CCOEFF_NORMED", "__version__"]

# OpenCV's TemplateMatchModes values: the reference takes them from cv2 (defaults at
# MTM/__init__.py:56, :95, :247); the drop-in defines them itself.
TM_SQDIFF, TM_SQDIFF_NORMED, TM_CCORR, TM_CCORR_NORMED, TM_CCOEFF, TM_CCOEFF_NORMED = range(6)

BBox = Tuple[int, int, int, int]            # (x, y, width, height), x,y = top left corner
TemplateTuple = Tuple[str, np.ndarray, Optional[np.ndarray]]

_MSG_MASK_METHOD = ("Template matching method not compatible with use of mask (only 0/TM_SQDIFF or "
                    "3/TM_CCORR_NORMED).\n-> Ignoring mask.")
_MSG_MASK_SHAPE = "Mask does not have the same dimension or bit depth than the template.\n-> Ignoring mask."
_MSG_MASK_UNSUPPORTED = ("Template matching method not supporting the use of Mask. "
                         "Use 0/TM_SQDIFF or 3/TM_CCORR_NORMED.")


def _apply_pixel_policy(template, image, method, mask):
    """dtype and mask policy of the reference's computeScoreMap (MTM/__init__.py:67-88).
    Returns (template, image, mask) ready for the library."""
    if template.dtype == "float64" or image.dtype == "float64":
        raise ValueError("64-bit images not supported, max 32-bit")

    # only 8-bit/8-bit stays 8-bit; every other combination is matched in float32
    native16 = template.dtype == "uint16" and image.dtype == "uint16"
    template16, image16 = template, image
    if not (template.dtype == "uint8" and image.dtype == "uint8") and not (native16 and mask is None):
        template = np.float32(template)
        image = np.float32(image)
        if mask is not None:
            mask = np.float32(mask)

    if mask is not None:
        if method not in (0, 3):
            mask = None
            warnings.warn(_MSG_MASK_METHOD)
        elif not (mask.shape == template.shape and mask.dtype == template.dtype):
            mask = None
            warnings.warn(_MSG_MASK_SHAPE)
    # uint16 / uint16 without a mask: the float32 cast of the reference is exact, so the library takes
    # the 16-bit pixels as they are (MTM_U16: exact integer matching on the int8 matrix cores)
    if native16 and mask is None:
        template, image = template16, image16
    return template, image, mask


def _check_opencv_preconditions(template, image):
    """What cv2.matchTemplate itself would reject (cv2.error in the reference)."""
    if template.ndim not in (2, 3) or image.ndim not in (2, 3):
        raise ValueError("image and template must be 2-D (grayscale) or 3-D (rows, cols, channels) arrays")
    tc = 1 if template.ndim == 2 else template.shape[2]
    ic = 1 if image.ndim == 2 else image.shape[2]
    if tc != ic:
        raise ValueError("image and template must have the same number of channels")
    if ic > 4:
        raise ValueError("at most 4 channels are supported")
    if template.shape[0] > image.shape[0] or template.shape[1] > image.shape[1]:
        raise ValueError("template is larger than the image")


def computeScoreMap(template: np.ndarray, image: np.ndarray, method: int = TM_CCOEFF_NORMED, mask=None):
    """
    Score map of one template over an image (reference MTM/__init__.py:56-92).

    The template must not be larger than the image.  A mask of the template's shape and dtype
    restricts the comparison to part of the template (methods 0/TM_SQDIFF and 3/TM_CCORR_NORMED
    only).  Anything that is not uint8/uint8 is matched in float32; float64 raises.

    Returns a float32 array of shape (H - h + 1, W - w + 1).
    """
    template, image, mask = _apply_pixel_policy(template, image, method, mask)
    _check_opencv_preconditions(template, image)
    ctx = _lib.default_context()
    with ctx.lock:
        ctx.set_image(image)
        ctx.set_templates([(template, mask)], method)
        return ctx.score_map(0, (image.shape[0] - template.shape[0] + 1, image.shape[1] - template.shape[1] + 1))


def _validate_search(listTemplates, image, N_object, searchBox):
    """Argument checks of the reference's findMatches, in its order (MTM/__init__.py:129-167).
    Returns (cropped image, xOffset, yOffset)."""
    if N_object != float("inf") and not isinstance(N_object, int):
        raise TypeError("N_object must be an integer")

    if image.shape[0] == 0:
        raise ValueError("Image has a height of 0.")
    if image.shape[1] == 0:
        raise ValueError("Image has a width of 0.")

    if searchBox is not None:
        xOffset, yOffset, searchWidth, searchHeight = searchBox
        image = image[yOffset: yOffset + searchHeight, xOffset: xOffset + searchWidth]
    else:
        xOffset = yOffset = 0

    ishape = image.shape
    # (a loop over images passes the SAME template tuples call after call: their checks against an image of the same shape
    # are remembered - see _ListMemo; an error is never remembered)
    memo = _list_memo
    if memo is not None and memo.ishape == ishape and memo.matches(listTemplates):
        return image, xOffset, yOffset
    for index, tempTuple in enumerate(listTemplates):
        if not isinstance(tempTuple, tuple) or len(tempTuple) < 2:
            raise ValueError("listTemplates should be a list of tuples as ('name','array') or ('name', 'array', 'mask')")
        tshape = tempTuple[1].shape
        if tshape[0] == 0:
            raise ValueError(f"Template '{tempTuple[0]}' has a height of 0.")
        if tshape[1] == 0:
            raise ValueError(f"Template '{tempTuple[0]}' has a width of 0.")
        # the reference compares the shapes dimension by dimension as far as both have one (channels included)
        if tshape[0] > ishape[0] or tshape[1] > ishape[1] or (len(tshape) > 2 and len(ishape) > 2 and tshape[2] > ishape[2]):
            where = "searchBox" if (searchBox is not None) else "image"
            raise ValueError("Template '{}' at index {} in the list of templates is larger than {}.".format(
                tempTuple[0], index, where))
    _remember_list(listTemplates, ishape)
    return image, xOffset, yOffset


class _ListMemo:
    """What the last validated template list was, by IDENTITY of its tuples plus the shapes of their arrays (the one thing
    about an array that can change in place and matters to the checks; changed PIXELS are the library's business - it
    compares the bytes with the copy it packed from in every call), and what the per-call loops over it produced: the
    reference's checks against an image of this shape (MTM/__init__.py:147-167), the engine units of the 8-bit path, the
    label column of the hit list.  One immutable snapshot, replaced as a whole: safe to read without a lock."""
    __slots__ = ("items", "shapes", "ishape", "units_pack", "labels")

    @staticmethod
    def _geometry(items):
        """Everything about a tuple's arrays that can change IN PLACE and matters to the checks and to the records built
        from them: shape and dtype of the template, shape and dtype of its mask (`arr.shape = ...`, `arr.dtype = ...`)."""
        return [(t[1].shape, t[1].dtype, (t[2].shape, t[2].dtype) if len(t) > 2 and t[2] is not None else None) for t in items]

    def __init__(self, listTemplates, ishape):
        self.items = tuple(listTemplates)            # (keeps the tuples alive: an id cannot be recycled)
        self.shapes = self._geometry(self.items)
        self.ishape = ishape
        self.units_pack = None                       # (key, units, ignored masks): ONE attribute, replaced as a whole
        self.labels = None

    def matches(self, listTemplates):
        it = self.items
        return len(listTemplates) == len(it) and all(map(_is, listTemplates, it)) and \
            self._geometry(listTemplates) == self.shapes


_list_memo = None


def _remember_list(listTemplates, ishape):
    global _list_memo
    try:
        _list_memo = _ListMemo(listTemplates, ishape)
    except Exception:  # noqa: BLE001 - exotic template objects: no memo, every call checks
        _list_memo = None


_U8 = np.dtype(np.uint8)


def _raw_matches(listTemplates, image, method, N_object, score_threshold, context=None, devices=None, nms=None):
    """Batched equivalent of one _multi_compute per template (MTM/__init__.py:179-244).
    Returns a structured array of hits (template index, box relative to `image`, score) ordered by
    template index, then descending quality, then row-major position.
    ``nms = [maxOverlap]`` (matchTemplates, N_object != 1, method != TM_SQDIFF): where the engine can run the non-maxima
    suppression inside the same native call (one GPU context, the usual 8-bit path) the hits returned are already the
    kept ones in NMS order, and ``nms`` is emptied to say so."""
    mode = _lib.PEAKS_GLOBAL if N_object == 1 else _lib.PEAKS_LOCAL
    ichans = image.shape[2] if image.ndim == 3 else 1
    # the usual call - 8-bit image, 8-bit templates - needs no pixel policy at all: one pass over the list
    units = None
    memo = _list_memo
    ukey = (method in (0, 3), image.ndim, ichans)
    ignored_masks = 0
    pack = memo.units_pack if memo is not None else None
    if image.dtype == _U8 and pack is not None and pack[0] == ukey and memo.matches(listTemplates):
        units, ignored_masks = pack[1], pack[2]                      # the same tuples as last time: the same units
    elif image.dtype == _U8 and image.ndim in (2, 3) and ichans <= 4:
        units = []
        ignored_masks = 0           # warnings only once the list is known to take this path (the general loop below warns itself)
        for tempTuple in listTemplates:
            t = tempTuple[1]
            mask = None
            if len(tempTuple) >= 3:
                if method in (0, 3):
                    mask = tempTuple[2]
                    if not (mask.shape == t.shape and mask.dtype == _U8):
                        units = None
                        break
                else:
                    ignored_masks += 1
            if t.dtype != _U8 or t.ndim != image.ndim or (t.ndim == 3 and t.shape[2] != ichans):
                units = None
                break
            units.append((t, mask))
        if units is not None and memo is not None and memo.matches(listTemplates):
            units = _lib.FrozenUnits(units)
            memo.units_pack = (ukey, units, ignored_masks)
    if units is not None:
        for _ in range(ignored_masks):      # one per template, as the reference's loop emits them (MTM/__init__.py:216-221)
            warnings.warn(_MSG_MASK_UNSUPPORTED)
        engine = context or _lib.engine_for(devices)
        with engine.lock:
            if nms and hasattr(engine, "search_nms"):
                n_obj = -1 if N_object == float("inf") else int(N_object)
                hits = engine.search_nms(units, image, method, score_threshold, nms[0], n_obj)
                del nms[:]
                return hits
            return engine.search(units, image, method, mode, score_threshold)

    # general case: the pixel policy is per template (MTM/__init__.py:67-88)
    units = []       # (index, template, image-as-matched, mask)
    for index, tempTuple in enumerate(listTemplates):
        template = tempTuple[1]
        mask = None
        if len(tempTuple) >= 3:
            if method in (0, 3):
                mask = tempTuple[2]
            else:
                warnings.warn(_MSG_MASK_UNSUPPORTED)
        t, im, m = _apply_pixel_policy(template, image, method, mask)
        _check_opencv_preconditions(t, im)
        units.append((index, t, im, m))
    parts = []
    engine = context or _lib.engine_for(devices)      # only now: argument errors come before "no GPU"
    with engine.lock:
        # group the units by the dtype their match runs in
        for code in ("uint8", "uint16", "float32"):
            group = [u for u in units if u[2].dtype == code]
            if not group:
                continue
            hits = engine.search([(u[1], u[3]) for u in group], group[0][2], method, mode, score_threshold)
            if len(group) != len(units):
                hits = hits.copy()
                hits["templ_idx"] = np.asarray([u[0] for u in group], dtype=np.int32)[hits["templ_idx"]]
            parts.append(hits)
    if not parts:
        return np.zeros(0, dtype=_lib.HIT_DTYPE)
    hits = parts[0] if len(parts) == 1 else np.concatenate(parts)
    if len(parts) > 1:
        hits = hits[np.argsort(hits["templ_idx"], kind="stable")]
    return hits


def _nms_raw(raw, scoreThreshold, sortAscending, N_object, maxOverlap):
    """MTM.NMS.NMS (reference MTM/NMS.py:53-84) evaluated on the raw hit array: same decisions on the
    same float32 scores, but Python tuples are only built for the hits that survive."""
    if len(raw) <= 1:
        return raw
    if N_object == 1:
        # python max() / min() as in the reference (MTM/NMS.py:61-69): the first best hit wins ties, and a NaN score
        # (masked TM_CCORR_NORMED is 0/0 over a window that is zero under the mask, as in OpenCV) is only ever selected
        # when it comes FIRST - every comparison with it is False.  numpy's argmax / argmin would return the first NaN.
        sc = raw["score"]
        if sc[0] != sc[0]:
            return raw[0:1]
        i = int(np.nanargmin(sc)) if sortAscending else int(np.nanargmax(sc))
        return raw[i:i + 1]
    # the 1 - score transform of sortAscending (float32 scores, python-float threshold) is done by mtm_nms
    idx = _lib.nms_hits(raw, scoreThreshold, maxOverlap, ascending=sortAscending)
    if N_object != float("inf"):
        idx = idx[:N_object]
    return raw[idx]


def _to_hit_list(raw, listTemplates, xOffset, yOffset):
    """Structured hit array -> the reference's list of (label, (x, y, w, h), np.float32 score)
    (MTM/__init__.py:241).  Column-wise tolist() keeps this cheap for thousands of hits."""
    memo = _list_memo
    if memo is not None and memo.labels is not None and memo.matches(listTemplates):
        labels = memo.labels
    else:
        labels = np.empty(len(listTemplates), dtype=object)
        for i, t in enumerate(listTemplates):          # element-wise: a label may be any object (even a tuple)
            labels[i] = t[0]
        if memo is not None and memo.matches(listTemplates):
            memo.labels = labels
    boxes = zip((raw["x"] + xOffset).tolist(), (raw["y"] + yOffset).tolist(), raw["w"].tolist(), raw["h"].tolist())
    return list(zip(labels[raw["templ_idx"]].tolist(), boxes, list(raw["score"])))


def findMatches(listTemplates: Sequence[TemplateTuple], image: np.ndarray, method: int = TM_CCOEFF_NORMED,
                N_object=float("inf"), score_threshold: float = 0.5, searchBox: Optional[BBox] = None,
                *, devices=None) -> List[Hit]:
    """
    All template locations satisfying the score threshold, before Non-Maxima Suppression
    (reference MTM/__init__.py:95-177).

    - listTemplates  : list of tuples (label, template[, mask]); template grayscale or RGB numpy
                       array, mask of the template's shape and dtype (methods 0 and 3 only)
    - image          : grayscale or RGB numpy array of the same bit depth and channel count
    - method         : one of the OpenCV template matching methods 0..5 (default 5)
    - N_object       : int or float("inf"); 1 returns the global extremum of every template
    - score_threshold: local maxima above it (minima below it for methods 0/1) are returned
    - searchBox      : optional (x, y, width, height) region of the image to search
    - devices        : (not in the reference) GPUs to search on: None = the default GPU, ``"all"``, a list of
                       device ids or ``"0,1,2"``; also the MTM_DEVICES environment variable.  With several
                       devices the templates are sharded over them (the reference's thread pool over templates,
                       MTM/__init__.py:172-175, with GPUs as workers); the result does not depend on it.

    Returns a list of hits ``(label, (x, y, width, height), score)``.  Hits come grouped by
    template in list order, each group by descending quality (the reference's cross-template
    order is the completion order of its worker threads).
    """
    image, xOffset, yOffset = _validate_search(listTemplates, image, N_object, searchBox)
    raw = _raw_matches(listTemplates, image, method, N_object, score_threshold, devices=devices)
    return _to_hit_list(raw, listTemplates, xOffset, yOffset)


def matchTemplates(listTemplates: List[TemplateTuple], image: np.ndarray, method: int = TM_CCOEFF_NORMED,
                   N_object=float("inf"), score_threshold: float = 0.5, maxOverlap: float = 0.25,
                   searchBox: Optional[BBox] = None, *, devices=None) -> List[Hit]:
    """
    Search each template in the image and return the best N_object locations that do not overlap
    more than maxOverlap (reference MTM/__init__.py:247-296).

    - method     : 1..5 (0/TM_SQDIFF is rejected: no NMS for an unbounded difference score)
    - maxOverlap : float in [0, 1], maximal Intersection-over-Union between two returned boxes
    Other arguments as in findMatches (``devices``: which GPUs; the result does not depend on it).

    Returns a list of hits ``(label, (x, y, width, height), score)``:
        N_object == 1   -> the best match, whatever its score
        N_object < inf  -> up to N_object best matches that passed the NMS
        N_object == inf -> every match that passed the NMS
    """
    if maxOverlap < 0 or maxOverlap > 1:
        raise ValueError("Maximal overlap between bounding box is in range [0-1]")

    image_s, xOffset, yOffset = _validate_search(listTemplates, image, N_object, searchBox)
    nms = [maxOverlap] if (N_object != 1 and method != 0) else None      # (N_object == 1: no suppression, the best hit)
    raw = _raw_matches(listTemplates, image_s, method, N_object, score_threshold, devices=devices, nms=nms)

    if method == 0:     # as in the reference, only after the search ran (MTM/__init__.py:291)
        raise ValueError("The method TM_SQDIFF is not supported. Use TM_SQDIFF_NORMED instead.")

    sortAscending = (method == 1)
    kept = raw if nms == [] else _nms_raw(raw, score_threshold, sortAscending, N_object, maxOverlap)
    return _to_hit_list(kept, listTemplates, xOffset, yOffset)


class TemplateMatcher:
    """
    The same templates over a stream of images (the "thousands of images" use of the reference,
    tutorials/Tutorial3-SpeedingUp.ipynb): templates are validated, packed and uploaded ONCE and stay
    resident on the GPU; every ``match(image)`` uploads one image and runs one batched search.
    ``matcher.match(image)`` returns exactly what ``matchTemplates(listTemplates, image, ...)`` with
    the constructor's arguments returns.

    All images must have the dtype the templates were prepared for (uint8 templates with uint8
    images, anything else float32) - mixing raises, use matchTemplates for one-off calls.
    """

    def __init__(self, listTemplates, method: int = TM_CCOEFF_NORMED, N_object=float("inf"),
                 score_threshold: float = 0.5, maxOverlap: float = 0.25, context=None):
        if maxOverlap < 0 or maxOverlap > 1:
            raise ValueError("Maximal overlap between bounding box is in range [0-1]")
        if N_object != float("inf") and not isinstance(N_object, int):
            raise TypeError("N_object must be an integer")
        self.listTemplates = list(listTemplates)
        self.method, self.N_object = method, N_object
        self.score_threshold, self.maxOverlap = score_threshold, maxOverlap
        self._ctx = context or _lib.Context()
        self._uploaded_for = None      # (dtype name, channel count) the resident templates were prepared for
        self._streaming = False        # a match_stream generator is being consumed: the context is its alone

    def _upload(self, image):
        units = []
        for tempTuple in self.listTemplates:
            mask = None
            if len(tempTuple) >= 3:
                if self.method in (0, 3):
                    mask = tempTuple[2]
                else:
                    warnings.warn(_MSG_MASK_UNSUPPORTED)
            t, im, m = _apply_pixel_policy(tempTuple[1], image, self.method, mask)
            _check_opencv_preconditions(t, im)
            units.append((t, m, im.dtype))
        kinds = {str(u[2]) for u in units}
        if len(kinds) > 1:
            raise ValueError("TemplateMatcher needs templates of one pixel type (all uint8, all uint16, or neither)")
        self._ctx.set_templates([(u[0], u[1]) for u in units], self.method)
        self._uploaded_for = (kinds.pop() if kinds else str(image.dtype), 1 if image.ndim == 2 else image.shape[2])

    def _prepare(self, image, searchBox):
        """Validation + pixel policy of one image (same order as findMatches); returns the array to
        upload and the searchBox offsets.  Caller holds the context lock."""
        image_s, xOffset, yOffset = _validate_search(self.listTemplates, image, self.N_object, searchBox)
        if self._uploaded_for is None:
            self._upload(image_s)
        native = self._uploaded_for[0]          # "uint8" | "uint16" (pixels taken as they are) | "float32"
        want = native if (native != "float32" and str(image_s.dtype) == native) else "float32"
        if image_s.dtype == "float64":
            raise ValueError("64-bit images not supported, max 32-bit")
        if want != native or (1 if image_s.ndim == 2 else image_s.shape[2]) != self._uploaded_for[1]:
            raise ValueError("TemplateMatcher: image pixel type / channel count differs from the resident templates")
        return (image_s if want != "float32" else np.float32(image_s)), xOffset, yOffset

    def _finish(self, raw, xOffset, yOffset):
        if self.method == 0:
            raise ValueError("The method TM_SQDIFF is not supported. Use TM_SQDIFF_NORMED instead.")
        kept = _nms_raw(raw, self.score_threshold, self.method == 1, self.N_object, self.maxOverlap)
        return _to_hit_list(kept, self.listTemplates, xOffset, yOffset)

    def _not_streaming(self):
        if self._streaming:
            raise RuntimeError("TemplateMatcher: a match_stream() generator of this matcher is still being consumed; "
                               "finish or close() it before calling match() / match_stream() again")

    def match(self, image: np.ndarray, searchBox: Optional[BBox] = None) -> List[Hit]:
        self._not_streaming()
        with self._ctx.lock:
            im, xOffset, yOffset = self._prepare(image, searchBox)
            mode = _lib.PEAKS_GLOBAL if self.N_object == 1 else _lib.PEAKS_LOCAL
            raw = self._ctx.find_matches_image(im, mode, self.score_threshold)
        return self._finish(raw, xOffset, yOffset)

    def match_stream(self, images, searchBox: Optional[BBox] = None):
        """
        Generator over an iterable of images: yields ``match(image)`` for each, in order.  The upload
        of image i+1 (PCIe transfer, plane conversion) is enqueued on a
        second stream while the kernels of image i run (mtm_find_matches_next), so a stream costs
        about the kernel time per image instead of upload + kernels.  The context stays locked while
        the generator is being consumed, and the matcher refuses other calls meanwhile - also from the
        consuming thread, whose loop body runs between two yields while a native call is in flight on the
        helper thread: finish the generator or close() it first.
        """
        self._not_streaming()
        mode = _lib.PEAKS_GLOBAL if self.N_object == 1 else _lib.PEAKS_LOCAL
        it = iter(images)
        try:
            first = next(it)
        except StopIteration:
            return
        # The native call of image i runs on a helper thread (ctypes releases the GIL) while this thread
        # builds the hit list of image i-1 and the consumer works on it: neither the Python host layer nor
        # the caller's own per-image code leaves the GPU idle.
        from concurrent.futures import ThreadPoolExecutor
        with self._ctx.lock, ThreadPoolExecutor(max_workers=1, thread_name_prefix="mtm-stream") as pool:
            pending = None          # (future of the raw hits, xOffset, yOffset) of the image in flight
            self._streaming = True
            try:
                cur = self._prepare(first, searchBox)
                self._ctx.set_image(cur[0])
                for nxt_image in it:
                    nxt = self._prepare(nxt_image, searchBox)
                    fut = pool.submit(self._ctx.find_matches, mode, self.score_threshold, nxt[0])
                    if pending is not None:
                        done, pending = pending, None
                        pending = (fut, cur[1], cur[2])
                        yield self._finish(done[0].result().copy(), done[1], done[2])
                    else:
                        pending = (fut, cur[1], cur[2])
                    cur = nxt
                fut = pool.submit(self._ctx.find_matches, mode, self.score_threshold)
                if pending is not None:
                    done, pending = pending, (fut, cur[1], cur[2])
                    yield self._finish(done[0].result().copy(), done[1], done[2])
                else:
                    pending = (fut, cur[1], cur[2])
                done, pending = pending, None
                yield self._finish(done[0].result().copy(), done[1], done[2])
            finally:
                if pending is not None:      # consumer stopped early: let the call in flight finish
                    try:
                        pending[0].result()
                    except Exception:        # noqa: BLE001 - nothing to report to: the generator is closing
                        pass
                self._streaming = False


# ---------------------------------------------------------------------------------------------
# drawing helpers (reference MTM/__init__.py:299-391), numpy only
# ---------------------------------------------------------------------------------------------
# Label text: cv2.putText(FONT_HERSHEY_SIMPLEX, LINE_AA) in the reference (MTM/__init__.py:335, :383).  The Hershey
# stroke tables are OpenCV data; labels are drawn here with the classic 5x7 dot-matrix font (columns, LSB = top
# row, ASCII 32..126), scaled to the height Hershey simplex has at the same fontScale, anchored like putText
# (org = bottom-left corner of the string).  Same place, same size, not the same glyph shapes.
_FONT_5X7 = bytes.fromhex(
    "0000000000" "00005f0000" "0007000700" "147f147f14" "242a7f2a12" "2313086462" "3649552250" "0005030000"
    "001c224100" "0041221c00" "14083e0814" "08083e0808" "0050300000" "0808080808" "0060600000" "2010080402"
    "3e5149453e" "00427f4000" "4261514946" "2141454b31" "1814127f10" "2745454539" "3c4a494930" "0171090503"
    "3649494936" "064949291e" "0036360000" "0056360000" "0814224100" "1414141414" "0041221408" "0201510906"
    "324979413e" "7e1111117e" "7f49494936" "3e41414122" "7f4141221c" "7f49494941" "7f09090901" "3e4149497a"
    "7f0808087f" "00417f4100" "2040413f01" "7f08142241" "7f40404040" "7f020c027f" "7f0408107f" "3e4141413e"
    "7f09090906" "3e4151215e" "7f09192946" "4649494931" "01017f0101" "3f4040403f" "1f2040201f" "3f4038403f"
    "6314081463" "0708700807" "6151494543" "007f414100" "0204081020" "0041417f00" "0402010204" "4040404040"
    "0001020400" "2054545478" "7f48444438" "3844444420" "384444487f" "3854545418" "087e090102" "0c5252523e"
    "7f08040478" "00447d4000" "2040443d00" "7f10284400" "00417f4000" "7c04180478" "7c08040478" "3844444438"
    "7c14141408" "081414187c" "7c08040408" "4854545420" "043f444020" "3c4040207c" "1c2040201c" "3c4030403c"
    "4428102844" "0c5050503c" "4464544c44" "0008364100" "00007f0000" "0041360800" "0804081008")


def _text_mask(text, scale):
    """Boolean (rows, cols) raster of `text` in the 5x7 font, every dot a k x k block; k follows the font scale
    (Hershey simplex is ~22 px tall at fontScale 1: k = round(22 * scale / 7), at least 1)."""
    k = max(1, int(round(22.0 * float(scale) / 7.0)))
    cols = []
    for ch in str(text):
        o = ord(ch)
        g = _FONT_5X7[(o - 32) * 5:(o - 32) * 5 + 5] if 32 <= o <= 126 else b"\x7f\x41\x41\x41\x7f"   # box for non-ASCII
        cols += [[(byte >> r) & 1 for r in range(7)] for byte in g] + [[0] * 7]
    if not cols:
        return np.zeros((0, 0), bool)
    m = np.array(cols[:-1], dtype=bool).T                    # 7 x (6 n - 1)
    return np.repeat(np.repeat(m, k, axis=0), k, axis=1)


def _draw_label(canvas, text, x, y, color, scale):
    m = _text_mask(text, scale)
    th, tw = m.shape
    H, W = canvas.shape[:2]
    y0, x0 = int(y) - th, int(x)                             # putText: org is the bottom-left corner of the text
    ya, yb, xa, xb = max(y0, 0), min(y0 + th, H), max(x0, 0), min(x0 + tw, W)
    if ya < yb and xa < xb:
        canvas[ya:yb, xa:xb][m[ya - y0:yb - y0, xa - x0:xb - x0]] = color


def _draw_boxes(canvas, listHit, boxThickness, color, showLabel=False, labelColor=None, labelScale=0.5):
    H, W = canvas.shape[:2]
    t = max(int(boxThickness), 1)
    # cv2.rectangle draws a line of thickness t centred on the box outline
    lo, hi = (t - 1) // 2, t // 2
    for label, (x, y, w, h), _ in listHit:
        x0, y0, x1, y1 = int(x), int(y), int(x + w), int(y + h)
        for (ya, yb, xa, xb) in ((y0 - lo, y0 + hi + 1, x0 - lo, x1 + hi + 1), (y1 - lo, y1 + hi + 1, x0 - lo, x1 + hi + 1),
                                 (y0 - lo, y1 + hi + 1, x0 - lo, x0 + hi + 1), (y0 - lo, y1 + hi + 1, x1 - lo, x1 + hi + 1)):
            ya, yb, xa, xb = max(ya, 0), min(yb, H), max(xa, 0), min(xb, W)
            if ya < yb and xa < xb:
                canvas[ya:yb, xa:xb] = color
        if showLabel:
            _draw_label(canvas, label, x0, y0, labelColor, labelScale)
    return canvas


def _rgb_to_gray(image):
    """cv2.cvtColor(image, cv2.COLOR_RGB2GRAY) (reference MTM/__init__.py:375): OpenCV's fixed-point weights for
    integer pixels - 8 bit: (9798 R + 19235 G + 3735 B + 2^14) >> 15, 16 bit: (4899 R + 9617 G + 1868 B + 2^13) >> 14 -
    and float32 weights 0.299 / 0.587 / 0.114 for float images (OpenCV imgproc color.hpp, not pinned: no cv2 here)."""
    r, g, b = image[..., 0], image[..., 1], image[..., 2]
    if image.dtype == np.uint8:
        v = (9798 * r.astype(np.int64) + 19235 * g.astype(np.int64) + 3735 * b.astype(np.int64) + (1 << 14)) >> 15
        return v.astype(np.uint8)
    if image.dtype == np.uint16:
        v = (4899 * r.astype(np.int64) + 9617 * g.astype(np.int64) + 1868 * b.astype(np.int64) + (1 << 13)) >> 14
        return v.astype(np.uint16)
    f = image.astype(np.float32, copy=False)
    return (f[..., 0] * np.float32(0.299) + f[..., 1] * np.float32(0.587) + f[..., 2] * np.float32(0.114)).astype(image.dtype)


def drawBoxesOnRGB(image: np.ndarray, listHit: Sequence[Hit], boxThickness: int = 2,
                   boxColor: Tuple[int, int, int] = (255, 255, 00), showLabel: bool = False,
                   labelColor=(255, 255, 0), labelScale=0.5) -> np.ndarray:
    """Return an RGB copy of the image with the hit boxes - and, with showLabel, the template names at the top
    left corner of each box - drawn on it (reference MTM/__init__.py:299-343).  Labels use a built-in dot-matrix
    font instead of OpenCV's Hershey strokes: same anchor and size, different glyph shapes."""
    if image.ndim == 2:
        out = np.stack([image] * 3, axis=2)
    else:
        out = image.copy()
    return _draw_boxes(out, listHit, boxThickness, np.asarray(boxColor, dtype=out.dtype), showLabel,
                       np.asarray(labelColor, dtype=out.dtype), labelScale)


def drawBoxesOnGray(image: np.ndarray, listHit: Sequence[Hit], boxThickness: int = 2, boxColor: int = 255,
                    showLabel: bool = False, labelColor: int = 255, labelScale=0.5) -> np.ndarray:
    """Return a grayscale copy of the image with the hit boxes (and labels) drawn on it
    (reference MTM/__init__.py:346-391)."""
    out = _rgb_to_gray(image) if image.ndim == 3 else image.copy()
    return _draw_boxes(out, listHit, boxThickness, boxColor, showLabel, labelColor, labelScale)


from . import augment  # noqa: E402,F401  (template augmentation / downscaled matching helpers)
