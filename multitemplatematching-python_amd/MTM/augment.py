"""
The step BEFORE the hot path in the reference's tutorials, without OpenCV: template augmentation
(tutorials/Tutorial2-Template_Augmentation.ipynb:313, np.rot90 / np.fliplr / np.flipud copies of a
template under derived labels), multi-scale copies, and the speed-up recipe of
tutorials/Tutorial3-SpeedingUp.ipynb:395-470 (downscale image and template with INTER_AREA, match,
scale the boxes back up).

Two ways to use it
  * the tutorial's way: ``rotations`` / ``flips`` / ``scales`` / ``expand`` build the augmented copies on the host
    and the caller passes the longer list to ``MTM.matchTemplates`` (works for every pixel type);
  * ``matchTemplatesAugmented(bases, variants(...), image, ...)``: the caller passes the BASES and a list of
    variants; the copies are never built on the host - every (base, variant) pair is a view of a source kept on
    the GPU (the base, or an area-resized copy a kernel makes of it) and the operand packs of the score kernel are
    gathered from those views (mtm_set_templates_augmented).  Same hits, labels and order as
    ``matchTemplates(expand(bases, variants), image, ...)``.

The search image can be downscaled ON THE DEVICE while it is laid out (mtm_set_image_downscaled): the full
resolution image crosses PCIe once and no host resize runs.

Resize rules.  ``downscale`` (integer factors) follows OpenCV's integer-factor INTER_AREA path: uint8 / uint16 ->
factor 2: (sum + 2) >> 2, otherwise rint(float32(sum) * float32(1 / factor**2)); float32 -> float32 row-major block
sum times float32(1 / factor**2).  ``resize_area`` (any target size, uint8) is the exact area average rounded half
up, in integer arithmetic.  OpenCV is absent from this image, so neither is pinned against cv2.resize itself: host,
device and oracle agree with each other byte for byte.
"""
from typing import List, Sequence

import numpy as np

from . import _lib
from . import (TM_CCOEFF_NORMED, _apply_pixel_policy, _check_opencv_preconditions, _nms_raw, _to_hit_list,
               _validate_search, _MSG_MASK_UNSUPPORTED)

import warnings


def rotations(listTemplates: Sequence[tuple], angles: Sequence[int] = (0, 90, 180, 270)) -> List[tuple]:
    """Every template (and mask) rotated by the given multiples of 90 degrees (np.rot90, counter-clockwise,
    as in the tutorial).  Labels become "<label>_<angle>"."""
    out = []
    for tup in listTemplates:
        for angle in angles:
            if angle % 90:
                raise ValueError("rotations: angles must be multiples of 90")
            k = (angle // 90) % 4
            out.append((f"{tup[0]}_{angle}",) + tuple(np.ascontiguousarray(np.rot90(a, k=k)) for a in tup[1:]))
    return out


def flips(listTemplates: Sequence[tuple], horizontal: bool = True, vertical: bool = True) -> List[tuple]:
    """The original plus its left-right ("<label>_lr") and/or up-down ("<label>_ud") mirror images."""
    out = []
    for tup in listTemplates:
        out.append(tuple(tup))
        if horizontal:
            out.append((f"{tup[0]}_lr",) + tuple(np.ascontiguousarray(np.fliplr(a)) for a in tup[1:]))
        if vertical:
            out.append((f"{tup[0]}_ud",) + tuple(np.ascontiguousarray(np.flipud(a)) for a in tup[1:]))
    return out


def downscale(image: np.ndarray, factor: int) -> np.ndarray:
    """Integer-factor area downscale (rows//factor x cols//factor; remainder rows / columns dropped).
    Same arithmetic as the device kernel (see the module docstring)."""
    factor = int(factor)
    if factor < 1:
        raise ValueError("downscale: factor must be a positive integer")
    if factor == 1:
        return image
    if image.dtype == np.float64:
        raise ValueError("64-bit images not supported, max 32-bit")
    r, c = image.shape[0] // factor, image.shape[1] // factor
    if r < 1 or c < 1:
        raise ValueError("downscale: factor larger than the image")
    a = image[:r * factor, :c * factor]
    blocks = a.reshape((r, factor, c, factor) + a.shape[2:])
    scale = np.float32(1.0) / np.float32(factor * factor)
    if a.dtype == np.uint8 or a.dtype == np.uint16:
        s = blocks.sum(axis=(1, 3), dtype=np.uint32)
        if factor == 2:
            return ((s + 2) >> 2).astype(a.dtype)
        return np.minimum(np.rint(s.astype(np.float32) * scale), np.iinfo(a.dtype).max).astype(a.dtype)
    acc = np.zeros((r, c) + a.shape[2:], np.float32)
    src = blocks.astype(np.float32, copy=False)
    for dy in range(factor):                 # float32 accumulation in row-major order, like the kernel
        for dx in range(factor):
            acc += src[:, dy, :, dx]
    return acc * scale


def resize_area(a: np.ndarray, rows: int, cols: int) -> np.ndarray:
    """Area-average resize of a uint8 (rows, cols[, C]) array to rows x cols, exact: output pixel (i, j) integrates
    the source over [i*H/rows, (i+1)*H/rows) x [j*W/cols, (j+1)*W/cols); the overlap lengths are integers in units
    of 1/rows (1/cols) source pixels and the mean is rounded half up, (2*num + H*W) // (2*H*W).  The device kernel
    (resize_area_kernel) does the same integer arithmetic."""
    if a.dtype != np.uint8:
        raise ValueError("resize_area takes uint8 arrays")
    H, W = a.shape[:2]
    rows, cols = int(rows), int(cols)
    if rows < 1 or cols < 1:
        raise ValueError("resize_area: the target size must be at least 1 x 1")

    def weights(n_out, n_in):
        w = np.zeros((n_out, n_in), dtype=np.int64)          # overlap lengths * n_out; every row sums to n_in
        for i in range(n_out):
            lo, hi = i * n_in, (i + 1) * n_in
            for j in range(lo // n_out, min((hi + n_out - 1) // n_out, n_in)):
                w[i, j] = max(0, min(hi, (j + 1) * n_out) - max(lo, j * n_out))
        return w
    wy, wx = weights(rows, H), weights(cols, W)
    den = H * W
    planes = a.reshape(H, W, -1).astype(np.int64)
    out = np.empty((rows, cols, planes.shape[2]), dtype=np.uint8)
    for c in range(planes.shape[2]):
        num = wy @ planes[:, :, c] @ wx.T                      # exact: < 2^63
        out[:, :, c] = np.clip((2 * num + den) // (2 * den), 0, 255).astype(np.uint8)
    return out.reshape((rows, cols) + a.shape[2:])


def variants(angles: Sequence[int] = (0,), flip_lr: bool = False, flip_ud: bool = False, sizes=None, factors=None):
    """The augmentation spec of matchTemplatesAugmented / expand: a list of (label suffix, record) pairs, one per
    copy made of every base, in this order: for every size (``sizes``: target (rows, cols) pairs or square sides for
    resize_area; ``factors``: integer downscale factors; neither: the base as it is) the unflipped copy, then the
    left-right mirror (flip_lr), then the up-down mirror (flip_ud), each under every angle (multiples of 90 degrees,
    np.rot90, counter-clockwise).  Suffixes follow the tutorial: "<label>_<angle>", with "_s<rows>x<cols>" /
    "_d<factor>" / "_lr" / "_ud" in front where they apply; the angle is omitted when ``angles`` is just (0,)."""
    if sizes is not None and factors is not None:
        raise ValueError("variants: give target sizes or integer factors, not both")
    resizes = [("", 0, 0, 0)]
    if sizes is not None:
        resizes = []
        for sz in sizes:
            r, c = (int(sz), int(sz)) if np.isscalar(sz) else (int(sz[0]), int(sz[1]))
            resizes.append(("_s%dx%d" % (r, c), r, c, 0))
    if factors is not None:
        resizes = [("_d%d" % int(f), 0, 0, int(f)) if int(f) > 1 else ("_d1", 0, 0, 0) for f in factors]
    angles = tuple(int(a) for a in angles)
    if any(a % 90 for a in angles):
        raise ValueError("variants: angles must be multiples of 90")
    flips_ = [("", 0, 0)] + ([("_lr", 1, 0)] if flip_lr else []) + ([("_ud", 0, 1)] if flip_ud else [])
    out = []
    for rs, r, c, d in resizes:
        for fs, lr, ud in flips_:
            for a in angles:
                suffix = rs + fs + ("" if angles == (0,) else "_%d" % a)
                out.append((suffix, (a // 90 % 4, lr, ud, r, c, d)))
    return out


def _apply_variant(a: np.ndarray, rec) -> np.ndarray:
    k, lr, ud, r, c, d = rec
    if r > 0:
        a = resize_area(a, r, c)
    elif d > 1:
        a = downscale(a, d)
    if lr:
        a = np.fliplr(a)
    if ud:
        a = np.flipud(a)
    return np.ascontiguousarray(np.rot90(a, k=k))


def expand(listTemplates: Sequence[tuple], spec) -> List[tuple]:
    """The augmented template list built on the host: every base under every variant of ``spec`` (see ``variants``),
    base-major - what matchTemplatesAugmented matches without building it."""
    return [(f"{tup[0]}{suffix}",) + tuple(_apply_variant(a, rec) for a in tup[1:]) for tup in listTemplates
            for suffix, rec in spec]


def matchTemplatesAugmented(listTemplates, spec, image: np.ndarray, method: int = TM_CCOEFF_NORMED,
                            N_object=float("inf"), score_threshold: float = 0.5, maxOverlap: float = 0.25,
                            searchBox=None, context=None):
    """
    ``matchTemplates(expand(listTemplates, spec), image, ...)`` without building the copies: the bases and the
    variant list go to the GPU, where every copy is a view of a device-resident source and the score kernel's
    operands are gathered from the views (mtm_set_templates_augmented).  uint8 bases and image (with uint8 masks
    for methods 0 / 3); anything else is expanded on the host and matched the ordinary way - same result.
    """
    from . import matchTemplates
    if maxOverlap < 0 or maxOverlap > 1:
        raise ValueError("Maximal overlap between bounding box is in range [0-1]")
    spec = list(spec)
    device_ok = image.dtype == np.uint8 and len(spec) > 0 and all(
        isinstance(t, tuple) and len(t) >= 2 and t[1].dtype == np.uint8 and t[1].ndim == image.ndim and
        (len(t) < 3 or (t[2].dtype == np.uint8 and t[2].shape == t[1].shape)) for t in listTemplates)
    if not device_ok:
        return matchTemplates(expand(listTemplates, spec), image, method, N_object, score_threshold, maxOverlap, searchBox)

    def unit_shape(shape, rec):
        k, _lr, _ud, r, c, d = rec
        h, w = (r, c) if r > 0 else ((shape[0] // d, shape[1] // d) if d > 1 else shape[:2])
        return ((w, h) if k % 2 else (h, w)) + tuple(shape[2:])
    labels = [(f"{t[0]}{suffix}", unit_shape(t[1].shape, rec)) for t in listTemplates for suffix, rec in spec]
    shapes_only = [(lab, np.empty(shp, np.uint8)) for lab, shp in labels]          # validation looks at shapes
    image_s, xOffset, yOffset = _validate_search(shapes_only, image, N_object, searchBox)
    use_mask = method in (0, 3)
    if not use_mask and any(len(t) >= 3 for t in listTemplates):
        warnings.warn(_MSG_MASK_UNSUPPORTED)
    bases = [(t[1], t[2] if (use_mask and len(t) >= 3) else None) for t in listTemplates]
    ctx = context or _lib.default_context()
    mode = _lib.PEAKS_GLOBAL if N_object == 1 else _lib.PEAKS_LOCAL
    with ctx.lock:
        ctx.set_templates_augmented(bases, [rec for _, rec in spec], method)
        raw = ctx.find_matches_image(image_s, mode, score_threshold)
    if method == 0:
        raise ValueError("The method TM_SQDIFF is not supported. Use TM_SQDIFF_NORMED instead.")
    kept = _nms_raw(raw, score_threshold, method == 1, N_object, maxOverlap)
    return _to_hit_list(kept, shapes_only, xOffset, yOffset)


def scales(listTemplates: Sequence[tuple], factors: Sequence[int]) -> List[tuple]:
    """Each template (and mask) downscaled by each integer factor; labels "<label>_d<factor>"."""
    return [(f"{tup[0]}_d{f}",) + tuple(downscale(a, f) for a in tup[1:]) for tup in listTemplates for f in factors]


def upscale_hits(listHits, factor: int):
    """Boxes found on a downscaled image, in full-resolution coordinates (x, y, w, h all * factor)."""
    return [(label, (x * factor, y * factor, w * factor, h * factor), score) for label, (x, y, w, h), score in listHits]


def matchTemplatesDownscaled(listTemplates, image: np.ndarray, factor: int, method: int = TM_CCOEFF_NORMED,
                             N_object=float("inf"), score_threshold: float = 0.5, maxOverlap: float = 0.25,
                             context=None):
    """
    matchTemplates on image and templates downscaled by `factor`, boxes returned in full-resolution
    coordinates: the reference's speed-up recipe (Tutorial3-SpeedingUp) in one call.  Equivalent to
        small = [(n, downscale(t, factor)) for n, t in listTemplates]
        upscale_hits(matchTemplates(small, downscale(image, factor), ...), factor)
    except that the image is downscaled on the GPU.
    """
    if maxOverlap < 0 or maxOverlap > 1:
        raise ValueError("Maximal overlap between bounding box is in range [0-1]")
    factor = int(factor)
    small = [(tup[0],) + tuple(downscale(a, factor) for a in tup[1:]) for tup in listTemplates]
    small_shape = (image.shape[0] // factor, image.shape[1] // factor) + image.shape[2:]
    if factor < 1 or min(small_shape[:2]) < 1:
        raise ValueError("downscale: factor larger than the image")
    placeholder = np.empty(small_shape, image.dtype)          # shapes only: validation, dtype policy
    _validate_search(small, placeholder, N_object, None)
    units, kinds = [], set()
    for tup in small:
        mask = None
        if len(tup) >= 3:
            if method in (0, 3):
                mask = tup[2]
            else:
                warnings.warn(_MSG_MASK_UNSUPPORTED)
        t, im, m = _apply_pixel_policy(tup[1], placeholder, method, mask)
        _check_opencv_preconditions(t, im)
        units.append((t, m))
        kinds.add(str(im.dtype))
    if len(kinds) > 1:
        raise ValueError("matchTemplatesDownscaled needs templates of one pixel type (all uint8, or none)")
    native = kinds.pop() if kinds else str(image.dtype)
    full = image if (native in ("uint8", "uint16") and str(image.dtype) == native) else np.float32(image)
    ctx = context or _lib.default_context()
    mode = _lib.PEAKS_GLOBAL if N_object == 1 else _lib.PEAKS_LOCAL
    with ctx.lock:
        ctx.set_image(full, downscale=factor)
        ctx.set_templates(units, method)
        raw = ctx.find_matches(mode, score_threshold).copy()
    if method == 0:
        raise ValueError("The method TM_SQDIFF is not supported. Use TM_SQDIFF_NORMED instead.")
    kept = _nms_raw(raw, score_threshold, method == 1, N_object, maxOverlap)
    return upscale_hits(_to_hit_list(kept, small, 0, 0), factor)
