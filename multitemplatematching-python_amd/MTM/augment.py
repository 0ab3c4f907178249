"""
The step BEFORE the hot path in the reference's tutorials, without OpenCV: template augmentation
(tutorials/Tutorial2-Template_Augmentation.ipynb:313, np.rot90 / np.fliplr / np.flipud copies of a
template under derived labels), multi-scale copies, and the speed-up recipe of
tutorials/Tutorial3-SpeedingUp.ipynb:395-470 (downscale image and template with INTER_AREA, match,
scale the boxes back up).

The image is downscaled ON THE DEVICE while it is laid out (mtm_set_image_downscaled): the full
resolution image crosses PCIe once and no host resize runs.  Templates are a few KB: they are
augmented / resized on the host (a device kernel would buy nothing) with the same arithmetic.

`downscale` follows OpenCV's integer-factor INTER_AREA path: uint8 / uint16 -> factor 2: (sum + 2) >> 2,
otherwise rint(float32(sum) * float32(1 / factor**2)); float32 -> float32 row-major block sum times
float32(1 / factor**2).  OpenCV is absent from this image, so this restatement is not pinned against
cv2.resize itself.
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
