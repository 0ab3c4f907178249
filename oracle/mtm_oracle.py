"""
CPU oracle for the MTM hot path (TEST INFRASTRUCTURE - not part of the product).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product package (``multitemplatematching-python_amd/MTM``) never does: it
fails loudly when its HIP library is missing.

What it restates, and from where (all file:line citations are into /root/reference):

* ``match_template``        cv2.matchTemplate, the call at MTM/__init__.py:92.  The arithmetic is
                            third-party (OpenCV ``imgproc/templmatch.cpp``, opencv-python-headless
                            >= 4.5.4 per setup.py:23; notebooks ran 4.7.0) and is NOT under
                            /root/reference: it is restated here from OpenCV's published algorithm
                            (``common_matchTemplate`` / ``matchTemplateMask``), with the sliding dot
                            products computed EXACTLY (integers for uint8, float64 otherwise) where
                            OpenCV uses a block DFT (float32 DFT for uint8 input, float64 DFT for
                            float32 input).
* ``min_max_loc``           cv2.minMaxLoc, call at MTM/__init__.py:226.
* ``find_local_max/min``    MTM/__init__.py:22-53 (shape dispatch), with
                            skimage.feature.peak_local_max (0.18.3 source read in this container:
                            skimage/feature/peak.py:28-50,114-261) and scipy.signal.find_peaks
                            restated natively.
* ``nms_boxes``             cv2.dnn.NMSBoxes, call at MTM/NMS.py:78 (OpenCV ``dnn/nms.cpp`` +
                            ``nms.inl.hpp``: strict score filter, stable sort, greedy IoU).
* ``NMS``                   MTM/NMS.py:20-84.
* ``compute_score_map``     MTM/__init__.py:56-92 (dtype policy, mask policy).
* ``find_matches``          MTM/__init__.py:95-177 + _multi_compute :179-244.
* ``match_templates``       MTM/__init__.py:247-296.

Pinning: this oracle is pinned against (a) the 72 hits printed in the reference's executed
notebooks (tutorials/Tutorial1-Introduction.ipynb cell 13, tutorials/WithMask.ipynb cells 8 and 11)
and (b) fixtures produced by running the UNMODIFIED reference package in this container with this
module standing in for ``cv2`` and the real skimage 0.18.3 / scipy doing peak finding
(tests/golden/make_golden.py).  See tests/test_oracle_golden.py.

The module is written to run under both numpy 2.x (python3.10) and numpy 1.26 (python3.9).
"""
import math
import numpy as np

TM_SQDIFF, TM_SQDIFF_NORMED, TM_CCORR, TM_CCORR_NORMED, TM_CCOEFF, TM_CCOEFF_NORMED = range(6)

FLT_EPSILON = float(np.finfo(np.float32).eps)
DBL_EPSILON = float(np.finfo(np.float64).eps)

# peak_local_max border handling.  skimage <= 0.18 (the only version importable here) pads the 3x3 maximum
# filter with the constant 0 ('constant'); releases >= 0.19 replicate the edge ('nearest').  The two agree
# whenever the filtered map is >= 0 at its border (methods 2..5 with a threshold >= 0); they differ for
# _findLocalMin_ (negated map: with zero padding a border minimum is never a peak) and for negative thresholds.
# Default = current scikit-image (the reference leaves it unpinned, setup.py:24); both rules are pinned by
# fixtures: "<call>@constant" from the real 0.18.3, "<call>@nearest" from the same code with mode='nearest'
# (tests/golden/make_golden.py).
DEFAULT_PEAK_BORDER = "nearest"


# --------------------------------------------------------------------------------------------
# sliding-window sums
# --------------------------------------------------------------------------------------------
def _as3d(a):
    a = np.asarray(a)
    return a[:, :, None] if a.ndim == 2 else a


def _next_fast(n):
    m = 1
    while m < n:
        m *= 2
    # allow 3*2^k and 5*2^k sizes too (smaller pads)
    best = m
    for f in (3, 5, 9, 15):
        k = f
        while k < n:
            k *= 2
        best = min(best, k)
    return best


def corr_fft(img2d, ker2d, cache=None, cache_key=None):
    """Valid-mode cross-correlation sum_{dy,dx} img[y+dy,x+dx]*ker[dy,dx] in float64 via FFT.
    ``cache`` (a dict owned by the caller) keeps the image spectrum under ``cache_key`` so that many
    templates over one image (full-size parity tests) transform the image once."""
    H, W = img2d.shape
    h, w = ker2d.shape
    fh, fw = _next_fast(H), _next_fast(W)
    if cache is not None and cache_key in cache:
        F = cache[cache_key]
    else:
        F = np.fft.rfft2(img2d.astype(np.float64), s=(fh, fw))
        if cache is not None:
            cache[cache_key] = F
    G = np.fft.rfft2(ker2d.astype(np.float64)[::-1, ::-1], s=(fh, fw))
    full = np.fft.irfft2(F * G, s=(fh, fw))
    return full[h - 1:H, w - 1:W]


def corr_direct(img2d, ker2d, acc_dtype):
    """Same, by explicit summation over template rows (exact for integer acc_dtype)."""
    H, W = img2d.shape
    h, w = ker2d.shape
    oh, ow = H - h + 1, W - w + 1
    out = np.zeros((oh, ow), dtype=acc_dtype)
    img = img2d.astype(acc_dtype)
    ker = ker2d.astype(acc_dtype)
    for dy in range(h):
        for dx in range(w):
            k = ker[dy, dx]
            if k != 0:
                out += k * img[dy:dy + oh, dx:dx + ow]
    return out


def sliding_corr(img2d, ker2d, exact_int=False, force=None, cache=None, cache_key=None):
    """Cross-correlation of one channel.  exact_int: both operands hold integers whose products
    fit in int64 -> result is the exact integer (as float64).  FFT in float64 + rint is exact for
    the sizes used here (|error| << 0.5); small problems use the direct sum."""
    h, w = ker2d.shape
    H, W = img2d.shape
    work = (H - h + 1) * (W - w + 1) * h * w
    use_direct = (work <= 4_000_000) if force is None else (force == "direct")
    if use_direct:
        if exact_int:
            return corr_direct(img2d, ker2d, np.int64).astype(np.float64)
        return corr_direct(img2d, ker2d, np.float64)
    out = corr_fft(img2d, ker2d, cache, cache_key)
    if exact_int:
        out = np.rint(out)
    return out


def window_sums(img2d_f64, h, w):
    """Window sums through a zero-padded 2-D cumulative sum in float64, i.e. what
    cv::integral(..., CV_64F) + the 4-corner lookup of common_matchTemplate compute."""
    H, W = img2d_f64.shape
    ii = np.zeros((H + 1, W + 1), dtype=np.float64)
    ii[1:, 1:] = np.cumsum(np.cumsum(img2d_f64, axis=1), axis=0)
    # p0 - p1 - p2 + p3 in OpenCV's order
    return ii[:H - h + 1, :W - w + 1] - ii[:H - h + 1, w:] - ii[h:, :W - w + 1] + ii[h:, w:]


# --------------------------------------------------------------------------------------------
# cv2.matchTemplate
# --------------------------------------------------------------------------------------------
def _templ_mean_sdv(t3, integer):
    """cv::meanStdDev per channel: mean = s/N, sdv = sqrt(max(sq/N - mean^2, 0))."""
    n = float(t3.shape[0] * t3.shape[1])
    means, sdvs = [], []
    for c in range(t3.shape[2]):
        ch = t3[:, :, c]
        if integer:
            s = float(int(ch.astype(np.int64).sum()))
            sq = float(int((ch.astype(np.int64) ** 2).sum()))
        else:
            ch64 = ch.astype(np.float64)
            s = float(ch64.sum())
            sq = float((ch64 * ch64).sum())
        mean = s / n
        var = sq / n - mean * mean
        means.append(mean)
        sdvs.append(math.sqrt(max(var, 0.0)))
    return means, sdvs


def match_template(image, templ, method, mask=None, corr="auto", cache=None):
    """cv2.matchTemplate(image, templ, method, mask=mask) -> float32 (H-h+1, W-w+1).

    Reference call site: MTM/__init__.py:92.  image/templ: uint8 or float32, (rows, cols) or
    (rows, cols, C) with equal C.  ``corr``: "auto" | "direct" | "fft" (how the exact sliding dot
    product is evaluated; results agree).  ``cache``: a dict the caller reuses across calls on the SAME image
    (image spectra are kept in it; pure speed-up)."""
    img3 = _as3d(image)
    t3 = _as3d(templ)
    if img3.dtype != t3.dtype or img3.shape[2] != t3.shape[2]:
        raise ValueError("image and template must have the same dtype and channel count")
    if img3.dtype not in (np.uint8, np.float32):
        raise ValueError("only uint8 and float32 are supported (as in OpenCV)")
    H, W, C = img3.shape
    h, w, _ = t3.shape
    if h > H or w > W:
        raise ValueError("template larger than image")
    force = None if corr == "auto" else corr
    integer = img3.dtype == np.uint8
    if mask is not None:
        return _match_template_mask(img3, t3, method, mask, force, cache)

    # exact sliding dot product, summed over channels
    corr_map = np.zeros((H - h + 1, W - w + 1), dtype=np.float64)
    for c in range(C):
        corr_map += sliding_corr(img3[:, :, c], t3[:, :, c], exact_int=integer, force=force, cache=cache, cache_key=("I", c))
    if method == TM_CCORR:
        return corr_map.astype(np.float32)

    num_type = 0 if method in (TM_CCORR, TM_CCORR_NORMED) else (1 if method in (TM_CCOEFF, TM_CCOEFF_NORMED) else 2)
    is_normed = method in (TM_SQDIFF_NORMED, TM_CCORR_NORMED, TM_CCOEFF_NORMED)
    inv_area = 1.0 / (float(h) * float(w))

    templ_mean, templ_sdv = _templ_mean_sdv(t3, integer)
    templ_norm = 0.0
    templ_sum2 = 0.0
    if method != TM_CCOEFF:
        templ_norm = sum(s * s for s in templ_sdv)
        if templ_norm < DBL_EPSILON and method == TM_CCOEFF_NORMED:
            return np.ones(corr_map.shape, dtype=np.float32)
        templ_sum2 = templ_norm + sum(m * m for m in templ_mean)
        if num_type != 1:
            templ_mean = [0.0] * C
            templ_norm = templ_sum2
        templ_sum2 /= inv_area
        templ_norm = math.sqrt(templ_norm)
        templ_norm /= math.sqrt(inv_area)

    num = corr_map.copy()
    wnd_mean2 = np.zeros_like(num)
    wnd_sum2 = np.zeros_like(num)
    if num_type == 1:
        for c in range(C):
            t = window_sums(img3[:, :, c].astype(np.float64), h, w)
            wnd_mean2 += t * t
            num -= t * templ_mean[c]
        wnd_mean2 *= inv_area
    if is_normed or num_type == 2:
        for c in range(C):
            ch = img3[:, :, c].astype(np.float64)
            wnd_sum2 += window_sums(ch * ch, h, w)
        if num_type == 2:
            num = wnd_sum2 - 2.0 * num + templ_sum2
            num = np.maximum(num, 0.0)
    if is_normed:
        diff2 = np.maximum(wnd_sum2 - wnd_mean2, 0.0)
        small = diff2 <= np.minimum(0.5, 10.0 * FLT_EPSILON * wnd_sum2)
        t = np.where(small, 0.0, np.sqrt(diff2) * templ_norm)
        absn = np.abs(num)
        with np.errstate(divide="ignore", invalid="ignore"):
            div = num / t
        sat = np.where(num > 0, 1.0, -1.0)
        other = 1.0 if method == TM_SQDIFF_NORMED else 0.0
        num = np.where(absn < t, div, np.where(absn < t * 1.125, sat, other))
    return num.astype(np.float32)


def _match_template_mask(img3, t3, method, mask, force, cache=None):
    """cv::matchTemplateMask (OpenCV >= 4.5.4), methods TM_SQDIFF and TM_CCORR_NORMED are the only
    ones MTM lets through (MTM/__init__.py:78, :216)."""
    m3 = _as3d(mask)
    if m3.shape[:2] != t3.shape[:2]:
        raise ValueError("mask and template sizes differ")
    if m3.dtype == np.uint8:
        m = (m3 > 0).astype(np.float64)      # CV_8U masks are binary masks
        mask_int = True
    else:
        m = m3.astype(np.float64)
        mask_int = False
    C = t3.shape[2]
    if m.shape[2] == 1 and C > 1:
        m = np.repeat(m, C, axis=2)
    integer = img3.dtype == np.uint8 and mask_int
    img = img3.astype(np.float64)
    t = t3.astype(np.float64)
    m2 = m * m
    tm2 = t * m2
    templ2_mask2_sum = float(((t * m) ** 2).sum())
    H, W, _ = img.shape
    h, w, _ = t.shape
    c_i_tm2 = np.zeros((H - h + 1, W - w + 1))
    c_i2_m2 = np.zeros_like(c_i_tm2)
    for c in range(C):
        c_i_tm2 += sliding_corr(img[:, :, c], tm2[:, :, c], exact_int=integer, force=force, cache=cache, cache_key=("I", c))
        c_i2_m2 += sliding_corr(img[:, :, c] ** 2, m2[:, :, c], exact_int=integer, force=force, cache=cache, cache_key=("I2", c))
    with np.errstate(divide="ignore", invalid="ignore"):
        if method == TM_SQDIFF:
            res = -2.0 * c_i_tm2 + c_i2_m2 + templ2_mask2_sum
        elif method == TM_SQDIFF_NORMED:
            res = (-2.0 * c_i_tm2 + c_i2_m2 + templ2_mask2_sum) / np.sqrt(templ2_mask2_sum * c_i2_m2)
        elif method == TM_CCORR:
            res = c_i_tm2
        elif method == TM_CCORR_NORMED:
            res = c_i_tm2 / np.sqrt(templ2_mask2_sum * c_i2_m2)
        else:
            raise ValueError("masked TM_CCOEFF* never reaches cv2 through MTM")
    return res.astype(np.float32)


# --------------------------------------------------------------------------------------------
# cv2.minMaxLoc
# --------------------------------------------------------------------------------------------
def min_max_loc(a):
    """cv2.minMaxLoc on a 2-D array: (minVal, maxVal, (minX, minY), (maxX, maxY)); the first
    occurrence in row-major order wins ties.  Call site MTM/__init__.py:226."""
    a = np.asarray(a)
    flat = a.ravel()
    imin = int(np.argmin(flat))
    imax = int(np.argmax(flat))
    cols = a.shape[1]
    return (float(flat[imin]), float(flat[imax]), (imin % cols, imin // cols), (imax % cols, imax // cols))


# --------------------------------------------------------------------------------------------
# peaks: MTM/__init__.py:22-53
# --------------------------------------------------------------------------------------------
def find_peaks_1d(x, height):
    """scipy.signal.find_peaks(x, height=height)[0]: strict local maxima, a flat plateau yields
    its middle sample (floor), end points are never peaks, height test is >=."""
    x = np.asarray(x)
    n = x.shape[0]
    peaks = []
    i = 1
    i_max = n - 1
    while i < i_max:
        if x[i - 1] < x[i]:
            i_ahead = i + 1
            while i_ahead < i_max and x[i_ahead] == x[i]:
                i_ahead += 1
            if x[i_ahead] < x[i]:
                left, right = i, i_ahead - 1
                peaks.append((left + right) // 2)
                i = i_ahead
        i += 1
    return [p for p in peaks if x[p] >= height]


def _max_filter_3x3(a, border):
    H, W = a.shape
    if border == "constant":
        p = np.zeros((H + 2, W + 2), dtype=a.dtype)
        p[1:-1, 1:-1] = a
    elif border == "nearest":
        p = np.pad(a, 1, mode="edge")
    else:
        raise ValueError("border must be 'constant' or 'nearest'")
    out = a.copy()
    for dy in range(3):
        for dx in range(3):
            np.maximum(out, p[dy:dy + H, dx:dx + W], out=out)
    return out


def peak_local_max_2d(corr_map, threshold, border=None):
    """skimage.feature.peak_local_max(corr_map, threshold_abs=threshold, exclude_border=False)
    for a 2-D map: pixel == max of its 3x3 neighbourhood AND pixel > threshold (strict); a map in
    which every pixel equals its local max has no peaks; all plateau pixels are returned.
    Order: descending value, ties in row-major order (skimage's own tie order is unspecified)."""
    border = border or DEFAULT_PEAK_BORDER
    a = np.asarray(corr_map)
    is_max = a == _max_filter_3x3(a, border)
    if is_max.all():
        is_max[:] = False
    is_max &= a > threshold
    rows, cols = np.nonzero(is_max)
    vals = a[rows, cols]
    order = np.argsort(-vals.astype(np.float64), kind="stable")
    return [[int(rows[i]), int(cols[i])] for i in order]


def find_local_max(corr_map, score_threshold=0.6, border=None):
    """MTM._findLocalMax_ (MTM/__init__.py:22-47)."""
    corr_map = np.asarray(corr_map)
    if corr_map.shape == (1, 1):
        return [[0, 0]] if corr_map[0, 0] >= score_threshold else []
    if corr_map.shape[0] == 1:
        return [[0, int(i)] for i in find_peaks_1d(corr_map[0], score_threshold)]
    if corr_map.shape[1] == 1:
        return [[int(i), 0] for i in find_peaks_1d(corr_map[:, 0], score_threshold)]
    return peak_local_max_2d(corr_map, score_threshold, border)


def find_local_min(corr_map, score_threshold=0.4, border=None):
    """MTM._findLocalMin_ (MTM/__init__.py:51-53)."""
    return find_local_max(-np.asarray(corr_map), -score_threshold, border)


# --------------------------------------------------------------------------------------------
# cv2.dnn.NMSBoxes and MTM.NMS
# --------------------------------------------------------------------------------------------
def _rect_overlap(a, b):
    """1.f - (float)jaccardDistance(a, b) for integer (x, y, w, h) rects."""
    aa = a[2] * a[3]
    ab = b[2] * b[3]
    if aa + ab <= 0:
        return np.float32(1.0)
    x1 = max(a[0], b[0])
    y1 = max(a[1], b[1])
    x2 = min(a[0] + a[2], b[0] + b[2])
    y2 = min(a[1] + a[3], b[1] + b[3])
    iw, ih = x2 - x1, y2 - y1
    aab = float(iw * ih) if (iw > 0 and ih > 0) else 0.0
    dist = 1.0 - aab / (float(aa) + float(ab) - aab)
    return np.float32(1.0) - np.float32(dist)


def nms_boxes(boxes, scores, score_threshold, nms_threshold):
    """cv2.dnn.NMSBoxes(boxes, scores, score_threshold, nms_threshold) -> list of kept indices.
    Call site MTM/NMS.py:78.  score > threshold (strict, float32), stable sort by descending
    score, greedy keep iff overlap <= nms_threshold with every kept box."""
    s32 = [np.float32(s) for s in scores]
    thr = np.float32(score_threshold)
    nthr = np.float32(nms_threshold)
    cand = [i for i, s in enumerate(s32) if s > thr]
    cand.sort(key=lambda i: -float(s32[i]))      # python sort is stable
    keep = []
    for i in cand:
        ok = True
        for k in keep:
            if not (_rect_overlap(boxes[i], boxes[k]) <= nthr):
                ok = False
                break
        if ok:
            keep.append(i)
    return keep


def NMS(listHit, scoreThreshold=0.5, sortAscending=False, N_object=float("inf"), maxOverlap=0.5):
    """MTM.NMS.NMS (MTM/NMS.py:20-84)."""
    if len(listHit) <= 1:
        return listHit[:]
    boxes = [h[1] for h in listHit]
    scores = [h[2] for h in listHit]
    if N_object == 1:
        best = min(listHit, key=lambda h: h[2]) if sortAscending else max(listHit, key=lambda h: h[2])
        return [best]
    if sortAscending:
        scores = [1 - s for s in scores]
        scoreThreshold = 1 - scoreThreshold
    idx = nms_boxes(boxes, scores, scoreThreshold, maxOverlap)
    if N_object != float("inf"):
        idx = idx[:N_object]
    return [listHit[i] for i in idx]


# --------------------------------------------------------------------------------------------
# MTM.computeScoreMap / findMatches / matchTemplates
# --------------------------------------------------------------------------------------------
def compute_score_map(template, image, method=TM_CCOEFF_NORMED, mask=None):
    """MTM.computeScoreMap (MTM/__init__.py:56-92) without the warnings."""
    if template.dtype == "float64" or image.dtype == "float64":
        raise ValueError("64-bit images not supported, max 32-bit")
    if not (template.dtype == "uint8" and image.dtype == "uint8"):
        template = np.float32(template)
        image = np.float32(image)
        if mask is not None:
            mask = np.float32(mask)
    if mask is not None:
        if method not in (0, 3):
            mask = None
        elif not (mask.shape == template.shape and mask.dtype == template.dtype):
            mask = None
    return match_template(image, template, method, mask=mask)


def find_matches(listTemplates, image, method=TM_CCOEFF_NORMED, N_object=float("inf"),
                 score_threshold=0.5, searchBox=None, border=None):
    """MTM.findMatches (MTM/__init__.py:95-177) with a deterministic hit order: templates in list
    order, peaks of one template by descending score then row-major position."""
    if N_object != float("inf") and not isinstance(N_object, int):
        raise TypeError("N_object must be an integer")
    if searchBox is not None:
        x_off, y_off, sw, sh = searchBox
        image = image[y_off:y_off + sh, x_off:x_off + sw]
    else:
        x_off = y_off = 0
    hits = []
    for tup in listTemplates:
        name, templ = tup[:2]
        mask = tup[2] if (len(tup) >= 3 and method in (0, 3)) else None
        cmap = compute_score_map(templ, image, method, mask)
        if N_object == 1:
            _, _, min_loc, max_loc = min_max_loc(cmap)
            peaks = [min_loc[::-1]] if method in (0, 1) else [max_loc[::-1]]
        elif method in (0, 1):
            peaks = find_local_min(cmap, score_threshold, border)
        else:
            peaks = find_local_max(cmap, score_threshold, border)
        th, tw = templ.shape[0:2]
        hits.extend((name, (int(p[1]) + x_off, int(p[0]) + y_off, tw, th), cmap[tuple(p)]) for p in peaks)
    return hits


def match_templates(listTemplates, image, method=TM_CCOEFF_NORMED, N_object=float("inf"),
                    score_threshold=0.5, maxOverlap=0.25, searchBox=None, border=None):
    """MTM.matchTemplates (MTM/__init__.py:247-296)."""
    if maxOverlap < 0 or maxOverlap > 1:
        raise ValueError("Maximal overlap between bounding box is in range [0-1]")
    hits = find_matches(listTemplates, image, method, N_object, score_threshold, searchBox, border)
    if method == 0:
        raise ValueError("The method TM_SQDIFF is not supported. Use TM_SQDIFF_NORMED instead.")
    return NMS(hits, score_threshold, method == 1, N_object, maxOverlap)


# --------------------------------------------------------------------------------------------
# CPU baseline port ("what OpenCV + a thread pool does"), used by bench.py's cpu_baseline leg only.
# --------------------------------------------------------------------------------------------
class FastPipeline:
    """A throughput-oriented CPU restatement of the reference pipeline for uint8 images and
    TM_CCOEFF_NORMED / TM_CCORR_NORMED / TM_SQDIFF_NORMED, structured like OpenCV's implementation:
    the sliding dot product comes from a float32 DFT (cv::crossCorr uses a float32 DFT for 8-bit
    input - this is where cv2's ~3e-6 deviation from exact arithmetic comes from), the window sums
    from float64 integral images, the normalisation from one float64 pass, the peaks from
    scipy.ndimage.maximum_filter as skimage does.  The image spectrum and the window statistics of a
    template size are computed once and shared by all templates (more than cv2 shares).
    Thread-safe: one instance is used from a thread pool with one task per template, as the
    reference does (MTM/__init__.py:172-175).  Checked against the exact oracle in
    tests/test_oracle_golden.py::test_fast_pipeline_matches_exact_oracle (1e-4)."""

    def __init__(self, image):
        import threading
        try:
            import scipy.fft as _fft
            self._fft = _fft
        except ImportError:          # numpy's pocketfft computes in float64
            self._fft = np.fft
        self.image = np.asarray(image)
        assert self.image.dtype == np.uint8 and self.image.ndim == 2
        H, W = self.image.shape
        self.fshape = (_next_fast(H), _next_fast(W))
        self.spec = self._fft.rfft2(self.image.astype(np.float32), s=self.fshape)
        f64 = self.image.astype(np.float64)
        self.ii1 = np.zeros((H + 1, W + 1))
        self.ii2 = np.zeros((H + 1, W + 1))
        self.ii1[1:, 1:] = np.cumsum(np.cumsum(f64, axis=1), axis=0)
        self.ii2[1:, 1:] = np.cumsum(np.cumsum(f64 * f64, axis=1), axis=0)
        self._stats = {}
        self._lock = threading.Lock()

    def _window_stats(self, h, w):
        with self._lock:
            st = self._stats.get((h, w))
        if st is None:
            H, W = self.image.shape
            def box(ii):
                return ii[:H - h + 1, :W - w + 1] - ii[:H - h + 1, w:] - ii[h:, :W - w + 1] + ii[h:, w:]
            s1, s2 = box(self.ii1), box(self.ii2)
            st = (s1, s2)
            with self._lock:
                self._stats[(h, w)] = st
        return st

    def score_map(self, templ, method=TM_CCOEFF_NORMED):
        H, W = self.image.shape
        h, w = templ.shape
        g = self._fft.rfft2(templ.astype(np.float32)[::-1, ::-1], s=self.fshape)
        corr = self._fft.irfft2(self.spec * g, s=self.fshape)[h - 1:H, w - 1:W].astype(np.float64)
        s1, s2 = self._window_stats(h, w)
        inv_area = 1.0 / (h * w)
        mean, sdv = _templ_mean_sdv(_as3d(templ), True)
        var = sdv[0] * sdv[0]
        if method == TM_CCOEFF_NORMED:
            if var < DBL_EPSILON:
                return np.ones(corr.shape, np.float32)
            num = corr - s1 * mean[0]
            diff2 = np.maximum(s2 - s1 * s1 * inv_area, 0.0)
            tnorm = math.sqrt(var) / math.sqrt(inv_area)
            other = 0.0
        else:
            tsum2 = (var + mean[0] * mean[0]) / inv_area
            num = corr if method == TM_CCORR_NORMED else np.maximum(s2 - 2.0 * corr + tsum2, 0.0)
            diff2 = s2
            tnorm = math.sqrt(var + mean[0] * mean[0]) / math.sqrt(inv_area)
            other = 1.0 if method == TM_SQDIFF_NORMED else 0.0
        t = np.sqrt(diff2)
        t[diff2 <= np.minimum(0.5, 10.0 * FLT_EPSILON * s2)] = 0.0
        t *= tnorm
        an = np.abs(num)
        with np.errstate(divide="ignore", invalid="ignore"):
            out = num / t
        sat = an >= t
        out[sat] = np.where(an[sat] < 1.125 * t[sat], np.where(num[sat] > 0, 1.0, -1.0), other)
        return out.astype(np.float32)

    def find(self, name, templ, method, score_threshold):
        """One _multi_compute task: score map + local maxima (minima for SQDIFF_NORMED) + hits."""
        m = self.score_map(templ, method)
        v = -m if method == TM_SQDIFF_NORMED else m
        thr = -score_threshold if method == TM_SQDIFF_NORMED else score_threshold
        try:
            import scipy.ndimage as ndi
            mx = ndi.maximum_filter(v, size=3, mode="constant")
        except ImportError:
            mx = _max_filter_3x3(v, "constant")
        is_max = v == mx
        if is_max.all():
            return []
        rows, cols = np.nonzero(is_max & (v > thr))
        order = np.argsort(-v[rows, cols].astype(np.float64), kind="stable")
        th, tw = templ.shape
        return [(name, (int(cols[i]), int(rows[i]), tw, th), m[rows[i], cols[i]]) for i in order]


# ---------------------------------------------------------------------------------------------
# cv2.resize(image, smallDim, interpolation=cv2.INTER_AREA) with an integer factor - the downscale of
# the reference's speed-up recipe (tutorials/Tutorial3-SpeedingUp.ipynb:395).  Restated from OpenCV's
# imgproc/resize.cpp integer-factor path (ResizeAreaFast: uint8 factor 2 -> (sum + 2) >> 2; otherwise
# saturate_cast<uchar>(sum * scale) with float scale = 1.f / area, i.e. round-half-even of a float32
# product; float: float32 sum in offset-table (row-major) order, times scale).  PARITY UNPINNED: no
# OpenCV in this image and no golden vector for it in the reference; checker of
# MTM.augment.downscale / mtm_set_image_downscaled only.
# ---------------------------------------------------------------------------------------------
def downscale_area(image, factor):
    factor = int(factor)
    if factor == 1:
        return image
    rows, cols = image.shape[0] // factor, image.shape[1] // factor
    chans = 1 if image.ndim == 2 else image.shape[2]
    src = image.reshape(image.shape[0], image.shape[1], chans)
    scale = np.float32(1.0) / np.float32(factor * factor)
    if image.dtype == np.uint8 or image.dtype == np.uint16:
        top = int(np.iinfo(image.dtype).max)
        total = np.zeros((rows, cols, chans), np.int64)
        for dy in range(factor):
            for dx in range(factor):
                total += src[dy:rows * factor:factor, dx:cols * factor:factor].astype(np.int64)
        if factor == 2:
            out = ((total + 2) // 4).astype(image.dtype)
        else:
            prod = total.astype(np.float32) * scale                      # float32 product
            out = np.clip(np.round(prod.astype(np.float64)), 0, top).astype(image.dtype)   # np.round: half to even
    else:
        acc = np.zeros((rows, cols, chans), np.float32)
        for dy in range(factor):
            for dx in range(factor):
                acc = (acc + src[dy:rows * factor:factor, dx:cols * factor:factor].astype(np.float32)).astype(np.float32)
        out = (acc * scale).astype(np.float32)
    return out.reshape((rows, cols) + image.shape[2:])
