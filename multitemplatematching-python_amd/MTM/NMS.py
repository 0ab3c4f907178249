"""
Non-Maxima Suppression for template matching (public ``MTM.NMS.NMS`` of the reference, MTM/NMS.py:20-84).

The suppression itself - cv2.dnn.NMSBoxes at reference MTM/NMS.py:78 - is ``mtm_nms`` in libmtm_hip.so
(float32-faithful C++ restatement, grid-accelerated).  This module only converts between the reference's
list of ``(label, (x, y, w, h), score)`` tuples and the structured hit records the library works on; the
package's own matchTemplates never builds the tuples before the suppression (``MTM._nms_raw``).
"""
from typing import List, Sequence, Tuple

import numpy as np

from . import _lib

Hit = Tuple[str, Tuple[int, int, int, int], float]


def _records(hits, quality):
    """Hit tuples -> structured records; ``quality`` (higher = better) goes into the float32 score field,
    which is the narrowing the cv2 binding applies to the reference's score list."""
    rec = np.zeros(len(hits), dtype=_lib.HIT_DTYPE)
    box = np.asarray([h[1] for h in hits], dtype=np.int64).reshape(len(hits), 4)
    rec["templ_idx"] = np.arange(len(hits), dtype=np.int32)
    rec["x"], rec["y"], rec["w"], rec["h"] = box[:, 0], box[:, 1], box[:, 2], box[:, 3]
    rec["score"] = np.asarray(quality, dtype=np.float32)
    return rec


def NMS(listHit: Sequence[Hit], scoreThreshold: float = 0.5, sortAscending: bool = False,
        N_object=float("inf"), maxOverlap: float = 0.5) -> List[Hit]:
    """
    Overlap-based Non-Maxima Suppression over hits ``(label, (x, y, width, height), score)``.

    - scoreThreshold: hits scoring below it (above it when ``sortAscending``) are discarded
    - sortAscending : True when a low score means a better match (difference-based scores)
    - N_object      : keep at most this many hits (``float("inf")`` = all that pass)
    - maxOverlap    : largest allowed Intersection-over-Union between two kept boxes

    Reference behaviours kept: a list of at most one hit comes back as a copy, unthresholded
    (MTM/NMS.py:53-55); ``N_object == 1`` returns the single best hit whatever its score, the first one
    winning ties (:61-69); difference scores are suppressed on ``1 - score`` against ``1 - scoreThreshold``
    computed on the caller's scalar types (:73-75).
    """
    if len(listHit) <= 1:
        return listHit[:]               # a copy of whatever sequence type came in (MTM/NMS.py:53-55)
    hits = list(listHit)
    scores = [h[2] for h in hits]
    if N_object == 1:
        # python's min() / max() as in the reference (:61-69): the first of equal scores, and a NaN score is only ever
        # selected when it comes first (every comparison with it is False)
        pick = min if sortAscending else max
        return [pick(hits, key=lambda hit: hit[2])]
    if sortAscending:
        quality, threshold = [1 - s for s in scores], 1 - scoreThreshold
    else:
        quality, threshold = scores, scoreThreshold
    keep = _lib.nms_hits(_records(hits, quality), threshold, maxOverlap)
    if N_object != float("inf"):
        keep = keep[:N_object]
    return [hits[int(i)] for i in keep]
