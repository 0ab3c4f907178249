"""
Non-Maxima Suppression for template matching: same public function, arguments, defaults and
results as the reference's MTM/NMS.py:20-84, with cv2.dnn.NMSBoxes (reference MTM/NMS.py:78)
replaced by the float32-faithful C++ restatement behind ``mtm_nms`` in libmtm_hip.so.
"""
from typing import List, Sequence, Tuple

from . import _lib

Hit = Tuple[str, Tuple[int, int, int, int], float]


def NMS(listHit: Sequence[Hit], scoreThreshold: float = 0.5, sortAscending: bool = False,
        N_object=float("inf"), maxOverlap: float = 0.5) -> List[Hit]:
    """
    Overlap-based Non-Maxima Suppression over hits ``(label, (x, y, width, height), score)``.

    - scoreThreshold: hits scoring below it (above it when ``sortAscending``) are discarded
    - sortAscending : True when a low score means a better match (difference-based scores)
    - N_object      : keep at most this many hits (``float("inf")`` = all that pass)
    - maxOverlap    : largest allowed Intersection-over-Union between two kept boxes

    A list of at most one hit is returned as a copy without thresholding, and ``N_object == 1``
    returns the single best hit, exactly as the reference does.
    """
    nHits = len(listHit)
    if nHits <= 1:
        return listHit[:]

    listLabel, listBoxes, listScores = zip(*listHit)

    if N_object == 1:
        if sortAscending:
            bestHit = min(listHit, key=lambda hit: hit[2])
        else:
            bestHit = max(listHit, key=lambda hit: hit[2])
        return [bestHit]

    if sortAscending:   # same arithmetic, on the same scalar types, as the reference
        listScores = [1 - score for score in listScores]
        scoreThreshold = 1 - scoreThreshold

    indexes = _lib.nms_indices(listBoxes, listScores, scoreThreshold, maxOverlap)

    if N_object != float("inf"):
        indexes = indexes[:N_object]

    return [listHit[x] for x in indexes]
