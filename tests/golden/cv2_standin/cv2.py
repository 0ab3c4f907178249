"""
Stand-in for the four cv2 entry points the reference's hot path touches (MTM/__init__.py:92,
:226; MTM/NMS.py:78) plus the TM_* constants, backed by oracle/mtm_oracle.py.

Used ONLY by tests/golden/make_golden.py, in the build container, to drive the UNMODIFIED
reference package (cv2 is not installable here: no network).  Never shipped, never imported by the
product or by the tests themselves.
"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", "oracle"))
import mtm_oracle as _o  # noqa: E402

TM_SQDIFF, TM_SQDIFF_NORMED, TM_CCORR, TM_CCORR_NORMED, TM_CCOEFF, TM_CCOEFF_NORMED = range(6)
__version__ = "standin-oracle"


def matchTemplate(image, templ, method, result=None, mask=None):
    return _o.match_template(np.asarray(image), np.asarray(templ), method, mask=mask)


def minMaxLoc(src, mask=None):
    return _o.min_max_loc(src)


class _Dnn:
    @staticmethod
    def NMSBoxes(bboxes, scores, score_threshold, nms_threshold, eta=1.0, top_k=0):
        keep = _o.nms_boxes(list(bboxes), list(scores), score_threshold, nms_threshold)
        return np.array(keep, dtype=np.int32) if keep else ()


dnn = _Dnn()
