// The two decisions of MTM's non-maxima suppression (reference MTM/NMS.py:53-84 -> cv2.dnn.NMSBoxes), written once for the
// host (mtm_host.cpp: checks, tests) and the device (mtm_k_nms.hip.h): which of two hits comes first, and whether the
// earlier one suppresses the later one.  Plain C++; MTM_HD marks the functions for both sides when hipcc compiles them.
#pragma once
#include <cstdint>

#include "../../include/mtm_hip.h"

#if defined(__HIPCC__)
#define MTM_HD __host__ __device__
#else
#define MTM_HD
#endif

namespace mtm {

// The score NMSBoxes sorts and thresholds: 1 - score in float32 for the difference methods (MTM/NMS.py:73-75), -0
// normalised (the comparison-based sort of the host treats -0 == +0).
MTM_HD inline float nms_score(const mtm_hit& h, int ascending) { return (ascending ? (1.0f - h.score) : h.score) + 0.0f; }

// Does `a` precede `b` in the order NMSBoxes works through?  Descending transformed score; ties stay in the order the hit
// list is in when MTM hands it over - the order mtm_find_matches returns: template, then descending quality (score, or
// -score for the difference methods), then row-major position.  (No NaN: hits with a NaN score fail the score threshold.)
MTM_HD inline bool nms_earlier(const mtm_hit& a, const mtm_hit& b, int ascending) {
    const float sa = nms_score(a, ascending), sb = nms_score(b, ascending);
    if (sa != sb) return sa > sb;
    if (a.templ_idx != b.templ_idx) return a.templ_idx < b.templ_idx;
    if (a.score != b.score) return ascending ? a.score < b.score : a.score > b.score;
    if (a.y != b.y) return a.y < b.y;
    return a.x < b.x;
}

// 1.f - (float)jaccardDistance(a, b) for Rect_<int> (OpenCV dnn/nms.inl.hpp), operation by operation
MTM_HD inline float nms_rect_overlap(const mtm_hit& a, const mtm_hit& b) {
    const long long aa = (long long)a.w * a.h, ab = (long long)b.w * b.h;
    if (aa + ab <= 0) return 1.0f;
    const int x1 = a.x > b.x ? a.x : b.x, y1 = a.y > b.y ? a.y : b.y;
    const int ax2 = a.x + a.w, bx2 = b.x + b.w, ay2 = a.y + a.h, by2 = b.y + b.h;
    const int x2 = ax2 < bx2 ? ax2 : bx2, y2 = ay2 < by2 ? ay2 : by2;
    const int iw = x2 - x1, ih = y2 - y1;
    if (iw <= 0 || ih <= 0) return 0.0f;      // disjoint: 1.f - (float)(1.0 - 0.0 / u), without the division
    const double aab = (double)((long long)iw * ih);
    const double dist = 1.0 - aab / ((double)aa + (double)ab - aab);
    return 1.0f - (float)dist;
}

}  // namespace mtm
