// Internal declarations shared by the host-only translation unit (mtm_host.cpp) and the HIP
// translation units (mtm_context / _placement / _launch / _api / _comm .hip).  Not part of the ABI.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mtm_hip.h"

namespace mtm {

void set_error(const std::string& msg);   // thread-local message behind mtm_last_error()

// cv::meanStdDev + the template constants of OpenCV's common_matchTemplate, in the same
// operation order as oracle/mtm_oracle.py::match_template (so that both sides round alike).
struct TemplStats {
    double mean[4] = {0, 0, 0, 0};   // templMean per channel (zeroed when numType != 1)
    double templ_norm = 0;           // sqrt(templNorm) / sqrt(invArea)
    double templ_sum2 = 0;           // templSum2 / invArea
    double inv_area = 0;
    int all_ones = 0;                // TM_CCOEFF_NORMED with a constant template: map == 1
    double templ2_mask2_sum = 0;     // masked path: sum((T*M)^2)
    double centred_sum2 = 0;         // sum over channels of sum (T - channel mean)^2, whatever the method (error bound of the
                                     // refined raw-sum extremum of float32 classes, mtm_bf16.hip.h)
};

// px: planar float64 copies of the template (and mask weights, or nullptr), chans planes of
// rows*cols each.  `integer` = values are exact integers (uint8 source).
TemplStats compute_templ_stats(const double* px, const double* mask, int rows, int cols, int chans,
                               int method, bool integer);

// the same from the per-channel sums (sum v, sum v^2), or - masked - from sum((v*m)^2) alone
TemplStats templ_stats_from_sums(const double* sum, const double* sumsq, double templ2_mask2_sum, bool masked, int rows,
                                 int cols, int chans, int method);

// exact sum v / sum v^2 of n bytes, added to *sum / *sumsq
void u8_run_sums(const uint8_t* p, size_t n, unsigned long long* sum, unsigned long long* sumsq);

// scipy.signal.find_peaks(x, height=h)[0]
std::vector<int> find_peaks_1d(const float* x, int n, int stride, float height, bool negate);

// float32-faithful restatement of cv2.dnn.NMSBoxes as called by MTM.NMS
void nms_boxes(const mtm_hit* hits, int64_t n, const float* scores, float score_threshold,
               float nms_threshold, std::vector<int32_t>& keep);

// the same selection from a list in any order, returned in NMSBoxes' order (score ties as mtm_find_matches' order resolves them)
// (`n_sure`: the first n_sure hits are known to be kept - no earlier hit overlaps them beyond the threshold -: they are not tested)
void nms_select(const mtm_hit* hits, int64_t n, int ascending, float score_threshold, float nms_threshold,
                std::vector<int32_t>& keep, int64_t n_sure = 0);

// The order in which mtm_find_matches returns its records: template, then descending quality (score, or -score for the
// difference methods), then row-major position.  Deterministic whatever order the GPU appended them in; thousands
// of records (smooth images at a low threshold) are sorted by an LSD radix sort on the same key.
void sort_hits(std::vector<mtm_hit>& hits, bool mode_min);

}  // namespace mtm
