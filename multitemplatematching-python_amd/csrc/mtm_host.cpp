// Host-only parts of libmtm_hip.so: error string, template statistics, 1-D peak finding and the
// NMS.  Nothing here touches the GPU, so these entry points also work on a machine without one
// (the CPU test-suite exercises mtm_nms through the C ABI).
#include <algorithm>
#include <cmath>
#include <cfloat>
#include <cstring>
#include <numeric>
#include <climits>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include "mtm_internal.h"
#include "mtm_nms_core.h"

namespace mtm {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

// sum v and sum v^2 of a run of bytes, exact: the per-call cost of a template set the library has not seen before is
// dominated by this pass on the host (32 templates of 64 x 64: 82 us as a scalar loop, 7 us this way)
void u8_run_sums(const uint8_t* p, size_t n, unsigned long long* sum, unsigned long long* sumsq) {
    unsigned long long s = 0, q = 0;
    size_t i = 0;
#if defined(__SSE2__)
    const __m128i zero = _mm_setzero_si128();
    while (n - i >= 16) {
        // 32-bit lanes: each step adds at most 4 * 255^2 to a lane of q -> 8192 steps stay below 2^31
        const size_t steps = std::min<size_t>((n - i) / 16, 8192);
        __m128i vs = zero, vq = zero;
        for (size_t k = 0; k < steps; ++k, i += 16) {
            const __m128i v = _mm_loadu_si128((const __m128i*)(p + i));
            vs = _mm_add_epi64(vs, _mm_sad_epu8(v, zero));
            const __m128i lo = _mm_unpacklo_epi8(v, zero), hi = _mm_unpackhi_epi8(v, zero);
            vq = _mm_add_epi32(vq, _mm_add_epi32(_mm_madd_epi16(lo, lo), _mm_madd_epi16(hi, hi)));
        }
        alignas(16) unsigned long long ls[2];
        alignas(16) uint32_t lq[4];
        _mm_store_si128((__m128i*)ls, vs);
        _mm_store_si128((__m128i*)lq, vq);
        s += ls[0] + ls[1];
        q += (unsigned long long)lq[0] + lq[1] + lq[2] + lq[3];
    }
#endif
    for (; i < n; ++i) {
        const unsigned v = p[i];
        s += v;
        q += v * v;
    }
    *sum += s;
    *sumsq += q;
}

// ---------------------------------------------------------------------------------------------
// Template constants.  Follows OpenCV's common_matchTemplate (the arithmetic behind the
// cv2.matchTemplate call at reference MTM/__init__.py:92) in the operation order that
// oracle/mtm_oracle.py::match_template uses, so that both sides round identically.
// ---------------------------------------------------------------------------------------------
TemplStats compute_templ_stats(const double* px, const double* mask, int rows, int cols, int chans,
                               int method, bool integer) {
    const size_t plane = (size_t)rows * cols;
    if (mask != nullptr) {
        // matchTemplateMask: templ2_mask2_sum = norm(templ.mul(mask), NORM_L2SQR)
        double s = 0.0;
        for (int c = 0; c < chans; ++c)
            for (size_t i = 0; i < plane; ++i) {
                const double v = px[c * plane + i] * mask[c * plane + i];
                s += v * v;
            }
        return templ_stats_from_sums(nullptr, nullptr, s, true, rows, cols, chans, method);
    }
    double sum[4] = {0, 0, 0, 0}, sumsq[4] = {0, 0, 0, 0};
    for (int c = 0; c < chans && c < 4; ++c) {
        double s = 0.0, sq = 0.0;
        if (integer) {
            long long is = 0, isq = 0;
            for (size_t i = 0; i < plane; ++i) {
                const long long v = (long long)px[c * plane + i];
                is += v;
                isq += v * v;
            }
            s = (double)is;
            sq = (double)isq;
        } else {
            for (size_t i = 0; i < plane; ++i) {
                const double v = px[c * plane + i];
                s += v;
                sq += v * v;
            }
        }
        sum[c] = s;
        sumsq[c] = sq;
    }
    return templ_stats_from_sums(sum, sumsq, 0.0, false, rows, cols, chans, method);
}

// The same from the per-channel sums (sum v, sum v^2) - or, masked, from sum (v*m)^2 alone: what the device
// reduction over a template source delivers (exact integers for uint8 pixels).
TemplStats templ_stats_from_sums(const double* sum, const double* sumsq, double templ2_mask2_sum, bool masked, int rows,
                                 int cols, int chans, int method) {
    TemplStats st;
    const double n = (double)rows * (double)cols;
    st.inv_area = 1.0 / ((double)rows * (double)cols);
    if (masked) {
        st.templ2_mask2_sum = templ2_mask2_sum;
        return st;
    }
    double mean[4] = {0, 0, 0, 0}, sdv[4] = {0, 0, 0, 0};
    for (int c = 0; c < chans && c < 4; ++c) {
        mean[c] = sum[c] / n;
        const double var = sumsq[c] / n - mean[c] * mean[c];
        sdv[c] = std::sqrt(std::max(var, 0.0));
        st.centred_sum2 += std::max(var, 0.0) * n + 1e-15 * sumsq[c];       // (+ the cancellation in sumsq / n - mean^2)
    }
    if (method == MTM_TM_CCORR) return st;
    const int num_type = (method == MTM_TM_CCORR || method == MTM_TM_CCORR_NORMED) ? 0
                       : (method == MTM_TM_CCOEFF || method == MTM_TM_CCOEFF_NORMED) ? 1 : 2;
    for (int c = 0; c < 4; ++c) st.mean[c] = mean[c];
    if (method != MTM_TM_CCOEFF) {
        double templ_norm = 0.0;
        for (int c = 0; c < chans && c < 4; ++c) templ_norm += sdv[c] * sdv[c];
        if (templ_norm < DBL_EPSILON && method == MTM_TM_CCOEFF_NORMED) {
            st.all_ones = 1;
            return st;
        }
        double msum = 0.0;
        for (int c = 0; c < chans && c < 4; ++c) msum += mean[c] * mean[c];
        double templ_sum2 = templ_norm + msum;
        if (num_type != 1) {
            for (int c = 0; c < 4; ++c) st.mean[c] = 0.0;
            templ_norm = templ_sum2;
        }
        templ_sum2 /= st.inv_area;
        templ_norm = std::sqrt(templ_norm);
        templ_norm /= std::sqrt(st.inv_area);
        st.templ_norm = templ_norm;
        st.templ_sum2 = templ_sum2;
    }
    return st;
}

// ---------------------------------------------------------------------------------------------
// scipy.signal.find_peaks(x, height=height)[0], the 1-D branch of MTM._findLocalMax_
// (reference MTM/__init__.py:33-41): strict local maxima, a plateau yields its middle sample,
// end points are never peaks, the height test is >=.  `negate` evaluates it on -x
// (MTM._findLocalMin_, :51-53).
// ---------------------------------------------------------------------------------------------
std::vector<int> find_peaks_1d(const float* x, int n, int stride, float height, bool negate) {
    std::vector<int> peaks;
    auto at = [&](int i) { const float v = x[(size_t)i * stride]; return negate ? -v : v; };
    int i = 1;
    const int i_max = n - 1;
    while (i < i_max) {
        if (at(i - 1) < at(i)) {
            int ahead = i + 1;
            while (ahead < i_max && at(ahead) == at(i)) ++ahead;
            if (at(ahead) < at(i)) {
                const int left = i, right = ahead - 1;
                const int mid = (left + right) / 2;
                if (at(mid) >= height) peaks.push_back(mid);
                i = ahead;
            }
        }
        ++i;
    }
    return peaks;
}

// ---------------------------------------------------------------------------------------------
// cv2.dnn.NMSBoxes (OpenCV dnn/nms.cpp + nms.inl.hpp) as called at reference MTM/NMS.py:78.
// ---------------------------------------------------------------------------------------------
static inline float rect_overlap(const mtm_hit& a, const mtm_hit& b) {
    // 1.f - (float)jaccardDistance(a, b) for Rect_<int>
    const long long aa = (long long)a.w * a.h, ab = (long long)b.w * b.h;
    if (aa + ab <= 0) return 1.0f;
    const int x1 = std::max(a.x, b.x), y1 = std::max(a.y, b.y);
    const int x2 = std::min(a.x + a.w, b.x + b.w), y2 = std::min(a.y + a.h, b.y + b.h);
    const int iw = x2 - x1, ih = y2 - y1;
    if (iw <= 0 || ih <= 0) return 0.0f;      // disjoint: 1.f - (float)(1.0 - 0.0 / u), without the division
    const double aab = (double)((long long)iw * ih);
    const double dist = 1.0 - aab / ((double)aa + (double)ab - aab);
    return 1.0f - (float)dist;
}

namespace {

// float -> uint32 whose unsigned order is the float order (-0 == +0 must be normalised by the caller; NaN sorts high)
inline uint32_t float_order(float v) {
    uint32_t b;
    std::memcpy(&b, &v, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// Stable LSD radix sort (8-bit digits) of `rec` by the `key_bytes` low bytes of Rec::key(), ascending.  Digits that are
// the same in every record cost one histogram pass and no move.
template <typename Rec, int KEY_BYTES>
void radix_sort(std::vector<Rec>& rec) {
    std::vector<Rec> tmp(rec.size());
    Rec* src = rec.data();
    Rec* dst = tmp.data();
    const size_t n = rec.size();
    for (int d = 0; d < KEY_BYTES; ++d) {
        size_t hist[256] = {0};
        for (size_t i = 0; i < n; ++i) ++hist[src[i].digit(d)];
        if (hist[src[0].digit(d)] == n) continue;
        size_t sum = 0;
        for (int b = 0; b < 256; ++b) {
            const size_t c = hist[b];
            hist[b] = sum;
            sum += c;
        }
        for (size_t i = 0; i < n; ++i) dst[hist[src[i].digit(d)]++] = src[i];
        std::swap(src, dst);
    }
    if (src != rec.data()) std::memcpy(rec.data(), src, n * sizeof(Rec));
}

struct HitRec {           // ascending (templ, ~order(quality), y, x)
    uint64_t lo;          // [~order(quality) : 32][y : 16][x : 16]
    uint32_t templ;
    uint32_t idx;
    unsigned digit(int d) const { return d < 8 ? (unsigned)(lo >> (8 * d)) & 255u : (unsigned)(templ >> (8 * (d - 8))) & 255u; }
};

struct ScoreRec {         // ascending ~order(score) = descending score
    uint32_t key, idx;
    unsigned digit(int d) const { return (key >> (8 * d)) & 255u; }
};

}  // namespace

void sort_hits(std::vector<mtm_hit>& hits, bool mode_min) {
    const size_t n = hits.size();
    bool radix = n >= 512;
    for (size_t i = 0; i < n && radix; ++i)
        radix = hits[i].x >= 0 && hits[i].x < 65536 && hits[i].y >= 0 && hits[i].y < 65536 && hits[i].templ_idx >= 0 &&
                hits[i].score == hits[i].score;
    if (!radix) {
        std::sort(hits.begin(), hits.end(), [&](const mtm_hit& a, const mtm_hit& b) {
            if (a.templ_idx != b.templ_idx) return a.templ_idx < b.templ_idx;
            const float qa = mode_min ? -a.score : a.score, qb = mode_min ? -b.score : b.score;
            if (qa != qb) return qa > qb;
            if (a.y != b.y) return a.y < b.y;
            return a.x < b.x;
        });
        return;
    }
    std::vector<HitRec> rec(n);
    for (size_t i = 0; i < n; ++i) {
        const float q = (mode_min ? -hits[i].score : hits[i].score) + 0.0f;      // -0 -> +0: equal qualities, equal keys
        rec[i].lo = ((uint64_t)(~float_order(q)) << 32) | ((uint64_t)(uint32_t)hits[i].y << 16) | (uint32_t)hits[i].x;
        rec[i].templ = (uint32_t)hits[i].templ_idx;
        rec[i].idx = (uint32_t)i;
    }
    radix_sort<HitRec, 12>(rec);
    std::vector<mtm_hit> out(n);
    for (size_t i = 0; i < n; ++i) out[i] = hits[rec[i].idx];
    hits.swap(out);
}

static void nms_greedy(const mtm_hit* hits, const std::vector<int32_t>& cand, float nms_threshold, std::vector<int32_t>& keep,
                       int64_t n_sure = 0);

void nms_boxes(const mtm_hit* hits, int64_t n, const float* scores, float score_threshold,
               float nms_threshold, std::vector<int32_t>& keep) {
    std::vector<int32_t> cand;
    cand.reserve((size_t)n);
    bool radix = n >= 512;
    for (int64_t i = 0; i < n; ++i)
        if (scores[i] > score_threshold) {            // (false for NaN)
            cand.push_back((int32_t)i);
        }
    if (radix) {
        // descending score, ties in input order: a stable radix sort on the ordered bits (-0 normalised: -0 == +0 in the
        // comparison-based sort below)
        std::vector<ScoreRec> rec(cand.size());
        for (size_t k = 0; k < cand.size(); ++k) {
            rec[k].key = ~float_order(scores[cand[k]] + 0.0f);
            rec[k].idx = (uint32_t)cand[k];
        }
        if (!rec.empty()) radix_sort<ScoreRec, 4>(rec);
        for (size_t k = 0; k < cand.size(); ++k) cand[k] = (int32_t)rec[k].idx;
    } else {
        std::stable_sort(cand.begin(), cand.end(),
                         [&](int32_t a, int32_t b) { return scores[a] > scores[b]; });
    }
    nms_greedy(hits, cand, nms_threshold, keep);
}

// MTM's NMS on a hit list in ANY order (mtm_find_matches_image_nms): what nms_boxes selects from the list in the order
// mtm_find_matches returns, without producing that order first - one radix sort by the transformed score, and only runs of
// equal scores (exact copies at 1.0) are put into the order they would arrive in (mtm_nms_core.h: nms_earlier).
void nms_select(const mtm_hit* hits, int64_t n, int ascending, float score_threshold, float nms_threshold,
                std::vector<int32_t>& keep, int64_t n_sure) {
    std::vector<ScoreRec> rec;
    rec.reserve((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const float s = nms_score(hits[i], ascending);
        if (s > score_threshold) rec.push_back(ScoreRec{~float_order(s), (uint32_t)i});      // (false for NaN)
    }
    if (rec.size() >= 64) radix_sort<ScoreRec, 4>(rec);
    else std::sort(rec.begin(), rec.end(), [](const ScoreRec& a, const ScoreRec& b) { return a.key < b.key; });
    for (size_t a = 0; a < rec.size();) {
        size_t b = a + 1;
        while (b < rec.size() && rec[b].key == rec[a].key) ++b;
        if (b - a > 1)
            std::sort(rec.begin() + (long)a, rec.begin() + (long)b,
                      [&](const ScoreRec& p, const ScoreRec& q) { return nms_earlier(hits[p.idx], hits[q.idx], ascending); });
        a = b;
    }
    std::vector<int32_t> cand(rec.size());
    for (size_t k = 0; k < rec.size(); ++k) cand[k] = (int32_t)rec[k].idx;
    nms_greedy(hits, cand, nms_threshold, keep, n_sure);
}

static void nms_greedy(const mtm_hit* hits, const std::vector<int32_t>& cand, float nms_threshold, std::vector<int32_t>& keep,
                       int64_t n_sure) {
    keep.clear();

    // Same greedy decisions as OpenCV's NMSFast_ (a candidate is kept iff its overlap with EVERY kept
    // box is <= nms_threshold), but a candidate is only compared with the kept boxes that can touch
    // it: disjoint boxes have overlap 0 <= nms_threshold.  Kept boxes are hashed by the grid cell of
    // their top-left corner, cell = largest box side, so 3x3 cells cover every possible partner.
    // O(n) instead of O(n^2) for the thousands of hits a multi-GPU gather produces.
    long long cell = 1;
    bool regular = nms_threshold >= 0.0f;
    for (int32_t i : cand) {
        if (hits[i].w <= 0 || hits[i].h <= 0) regular = false;
        cell = std::max<long long>(cell, std::max(hits[i].w, hits[i].h));
    }
    if (!regular || cand.size() < 64) {          // degenerate boxes / tiny lists: plain double loop
        for (int32_t idx : cand) {
            bool ok = true;
            for (size_t k = 0; k < keep.size() && ok; ++k)
                ok = rect_overlap(hits[idx], hits[keep[k]]) <= nms_threshold;
            if (ok) keep.push_back(idx);
        }
        return;
    }
    auto fdiv = [](long long a, long long b) { return a >= 0 ? a / b : -((-a + b - 1) / b); };
    // dense grid of singly linked lists (head per cell, next per box) over the candidates' extent
    long long cx0 = LLONG_MAX, cy0 = LLONG_MAX, cx1 = LLONG_MIN, cy1 = LLONG_MIN;
    for (int32_t i : cand) {
        const long long cx = fdiv(hits[i].x, cell), cy = fdiv(hits[i].y, cell);
        cx0 = std::min(cx0, cx); cx1 = std::max(cx1, cx);
        cy0 = std::min(cy0, cy); cy1 = std::max(cy1, cy);
    }
    const long long gw = cx1 - cx0 + 3, gh = cy1 - cy0 + 3;      // one empty ring around the extent
    if (gw * gh > (1ll << 24)) {                                  // absurdly sparse: plain double loop
        for (int32_t idx : cand) {
            bool ok = true;
            for (size_t k = 0; k < keep.size() && ok; ++k)
                ok = rect_overlap(hits[idx], hits[keep[k]]) <= nms_threshold;
            if (ok) keep.push_back(idx);
        }
        return;
    }
    // Lists are kept in insertion order (head = the cell's best box): a candidate of a dense cluster is almost always
    // suppressed by the cluster's top, which is then the first box it meets; the own cell is visited first.  The kept
    // boxes live in one compact array in the order they were kept (the lists link into it).
    //
    // The decision "overlap <= nms_threshold" is OpenCV's float expression 1.f - (float)(1.0 - inter / union), a
    // non-decreasing step function of r = inter / union: there is one r* with  overlap <= threshold  <=>  r <= r*.  It is
    // found by bisection on the expression itself; a pair is then decided by inter vs r* x union - no division - unless it
    // falls within 1e-12 (relative) of the step, where the original expression decides.
    auto overlap_of = [](double r) { return 1.0f - (float)(1.0 - r); };
    double r_star = 2.0;                                  // threshold >= 1: nothing is ever suppressed
    if (overlap_of(1.0) > nms_threshold) {
        double lo = 0.0, hi = 1.0;                        // overlap_of(lo) <= threshold < overlap_of(hi)
        for (int it = 0; it < 100 && hi - lo > 0.0; ++it) {
            const double mid = lo + 0.5 * (hi - lo);
            if (mid <= lo || mid >= hi) break;
            if (overlap_of(mid) <= nms_threshold) lo = mid;
            else hi = mid;
        }
        r_star = lo;
    }
    const double r_lo = r_star * (1.0 - 1e-12), r_hi = r_star * (1.0 + 1e-12);
    struct Kept { int x, y, x2, y2; double area; int32_t next; };
    std::vector<Kept> kb;
    kb.reserve(cand.size());
    std::vector<int32_t> head((size_t)(gw * gh), -1), tail((size_t)(gw * gh), -1);
    static const int kOrder[9][2] = {{0, 0}, {0, -1}, {0, 1}, {-1, 0}, {1, 0}, {-1, -1}, {-1, 1}, {1, -1}, {1, 1}};
    const bool cell_pow2 = (cell & (cell - 1)) == 0;
    int cell_shift = 0;
    while ((1ll << cell_shift) < cell) ++cell_shift;
    for (int32_t idx : cand) {
        const mtm_hit& b = hits[idx];
        const long long cx = (cell_pow2 && b.x >= 0 ? (long long)(b.x >> cell_shift) : fdiv(b.x, cell)) - cx0 + 1;
        const long long cy = (cell_pow2 && b.y >= 0 ? (long long)(b.y >> cell_shift) : fdiv(b.y, cell)) - cy0 + 1;
        const int bx2 = b.x + b.w, by2 = b.y + b.h;
        const double barea = (double)((long long)b.w * b.h);
        bool ok = true;
        for (int o = 0; o < 9 && ok && idx >= n_sure; ++o)       // (idx < n_sure: kept for certain, straight into the grid)
            for (int32_t k = head[(size_t)((cy + kOrder[o][0]) * gw + cx + kOrder[o][1])]; k >= 0; k = kb[(size_t)k].next) {
                const Kept& q = kb[(size_t)k];
                const int iw = std::min(bx2, q.x2) - std::max(b.x, q.x);
                if (iw <= 0) continue;
                const int ih = std::min(by2, q.y2) - std::max(b.y, q.y);
                if (ih <= 0) continue;                               // disjoint: overlap 0 <= threshold
                const double inter = (double)((long long)iw * ih), uni = barea + q.area - inter;
                if (inter <= r_lo * uni) continue;
                if (inter >= r_hi * uni || !(rect_overlap(b, hits[keep[(size_t)k]]) <= nms_threshold)) {
                    ok = false;
                    break;
                }
            }
        if (ok) {
            const int32_t slot = (int32_t)kb.size();              // == keep.size(): kb[i] is the box of keep[i]
            keep.push_back(idx);
            kb.push_back(Kept{b.x, b.y, bx2, by2, barea, -1});
            const size_t cellidx = (size_t)(cy * gw + cx);
            if (tail[cellidx] >= 0) kb[(size_t)tail[cellidx]].next = slot;
            else head[cellidx] = slot;
            tail[cellidx] = slot;
        }
    }
}

}  // namespace mtm

extern "C" {

const char* mtm_last_error(void) { return mtm::g_last_error.c_str(); }

int mtm_abi_version(void) { return MTM_ABI_VERSION; }

int mtm_nms(const mtm_hit* hits, int64_t n, double score_threshold, int ascending,
            int64_t n_object, double max_overlap, int32_t* keep, int64_t* n_keep) {
    if (n < 0 || (n > 0 && (hits == nullptr || keep == nullptr)) || n_keep == nullptr) {
        mtm::set_error("mtm_nms: bad arguments");
        return MTM_E_INVALID;
    }
    std::vector<float> scores((size_t)n);
    // MTM/NMS.py:73-75: scores are np.float32, so 1-score is a float32 subtraction; the threshold
    // is a python float, transformed in double and narrowed by the cv2 binding.
    for (int64_t i = 0; i < n; ++i) scores[i] = ascending ? (1.0f - hits[i].score) : hits[i].score;
    const float thr = (float)(ascending ? (1.0 - score_threshold) : score_threshold);
    std::vector<int32_t> kept;
    mtm::nms_boxes(hits, n, scores.data(), thr, (float)max_overlap, kept);
    int64_t m = (int64_t)kept.size();
    if (n_object >= 0 && m > n_object) m = n_object;   // MTM/NMS.py:81-82
    for (int64_t i = 0; i < m; ++i) keep[i] = kept[(size_t)i];
    *n_keep = m;
    return MTM_OK;
}

}  // extern "C"
