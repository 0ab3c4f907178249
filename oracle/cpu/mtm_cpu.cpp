// libmtm_cpu.so - C++ CPU port of the reference pipeline for the baseline leg of bench.py (TEST INFRASTRUCTURE: only
// tests/, __graft_entry__ and bench.py's cpu_baseline may build, load or call it; the product never does).
//
// What the reference does per template on a worker thread of its pool (MTM/__init__.py:172-175, :179-244):
//   cv2.matchTemplate(image, template, method)          OpenCV imgproc/templmatch.cpp: block-wise float32 DFT
//                                                        cross-correlation + integral images (double) + the
//                                                        per-pixel normalisation of common_matchTemplate
//   skimage.feature.peak_local_max(map, threshold_abs)  3x3 maximum filter (edge replicated) == map, > threshold
// followed by cv2.dnn.NMSBoxes once (MTM/NMS.py:78).  This file restates exactly that structure in portable C++
// (own radix-2 FFT, no library): uint8 single-channel images and templates, TM_CCOEFF_NORMED / TM_CCORR_NORMED /
// TM_SQDIFF_NORMED, one task per template on `n_threads` std::threads (the reference uses round(cpu_count / 2)).
// Two things are SHARED across templates where the reference recomputes them per call - the block spectra of the
// image and the integral images - which makes this baseline faster than the reference's own structure, never
// slower.  Correlation blocks: 512 x 512 DFTs (OpenCV picks its block size the same way: a power of two a few
// times the template).  Accuracy is that of a float32 DFT, like cv2 (|score error| ~ 1e-6); the hit lists are
// checked against the float64 oracle in tests/test_cpu_baseline_cpu.py.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {

typedef std::complex<float> cf;

struct Fft {                         // in-place radix-2 decimation-in-time, length n = 2^k
    int n = 0;
    std::vector<int> rev;
    std::vector<cf> twf, twi;        // per stage, contiguous: stage with half-length m uses [m .. 2m): exp(-/+ 2 pi i k / (2m))
    explicit Fft(int n_) : n(n_), rev((size_t)n_), twf((size_t)n_), twi((size_t)n_) {
        int lg = 0;
        while ((1 << lg) < n) ++lg;
        for (int i = 0; i < n; ++i) {
            int r = 0;
            for (int b = 0; b < lg; ++b) r |= ((i >> b) & 1) << (lg - 1 - b);
            rev[(size_t)i] = r;
        }
        for (int m = 1; m < n; m <<= 1)
            for (int k = 0; k < m; ++k) {
                const double a = -M_PI * k / m;
                twf[(size_t)(m + k)] = cf((float)std::cos(a), (float)std::sin(a));
                twi[(size_t)(m + k)] = cf((float)std::cos(a), (float)-std::sin(a));
            }
    }
    void run(cf* x, bool inverse) const {
        for (int i = 0; i < n; ++i)
            if (rev[(size_t)i] > i) std::swap(x[i], x[rev[(size_t)i]]);
        const cf* tw = inverse ? twi.data() : twf.data();
        for (int i = 0; i < n; i += 2) {                       // first stage: w = 1
            const cf u = x[i], v = x[i + 1];
            x[i] = u + v;
            x[i + 1] = u - v;
        }
        for (int half = 2; half < n; half <<= 1) {
            const cf* w = tw + half;
            for (int i = 0; i < n; i += 2 * half) {
                cf* a = x + i;
                cf* b = x + i + half;
                for (int k = 0; k < half; ++k) {
                    const cf u = a[k], v = b[k] * w[k];
                    a[k] = u + v;
                    b[k] = u - v;
                }
            }
        }
    }
};

constexpr int D = 512;               // DFT block edge

void transpose(const cf* a, cf* b) { // D x D, blocked
    constexpr int B = 16;
    for (int i0 = 0; i0 < D; i0 += B)
        for (int j0 = 0; j0 < D; j0 += B)
            for (int i = i0; i < i0 + B; ++i)
                for (int j = j0; j < j0 + B; ++j) b[(size_t)j * D + i] = a[(size_t)i * D + j];
}

// 2-D transform of a D x D block held row-major in `a` (scratch `t`); rows [0, nrows) are non-zero on input
// (forward) / needed on output (inverse)
void fft2(const Fft& f, cf* a, cf* t, bool inverse, int nrows) {
    if (!inverse) {
        for (int r = 0; r < nrows; ++r) f.run(a + (size_t)r * D, false);
        transpose(a, t);
        for (int c = 0; c < D; ++c) f.run(t + (size_t)c * D, false);
        transpose(t, a);
    } else {
        transpose(a, t);
        for (int c = 0; c < D; ++c) f.run(t + (size_t)c * D, true);
        transpose(t, a);
        for (int r = 0; r < nrows; ++r) f.run(a + (size_t)r * D, true);
    }
}

struct Hit {
    int32_t templ_idx, x, y, w, h;
    float score;
};

}  // namespace

extern "C" {

// Returns 0 on success.  hits: (templ_idx, x, y, w, h, score) records in template order, then descending quality.
// method: 5 TM_CCOEFF_NORMED, 3 TM_CCORR_NORMED, 1 TM_SQDIFF_NORMED (local minima below the threshold).
// seconds_out[0..2] = shared precomputation (image spectra + integral images), per-template phase (wall), total.
int mtm_cpu_find_matches(const uint8_t* img, int rows, int cols, int64_t row_stride, const uint8_t* const* templs,
                         const int32_t* th_, const int32_t* tw_, int n_templ, int method, float thr, int n_threads,
                         void* hits_out, int64_t capacity, int64_t* n_out, double* seconds_out) {
    if (!img || rows <= 0 || cols <= 0 || n_templ < 0 || !n_out || (method != 5 && method != 3 && method != 1)) return -1;
    for (int t = 0; t < n_templ; ++t)
        if (th_[t] <= 0 || tw_[t] <= 0 || th_[t] > rows || tw_[t] > cols || th_[t] > D / 2 || tw_[t] > D / 2) return -1;
    const auto t_start = std::chrono::steady_clock::now();
    n_threads = std::max(1, n_threads);
    const Fft fft(D);
    // ---- shared: integral images (double), as cv::integral(..., CV_64F)
    std::vector<double> ii((size_t)(rows + 1) * (cols + 1), 0.0), ii2((size_t)(rows + 1) * (cols + 1), 0.0);
    for (int y = 0; y < rows; ++y) {
        double rs = 0.0, rs2 = 0.0;
        const uint8_t* p = img + (size_t)y * row_stride;
        for (int x = 0; x < cols; ++x) {
            rs += p[x];
            rs2 += (double)p[x] * p[x];
            ii[(size_t)(y + 1) * (cols + 1) + x + 1] = ii[(size_t)y * (cols + 1) + x + 1] + rs;
            ii2[(size_t)(y + 1) * (cols + 1) + x + 1] = ii2[(size_t)y * (cols + 1) + x + 1] + rs2;
        }
    }
    // ---- shared: spectra of the image blocks.  Blocks step by D - (max template - 1) so that one set serves all.
    int max_h = 1, max_w = 1;
    for (int t = 0; t < n_templ; ++t) {
        max_h = std::max(max_h, (int)th_[t]);
        max_w = std::max(max_w, (int)tw_[t]);
    }
    const int step_y = D - (max_h - 1), step_x = D - (max_w - 1);
    const int nby = (rows - max_h + 1 + step_y - 1) / step_y + ((rows - max_h + 1) <= 0 ? 1 : 0);
    const int nbx = (cols - max_w + 1 + step_x - 1) / step_x + ((cols - max_w + 1) <= 0 ? 1 : 0);
    const int nby_ = std::max(1, (rows - 1) / step_y + 1), nbx_ = std::max(1, (cols - 1) / step_x + 1);
    (void)nby;
    (void)nbx;
    const int NBY = nby_, NBX = nbx_;
    std::vector<cf> spectra((size_t)NBY * NBX * D * D);
    {
        std::atomic<int> next(0);
        auto work = [&]() {
            std::vector<cf> scratch((size_t)D * D);
            for (;;) {
                const int b = next.fetch_add(1);
                if (b >= NBY * NBX) break;
                const int by = (b / NBX) * step_y, bx = (b % NBX) * step_x;
                cf* a = spectra.data() + (size_t)b * D * D;
                int nr = 0;
                for (int r = 0; r < D; ++r) {
                    const int y = by + r;
                    for (int c = 0; c < D; ++c) {
                        const int x = bx + c;
                        a[(size_t)r * D + c] = (y < rows && x < cols) ? cf((float)img[(size_t)y * row_stride + x], 0.f) : cf(0.f, 0.f);
                    }
                    if (y < rows) nr = r + 1;
                }
                fft2(fft, a, scratch.data(), false, nr);
            }
        };
        std::vector<std::thread> pool;
        for (int i = 0; i < n_threads; ++i) pool.emplace_back(work);
        for (auto& th : pool) th.join();
    }
    const auto t_shared = std::chrono::steady_clock::now();
    // ---- one task per template (the reference's thread pool)
    std::vector<std::vector<Hit>> per_t((size_t)n_templ);
    {
        std::atomic<int> next(0);
        auto work = [&]() {
            std::vector<cf> tsp((size_t)D * D), blk((size_t)D * D), scratch((size_t)D * D);
            std::vector<float> map;
            for (;;) {
                const int t = next.fetch_add(1);
                if (t >= n_templ) break;
                const int h = th_[t], w = tw_[t], oh = rows - h + 1, ow = cols - w + 1;
                const uint8_t* T = templs[t];
                // template constants (cv::meanStdDev), spectrum of the zero-padded template
                double s = 0.0, sq = 0.0;
                std::fill(tsp.begin(), tsp.end(), cf(0.f, 0.f));
                for (int y = 0; y < h; ++y)
                    for (int x = 0; x < w; ++x) {
                        const double v = T[(size_t)y * w + x];
                        s += v;
                        sq += v * v;
                        tsp[(size_t)y * D + x] = cf((float)v, 0.f);
                    }
                fft2(fft, tsp.data(), scratch.data(), false, h);
                const double area = (double)h * w, inv_area = 1.0 / area;
                const double mean = s / area, var = std::max(sq / area - mean * mean, 0.0);
                double templ_norm = var, templ_sum2 = var + mean * mean, tmean = mean;
                if (method != 5) {
                    tmean = 0.0;
                    templ_norm = templ_sum2;
                }
                templ_sum2 /= inv_area;
                templ_norm = std::sqrt(templ_norm) / std::sqrt(inv_area);
                const bool all_ones = method == 5 && var < 2.220446049250313e-16;
                map.assign((size_t)oh * ow, 0.f);
                for (int b = 0; b < NBY * NBX; ++b) {
                    const int by = (b / NBX) * step_y, bx = (b % NBX) * step_x;
                    if (by >= oh || bx >= ow) continue;
                    const int vy = std::min(D - h + 1, oh - by), vx = std::min(D - w + 1, ow - bx);     // valid outputs of this block
                    const int uy = std::min(vy, step_y), ux = std::min(vx, step_x);                      // (the rest belongs to the next block)
                    const cf* isp = spectra.data() + (size_t)b * D * D;
                    for (size_t k = 0; k < (size_t)D * D; ++k) blk[k] = isp[k] * std::conj(tsp[k]);
                    fft2(fft, blk.data(), scratch.data(), true, uy);
                    const float scale = 1.0f / ((float)D * (float)D);
                    for (int y = 0; y < uy; ++y) {
                        const int Y = by + y;
                        for (int x = 0; x < ux; ++x) {
                            const int X = bx + x;
                            const double corr = (double)(blk[(size_t)y * D + x].real() * scale);
                            const size_t p0 = (size_t)Y * (cols + 1) + X, p1 = p0 + w, p2 = (size_t)(Y + h) * (cols + 1) + X, p3 = p2 + w;
                            const double s1 = ii[p0] - ii[p1] - ii[p2] + ii[p3], s2 = ii2[p0] - ii2[p1] - ii2[p2] + ii2[p3];
                            double num = corr, wnd_mean2 = 0.0;
                            if (method == 5) {
                                wnd_mean2 = s1 * s1 * inv_area;
                                num -= s1 * tmean;
                            } else if (method == 1) {
                                num = std::max(s2 - 2.0 * num + templ_sum2, 0.0);
                            }
                            const double diff2 = std::max(s2 - wnd_mean2, 0.0);
                            const double tt = diff2 <= std::min(0.5, 10.0 * 1.1920928955078125e-07 * s2) ? 0.0 : std::sqrt(diff2) * templ_norm;
                            double r;
                            if (std::fabs(num) < tt) r = num / tt;
                            else if (std::fabs(num) < tt * 1.125) r = num > 0 ? 1.0 : -1.0;
                            else r = method == 1 ? 1.0 : 0.0;
                            map[(size_t)Y * ow + X] = all_ones ? 1.0f : (float)r;
                        }
                    }
                }
                // peak_local_max(threshold_abs = thr, exclude_border = False), edge-replicating 3x3 maximum filter;
                // minima of the map (method 1) = maxima of its negative above -thr
                std::vector<Hit>& out = per_t[(size_t)t];
                const float sign = method == 1 ? -1.f : 1.f, tq = method == 1 ? -thr : thr;
                bool any_diff = false;
                for (int y = 0; y < oh; ++y)
                    for (int x = 0; x < ow; ++x) {
                        const float v = sign * map[(size_t)y * ow + x];
                        float mx = v;
                        for (int dy = -1; dy <= 1; ++dy)
                            for (int dx = -1; dx <= 1; ++dx) {
                                const int yy = std::min(std::max(y + dy, 0), oh - 1), xx = std::min(std::max(x + dx, 0), ow - 1);
                                mx = std::max(mx, sign * map[(size_t)yy * ow + xx]);
                            }
                        if (v != mx) any_diff = true;
                        else if (v > tq) out.push_back(Hit{t, x, y, w, h, map[(size_t)y * ow + x]});
                    }
                if (!any_diff || oh < 2 || ow < 2) out.clear();          // trivial map: no peaks (2-D maps only are handled here)
                std::stable_sort(out.begin(), out.end(), [&](const Hit& a, const Hit& b) { return sign * a.score > sign * b.score; });
            }
        };
        std::vector<std::thread> pool;
        for (int i = 0; i < std::min(n_threads, std::max(1, n_templ)); ++i) pool.emplace_back(work);
        for (auto& th : pool) th.join();
    }
    const auto t_end = std::chrono::steady_clock::now();
    int64_t total = 0;
    for (const auto& v : per_t) total += (int64_t)v.size();
    *n_out = total;
    if (seconds_out) {
        seconds_out[0] = std::chrono::duration<double>(t_shared - t_start).count();
        seconds_out[1] = std::chrono::duration<double>(t_end - t_shared).count();
        seconds_out[2] = std::chrono::duration<double>(t_end - t_start).count();
    }
    if (total > capacity) return -5;
    Hit* o = static_cast<Hit*>(hits_out);
    for (const auto& v : per_t) {
        if (!v.empty()) std::memcpy(o, v.data(), sizeof(Hit) * v.size());
        o += v.size();
    }
    return 0;
}

}  // extern "C"
