// float32 images: exact (float64) decisions on top of the bf16 matrix-core score kernel.
//
// The reference casts every non-uint8 input to float32 and cv2.matchTemplate correlates it with a float64 DFT
// (MTM/__init__.py:71-74, :92); ncc_f64_kernel reproduces that to rounding, ncc_bf16_kernel is 10x faster and within
// ~1e-5 of it.  A 1e-5 perturbation does not move a score map outside north_star's tolerance - but it decides which
// pixel of a plateau "equals its 3x3 maximum" (skimage.feature.peak_local_max, MTM/__init__.py:45), which of two
// near-ties cv2.minMaxLoc returns (:226), and on which side of score_threshold a borderline score falls.  So the
// fast kernel only SCREENS: whatever could be a peak by its approximate score (a margin of 1e-4 around the
// threshold, 5e-5 around the neighbourhood maximum - five to ten times the kernel's worst observed error) is
// re-scored here with the float64 FMA chain of ncc_f64_kernel - same operations, same order, same statistics
// planes, hence the same float32 value bit for bit - and the decisions are taken on those values.  A few thousand
// windows instead of eight million.
#pragma once
#include "mtm_device_util.hip.h"

namespace mtm {

constexpr float kRefineThrMargin = 1e-4f;   // candidates: approximate quality > threshold - margin * max(1, |threshold|)
constexpr float kRefineNbrTol = 5e-5f;      // potential peaks of a map scan: approximate value >= 3x3 maximum - tolerance

struct RefineParams {
    ImageDev img;
    const TemplDev* td;
    const double* weights;      // float64 template pixels (TemplDev::k1_off), as ncc_f64_kernel reads them
    StatPlanes st;              // statistics planes of the size class being refined (live right after its score launch)
    int method;
    int cls;                    // only records of templates in this size class (TemplDev::cls)
    int ring;                   // 0: one window per record (its score is replaced, and the map value if maps != null);
                                // 1: the record's 3x3 neighbourhood, nine windows, written to the maps (centre: the record too)
    mtm_hit* list;
    const unsigned long long* count;
    unsigned long long cap;
    float* maps;
};

// The score ncc_f64_kernel<false> stores for (template T, output pixel x, y): per channel one float64 FMA chain over
// the template in that kernel's order - 16-row x 32-column chunks, row-major inside a chunk - channel totals added in
// channel order, then finish_unmasked.
__device__ __forceinline__ float exact_score_f32(const RefineParams& p, const TemplDev& T, int x, int y) {
    constexpr int kCh = 16, kCw = 32;           // kF64ChunkH, kF64ChunkW
    const int h = T.rows, w = T.cols;
    double tot = 0.0;
    for (int c = 0; c < p.img.chans; ++c) {
        const float* plane = p.img.f32 + c * p.img.f32_plane + (size_t)y * p.img.f32_pitch + x;
        const double* k1 = p.weights + T.k1_off + (size_t)c * h * w;
        double acc = 0.0;
        for (int cy0 = 0; cy0 < h; cy0 += kCh) {
            const int ch = min(kCh, h - cy0);
            for (int cx0 = 0; cx0 < w; cx0 += kCw) {
                const int cw = min(kCw, w - cx0);
                for (int dy = 0; dy < ch; ++dy) {
                    const float* r = plane + (size_t)(cy0 + dy) * p.img.f32_pitch + cx0;
                    const double* k = k1 + (size_t)(cy0 + dy) * w + cx0;
                    int dx = 0;
                    for (; dx + 8 <= cw; dx += 8) {             // loads of eight taps in flight, one chain
                        float v[8];
                        double kk[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            v[u] = r[dx + u];
                            kk[u] = k[dx + u];
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) acc = fma((double)v[u], kk[u], acc);
                    }
                    for (; dx < cw; ++dx) acc = fma((double)r[dx], k[dx], acc);
                }
            }
        }
        tot += acc;
    }
    return finish_unmasked(p.method, tot, p.st, (size_t)y * p.st.pitch + x, T, p.img.chans);
}

__global__ __launch_bounds__(256) void refine_rescore_kernel(RefineParams p) {
    const unsigned long long n = min(*p.count, p.cap);
    const unsigned long long g = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const unsigned long long i = p.ring ? g / 9 : g;
    if (i >= n) return;
    const mtm_hit rec = p.list[i];
    const TemplDev T = p.td[rec.templ_idx];
    if (T.cls != p.cls) return;
    int x = rec.x, y = rec.y;
    bool centre = true;
    if (p.ring) {
        const int r = (int)(g - i * 9);
        x += r % 3 - 1;
        y += r / 3 - 1;
        centre = r == 4;
        if (x < 0 || x >= T.ow || y < 0 || y >= T.oh) return;
    }
    const float s = exact_score_f32(p, T, x, y);
    if (centre) p.list[i].score = s;
    if (p.maps) p.maps[T.map_off + (size_t)y * T.map_pitch + x] = s;
}

// Map scan of the refined route (peaks_kernel with tolerances): every pixel that COULD be a peak of the exact map - its
// approximate quality is above the lowered threshold and within `tol` of its approximate 3x3 maximum - is appended to
// `list`.  refine_rescore_kernel (ring mode) then replaces the neighbourhoods of the listed pixels by exact scores and
// verify_peaks_kernel takes the decisions.
__global__ __launch_bounds__(256) void refine_scan_kernel(const float* __restrict__ maps, const TemplDev* __restrict__ td,
                                                          const int* __restrict__ tlist, int mode_min, float thr_lo,
                                                          float tol, int border, mtm_hit* __restrict__ list,
                                                          unsigned long long cap, unsigned long long* __restrict__ counter) {
    const int t = tlist[blockIdx.z];
    const TemplDev T = td[t];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xs = blockIdx.x * kPkCols;
    const int y0 = (blockIdx.y * 4 + wave) * kPkRows;
    if (xs >= T.ow || y0 >= T.oh) return;
    const float* m = maps + T.map_off;
    const float padv = (border == MTM_BORDER_CONSTANT) ? 0.0f : -INFINITY;
    const int xb = xs + 4 * lane;
    float va[4], vb[4], vc[4], hl, hr;
    float hm_a[4], hm_b[4], hm_c[4];
    auto hmax = [](const float (&v)[4], float hl, float hr, float (&h)[4]) {
        h[0] = fmaxf(fmaxf(hl, v[0]), v[1]);
        h[1] = fmaxf(fmaxf(v[0], v[1]), v[2]);
        h[2] = fmaxf(fmaxf(v[1], v[2]), v[3]);
        h[3] = fmaxf(fmaxf(v[2], v[3]), hr);
    };
    peaks_load_row(m, T.map_pitch, T.oh, T.ow, y0 - 1, xb, lane, mode_min, padv, va, hl, hr);
    hmax(va, hl, hr, hm_a);
    peaks_load_row(m, T.map_pitch, T.oh, T.ow, y0, xb, lane, mode_min, padv, vb, hl, hr);
    hmax(vb, hl, hr, hm_b);
    const int y1 = min(y0 + kPkRows, T.oh);
    for (int y = y0; y < y1; ++y) {
        peaks_load_row(m, T.map_pitch, T.oh, T.ow, y + 1, xb, lane, mode_min, padv, vc, hl, hr);
        hmax(vc, hl, hr, hm_c);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = xb + k;
            const float v = vb[k];
            const float mx = fmaxf(fmaxf(hm_a[k], hm_b[k]), hm_c[k]);
            mtm_hit hrec;
            hrec.templ_idx = t;
            hrec.x = x;
            hrec.y = y;
            hrec.w = T.cols;
            hrec.h = T.rows;
            hrec.score = mode_min ? -v : v;
            cand_append(x < T.ow && v > thr_lo && v >= mx - tol, counter, cap, list, hrec);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            vb[k] = vc[k];
            hm_a[k] = hm_b[k];
            hm_b[k] = hm_c[k];
        }
    }
}

// N_object == 1 on the refined route: the score kernel listed every output within the margin of the running best of its
// template, refine_rescore_kernel made the listed scores exact; this folds them into the extremum keys
// (cv2.minMaxLoc: first occurrence in row-major order wins ties, NaN never wins) the plain route's kernels produce.
__global__ __launch_bounds__(256) void refine_extremum_kernel(const mtm_hit* __restrict__ list,
                                                              const unsigned long long* __restrict__ count,
                                                              unsigned long long cap, const TemplDev* __restrict__ td,
                                                              int mode_min, unsigned long long* __restrict__ best) {
    const unsigned long long n = min(*count, cap);
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const mtm_hit rec = list[i];
    const float v = rec.score;
    if (v != v) return;
    const uint32_t o = mf_float_order(v);
    const uint32_t idx = (uint32_t)(rec.y * td[rec.templ_idx].ow + rec.x);
    const unsigned long long key = ((unsigned long long)(mode_min ? ~o : o) << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
    atomicMax(&best[2 * rec.templ_idx + (mode_min ? 1 : 0)], key);
}

}  // namespace mtm
