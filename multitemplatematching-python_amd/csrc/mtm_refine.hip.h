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
//
// One wave per record.  A chain of w*h*C dependent FMAs cannot be split without changing its rounding, so what the wave
// parallelises is everything around it: all 64 lanes fetch the next chunk of the image patch and of the weights
// (coalesced, a register set in flight) while the chains run out of LDS on the chunk before - one chain in lane 0, or
// (ring mode) the nine chains of a 3x3 neighbourhood in lanes 0..8, which share a patch with a one-pixel ring.
// (A thread per window with loads straight from memory measured 0.35 ms for the 128 candidates of a 4K x 32 call -
// every tap waited for its two loads; this form is bound by the FMA latency, ~25 us.)
constexpr int kRfCh = 16, kRfCw = 32;                       // kF64ChunkH, kF64ChunkW
constexpr int kRfPw = kRfCw + 2, kRfPh = kRfCh + 2;         // patch chunk with the ring
constexpr int kRfPatch = kRfPw * kRfPh;                     // 612 floats
constexpr int kRfPxPerLane = (kRfPatch + 63) / 64;          // 10
constexpr int kRfKPerLane = kRfCh * kRfCw / 64;             // 8

__global__ __launch_bounds__(64) void refine_rescore_kernel(RefineParams p) {
    __shared__ float s_px[2][kRfPatch];
    __shared__ double s_k[2][kRfCh * kRfCw];
    const int lane = threadIdx.x;
    const unsigned long long n = min(*p.count, p.cap);
    const int R = p.ring ? 1 : 0;
    for (unsigned long long i = blockIdx.x; i < n; i += gridDim.x) {
        const mtm_hit rec = p.list[i];
        const TemplDev T = p.td[rec.templ_idx];
        if (T.cls != p.cls) continue;                                   // wave-uniform
        const int h = T.rows, w = T.cols;
        // this lane's window: the record itself (lane 0), or neighbour `lane` of its 3x3 ring (lanes 0..8)
        const int ox = p.ring ? lane % 3 : 0, oy = p.ring ? lane / 3 : 0;      // offsets inside the ringed patch
        const int wx = rec.x + ox - R, wy = rec.y + oy - R;
        const bool mine = lane < (p.ring ? 9 : 1) && wx >= 0 && wx < T.ow && wy >= 0 && wy < T.oh;
        const int ncy = (h + kRfCh - 1) / kRfCh, ncx = (w + kRfCw - 1) / kRfCw;
        const int per_chan = ncy * ncx, n_chunks = per_chan * p.img.chans;
        float fv[kRfPxPerLane];
        double kv[kRfKPerLane];
        auto fetch = [&](int k) {                       // chunk k -> registers (requests only)
            const int c = k / per_chan, r = k - c * per_chan;
            const int cy0 = (r / ncx) * kRfCh, cx0 = (r % ncx) * kRfCw;
            const float* plane = p.img.f32 + c * p.img.f32_plane;
            const double* k1 = p.weights + T.k1_off + (size_t)c * h * w;
#pragma unroll
            for (int u = 0; u < kRfPxPerLane; ++u) {
                const int e = lane + 64 * u;
                const int pr = e / kRfPw, pc = e - pr * kRfPw;
                // rows / columns of the ring that fall outside the image are never used: clamp the address
                const int yy = max(rec.y - R + cy0 + pr, 0), xx = max(rec.x - R + cx0 + pc, 0);
                fv[u] = e < kRfPatch ? plane[(size_t)yy * p.img.f32_pitch + xx] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < kRfKPerLane; ++u) {
                const int e = lane + 64 * u;
                const int dy = e / kRfCw, dx = e - dy * kRfCw;
                kv[u] = (cy0 + dy < h && cx0 + dx < w) ? k1[(size_t)(cy0 + dy) * w + cx0 + dx] : 0.0;
            }
        };
        auto park = [&](int buf) {                      // registers -> LDS buffer
#pragma unroll
            for (int u = 0; u < kRfPxPerLane; ++u) {
                const int e = lane + 64 * u;
                if (e < kRfPatch) s_px[buf][e] = fv[u];
            }
#pragma unroll
            for (int u = 0; u < kRfKPerLane; ++u) s_k[buf][lane + 64 * u] = kv[u];
        };
        __syncthreads();                                // the previous record's chains are done with LDS
        fetch(0);
        park(0);
        __syncthreads();
        double tot = 0.0, acc = 0.0;
        for (int k = 0; k < n_chunks; ++k) {
            const int buf = k & 1;
            if (k + 1 < n_chunks) fetch(k + 1);
            if (mine) {
                const int r = k % per_chan;
                const int cy0 = (r / ncx) * kRfCh, cx0 = (r % ncx) * kRfCw;
                const int ch = min(kRfCh, h - cy0), cw = min(kRfCw, w - cx0);
                const float* px = &s_px[buf][oy * kRfPw + ox];
                const double* kk = &s_k[buf][0];
                for (int dy = 0; dy < ch; ++dy) {
                    int dx = 0;
                    for (; dx + 16 <= cw; dx += 16) {           // 16 operand pairs out of LDS, then their 16 chained FMAs
                        float v[16];
                        double kw[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) {
                            v[u] = px[dy * kRfPw + dx + u];
                            kw[u] = kk[dy * kRfCw + dx + u];
                        }
#pragma unroll
                        for (int u = 0; u < 16; ++u) acc = fma((double)v[u], kw[u], acc);
                    }
                    for (; dx < cw; ++dx) acc = fma((double)px[dy * kRfPw + dx], kk[dy * kRfCw + dx], acc);
                }
                if (r == per_chan - 1) {                // channel done
                    tot += acc;
                    acc = 0.0;
                }
            }
            if (k + 1 < n_chunks) park(buf ^ 1);
            __syncthreads();
        }
        if (mine) {
            const float s = finish_unmasked(p.method, tot, p.st, (size_t)wy * p.st.pitch + wx, T, p.img.chans);
            if (!p.ring || lane == 4) p.list[i].score = s;
            if (p.maps) p.maps[T.map_off + (size_t)wy * T.map_pitch + wx] = s;
        }
    }
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
