// float32 templates WITH a mask (float weights or binary; reference MTM/__init__.py:76-88, :212-219 -> cv2's
// matchTemplateMask, methods TM_SQDIFF and TM_CCORR_NORMED) on the bf16 matrix cores - round 6.
//
// cv2 needs two correlations per output:  c1 = sum I * (T M^2)  and  c2 = sum I^2 * M^2, and returns
//   TM_CCORR_NORMED: c1 / sqrt(tms c2)        TM_SQDIFF: -2 c1 + c2 + tms        (tms = sum (T M)^2).
// ncc_f64_kernel<true> computes both as float64 FMA chains at the fp64 vector rate: 50 ms at 4K x 32 templates.  Here
// ncc_bf16_kernel - the unmasked kernel, as it is - runs twice in raw (TM_CCORR) mode: image I against the templates
// U = T M^2, image J = float32(I^2) against the "templates" V = M^2.  Its scores are a SCREEN, exactly as on the
// unmasked float32 routes: every output carries a rigorous bound of what the two approximate sums can be off by,
//   E1 = eps sqrt(sum (I - mu_I)^2  sum (U - mean U)^2),   E2 = eps sqrt(sum (J - mu_J)^2  sum (V - mean V)^2) + 2^-23 c2
// (Cauchy-Schwarz over the dropped piece products, the 16-bit representations and the float32 accumulation - eps =
// bf16_rig_eps, already doubled; mu the constant each work item subtracted, handed over in Bf16Params::mu_out; the last term
// the float32 rounding of J), maskf32_combine_kernel turns them into an UPPER bound of the output's quality, and
//   * an output whose upper bound stays below the threshold gets "below" (-inf; minima: +inf) in the score map: a value
//     certainly below the threshold can neither be a peak nor beat one, whatever it is exactly;
//   * every other output is listed and refine_rescore_masked_kernel replaces it by ncc_f64_kernel<true>'s own value -
//     same FMA chains, same order, bit for bit.
// The peak pass then runs on a map that equals the float64 kernel's wherever that matters: identical hit lists by
// construction.  N_object == 1 (cv2.minMaxLoc) works the same way with the templates' own best LOWER bound in the place of the
// threshold (two passes: maskf32_best_kernel, maskf32_list_best_kernel) - the exact extremum and every exact tie with it are
// re-scored, the extremum search over the maps does the rest.  The maps are not publishable (mtm_last_score_map refuses);
// mtm_score_map keeps the float64 kernel.  A list that overflows sends the class to the float64 kernel for this call.
#pragma once
#include "mtm_device_util.hip.h"
#include "mtm_refine.hip.h"

namespace mtm {

// J = float32(I * I), same geometry as the float32 plane (the zero padding stays zero)
__global__ __launch_bounds__(256) void square_f32_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = reinterpret_cast<const float4*>(in)[i];
    reinterpret_cast<float4*>(out)[i] = make_float4(v.x * v.x, v.y * v.y, v.z * v.z, v.w * v.w);
}

struct MaskF32Params {
    const float* m1;            // approximate c1 / c2 maps: template t at td_u[t].map_off / td_v[t].map_off
    const float* m2;
    const TemplDev* td;         // the class's real table (tms, the final map's offset and pitch)
    const TemplDev* td_u;       // ... with centred_sum2 = sum (U - mean U)^2 / sum (V - mean V)^2
    const TemplDev* td_v;
    const int* tlist;
    const double* s1i; const double* s2i;     // window sums of I and I^2 (float64)
    const double* s1j; const double* s2j;     // ... of J and J^2
    int st_pitch;
    const float* mu_i; const float* mu_j;     // the launches' tile constants, [yb][seg]
    int nseg;                   // segments (kBfSeg outputs) per row
    int method, mode_min;
    float thr;                  // score_threshold as float32 (what the peak pass compares the float32 scores with)
    float eps;
    int h, w, oh, ow;
    float* maps;                // the final score maps
    mtm_hit* list;
    unsigned long long* counter;
    unsigned long long cap;
};

// What the bounds need of the PIXEL (template-independent): sum (I - mu_I)^2 and sum (J - mu_J)^2 with their own cancellation
// covered, as in ncc_bf16_kernel's epilogue.  One thread owns a pixel and walks the class's templates: the four statistics
// planes are read once per pixel, not once per (pixel, template) - 10 GB of traffic at 4K x 32 otherwise.
struct MaskF32Pixel {
    double si, sj;
};
__device__ __forceinline__ MaskF32Pixel maskf32_pixel(const MaskF32Params& p, int x, int y) {
    const size_t sidx = (size_t)y * p.st_pitch + x;
    const size_t tile = (size_t)(y / kBfRows) * p.nseg + x / kBfSeg;
    const double mi = (double)p.mu_i[tile], mj = (double)p.mu_j[tile];
    const double area = (double)p.h * (double)p.w;
    const double s2i = p.s2i[sidx], s2j = p.s2j[sidx];
    double si = s2i + mi * (area * mi - 2.0 * p.s1i[sidx]);
    double sj = s2j + mj * (area * mj - 2.0 * p.s1j[sidx]);
    MaskF32Pixel o;
    o.si = fmax(si, 0.0) * 1.000001 + 1e-12 * (fabs(s2i) + area * mi * mi);
    o.sj = fmax(sj, 0.0) * 1.000001 + 1e-12 * (fabs(s2j) + area * mj * mj);
    return o;
}
// Bounds of the QUALITY (the score; minima: minus the score) of output (x, y) of template t from the two approximate sums:
// *lb <= exact quality <= *ub for finite inputs (+-inf where a bound does not exist: c2 - E2 <= 0, tms <= 0).
__device__ __forceinline__ void maskf32_bounds(const MaskF32Params& p, int t, const TemplDev& T, const MaskF32Pixel& px, int x, int y,
                                               float* approx, double* lb, double* ub) {
    const TemplDev& U = p.td_u[t];
    const TemplDev& V = p.td_v[t];
    const double c1 = (double)p.m1[U.map_off + (size_t)y * U.map_pitch + x];
    const double c2 = (double)p.m2[V.map_off + (size_t)y * V.map_pitch + x];
    const double e1 = (double)p.eps * sqrt(px.si * U.centred_sum2) * 1.000002 + 3e-7 * fabs(c1) + 1e-30;
    const double e2 = (double)p.eps * sqrt(px.sj * V.centred_sum2) * 1.000002 + 6e-7 * fabs(c2) + 1e-30;
    const double tms = T.templ2_mask2_sum;
    if (p.method == MTM_TM_SQDIFF) {
        // minima: quality = -score, score = -2 c1 + c2 + tms
        const double s = -2.0 * c1 + c2 + tms;
        const double slack = (2.0 * e1 + e2) + 4e-7 * (2.0 * fabs(c1) + fabs(c2) + fabs(tms));
        *ub = -(s - slack);
        *lb = -(s + slack);
        *approx = (float)s;
        return;
    }
    // TM_CCORR_NORMED: c1 / sqrt(tms c2), no guards (0 / 0 is NaN, as in OpenCV)
    const double num_hi = c1 + e1, num_lo = c1 - e1;
    const double c2_lo = c2 - e2, c2_hi = c2 + e2;
    if (!(tms > 0.0)) {
        *ub = INFINITY;
        *lb = -INFINITY;
    } else {
        if (num_hi <= 0.0) *ub = c2_hi > 0.0 ? num_hi / sqrt(tms * c2_hi) : 0.0;
        else *ub = c2_lo > 0.0 ? num_hi / sqrt(tms * c2_lo) : INFINITY;
        if (num_lo >= 0.0) *lb = c2_hi > 0.0 ? num_lo / sqrt(tms * c2_hi) : 0.0;
        else *lb = c2_lo > 0.0 ? num_lo / sqrt(tms * c2_lo) : -INFINITY;
        *ub += 4e-7 * fmax(1.0, fabs(*ub));          // (the float32 rounding of the exact score)
        *lb -= 4e-7 * fmax(1.0, fabs(*lb));
    }
    *approx = (float)(c1 / sqrt(tms * fmax(c2, 1e-300)));
}

// Local extrema against a threshold: "below" placeholders where the upper bound stays below it, the rest listed.
// Grid: (ceil(ow / 256), oh); every thread walks the n_list templates of the class for its pixel.
__global__ __launch_bounds__(256) void maskf32_combine_kernel(MaskF32Params p, int n_list) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    const bool on = x < p.ow && y < p.oh;
    const MaskF32Pixel px = maskf32_pixel(p, min(x, p.ow - 1), min(y, p.oh - 1));
    // quality threshold: the score threshold for maxima, minus it for minima (a hit needs quality > threshold in float32)
    const double thr_q = p.mode_min ? -(double)p.thr : (double)p.thr;
    for (int li = 0; li < n_list; ++li) {
        const int t = p.tlist[li];
        const TemplDev& T = p.td[t];
        bool list_it = false;
        float approx = 0.0f;
        if (on) {
            double lb, ub;
            maskf32_bounds(p, t, T, px, x, y, &approx, &lb, &ub);
            list_it = !(ub < thr_q);         // (inputs that are not finite: the comparison is false - listed, the exact chain decides)
            if (!list_it) p.maps[T.map_off + (size_t)y * T.map_pitch + x] = p.mode_min ? INFINITY : -INFINITY;
        }
        mtm_hit rec;
        rec.templ_idx = t;
        rec.x = x;
        rec.y = y;
        rec.w = p.w;
        rec.h = p.h;
        rec.score = approx;
        cand_append(on && list_it, p.counter, p.cap, p.list, rec);
    }
}

// N_object == 1 (cv2.minMaxLoc), pass 1: the best LOWER bound of the quality per template (`best[t]`: an ordered-float key,
// atomicMax; NaN never takes part), and every output's UPPER bound parked in its score-map slot (rounded up).
__global__ __launch_bounds__(256) void maskf32_best_kernel(MaskF32Params p, int n_list, unsigned int* __restrict__ best) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    const bool on = x < p.ow && y < p.oh;
    const MaskF32Pixel px = maskf32_pixel(p, min(x, p.ow - 1), min(y, p.oh - 1));
    for (int li = 0; li < n_list; ++li) {
        const int t = p.tlist[li];
        const TemplDev& T = p.td[t];
        uint32_t key = 0u;
        if (on) {
            float approx;
            double lb, ub;
            maskf32_bounds(p, t, T, px, x, y, &approx, &lb, &ub);
            float ubf = (float)ub;
            if ((double)ubf < ub) ubf = nextafterf(ubf, INFINITY);
            if (!(ub == ub)) ubf = INFINITY;                      // not finite: to be re-scored whatever the others say
            p.maps[T.map_off + (size_t)y * T.map_pitch + x] = ubf;
            float lbf = (float)lb;
            if ((double)lbf > lb) lbf = nextafterf(lbf, -INFINITY);
            if (lb == lb) key = mf_float_order(lbf);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t other = (uint32_t)__shfl_down((int)key, off);
            key = other > key ? other : key;
        }
        // (a plain look first: 4 M waves x 32 templates of atomics on 32 addresses cost 25 ms at 4K x 32 - same-address
        // atomics serialise at ~9 ns apiece -, and hardly any wave improves on what is already there)
        if ((threadIdx.x & 63) == 0 && key && key > __hip_atomic_load(&best[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(&best[t], key);
    }
}
// ... pass 2: every output whose upper bound reaches its template's best lower bound is listed (the exact extremum and
// every exact tie with it are among them), the others get the placeholder no extremum search can pick.
__global__ __launch_bounds__(256) void maskf32_list_best_kernel(MaskF32Params p, int n_list, const unsigned int* __restrict__ best) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    const bool on = x < p.ow && y < p.oh;
    for (int li = 0; li < n_list; ++li) {
        const int t = p.tlist[li];
        const TemplDev& T = p.td[t];
        bool list_it = false;
        if (on) {
            float* slot = p.maps + T.map_off + (size_t)y * T.map_pitch + x;
            const float ub = *slot;
            const uint32_t bk = best[t];
            const float lb_best = bk ? mf_order_float(bk) : -INFINITY;
            list_it = !(ub < lb_best);
            if (!list_it) *slot = p.mode_min ? INFINITY : -INFINITY;
        }
        mtm_hit rec;
        rec.templ_idx = t;
        rec.x = x;
        rec.y = y;
        rec.w = p.w;
        rec.h = p.h;
        rec.score = 0.0f;
        cand_append(on && list_it, p.counter, p.cap, p.list, rec);
    }
}

// refine_rescore_kernel for masked templates: the two FMA chains of ncc_f64_kernel<true> - c1 over K1 = T M^2, c2 over
// K2 = M^2 with the float64 square of the pixel - in that kernel's order (16-row x 32-column chunks, row-major inside a
// chunk, one accumulator pair across the chunks of a channel), then finish_masked.  One wave per record; the record's
// window only (no ring).  The exact score goes to the record and to the score map.
__global__ __launch_bounds__(64) void refine_rescore_masked_kernel(RefineParams p) {
    __shared__ float s_px[2][kRfCh * kRfCw];
    __shared__ double s_k1[2][kRfCh * kRfCw];
    __shared__ double s_k2[2][kRfCh * kRfCw];
    const int lane = threadIdx.x;
    const unsigned long long n = min(*p.count, p.cap);
    constexpr int kPer = kRfCh * kRfCw / 64;          // 8 elements of a chunk per lane
    for (unsigned long long i = blockIdx.x; i < n; i += gridDim.x) {
        const mtm_hit rec = p.list[i];
        const TemplDev T = p.td[rec.templ_idx];
        if (T.cls != p.cls) continue;                                   // wave-uniform
        const int h = T.rows, w = T.cols;
        const int ncy = (h + kRfCh - 1) / kRfCh, ncx = (w + kRfCw - 1) / kRfCw;
        const int per_chan = ncy * ncx, n_chunks = per_chan * p.img.chans;
        float fv[kPer];
        double k1v[kPer], k2v[kPer];
        auto fetch = [&](int k) {
            const int c = k / per_chan, r = k - c * per_chan;
            const int cy0 = (r / ncx) * kRfCh, cx0 = (r % ncx) * kRfCw;
            const float* plane = p.img.f32 + c * p.img.f32_plane;
            const double* k1 = p.weights + T.k1_off + (size_t)c * h * w;
            const double* k2 = p.weights + T.k2_off + (size_t)c * h * w;
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const int e = lane + 64 * u;
                const int dy = e / kRfCw, dx = e - dy * kRfCw;
                const bool in = cy0 + dy < h && cx0 + dx < w;
                fv[u] = in ? plane[(size_t)(rec.y + cy0 + dy) * p.img.f32_pitch + rec.x + cx0 + dx] : 0.0f;
                k1v[u] = in ? k1[(size_t)(cy0 + dy) * w + cx0 + dx] : 0.0;
                k2v[u] = in ? k2[(size_t)(cy0 + dy) * w + cx0 + dx] : 0.0;
            }
        };
        auto park = [&](int buf) {
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                s_px[buf][lane + 64 * u] = fv[u];
                s_k1[buf][lane + 64 * u] = k1v[u];
                s_k2[buf][lane + 64 * u] = k2v[u];
            }
        };
        __syncthreads();                                // the previous record's chains are done with LDS
        fetch(0);
        park(0);
        __syncthreads();
        double tot1 = 0.0, tot2 = 0.0, a1 = 0.0, a2 = 0.0;
        for (int k = 0; k < n_chunks; ++k) {
            const int buf = k & 1;
            if (k + 1 < n_chunks) fetch(k + 1);
            if (lane == 0) {
                const int r = k % per_chan;
                const int cy0 = (r / ncx) * kRfCh, cx0 = (r % ncx) * kRfCw;
                const int ch = min(kRfCh, h - cy0), cw = min(kRfCw, w - cx0);
                for (int dy = 0; dy < ch; ++dy)
                    for (int dx = 0; dx < cw; ++dx) {
                        const double v = (double)s_px[buf][dy * kRfCw + dx];
                        a1 = fma(v, s_k1[buf][dy * kRfCw + dx], a1);
                        a2 = fma(v * v, s_k2[buf][dy * kRfCw + dx], a2);
                    }
                if (r == per_chan - 1) {                // channel done
                    tot1 += a1;
                    tot2 += a2;
                    a1 = a2 = 0.0;
                }
            }
            if (k + 1 < n_chunks) park(buf ^ 1);
            __syncthreads();
        }
        if (lane == 0) {
            const float s = finish_masked(p.method, tot1, tot2, T);
            p.list[i].score = s;
            if (p.maps) p.maps[T.map_off + (size_t)rec.y * T.map_pitch + rec.x] = s;
        }
    }
}

}  // namespace mtm
