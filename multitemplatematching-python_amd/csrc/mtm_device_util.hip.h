// Inline device helpers shared by every kernel translation unit of libmtm_hip.so: the float64 normalisation
// epilogue (OpenCV common_matchTemplate / matchTemplateMask) and the wave-aggregated candidate append.
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>

#include "mtm_kernels.h"
#include "../../include/mtm_hip.h"

namespace mtm {

// ---------------------------------------------------------------------------------------------
// normalisation epilogue (common_matchTemplate / matchTemplateMask), float64 -> float32
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float finish_unmasked(int method, double corr, const StatPlanes& st,
                                                 size_t sidx, const TemplDev& T, int chans) {
    if (T.all_ones) return 1.0f;
    if (method == MTM_TM_CCORR) return (float)corr;
    const int num_type = (method == MTM_TM_CCORR_NORMED) ? 0
                       : (method == MTM_TM_CCOEFF || method == MTM_TM_CCOEFF_NORMED) ? 1 : 2;
    const bool normed = (method == MTM_TM_SQDIFF_NORMED) || (method == MTM_TM_CCORR_NORMED) ||
                        (method == MTM_TM_CCOEFF_NORMED);
    double num = corr;
    if (num_type == 1) {
#pragma unroll
        for (int c = 0; c < kMaxChans; ++c)
            if (c < chans) num -= st.t[c][sidx] * T.mean[c];
    } else if (num_type == 2) {
        num = st.sum2[sidx] - 2.0 * num + T.templ_sum2;
        num = fmax(num, 0.0);
    }
    if (normed) {
        const double t = st.sq[sidx] * T.templ_norm;
        const double an = fabs(num);
        if (an < t) num = num / t;
        else if (an < t * 1.125) num = (num > 0.0) ? 1.0 : -1.0;
        else num = (method == MTM_TM_SQDIFF_NORMED) ? 1.0 : 0.0;
    }
    return (float)num;
}

__device__ __forceinline__ float finish_masked(int method, double c_i_tm2, double c_i2_m2,
                                               const TemplDev& T) {
    const double tms = T.templ2_mask2_sum;
    double res;
    switch (method) {
        case MTM_TM_SQDIFF:        res = -2.0 * c_i_tm2 + c_i2_m2 + tms; break;
        case MTM_TM_SQDIFF_NORMED: res = (-2.0 * c_i_tm2 + c_i2_m2 + tms) / sqrt(tms * c_i2_m2); break;
        case MTM_TM_CCORR:         res = c_i_tm2; break;
        default:                   res = c_i_tm2 / sqrt(tms * c_i2_m2); break;   // TM_CCORR_NORMED
    }
    return (float)res;
}


// Append `rec` to the candidate list for every lane with `pred` - one atomic per wave (ballot + leader), so that
// smooth score maps with millions of candidates do not serialise on the counter.  Call from wave-convergent or
// divergent code alike (lanes that are not here count as pred = false).  Slots beyond `cap` are counted, not written.
__device__ __forceinline__ void cand_append(bool pred, unsigned long long* counter, unsigned long long cap, mtm_hit* list,
                                            const mtm_hit& rec) {
    const unsigned long long b = __builtin_amdgcn_ballot_w64(pred);
    if (b == 0ull) return;                                      // uniform over the lanes that are here
    const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
    const int leader = (int)__builtin_ctzll(act);
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    unsigned long long base = 0ull;
    if (lane == leader) base = atomicAdd(counter, (unsigned long long)__popcll(b));
    const uint32_t blo = __builtin_amdgcn_readlane((uint32_t)base, leader);
    const uint32_t bhi = __builtin_amdgcn_readlane((uint32_t)(base >> 32), leader);
    if (pred) {
        const unsigned below = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
        const unsigned long long slot = (((unsigned long long)bhi << 32) | blo) + below;
        if (slot < cap) list[slot] = rec;
    }
}

// order-preserving image of a float32 (the high word of the extremum keys, see extremum_kernel) and back
__device__ __forceinline__ uint32_t mf_float_order(float v) {
    if (v == 0.0f) v = 0.0f;     // -0 -> +0
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float mf_order_float(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o ^ 0x80000000u) : ~o);
}

// rows of a score map for the 3x3 scans (peaks_kernel, refine_scan_kernel)
constexpr int kPkCols = 256, kPkRows = 32;

__device__ __forceinline__ void peaks_load_row(const float* __restrict__ m, int pitch, int oh, int ow, int y, int xb,
                                               int lane, bool mode_min, float padv, float (&v)[4], float& hl,
                                               float& hr) {
    // v[k] = value of pixel (y, xb + k); hl / hr = pixels xb - 1 and xb + 4; everything outside the
    // map is the pad value
    if (y < 0 || y >= oh) {
        v[0] = v[1] = v[2] = v[3] = hl = hr = padv;
        return;
    }
    const float* r = m + (size_t)y * pitch;
    if (xb + 3 < ow) {
        const float4 q4 = *reinterpret_cast<const float4*>(r + xb);
        v[0] = q4.x; v[1] = q4.y; v[2] = q4.z; v[3] = q4.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (xb + k < ow) ? r[xb + k] : padv;
    }
    if (mode_min) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (xb + k < ow) ? -v[k] : padv;
    }
    // neighbours across lanes
    const float left = __shfl_up(v[3], 1), right = __shfl_down(v[0], 1);
    hl = left;
    hr = right;
    if (lane == 0) {
        const int x = xb - 1;
        hl = (x >= 0) ? (mode_min ? -r[x] : r[x]) : padv;
    }
    if (lane == 63) {
        const int x = xb + 4;
        hr = (x < ow) ? (mode_min ? -r[x] : r[x]) : padv;
    }
}

// peaks_load_row in two halves, so that a kernel can request a row's pixels rows ahead of using them: peaks_fetch_row
// issues the loads (nothing depends on their values), peaks_finish_row negates for minima and exchanges the edge pixels
// with the neighbouring lanes.  Same values as peaks_load_row.
struct PeakRow {
    float4 q;       // pixels xb .. xb + 3 as loaded (pad value outside the map)
    float el, er;   // pixels xb - 1 / xb + 4 for lanes 0 / 63 (the strip's neighbours in memory), pad value otherwise
};
__device__ __forceinline__ PeakRow peaks_fetch_row(const float* __restrict__ m, int pitch, int oh, int ow, int y, int xb, int lane,
                                                   float padv_raw) {
    PeakRow r;
    r.q = make_float4(padv_raw, padv_raw, padv_raw, padv_raw);
    r.el = r.er = padv_raw;
    if (y < 0 || y >= oh) return r;
    const float* row = m + (size_t)y * pitch;
    if (xb + 3 < ow) {
        r.q = *reinterpret_cast<const float4*>(row + xb);
    } else {
        if (xb < ow) r.q.x = row[xb];
        if (xb + 1 < ow) r.q.y = row[xb + 1];
        if (xb + 2 < ow) r.q.z = row[xb + 2];
    }
    if (lane == 0 && xb - 1 >= 0) r.el = row[xb - 1];
    if (lane == 63 && xb + 4 < ow) r.er = row[xb + 4];
    return r;
}
// padv_raw = the pad value as it would sit in memory (mode_min: the negated pad), so that one negation serves all
__device__ __forceinline__ void peaks_finish_row(const PeakRow& r, int lane, bool mode_min, float (&v)[4], float& hl, float& hr) {
    const float s = mode_min ? -1.0f : 1.0f;
    v[0] = s * r.q.x; v[1] = s * r.q.y; v[2] = s * r.q.z; v[3] = s * r.q.w;
    const float left = __shfl_up(v[3], 1), right = __shfl_down(v[0], 1);
    hl = lane == 0 ? s * r.el : left;
    hr = lane == 63 ? s * r.er : right;
}


// The float32 value of the IEEE quotient num / tt without the IEEE division sequence (round 6; the IEEE-division epilogues of
// the single-channel uint8 score kernel).  Only the float32 rounding of the float64 quotient leaves the kernel, so a float64
// quotient that is a few ulp off gives the same float unless it lies next to a float32 rounding boundary: q0 = num * rr with
// rr = RN(RN(1 / sq) * RN(1 / templ_norm)) carries four roundings, the reference RN(num / RN(sq * templ_norm)) two - they
// differ by <= 6 ulp(double) (mtm_debug_quotient_check measures it: 4).  A float32 rounding boundary is a double whose low 29
// significand bits read 0x10000000; q0 within 32 ulp of one (1.2e-7 of the outputs) takes the division itself.  Zero, the
// 0 * x of a flat window and the unused quotients of |num| >= tt are "far from a boundary" by the same integer test.
// TINY: non-zero quotients below 2^-120 take the division too - there the float is (nearly) denormal and rounds at other
// bits.  The score kernel's epilogues instantiate TINY = false, because their operands cannot produce such a quotient:
//   * num is 0 or |num| >= 2^-53.  TM_CCORR_NORMED / TM_SQDIFF_NORMED: an integer.  TM_CCOEFF_NORMED: corr - RN(S1 mean)
//     with corr and S1 integers; p = RN(S1 mean) is 0 (then corr = 0 too), or an integer, or a double that is not an integer:
//     for p >= 1 integers are multiples of ulp(p), so |corr - p| >= ulp(p) >= 2^-52; for p < 1 either corr = 0 and
//     |num| = p >= S1 mean (1 - 2^-53) >= 1 / (2 area), or |corr| >= 1 and |num| >= 1 - p >= 2^-53; the subtraction of two
//     doubles that far apart does not round below that.
//   * tt = sq templ_norm <= 255^2 area (both are square roots of sums of at most `area` squares of 8-bit values).
//   With area < 2^31 (an int32-indexed image): |q| >= min(2^-53, 2^-32) / 2^47 = 2^-100.
// quotient_check_kernel runs both instantiations, the general one also on quotients in the float32 denormal range.
template <bool TINY = true>
__device__ __forceinline__ bool quotient_needs_division(double q0) {
    const uint32_t lo = (uint32_t)__double2loint(q0), hi = (uint32_t)__double2hiint(q0);
    const bool near_boundary = ((lo & 0x1fffffffu) - (0x10000000u - 32u)) <= 64u;
    if constexpr (!TINY) return near_boundary;
    const bool tiny = ((hi & 0x7fffffffu) - 1u) < (0x38700000u - 1u);   // 0 < |q0| < 2^-120 (hi == 0: 0 or a double denormal -> +-0)
    return near_boundary || tiny;
}
template <bool TINY = true>
__device__ __forceinline__ float quotient_as_float(double num, double tt, double rr) {
    const double q0 = num * rr;
    float qf = (float)q0;
    if (quotient_needs_division<TINY>(q0)) qf = (float)(num / tt);
    return qf;
}

}  // namespace mtm
