// MFMA score-map kernel: the uint8 sliding-window correlation as an implicit GEMM on the int8
// matrix cores (v_mfma_i32_16x16x64_i8), exact integer arithmetic.
//
//   out[t][y][x] = sum_{dy,dx} I[y+dy][x+dx] * T[t][dy][dx]           (uint8 x uint8)
//
// Signed operands: the matrix cores multiply int8, so both operands are biased by -128
// (I' = I ^ 0x80, T' = T ^ 0x80) and the exact correlation is recovered in the epilogue from the
// window sum S1 the statistics pass already provides:
//   sum I*T = sum I'*T' + 128*S1 + 128*sum(T) - 16384*w*h*C
// |sum I'T'| <= 16384*w*h*C must fit int32: the launcher only takes w*h*C <= 131071.
//
// GEMM mapping (one v_mfma_i32_16x16x64_i8 = 16 templates x 16 pixels x 64 taps):
//   A (16 x 64)  = one 64-tap template-row segment of 16 templates; lane (i = lane&15, q = lane>>4)
//                  holds taps 16q..16q+15 of template i: 16 contiguous bytes, pre-packed on the host
//                  in exactly this order (coalesced 1 KiB global_load_dwordx4 per operand).
//   B (64 x 16)  = the matching image bytes of 16 output pixels; lane (j = lane&15, q) holds
//                  I'[y+dy][x_j + 16q .. +15].
//   The pairing only relies on A and B using the same (q, byte) slot for the same tap.
//   The 16 pixels of one MFMA are 16 apart: x_j = x0 + 16 j + c, "phase" c = 0..15.  All 16 phases
//   of a lane read the SAME 32 bytes of LDS (two aligned 16-byte chunks j+q and j+q+1), shifted by
//   c bytes: 2 ds_read_b128 + 21 v_alignbyte_b32 feed 16*MB MFMAs.  A wave therefore owns
//   256 consecutive output pixels of one row for 16*MB templates (64*MB accumulator registers).
//   * work-group = 4 waves = 4 consecutive output rows of the same 256-pixel segment, sharing one
//     LDS image tile (biased to int8 while staging); template rows beyond 64 are processed in
//     64-row chunks (tile re-staged), widths beyond 64 in 64-tap blocks.
//   * epilogue: accumulators go through LDS (transposed) so that consecutive lanes own consecutive
//     output columns - coalesced statistics loads and score-map stores - then the float64
//     normalisation shared with the other kernels (finish_unmasked).
#pragma once
#include "mtm_device_util.hip.h"
#include "mtm_mfma_params.h"

namespace mtm {

__device__ __forceinline__ int mf_epi_rot(int j) { return 4 * ((j >> 1) & 3); }
// A launch parameter read where it is used: as a plain `p.x` the compiler evaluates the test once ahead of the item loop and,
// out of scalar registers there, parks the boolean in a VECTOR register across the K loop - i.e. in scratch memory (one dword
// per lane stored per work-group, reloaded per work item, in the headline instantiation).
__device__ __forceinline__ int mf_opaque_sgpr(int v) {
    asm volatile("" : "+s"(v));
    return v;
}
// finish_unmasked on values already in registers (same arithmetic, same order).  METHOD >= 0 fixes
// the matching method at compile time: the per-output code is then branch-free apart from the
// data-dependent normalisation cases (the runtime-method version spends most of its time in
// wave-uniform scalar branches).
template <int METHOD>
__device__ __forceinline__ float finish_vals(int method_rt, double corr, const double (&t)[kMaxChans], double sum2,
                                             double sq, const MfTemplConst& T, int chans) {
    const int method = METHOD >= 0 ? METHOD : method_rt;
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
            if (c < chans) num -= t[c] * T.mean[c];
    } else if (num_type == 2) {
        num = sum2 - 2.0 * num + T.templ_sum2;
        num = fmax(num, 0.0);
    }
    if (normed) {
        const double tt = sq * T.templ_norm;
        const double an = fabs(num);
        const double other = (method == MTM_TM_SQDIFF_NORMED) ? 1.0 : 0.0;
        const double sat = (an < tt * 1.125) ? ((num > 0.0) ? 1.0 : -1.0) : other;
        num = (an < tt) ? num / tt : sat;
    }
    return (float)num;
}

// Lean single-channel epilogue with the method fixed at compile time.  Same quantities and the same
// case analysis as finish_unmasked / OpenCV's common_matchTemplate; the exact integer correlation is
// rebuilt from the biased MFMA accumulator (a32 + 128*S1 + K, all exact in float64).  EXACT_DIV keeps
// the IEEE division num / t (bit-identical to the other kernels and to the oracle); otherwise the
// quotient is num * (1/sq) * (1/templ_norm) with both reciprocals correctly rounded: <= 2 ulp(double)
// from the exact quotient, i.e. the float32 result differs in the last bit for ~1e-8 of the pixels.
template <int METHOD, bool EXACT_DIV>
__device__ __forceinline__ float finish_lean(int a32, double s1, double p1, double sum2, double sq, double rsq,
                                             const MfTemplConst& T) {
    constexpr bool normed = METHOD == MTM_TM_SQDIFF_NORMED || METHOD == MTM_TM_CCORR_NORMED ||
                            METHOD == MTM_TM_CCOEFF_NORMED;
    const double corr = (double)a32 + (p1 + T.mfma_k);
    double num = corr;
    if (METHOD == MTM_TM_CCOEFF || METHOD == MTM_TM_CCOEFF_NORMED) num = corr - s1 * T.mean[0];
    if (METHOD == MTM_TM_SQDIFF || METHOD == MTM_TM_SQDIFF_NORMED) num = fmax(sum2 - 2.0 * corr + T.templ_sum2, 0.0);
    if (!normed) return (float)num;
    const double tt = sq * T.templ_norm;
    const double an = fabs(num);
    const float qf = EXACT_DIV ? (float)(num / tt) : (float)(num * (rsq * T.rtempl_norm));
    const float satf = (num > 0.0) ? 1.0f : -1.0f;
    const float other = (METHOD == MTM_TM_SQDIFF_NORMED) ? 1.0f : 0.0f;
    return (an < tt) ? qf : ((an < tt * 1.125) ? satf : other);
}

// (quotient_as_float: mtm_device_util.hip.h)
// finish_lean with the quotient computed unconditionally (the empty asm keeps the compiler from
// sinking it into a divergent branch): straight-line code, the four pixels of a lane interleave.
// Same operations in the same order as finish_lean: bit-identical results.
// DEFER (IEEE-division builds, one channel): the quotient is quotient_as_float's reciprocal product and *redo says whether
// it sits next to a float32 rounding boundary - the caller repeats those through the division behind ONE wave-uniform
// branch for the lane's four pixels (as a per-lane `if` the compiler turned the division into a select and computed both).
template <int METHOD, bool EXACT_DIV, int CH = 1, bool DEFER = false>
__device__ __forceinline__ float finish_fast(int a32, const double (&s1)[CH], double p1, double sum2, double sq,
                                             double rsq, const MfTemplConst& T, bool* redo = nullptr, bool* sat = nullptr) {
    constexpr bool normed = METHOD == MTM_TM_SQDIFF_NORMED || METHOD == MTM_TM_CCORR_NORMED ||
                            METHOD == MTM_TM_CCOEFF_NORMED;
    if constexpr (!EXACT_DIV && (METHOD == MTM_TM_CCORR_NORMED || METHOD == MTM_TM_CCOEFF_NORMED)) {
        // Default (reciprocal) mode of the two north-star methods: 5-6 float64 operations per output.
        //   num = a32 + K + 128 S1 [- mean S1]  as one fma on the biased accumulator,
        //   q   = num * (1/sq) * (1/templ_norm)   (both reciprocals correctly rounded; 0 for a flat
        //         window or a constant template, which yields the 0 OpenCV returns for t == 0),
        //   |q| < 1 -> q,  |q| < 1.125 -> +-1,  else 0: the case analysis of common_matchTemplate
        //   evaluated on the float32 quotient (it only differs from the float64 comparison when q is
        //   within one float64 rounding of 1.125; at 1 both give +-1.0f).
        double num = (double)a32 + T.mfma_k;
#pragma unroll
        for (int cc = 0; cc < CH; ++cc) num = fma(s1[cc], METHOD == MTM_TM_CCOEFF_NORMED ? T.m128[cc] : 128.0, num);
        const float qf = (float)(num * (rsq * T.rtempl_norm));
        const float aq = fabsf(qf);
        const float sat = (aq < 1.125f) ? copysignf(1.0f, qf) : 0.0f;
        return (aq < 1.0f) ? qf : sat;
    }
    const double corr = (double)a32 + (p1 + T.mfma_k);
    double num = corr;
    if (METHOD == MTM_TM_CCOEFF || METHOD == MTM_TM_CCOEFF_NORMED) {
#pragma unroll
        for (int cc = 0; cc < CH; ++cc) num -= s1[cc] * T.mean[cc];      // same order as finish_unmasked
    }
    if (METHOD == MTM_TM_SQDIFF || METHOD == MTM_TM_SQDIFF_NORMED) num = fmax(sum2 - 2.0 * corr + T.templ_sum2, 0.0);
    if (!normed) return (float)num;
    const double tt = sq * T.templ_norm;
    // (IEEE-division builds of the normalised methods defer; three channels with the function's general form - 16-44 B more
    // scratch in eight of those instantiations, RGB map-mode kernel 2.83 -> 2.68 ms at 4K x 32)
    float qf;
    if constexpr (EXACT_DIV && DEFER) {
        // |num| >= t (a flat window, a saturated quotient: the rules' constants, finish_saturated) is the caller's second
        // wave-uniform branch - the comparison with 1.125 t and the selects leave the common path as well
        const double q0 = num * (rsq * T.rtempl_norm);
        *redo = quotient_needs_division<(CH > 1)>(q0);   // (one channel: no tiny quotients from these operands, mtm_device_util.hip.h)
        *sat = !(fabs(num) < tt);
        return (float)q0;
    } else {
        qf = EXACT_DIV ? (float)(num / tt) : (float)(num * (rsq * T.rtempl_norm));
    }
    asm volatile("" : "+v"(qf));
    const double an = fabs(num);
    const float satf = (num > 0.0) ? 1.0f : -1.0f;
    const float other = (METHOD == MTM_TM_SQDIFF_NORMED) ? 1.0f : 0.0f;
    const float r2 = (an < tt * 1.125) ? satf : other;
    return (an < tt) ? qf : r2;
}

// What finish_fast returns where |num| >= t (same quantities, same order): +-1 below 1.125 t, else the rules' constant.
template <int METHOD, int CH = 1>
__device__ __forceinline__ float finish_saturated(int a32, const double (&s1)[CH], double p1, double sum2, double sq,
                                                  const MfTemplConst& T) {
    const double corr = (double)a32 + (p1 + T.mfma_k);
    double num = corr;
    if (METHOD == MTM_TM_CCOEFF || METHOD == MTM_TM_CCOEFF_NORMED) {
#pragma unroll
        for (int cc = 0; cc < CH; ++cc) num -= s1[cc] * T.mean[cc];
    }
    if (METHOD == MTM_TM_SQDIFF || METHOD == MTM_TM_SQDIFF_NORMED) num = fmax(sum2 - 2.0 * corr + T.templ_sum2, 0.0);
    const double tt = sq * T.templ_norm;
    const float satf = (num > 0.0) ? 1.0f : -1.0f;
    const float other = (METHOD == MTM_TM_SQDIFF_NORMED) ? 1.0f : 0.0f;
    return (fabs(num) < tt * 1.125) ? satf : other;
}

// finish_lean with the method chosen at run time (one channel): `corr` is the exact correlation.
template <bool EXACT_DIV>
__device__ __forceinline__ float finish_rt(int method, double corr, double s1, double sum2, double sq, double rsq,
                                           const MfTemplConst& T) {
    const bool normed = method == MTM_TM_SQDIFF_NORMED || method == MTM_TM_CCORR_NORMED || method == MTM_TM_CCOEFF_NORMED;
    double num = corr;
    if (method == MTM_TM_CCOEFF || method == MTM_TM_CCOEFF_NORMED) num = corr - s1 * T.mean[0];
    if (method == MTM_TM_SQDIFF || method == MTM_TM_SQDIFF_NORMED) num = fmax(sum2 - 2.0 * corr + T.templ_sum2, 0.0);
    if (!normed) return (float)num;
    const double tt = sq * T.templ_norm;
    float qf = EXACT_DIV ? (float)(num / tt) : (float)(num * (rsq * T.rtempl_norm));
    asm volatile("" : "+v"(qf));
    const double an = fabs(num);
    const float satf = (num > 0.0) ? 1.0f : -1.0f;
    const float other = (method == MTM_TM_SQDIFF_NORMED) ? 1.0f : 0.0f;
    const float r2 = (an < tt * 1.125) ? satf : other;
    return (an < tt) ? qf : r2;
}

// Masked templates (OpenCV's matchTemplateMask, binary uint8 mask, reference MTM/__init__.py:78,:216):
// c1 = sum I*(T*M) comes from the MFMA accumulator exactly like an unmasked correlation (the packed
// template is T*M), c2 = sum I^2*M from the class's raw row-multiplexed pass (MfmaParams::sq_fused; MTM_ROW_MUX=0: the MASKSQ dot4 pass).  No guards, as in OpenCV: 0/0 is NaN.
// DEFER (IEEE builds, round 6): the float32 value of num / sqrt(tms c2) from num * (RN(1 / sqrt(c2)) * RN(1 / sqrt(tms))) - six
// roundings against the reference's three, <= 9 ulp(double) apart - with quotient_needs_division's general form; a quotient
// that is not finite (c2 == 0: inf or the 0 / 0 NaN of OpenCV) takes the reference sequence as well, so that even the NaN's bits
// are the division's.  *redo: the caller repeats those behind one wave-uniform branch (see finish_fast).
template <int METHOD, bool EXACT_DIV, bool DEFER = false>
__device__ __forceinline__ float finish_lean_masked(int a32, double p1, double c2, double rsqrt_c2,
                                                    const MfTemplConst& T, bool* redo = nullptr) {
    const double c1 = (double)a32 + (p1 + T.mfma_k);
    if (METHOD == MTM_TM_CCORR) return (float)c1;
    if (METHOD == MTM_TM_SQDIFF) return (float)(-2.0 * c1 + c2 + T.tms);
    const double num = (METHOD == MTM_TM_SQDIFF_NORMED) ? (-2.0 * c1 + c2 + T.tms) : c1;
    if constexpr (EXACT_DIV && DEFER) {
        const double q0 = num * (rsqrt_c2 * T.rsqrt_tms);
        const bool finite = ((uint32_t)__double2hiint(q0) & 0x7ff00000u) != 0x7ff00000u;
        *redo = quotient_needs_division<true>(q0) || !finite;
        return (float)q0;
    }
    return EXACT_DIV ? (float)(num / sqrt(T.tms * c2)) : (float)(num * (rsqrt_c2 * T.rsqrt_tms));
}

// One K step: 16 phases x MB template groups, operands already in registers.
template <int MB>
__device__ __forceinline__ void mfma_step(v4i (&acc)[MB][16], const v4i qa, const v4i qb, const v4i (&a)[MB]) {
    const int W[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
    // phases with a whole-dword shift: no VALU work
#pragma unroll
    for (int cq = 0; cq < 4; ++cq) {
        const v4i bv = {W[cq], W[cq + 1], W[cq + 2], W[cq + 3]};
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
            acc[mb][4 * cq] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[mb], bv, acc[mb][4 * cq], 0, 0, 0);
    }
    // byte shifts 1..3: 7 v_alignbyte_b32 serve four phases each
#pragma unroll
    for (int cr = 1; cr < 4; ++cr) {
        int E[7];
#pragma unroll
        for (int m = 0; m < 7; ++m) E[m] = (int)__builtin_amdgcn_alignbyte((uint32_t)W[m + 1], (uint32_t)W[m], cr);
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) {
            const v4i bv = {E[cq], E[cq + 1], E[cq + 2], E[cq + 3]};
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
                acc[mb][4 * cq + cr] =
                    __builtin_amdgcn_mfma_i32_16x16x64_i8(a[mb], bv, acc[mb][4 * cq + cr], 0, 0, 0);
        }
    }
}

#ifndef MTM_MFMA_NO_ASM
#include "mtm_mfma_step_asm.inc"
#else
__device__ __forceinline__ void mfma_step_one_set(v4i (&acc)[2][16], const v4i qa, const v4i qb, const v4i (&a)[2]) {
    mfma_step<2>(acc, qa, qb, a);
}
#endif
// The K step of two MFMA groups.  MTM_STEP_VARIANT (build-time experiment switch): 2 = the whole step as ONE asm
// statement with the odd-offset windows computed from the operand (default; 44 VALU), 1 = one statement with the odd-offset
// windows copied from the even ones (47 VALU), 0 = round 2's four statements.  Measured on one box, 4K x 32 templates,
// kernel per step banded / single launch (profiles/r04a/lib_ab.txt): 0: 0.6636 / 0.6262 ms at 1943 / 2036 MHz,
// 1: 0.6499 / 0.6131 at 1893 / 1982, 2: 0.6490 / 0.6116 at 1873 / 1967 - 6 % fewer cycles, 2.3 % less time (the chip
// clocks to its power budget); register-only step: 17.4 -> 16.2 cycles per MFMA (tools/ubench/step).
#ifndef MTM_STEP_VARIANT
#define MTM_STEP_VARIANT 2
#endif
__device__ __forceinline__ void mfma_step2(v4i (&acc)[2][16], const v4i qa, const v4i qb, const v4i (&a)[2]) {
#if defined(MTM_MFMA_NO_ASM) || MTM_STEP_VARIANT == 0
    mfma_step<2>(acc, qa, qb, a);
#elif MTM_STEP_VARIANT == 2
    mfma_step2_fused_b(acc, qa, qb, a);
#else
    mfma_step2_fused(acc, qa, qb, a);
#endif
}
// The K step of the row-multiplexed tilings: both groups, or - edge steps, see the generic K loop - only one of them
// (mode 1: group 0, 2: group 1).  One asm statement with three paths (tools/gen_mfma_step.py says why).
__device__ __forceinline__ void mfma_step2_rm(v4i (&acc)[2][16], const v4i qa, const v4i qb, const v4i (&a)[2], const int mode) {
#if defined(MTM_MFMA_NO_ASM) || MTM_STEP_VARIANT == 0
    if (mode == 0) {
        mfma_step<2>(acc, qa, qb, a);
    } else {
        v4i (&acc1)[1][16] = reinterpret_cast<v4i (&)[1][16]>(acc[mode - 1]);
        const v4i a1[1] = {a[mode - 1]};
        mfma_step<1>(acc1, qa, qb, a1);
    }
#else
    mfma_step2_edges(acc, qa, qb, a, mode);
#endif
}
// the K step of an instantiation: the uint16 kernel (three accumulator sets) takes the one-scratch-set form
template <int METHOD_, int MB>
__device__ __forceinline__ void mfma_kstep(v4i (&acc)[MB][16], const v4i qa, const v4i qb, const v4i (&a)[MB]) {
    if constexpr (METHOD_ == 7 && MB == 2) mfma_step_one_set(acc, qa, qb, a);       // kMfU16
    else if constexpr (MB == 2) mfma_step2(acc, qa, qb, a);
    else mfma_step<MB>(acc, qa, qb, a);
}

// METHOD >= 0: single-channel image, method fixed at compile time.  METHOD < 0: generic (any channel
// count, runtime method).
// R2 (two-row variant; classes of more than 16 templates up to 64 wide, one channel, methods 2..5): the two MFMA
// groups of a wave are the SAME 16 templates on two consecutive output rows y, y + 1 instead of 32 templates on one
// row.  Image row r is template row r - y for the first and r - y - 1 for the second, so the second group's A operand
// is the one the first group used one step earlier: it stays in registers.  A K step then needs ONE 1 KiB template
// load instead of two (the B operand and its byte shifts were shared already); a wave walks h + 1 image rows (the
// first / last step use a zero operand for one of the rows) and a work-group covers 8 output rows.
template <int MB, int METHOD, bool EXACT_DIV, bool MASKED = false, bool RM = false, int CH = 1, bool EXT = false,
          bool R2 = false, bool KP = false>
__global__ __launch_bounds__(256, 2) void ncc_mfma_kernel(MfmaParams p, const TemplDev* __restrict__ td,
                                                          const int* __restrict__ tlist,
                                                          const uint8_t* __restrict__ apack,
                                                          StatPlanes st, float* __restrict__ maps) {
    constexpr bool C1 = METHOD >= 0;         // compile-time method: CH (1 or 3) channels, lean epilogue
    static_assert(CH == 1 || !MASKED, "multi-channel: unmasked paths only");
    static_assert(!EXT || (METHOD >= 0 && METHOD != kMfRaw), "fused extremum: compile-time-method paths");
    static_assert(METHOD != kMfU16 || (MB == 2 && !MASKED && !RM && CH == 1 && !R2), "uint16 finishing pass");
    // packed K is a compile-time variant (its own instantiations): two K loops in one kernel - a run-time choice -
    // push the register allocator of the 256-VGPR kernel into spilling accumulators
    static_assert(!KP || (!R2 && METHOD >= 0 && (METHOD != kMfRaw || !RM)), "packed K");
    // (a three-row form of this variant - 192 accumulators, 48 MFMAs per step - shipped as an opt-in through round 5: 3 % fewer
    // cycles, 2.4 % less clock, no gain, and the only uint8 instantiations with hundreds of bytes of spills; removed in round 6)
    static_assert(!R2 || (MB == 2 && METHOD >= 2 && METHOD <= 5 && !MASKED && !RM && CH == 1), "two-row variant");
    static_assert(MB <= 2, "at most two MFMA groups per wave");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    // One work item per work-group; the block index orders the items so that the rows of the image an XCD reads stay in
    // its own L2 (block b runs on XCD b % 8: XCD x works through the contiguous item range [x per_xcd, (x + 1) per_xcd)).
    // (Rounds 2-5 also carried a persistent form - a grid of co-resident work-groups drawing items from atomic counters,
    // with and without a start stagger, per chip and per XCD: measured several times, never faster, removed in round 6.)
    int* s_item = reinterpret_cast<int*>(smem + p.tc_off + 32 * (int)sizeof(MfTemplConst));
    // effective shader clock under this very load (the chip clocks to its power budget: MFMA-dense code runs well
    // below the 2.4 GHz the peak figures assume)
    // (the two start values wait in LDS: kept in registers they were live across the whole kernel and the compiler
    // spilled them - 16 bytes of scratch per lane and work-group, 25 MB of HBM writes per 4K launch)
    const bool clk_probe = p.clk_out != nullptr && blockIdx.x == (gridDim.x >> 1) && threadIdx.x == 0;
    if (clk_probe) {
        unsigned long long* clk0 = reinterpret_cast<unsigned long long*>(s_item + 4);
        clk0[0] = __builtin_readcyclecounter();
        clk0[1] = __builtin_amdgcn_s_memrealtime();
    }
    const int per_xcd = (p.n_work + 7) >> 3;
    do {        // (a scope to leave: `continue` / `break` below = this wave is done with the item)
    const int wid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (wid >= p.n_work) break;
    if (!EXT && METHOD != kMfRaw && p.hits_only) {
        // hits-only launch whose candidate list overflowed: the call will be repeated with the maps in memory
        // (dense maps), nothing this launch still computes is used - leave (work-group-uniform decision)
        if (threadIdx.x == 0)
            s_item[2] = __hip_atomic_load(p.cand_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > p.cand_cap ? 1 : 0;
        __syncthreads();
        if (s_item[2]) break;
    }
    const int tg = wid % p.ntg;
    const int rest = wid / p.ntg;
    const int seg = rest % p.nseg;
    int yb = rest / p.nseg;
    // several slabs in one launch (MfmaParams::n_slab): which slab, where its image window, A pack and raw maps are
    long long slab_img = 0, slab_ap = 0, slab_raw = 0;
    if constexpr (RM && METHOD == kMfRaw && !KP) {
        if (p.n_slab > 1) {
            const int k = yb / p.nyb;
            yb -= k * p.nyb;
            const int cb = k % p.slab_ncb, rbk = k / p.slab_ncb;
            const int rb = rbk % p.slab_nrb, ch = rbk / p.slab_nrb;
            slab_img = (long long)ch * p.plane + (long long)rb * p.slab_rh * p.pitch + (long long)cb * p.slab_cw;
            slab_ap = (long long)k * p.slab_ap_step;
            slab_raw = (long long)k * p.slab_raw_step;
        }
    }
    const int x0 = seg * kMfSeg, y0 = (yb + p.yb0) * (RM ? 8 * p.rm_R : (R2 ? MB * kMfRows : kMfRows));
    // Lane coordinates of this work item, derived from an opaque copy of the thread index: whatever the prologue
    // computes from them is then computed HERE, per item, instead of once ahead of the item loop - where it stayed
    // live across the K loop (which owns every register) and was spilled to scratch memory.
    int tid_i = threadIdx.x;
    asm volatile("" : "+v"(tid_i));
    const int wave = __builtin_amdgcn_readfirstlane(tid_i >> 6), lane = tid_i & 63;
    const int j = lane & 15, q = lane >> 4;
    const int wave_rows = RM ? 2 * p.rm_R : (R2 ? MB : 1);   // output rows per wave = tile-row stride between waves
    constexpr int kTG = R2 ? 16 : 16 * MB;              // templates per work item

    v4i acc[MB][16];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[mb][c] = v4i{0, 0, 0, 0};
    // uint16 (kMfU16): the image's two byte planes are the two "channels" of ONE launch.  After the high-byte plane the
    // work item holds a_hh = acc[0] (I_hi x T_hi) and a_hl = acc[1] (I_hi x T_lo); a_hh moves to this third set, a_hl
    // becomes the start value of acc[0], which then collects I_lo x T_hi on top of it (the two middle terms of the 16-bit
    // product carry the same weight, 256), and acc[1] restarts at zero for I_lo x T_lo.  Round 2 wrote a_hh / a_hl to
    // memory in a launch of their own and read them back in the second: 4.3 GB of traffic for 135 MB of input.
    v4i u16_hh[METHOD == kMfU16 ? 16 : 1];

    // per-template constants -> LDS (read back in the epilogue; the staging barriers below order it)
    MfTemplConst* tcl = reinterpret_cast<MfTemplConst*>(smem + p.tc_off);
    constexpr int kTGc = METHOD == kMfU16 ? 16 : kTG;   // templates with constants (uint16: both groups are the same 16)
    if (METHOD != kMfRaw && tid_i < (RM ? p.rm_nt : kTGc)) {
        const int li = tg * kTGc + tid_i;
        if (li < p.n_list) {
            const TemplDev& T = td[tlist[li]];
            MfTemplConst k;
#pragma unroll
            for (int cc = 0; cc < kMaxChans; ++cc) k.mean[cc] = T.mean[cc];
            k.templ_norm = T.templ_norm;
            k.templ_sum2 = T.templ_sum2;
            k.mfma_k = T.mfma_k;
            if constexpr (METHOD == kMfU16) {
                // bias of the 16-bit combination: 257 (256 K_hi + K_lo), K_y = 128 sum(T_y) - 16384 A (exact integers)
                const double kh = 128.0 * p.u16_tsum[li] - 16384.0 * p.u16_area;
                const double kl = 128.0 * p.u16_tsum[p.u16_npad + li] - 16384.0 * p.u16_area;
                k.mfma_k = 257.0 * (256.0 * kh + kl);
            }
            k.rtempl_norm = T.templ_norm > 0.0 ? 1.0 / T.templ_norm : 0.0;
#pragma unroll
            for (int cc = 0; cc < kMaxChans; ++cc) k.m128[cc] = 128.0 - T.mean[cc];
            k.tms = T.templ2_mask2_sum;
            k.rsqrt_tms = 1.0 / sqrt(T.templ2_mask2_sum);
            k.map_off = T.map_off;
            k.map_pitch = T.map_pitch;
            k.all_ones = T.all_ones;
            {   // (built behind an opaque asm: as a plain constant the compiler hoists it out of the item loop, where it
                // lives across the K loop - which owns every register - in scratch memory)
                int ninf_hi = (int)0xfff00000;
                asm volatile("" : "+v"(ninf_hi));
                k.ext_thr_lo = __hiloint2double(ninf_hi, 0);
            }
            k.ext_hi = 0u;
            k.flag_base = tlist[li] * p.flag_tstride;
            if (EXT) {
                const unsigned long long bk = __hip_atomic_load(&p.ext_best[2 * tlist[li] + p.cand_min], __ATOMIC_RELAXED,
                                                                __HIP_MEMORY_SCOPE_AGENT);
                if (bk) {
                    const uint32_t hiw = (uint32_t)(bk >> 32);
                    const double sc = (double)mf_order_float(p.cand_min ? ~hiw : hiw);
                    const double ql = p.cand_min ? -sc : sc;
                    k.ext_hi = hiw;
                    k.ext_thr_lo = ql - 1e-6 * fmax(1.0, fabs(ql));
                }
            }
            tcl[tid_i] = k;
        } else if (!RM) {
            // Beyond the list (zero-padded A rows): constants with which no screen can pass and nothing is read uninitialised -
            // the bound acc + K + ... is hugely negative whatever the statistics are, the running best is out of reach.  The
            // screens then need no per-lane "is this template in the list" test (three vector registers of indices that lived
            // across them next to 128 accumulators, one of them in scratch memory); the epilogue's own loops stop at n_list.
            MfTemplConst k;
#pragma unroll
            for (int cc = 0; cc < kMaxChans; ++cc) k.mean[cc] = 0.0, k.m128[cc] = 0.0;
            k.templ_norm = 1.0;
            k.templ_sum2 = 0.0;
            k.mfma_k = -1e300;
            k.rtempl_norm = 0.0;
            k.tms = 1.0;
            k.rsqrt_tms = 0.0;
            k.map_off = 0;
            k.map_pitch = 0;
            k.all_ones = 0;
            k.ext_thr_lo = 1e300;
            k.ext_hi = 0xFFFFFFFFu;
            k.flag_base = 0;
            tcl[tid_i] = k;
        }
    }

    // window statistics of this wave's 256 pixels -> LDS (LDS-DMA, no registers; drained by the
    // staging barriers below, read back in the epilogue).  Lane L fetches pixels 4L..4L+3.
    constexpr bool kNeedSum2 = MASKED || METHOD == MTM_TM_SQDIFF || METHOD == MTM_TM_SQDIFF_NORMED;
    constexpr bool kNormed = !MASKED && (METHOD == MTM_TM_SQDIFF_NORMED || METHOD == MTM_TM_CCORR_NORMED ||
                                         METHOD == MTM_TM_CCOEFF_NORMED);
    constexpr bool kMaskedNormed = MASKED && (METHOD == MTM_TM_SQDIFF_NORMED || METHOD == MTM_TM_CCORR_NORMED);
    if constexpr (R2 && kNormed) {
        // MB rows per wave.  Only the hits-only screen's first level is on every item's path, and all it needs of the
        // statistics are their ranges over each 16-pixel column block (StatPlanes::blk, written by stats_u8_kernel:
        // [S1 min, S1 max, sqrt min, -] per block): 512 bytes per wave and row instead of 4 KiB - which is what lets
        // three rows per wave fit two work-groups on a CU.  One LDS-DMA instruction fetches two rows (lanes 0..31 /
        // 32..63: 16 blocks x 32 bytes each).  The per-pixel statistics of the rare paths behind the screen come
        // straight from memory.
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        uint8_t* sbase = smem + p.st_off + wave * (((MB + 1) / 2) * 1024);
        const int bj = min((x0 >> 4) + ((lane & 31) >> 1), st.blk_pitch - 1);
#pragma unroll
        for (int r0 = 0; r0 < MB; r0 += 2) {
            const int yr = min(y0 + MB * wave + min(r0 + (lane >> 5), MB - 1), p.oh - 1);
            const double* src = st.blk + ((size_t)yr * st.blk_pitch + bj) * 4 + 2 * (lane & 1);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sbase + (r0 / 2) * 1024), 16, 0, 0);
        }
    } else if constexpr (C1 && METHOD != kMfRaw && METHOD != kMfU16 && !RM && !R2) {
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        const int yc = min(y0 + wave, p.oh - 1), xc = min(x0 + 4 * lane, st.pitch - 4);
        const size_t sidx = (size_t)yc * st.pitch + xc;
        uint8_t* sbase = smem + p.st_off + wave * mf_stat_bytes_per_wave(CH);
        // planes: S1 of channel 0..CH-1, then S2, then sqrt
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
            for (int cc = 0; cc < CH; ++cc)
                __builtin_amdgcn_global_load_lds((gptr_t)(st.t[cc] + sidx + 2 * hh), (lptr_t)(sbase + (2 * cc + hh) * 1024), 16, 0, 0);
            if (kNeedSum2)
                __builtin_amdgcn_global_load_lds((gptr_t)(st.sum2 + sidx + 2 * hh), (lptr_t)(sbase + (2 * CH + hh) * 1024), 16, 0, 0);
            if (kNormed)
                __builtin_amdgcn_global_load_lds((gptr_t)(st.sq + sidx + 2 * hh), (lptr_t)(sbase + (2 * CH + 2 + hh) * 1024), 16, 0, 0);
        }
    }

    const uint8_t* apack_g = apack + slab_ap + (long long)tg * (R2 ? 1 : MB) * p.group_bytes + (size_t)lane * 16;

    for (int c = 0; c < p.chans; ++c) {
        const uint8_t* plane = p.img + slab_img + c * p.plane;
        // channel of the A pack (uint16: both byte planes meet the same [T_hi | T_lo]; sum I^2 M: both planes meet the mask)
        const int cpk = (METHOD == kMfU16 || (RM && METHOD == kMfRaw && p.sq_fused)) ? 0 : c;
        if constexpr (RM && METHOD == kMfRaw && !KP) {
            // sum I^2 M in ONE launch (round 4): the byte planes of I^2 are the two "channels".  The high-byte sums
            // a_h = sum (J_h - 128) M are scaled by 256 in place and the low-byte plane accumulates on top of them:
            // 256 a_h + a_l, |.| <= 256 * 128 * w h + 128 w h < 2^31 for w h <= 65025 (the launcher's bound).
            if (p.sq_fused && c == 1) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[mb][k] = acc[mb][k] * 256;
            }
        }
        if constexpr (METHOD == kMfU16) {
            if (c == 1) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    u16_hh[k] = acc[0][k];
                    acc[0][k] = acc[1][k];
                    acc[1][k] = v4i{0, 0, 0, 0};
                }
            }
        }
        const int krows = RM ? p.rm_steps : (R2 ? p.h + MB - 1 : p.h);        // image rows a wave walks (K steps / nb)
        constexpr int kChunk = R2 ? kMfChunkR2 : kMfChunkH;
        v4i r2_prev = v4i{0, 0, 0, 0};      // R2: the A operand of the previous step (the int8 zero before template row 0)
        for (int cy0 = 0; cy0 < krows; cy0 += kChunk) {
            const int ch = min(kChunk, krows - cy0);
            // ---- stage (ch + 3) rows of the int8 image plane by LDS-DMA: the tile is one linear
            // array of 16-byte chunks [row][cpr]; wave-instruction k of the work-group fills chunks
            // 64 k .. 64 k + 63 (LDS destination = wave-uniform base + lane * 16), each lane
            // supplying the global address of its chunk.  No registers, no VALU conversion.
            __syncthreads();
            {
                typedef const __attribute__((address_space(1))) void* gptr_t;
                typedef __attribute__((address_space(3))) void* lptr_t;
                const int nchunk = (ch + (kMfRows - 1) * wave_rows) * p.cpr;
                int ci = tid_i;
                int r = (ci * p.cpr_magic) >> 16, d = ci - r * p.cpr;     // ci / cpr for ci < 256 (cpr <= 33), without a division
                const uint8_t* grow = plane + (size_t)(y0 + cy0) * p.pitch + x0;
                uint8_t* lbase_w = smem + wave * 1024;
                for (; ci - lane < nchunk; ci += 256) {
                    if (ci < nchunk)
                        __builtin_amdgcn_global_load_lds((gptr_t)(grow + (size_t)r * p.pitch + 16 * d), (lptr_t)lbase_w,
                                                         16, 0, 0);
                    lbase_w += 4096;
                    r += p.cpr_rstep;
                    d += p.cpr_dstep;
                    if (d >= p.cpr) {
                        d -= p.cpr;
                        ++r;
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            if constexpr (R2) {
                // ---- K loop of the two-row variant (nb == 1: one step per image row).  B operands alternate between
                // two register sets, A operands rotate through three (this step's, the previous step's - the second
                // row's operand - and the one being loaded), so the body is unrolled six times; every request runs
                // one step ahead and up to six steps past the end of the chunk (pack slack / inside the LDS
                // allocation; never used).
                const uint8_t* aptr = apack_g + (size_t)cy0 * 1024;
                const uint8_t* lbase = smem + wave * wave_rows * p.lds_pitch + (j + q) * 16;
                int loff = 0;
                v4i qx, qy, qx2, qy2, aA, aB, aC = r2_prev;
                if (mf_opaque_sgpr(p.hits_only)) __builtin_amdgcn_s_setprio(3);
#define MTM_R2_LOAD(QA, QB, A)                                              \
                QA = *reinterpret_cast<const v4i*>(lbase + loff);           \
                QB = *reinterpret_cast<const v4i*>(lbase + loff + 16);      \
                A = *reinterpret_cast<const v4i*>(aptr);
#define MTM_R2_STEP(QA, QB, ACUR, APREV, QA2, QB2, ANEXT)                   \
                {                                                           \
                    aptr += 1024;                                           \
                    loff += p.lds_pitch;                                    \
                    MTM_R2_LOAD(QA2, QB2, ANEXT)                            \
                    __builtin_amdgcn_sched_barrier(0);                      \
                    const v4i ar_[2] = {ACUR, APREV};                       \
                    mfma_step2(acc, QA, QB, ar_);                           \
                    __builtin_amdgcn_sched_barrier(0);                      \
                }
                MTM_R2_LOAD(qx, qy, aA)           // step 0 of the chunk; aC = the step before it
                int ks = 0;
                for (; ks + 6 <= ch; ks += 6) {
                    MTM_R2_STEP(qx, qy, aA, aC, qx2, qy2, aB)
                    MTM_R2_STEP(qx2, qy2, aB, aA, qx, qy, aC)
                    MTM_R2_STEP(qx, qy, aC, aB, qx2, qy2, aA)
                    MTM_R2_STEP(qx2, qy2, aA, aC, qx, qy, aB)
                    MTM_R2_STEP(qx, qy, aB, aA, qx2, qy2, aC)
                    MTM_R2_STEP(qx2, qy2, aC, aB, qx, qy, aA)
                }
                // remainder (0..5 steps), same rotation; r2_prev = the operand of the chunk's last step
                r2_prev = aC;
                if (ks < ch) { MTM_R2_STEP(qx, qy, aA, aC, qx2, qy2, aB) r2_prev = aA; }
                if (ks + 1 < ch) { MTM_R2_STEP(qx2, qy2, aB, aA, qx, qy, aC) r2_prev = aB; }
                if (ks + 2 < ch) { MTM_R2_STEP(qx, qy, aC, aB, qx2, qy2, aA) r2_prev = aC; }
                if (ks + 3 < ch) { MTM_R2_STEP(qx2, qy2, aA, aC, qx, qy, aB) r2_prev = aA; }
                if (ks + 4 < ch) { MTM_R2_STEP(qx, qy, aB, aA, qx2, qy2, aC) r2_prev = aB; }
#undef MTM_R2_LOAD
#undef MTM_R2_STEP
                asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
                __builtin_amdgcn_s_setprio(0);
            } else if constexpr (KP) {
            // ---- K loop over the segment stream (packed K, see MfmaParams): as below, but every lane group q walks
            // its own (row, segment) position - five VALU instructions of bookkeeping per step instead of scalar ones
            const int nseg = p.kp_nseg;
            const uint8_t* aptr = apack_g + (RM ? (size_t)cpk * p.rm_cstride : (size_t)cpk * p.kp_blocks * 1024) +
                                  (size_t)((cy0 * nseg) >> 2) * 1024;           // cy0 is a multiple of 64
            const uint8_t* lbase = smem + wave * wave_rows * p.lds_pitch + j * 16;
            const int nsteps = (ch * nseg + 3) >> 2;
            const int drow = 4 / nseg, dseg = 4 - drow * nseg;                 // uniform: stream advance of one step
            const int adv = drow * p.lds_pitch + dseg * 16, wrapfix = p.lds_pitch - nseg * 16;
            int seg = q % nseg;                                                 // this lane group's segment of step 0
            int loff = (q / nseg) * p.lds_pitch + seg * 16;
            v4i qa0, qb0, qa1, qb1, a0[MB], a1[MB];
#define MTM_KP_ADVANCE()                                            \
            {                                                       \
                aptr += 1024;                                       \
                seg += dseg;                                        \
                const bool wrap_ = seg >= nseg;                     \
                seg -= wrap_ ? nseg : 0;                            \
                loff += adv + (wrap_ ? wrapfix : 0);                \
            }
#define MTM_KP_LOAD(QA, QB, A)                                                          \
            QA = *reinterpret_cast<const v4i*>(lbase + loff);                           \
            QB = *reinterpret_cast<const v4i*>(lbase + loff + 16);                      \
            _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                           \
                A[mb] = *reinterpret_cast<const v4i*>(aptr + mb * p.group_bytes);
            if (mf_opaque_sgpr(p.hits_only)) __builtin_amdgcn_s_setprio(3);
            MTM_KP_LOAD(qa0, qb0, a0)            // step 0
            int ks = 0;
            for (; ks + 2 <= nsteps; ks += 2) {
                MTM_KP_ADVANCE()
                MTM_KP_LOAD(qa1, qb1, a1)
                __builtin_amdgcn_sched_barrier(0);
                mfma_kstep<METHOD, MB>(acc, qa0, qb0, a0);
                __builtin_amdgcn_sched_barrier(0);
                MTM_KP_ADVANCE()
                MTM_KP_LOAD(qa0, qb0, a0)
                __builtin_amdgcn_sched_barrier(0);
                mfma_kstep<METHOD, MB>(acc, qa1, qb1, a1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ks < nsteps) {
                mfma_kstep<METHOD, MB>(acc, qa0, qb0, a0);
            }
#undef MTM_KP_ADVANCE
#undef MTM_KP_LOAD
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            __builtin_amdgcn_s_setprio(0);
            } else {
            // ---- K loop: template rows of this chunk x 64-tap blocks, software pipelined with two
            // register sets: the operands of the next step (2 LDS chunks + MB packed template rows)
            // are requested before the 16*MB MFMAs of the current step issue.  sched_barrier keeps
            // the compiler from sinking the requests below the MFMAs.
            const uint8_t* aptr = apack_g + (RM ? (size_t)cpk * p.rm_cstride + (size_t)cy0 * p.nb * 1024     // + ks * 1024
                                               : ((size_t)(cpk * p.h + cy0) * p.nb) * 1024);
            const uint8_t* lbase = smem + wave * wave_rows * p.lds_pitch + (j + q) * 16;
            const int nsteps = ch * p.nb;
            int nb_i = 0;                       // 64-tap block of the step last requested
            int loff = 0;                       // its LDS offset: dy * lds_pitch + b * 64
            const int row_adv = p.lds_pitch - (p.nb - 1) * 64;
            v4i qa0, qb0, qa1, qb1, a0[MB], a1[MB];
            // The loop body is branch-free: the bookkeeping of the next step is scalar selects, and the
            // requests run up to two steps past the end of the chunk (the pack arena has slack, LDS reads
            // past the tile stay inside the allocation; nothing of it is used).  Taken scalar branches
            // cost more than the loads they would save.
#define MTM_MF_ADVANCE()                                            \
            {                                                       \
                MTM_MF_APTR_STEP                                    \
                const bool wrap_ = nb_i + 1 == p.nb;                \
                loff += wrap_ ? row_adv : 64;                       \
                nb_i = wrap_ ? 0 : nb_i + 1;                        \
            }
#define MTM_MF_APTR_STEP aptr += 1024;
#define MTM_MF_LOAD(QA, QB, A)                                                          \
            QA = *reinterpret_cast<const v4i*>(lbase + loff);                           \
            QB = *reinterpret_cast<const v4i*>(lbase + loff + 16);                      \
            _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                           \
                A[mb] = *reinterpret_cast<const v4i*>(aptr + mb * p.group_bytes);
#define MTM_MF_LOAD_LOOP(QA, QB, A) MTM_MF_LOAD(QA, QB, A)
            // Row-multiplexed tilings (round 4): a wave's two MFMA groups are output rows [0, R) and [R, 2 R) of the same
            // templates, and image row s of the wave's h + 2 R - 1 holds template rows of group 0 only for s < h + R - 1 and
            // of group 1 only for s >= R - outside, the group's A operand is all zero.  Those edge steps run the
            // one-group step (16 MFMAs instead of 32): 2 R of the h + 2 R - 1 steps, i.e. for ONE template or mask
            // (R = 16, h = 64) 79 step-equivalents instead of 95.
            constexpr bool kRmEdges = RM && MB == 2 && METHOD != kMfU16;
            int srow = cy0, sblk = 0;                       // image row / 64-tap block of the step being executed
            const int edge_lo = RM ? p.rm_R : 0, edge_hi = RM ? p.h + p.rm_R - 1 : 0x7fffffff;
#define MTM_MF_STEP(QA, QB, A, K)                                                       \
            if constexpr (kRmEdges) {                                                   \
                const int mode_ = srow < edge_lo ? 1 : srow >= edge_hi ? 2 : 0; \
                mfma_step2_rm(acc, QA, QB, A, __builtin_amdgcn_readfirstlane(mode_));   \
                const bool last_ = sblk + 1 == p.nb;                                    \
                srow += last_ ? 1 : 0;                                                  \
                sblk = last_ ? 0 : sblk + 1;                                            \
            } else {                                                                    \
                mfma_kstep<METHOD, MB>(acc, QA, QB, A);                                 \
            }
            // hits-only launches: the MFMA main loop outranks the (short) epilogue of the co-resident work-group
            // (-0.9 % kernel time; with the maps written the long epilogue is the one that must not starve: +1 %)
            if (mf_opaque_sgpr(p.hits_only)) __builtin_amdgcn_s_setprio(3);
            MTM_MF_LOAD(qa0, qb0, a0)            // step 0
            int ks = 0;
            for (; ks + 2 <= nsteps; ks += 2) {
                MTM_MF_ADVANCE()
                MTM_MF_LOAD_LOOP(qa1, qb1, a1)    // step ks + 1
                __builtin_amdgcn_sched_barrier(0);
                MTM_MF_STEP(qa0, qb0, a0, 0)
                __builtin_amdgcn_sched_barrier(0);
                MTM_MF_ADVANCE()
                MTM_MF_LOAD_LOOP(qa0, qb0, a0)    // step ks + 2 (one past the end in the last iteration)
                __builtin_amdgcn_sched_barrier(0);
                MTM_MF_STEP(qa1, qb1, a1, 1)
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ks < nsteps) {                    // odd number of steps: the last one is already in set 0
                MTM_MF_STEP(qa0, qb0, a0, 0)
            }
#undef MTM_MF_ADVANCE
#undef MTM_MF_APTR_STEP
#undef MTM_MF_LOAD
#undef MTM_MF_LOAD_LOOP
#undef MTM_MF_STEP
            // the last MFMAs may have been issued from inline asm: give their results time to land
            // before compiler-generated code reads the accumulators (hipcc does not see asm MFMAs)
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            __builtin_amdgcn_s_setprio(0);
            }   // !R2
        }
    }

    // ---- epilogue: per wave, 8 templates at a time through LDS ([8 templates][pixel] int32).
    // A lane owns 4 consecutive pixels: their statistics are loaded ONCE into registers (they do
    // not depend on the template), per-template constants come from LDS, every (lane, template)
    // pair is one ds_read_b128, four normalisations and one float4 store.  The stage loop is
    // rolled (one copy of the float64 normalisation in the binary).  The buffer of a wave is
    // private to it: LDS executes a wave's instructions in order, so no work-group barrier is
    // needed between the stages.
    {   // epilogue scope
    // Lane coordinates are re-derived behind an opaque asm: everything the epilogue computes from
    // them is then computed HERE instead of being hoisted above the K loop and spilled across it.
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int wave = tid_e >> 6, lane = tid_e & 63;
    const int j = lane & 15, q = lane >> 4;
    const int y = y0 + (R2 ? MB : 1) * wave;             // R2: the wave's first row; MFMA group mb is row y + mb
    int* epi = reinterpret_cast<int*>(smem + wave * kMfEpiBytesPerWave);
    const int xq = x0 + 4 * lane;                       // first of this lane's 4 pixels
    // is the candidate list full already?  (dense maps; see emit_at)
    bool emit_full = false;
    if (!EXT && METHOD != kMfRaw && p.cand_on)
        emit_full = __hip_atomic_load(p.cand_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > p.cand_cap;   // strictly: the overflow is on record
    const int rd_off = 16 * (lane >> 2) + ((4 * (lane & 3) + mf_epi_rot(lane >> 2)) & 15);
    // registers e = 0..3 of phase c hold templates 4 (q & 1) + e of a stage, pixel 16 j + c
    auto put = [&](const v4i (&blk)[16]) {
        int* dst = &epi[(4 * (q & 1)) * kMfEpiPitch + 16 * j];
        const int rot = mf_epi_rot(j);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int col = (c + rot) & 15;
            dst[0 * kMfEpiPitch + col] = blk[c].x;
            dst[1 * kMfEpiPitch + col] = blk[c].y;
            dst[2 * kMfEpiPitch + col] = blk[c].z;
            dst[3 * kMfEpiPitch + col] = blk[c].w;
        }
    };
    // append the outputs above the threshold to the candidate list.  Rare on sparse maps, massive on smooth ones (a
    // photograph at threshold 0.5: millions), so the list slots of a wave come from ONE atomic: the lanes that reach
    // this point (the call sits under a divergent any-of-4 test) count their candidates with four ballots, the first
    // of them reserves the total.  Once the list is full (emit_full, sampled when the epilogue starts) nothing is
    // appended or counted any more - the host only needs to see count > capacity to fall back to the full peak pass.
    // Round 4: a wave first collects its candidates in a wave-private LDS buffer (kMfCandStage records + a count) and
    // appends them to the global list with ONE atomic when its epilogue ends (or the buffer is full).  A wave over a
    // bright region of a photograph-like image emits for most of its 16 x 2 (template, row) pairs; one round trip to the
    // list's counter per pair - on which the wave waits before it can store - made such launches 3x slower than the
    // sparse case (2.0 against 0.7 ms at 4K x 32 templates, 3e5-1e6 candidates).
    int* cs_cnt = reinterpret_cast<int*>(smem + p.cs_off + wave * kMfCandStageBytes);
    mtm_hit* cs_rec = reinterpret_cast<mtm_hit*>(smem + p.cs_off + wave * kMfCandStageBytes + 16);
    const bool cs_on = p.cs_off > 0 && !EXT && METHOD != kMfRaw && p.cand_on;
    if (cs_on) {
        if (lane == 0) cs_cnt[0] = 0;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    auto emit_at = [&](const float (&out)[4], int li, int yrow, unsigned allow = 15u) {
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = p.cand_min ? -out[i] : out[i];
            if (xq + i < p.ow && v > p.cand_thr) m |= 1u << i;
        }
        m &= allow;
        if (emit_full) m = 0;
        const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
        const unsigned long long b0 = __builtin_amdgcn_ballot_w64((m & 1u) != 0), b1 = __builtin_amdgcn_ballot_w64((m & 2u) != 0);
        const unsigned long long b2 = __builtin_amdgcn_ballot_w64((m & 4u) != 0), b3 = __builtin_amdgcn_ballot_w64((m & 8u) != 0);
        const unsigned n0 = __popcll(b0), n1 = __popcll(b1), n2 = __popcll(b2), n3 = __popcll(b3);
        const unsigned total = n0 + n1 + n2 + n3;
        if (total == 0) return;                                     // uniform over the lanes that are here
        const int leader = (int)__builtin_ctzll(act);
        if (cs_on) {
            int cur = cs_cnt[0];                                    // (same address for every lane here: a broadcast)
            bool flushed = false;
            if (cur > 0 && cur + (int)total > kMfCandStage) {
                // no room for this call's candidates: the staged ones go to the list now - one atomic for up to
                // kMfCandStage records, the lanes that are here share the copy - and the buffer starts over.  (A wave over a
                // bright region lists for most of its 32 (template, row) pairs; without this every further pair would
                // pay its own round trip to the list's counter.)
                unsigned long long fb = 0ull;
                if (lane == leader) fb = atomicAdd(p.cand_counter, (unsigned long long)cur);
                const uint32_t flo = __builtin_amdgcn_readlane((uint32_t)fb, leader);
                const uint32_t fhi = __builtin_amdgcn_readlane((uint32_t)(fb >> 32), leader);
                fb = ((unsigned long long)fhi << 32) | flo;
                const int n_act = __popcll(act);
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
                for (int r = rank; r < cur; r += n_act)
                    mf_put_cand(p, fb + (unsigned long long)r, cs_rec[r]);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // the records are read: their slots may be rewritten
                cur = 0;
                flushed = true;
            }
            if (cur + (int)total <= kMfCandStage) {
                if (m) {
                    const int tglob = tlist[li];
                    const unsigned long long bb[4] = {b0, b1, b2, b3};
                    const unsigned pre[4] = {0u, n0, n0 + n1, n0 + n1 + n2};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if ((m >> i) & 1u) {
                            const unsigned below = __builtin_amdgcn_mbcnt_hi((uint32_t)(bb[i] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bb[i], 0u));
                            mtm_hit hrec;
                            hrec.templ_idx = tglob;
                            hrec.x = xq + i;
                            hrec.y = yrow;
                            hrec.w = p.w;
                            hrec.h = p.h;
                            hrec.score = out[i];
                            cs_rec[cur + (int)(pre[i] + below)] = hrec;
                        }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // every lane has read the count: now it moves
                if (lane == leader) cs_cnt[0] = cur + (int)total;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                return;
            }
            // (more candidates in this one call than the buffer holds: they go straight to the list, as before)
            if (flushed) {
                if (lane == leader) cs_cnt[0] = 0;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            }
        }
        unsigned long long base = 0ull;
        if (lane == leader) base = atomicAdd(p.cand_counter, (unsigned long long)total);
        const uint32_t blo = __builtin_amdgcn_readlane((uint32_t)base, leader);
        const uint32_t bhi = __builtin_amdgcn_readlane((uint32_t)(base >> 32), leader);
        base = ((unsigned long long)bhi << 32) | blo;
        if (m) {
            const int tglob = tlist[li];
            const unsigned long long bb[4] = {b0, b1, b2, b3};
            const unsigned pre[4] = {0u, n0, n0 + n1, n0 + n1 + n2};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if ((m >> i) & 1u) {
                    const unsigned below = __builtin_amdgcn_mbcnt_hi((uint32_t)(bb[i] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bb[i], 0u));
                    const unsigned long long slot = base + pre[i] + below;
                    if (slot < p.cand_cap) {
                        mtm_hit hrec;
                        hrec.templ_idx = tglob;
                        hrec.x = xq + i;
                        hrec.y = yrow;
                        hrec.w = p.w;
                        hrec.h = p.h;
                        hrec.score = out[i];
                        mf_put_cand(p, slot, hrec);
                    }
                }
        }
    };
    // global extremum (EXT): the lane's best key not below the template's running best goes to the wave's
    // LDS slot (cv2.minMaxLoc: the first index wins ties, NaN never wins)
    auto ext_update = [&](const float (&out)[4], int yrow, uint32_t best_hi, unsigned long long* slot, bool on = true) {
        unsigned long long bestk = 0ull;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = out[i];
            const uint32_t o = mf_float_order(v);
            const uint32_t hiw = p.cand_min ? ~o : o;
            if (on && xq + i < p.ow && v == v && hiw >= best_hi) {
                const unsigned long long key = ((unsigned long long)hiw << 32) |
                                               (unsigned long long)(0xFFFFFFFFu - (uint32_t)(yrow * p.ow + xq + i));
                bestk = key > bestk ? key : bestk;
            }
        }
        if (bestk) atomicMax(slot, bestk);
    };
    const bool full4 = xq + 3 < p.ow;         // (a lane mask in scalar registers: as `xq + 3` recomputed per store it cost a vector register)
    auto store4 = [&](float* orow, const float (&out)[4]) {
        if (full4) {
            *reinterpret_cast<float4*>(orow) = make_float4(out[0], out[1], out[2], out[3]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (xq + i < p.ow) orow[i] = out[i];
        }
    };

    if constexpr (RM && METHOD == kMfRaw) {
        // ---- row-multiplexed raw mode (sum I^2 M of masked classes): int32 accumulators of
        // (template i % nt, output row mb R + i / nt) to raw_out + t * raw_map + row * raw_pitch
        __syncthreads();
        const int R = p.rm_R, ntm = p.rm_nt - 1, lg = p.rm_log2nt;
        if (p.sq_fused) {
            // ---- sum I^2 M, finished here (round 4; round 3: two raw launches + masksq_combine_kernel): the accumulator
            // holds 256 a_h + a_l, c2 = that + 257 * 128 * sum(M) (p.sq_k; the mask operand is not biased) goes to the
            // class's sum2 plane as float64, and the smallest c2 of every 16-pixel column block - lanes 4 b .. 4 b + 3 of
            // a row - to the third slot of the block record (the masked hits-only screen bounds sqrt(tms c2) with it;
            // +inf right of the last output column).
            double* sum2_out = const_cast<double*>(st.sum2);
            double* blk_out = const_cast<double*>(st.blk);
            const int bj = (x0 >> 4) + (lane >> 2);
            auto xmin = [&](double v, int off) {
                const int addr = (lane ^ off) << 2;
                const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
                const int hi2 = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
                return fmin(v, __hiloint2double(hi2, lo));
            };
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
#pragma unroll 1
                for (int round = 0; round < 2; ++round) {
                    if ((q >> 1) == round) put(acc[mb]);
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
                    for (int s8 = 0; s8 < 8; ++s8) {
                        const int i = 8 * round + s8;
                        const int t = i & ntm, rho = i >> lg;
                        const int yy = y0 + wave * wave_rows + mb * R + rho;
                        if (t >= p.n_list || yy >= p.oh) continue;              // wave-uniform
                        const v4i a4 = *reinterpret_cast<const v4i*>(&epi[s8 * kMfEpiPitch + rd_off]);
                        const int a32[4] = {a4.x, a4.y, a4.z, a4.w};
                        double c2v[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) c2v[k] = xq + k < p.ow ? (double)a32[k] + p.sq_k : INFINITY;
                        double* orow = sum2_out + (size_t)yy * st.pitch + xq;
                        if (xq + 3 < p.ow) {
                            *reinterpret_cast<double2*>(orow) = make_double2(c2v[0], c2v[1]);
                            *reinterpret_cast<double2*>(orow + 2) = make_double2(c2v[2], c2v[3]);
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (xq + k < p.ow) orow[k] = c2v[k];
                        }
                        double bmin = fmin(fmin(c2v[0], c2v[1]), fmin(c2v[2], c2v[3]));
                        bmin = xmin(bmin, 1);
                        bmin = xmin(bmin, 2);
                        if (blk_out != nullptr && (lane & 3) == 0 && bj < st.blk_pitch)
                            blk_out[((size_t)yy * st.blk_pitch + bj) * 4 + 2] = bmin;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            continue;           // done (nothing behind the scope needs a work-group barrier)
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
#pragma unroll 1
            for (int round = 0; round < 2; ++round) {
                if ((q >> 1) == round) put(acc[mb]);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                {   // (the only divergent code between the transposition stages are the stores themselves)
#pragma unroll 1
                    for (int s8 = 0; s8 < 8; ++s8) {
                        const int i = 8 * round + s8;
                        const int t = i & ntm, rho = i >> lg;
                        const int yy = y0 + wave * wave_rows + mb * R + rho;
                        if (t >= p.n_list || yy >= p.oh) continue;              // wave-uniform
                        const v4i a4 = *reinterpret_cast<const v4i*>(&epi[s8 * kMfEpiPitch + rd_off]);
                        int* orow = p.raw_out + slab_raw + (size_t)t * p.raw_map + (size_t)yy * p.raw_pitch + xq;
                        if (xq + 3 < p.ow) {
                            *reinterpret_cast<v4i*>(orow) = a4;
                        } else {
                            const int a32[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (xq + k < p.ow) orow[k] = a32[k];
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    } else if constexpr (RM) {
        // ---- row-multiplexed mode: A row i of group mb is (template i % nt, output row mb R + i / nt) of
        // this wave.  Statistics are per output row here, so they are loaded (S1 and 1/sqrt, both from
        // the statistics pass) whenever the row changes; otherwise the same normalisation, hits-only
        // pre-test, candidate emission and stores as the single-row path.
        __syncthreads();              // every wave is done reading the image tile: the buffers alias it
        const int R = p.rm_R, ntm = p.rm_nt - 1, lg = p.rm_log2nt;
        // ---- hits-only screen of the row-multiplexed tiling (one channel; the normalised unmasked methods and masked
        // TM_CCORR_NORMED): the per-lane bound of the single-row path's first level (see there), straight from the
        // accumulator registers.  In the C/D layout lane (j, q) holds, per MFMA group and A row i = 4 q + e, the 16
        // consecutive outputs 16 j + c of (template i % nt, output row mb R + i / nt); the ranges of the statistics over
        // that 16-pixel block come from StatPlanes::blk (a 32-byte load per row, from memory: the item is long and
        // this is once per item).  Masked: c1 = acc + 128 S1 + K against thr sqrt(tms c2), c2 = sum I^2 M >= its block
        // minimum (third slot of the block record, written by masksq_combine_kernel).  A wave none of whose lanes can
        // hold a candidate skips the transposition and the per-output normalisation below.
        if constexpr (CH == 1 && ((!MASKED && (METHOD == MTM_TM_CCORR_NORMED || METHOD == MTM_TM_CCOEFF_NORMED)) ||
                                  (MASKED && METHOD == MTM_TM_CCORR_NORMED))) {
            if ((p.hits_only || (!MASKED && !EXT && p.seg_skip)) && p.screen_l1 && st.blk != nullptr && (EXT || p.cand_thr_lo >= 0.0)) {
                bool pass1 = false;
                const int bj = min((x0 >> 4) + j, st.blk_pitch - 1);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    int amax[4] = {INT_MIN, INT_MIN, INT_MIN, INT_MIN};
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        const v4i a = acc[mb][c];
                        amax[0] = max(amax[0], a.x);
                        amax[1] = max(amax[1], a.y);
                        amax[2] = max(amax[2], a.z);
                        amax[3] = max(amax[3], a.w);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = 4 * q + e;
                        const int t = i & ntm, rho = i >> lg;
                        const int yy = y0 + wave * wave_rows + mb * R + rho;
                        const bool live = t < p.n_list && yy < p.oh;
                        const double* bp = st.blk + ((size_t)min(yy, p.oh - 1) * st.blk_pitch + bj) * 4;
                        const double2 lohi = *reinterpret_cast<const double2*>(bp);
                        const double third = bp[2];
                        const MfTemplConst& T = tcl[t];
                        const double thr_lo_e = EXT ? T.ext_thr_lo : p.cand_thr_lo;
                        const double hi = EXT ? fmin(thr_lo_e, 0.999999) - 1e-6 : p.screen_hi;
                        bool could;
                        if constexpr (MASKED) {
                            const double bound = ((double)amax[e] + T.mfma_k) + 128.0 * lohi.y;
                            could = bound > hi * sqrt(T.tms * third);
                        } else {
                            const double m = METHOD == MTM_TM_CCOEFF_NORMED ? T.m128[0] : 128.0;
                            const double bound = ((double)amax[e] + T.mfma_k) + fmax(m * lohi.x, m * lohi.y);
                            could = T.all_ones != 0 || bound > hi * T.templ_norm * fmax(third, p.sq_floor);
                        }
                        pass1 = pass1 || (live && (thr_lo_e < 0.0 || could));
                    }
                }
                if (__builtin_amdgcn_ballot_w64(pass1) == 0ull) continue;      // wave-uniform; no work-group barrier below
            }
        }
        const bool col_on = xq < p.ow;
        const int xs = min(xq, st.pitch - 4);
        unsigned long long* ext_slot = reinterpret_cast<unsigned long long*>(smem + p.ext_off) + wave * 32;
        if (EXT && lane < 32) ext_slot[lane] = 0ull;   // ordered before the first update by the fences below
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
#pragma unroll 1
            for (int round = 0; round < 2; ++round) {
                if ((q >> 1) == round) put(acc[mb]);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                {   // (no divergent region: lanes right of the map compute on the clamped column xs - see the plain tiling)
                    int cur_rho = -1;
                    double ps1[4][CH], pp1[4] = {0, 0, 0, 0}, psum2[4] = {0, 0, 0, 0}, psq[4] = {0, 0, 0, 0},
                           prsq[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int cc = 0; cc < CH; ++cc) ps1[k][cc] = 0.0;
#pragma unroll 1
                    for (int s8 = 0; s8 < 8; ++s8) {
                        const int i = 8 * round + s8;
                        const int t = i & ntm, rho = i >> lg;
                        const int yy = y0 + wave * wave_rows + mb * R + rho;
                        if (t >= p.n_list || yy >= p.oh || (p.only_li >= 0 && t != p.only_li)) continue;   // wave-uniform
                        if (rho != cur_rho) {
                            cur_rho = rho;
                            const size_t sidx = (size_t)yy * st.pitch + xs;
#pragma unroll
                            for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                                for (int cc = 0; cc < CH; ++cc) {
                                    const double2 a = *reinterpret_cast<const double2*>(st.t[cc] + sidx + 2 * hh);
                                    ps1[2 * hh][cc] = a.x;
                                    ps1[2 * hh + 1][cc] = a.y;
                                }
                                if (kNeedSum2) {
                                    const double2 b = *reinterpret_cast<const double2*>(st.sum2 + sidx + 2 * hh);
                                    psum2[2 * hh] = b.x;
                                    psum2[2 * hh + 1] = b.y;
                                }
                                if (kNormed) {
                                    const double2 d = *reinterpret_cast<const double2*>(st.sq + sidx + 2 * hh);
                                    psq[2 * hh] = d.x;
                                    psq[2 * hh + 1] = d.y;
                                    if constexpr (CH == 1) {     // the single-channel statistics pass writes 1/sqrt too
                                        const double2 e = *reinterpret_cast<const double2*>(p.rm_rsq + sidx + 2 * hh);
                                        prsq[2 * hh] = e.x;
                                        prsq[2 * hh + 1] = e.y;
                                    } else {
                                        prsq[2 * hh] = d.x > 0.0 ? 1.0 / d.x : 0.0;
                                        prsq[2 * hh + 1] = d.y > 0.0 ? 1.0 / d.y : 0.0;
                                    }
                                }
                            }
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                double s1all = ps1[k][0];       // bias term: 128 * sum over channels of S1 (exact integers)
#pragma unroll
                                for (int cc = 1; cc < CH; ++cc) s1all += ps1[k][cc];
                                pp1[k] = 128.0 * s1all;
                                if (kMaskedNormed) prsq[k] = 1.0 / sqrt(psum2[k]);
                            }
                        }
                        const MfTemplConst T = tcl[t];
                        const v4i a4 = *reinterpret_cast<const v4i*>(&epi[s8 * kMfEpiPitch + rd_off]);
                        const int a32[4] = {a4.x, a4.y, a4.z, a4.w};
                        bool wanted = col_on;          // (see the plain tiling's epilogue)
                        const bool pretest = kNormed && (p.hits_only || (!MASKED && !EXT && p.seg_skip));
                        if (pretest) {
                            bool pass = T.all_ones != 0;
                            const double rt = T.rtempl_norm;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                double num = (double)a32[k] + T.mfma_k;
#pragma unroll
                                for (int cc = 0; cc < CH; ++cc)
                                    num = fma(ps1[k][cc], METHOD == MTM_TM_CCOEFF_NORMED ? T.m128[cc] : 128.0, num);
                                if (METHOD == MTM_TM_SQDIFF_NORMED) num = fmax(psum2[k] - 2.0 * num + T.templ_sum2, 0.0);
                                const double qd = num * (prsq[k] * rt);
                                const double quality = METHOD == MTM_TM_SQDIFF_NORMED ? -qd : qd;
                                pass = pass || quality > (EXT ? T.ext_thr_lo : p.cand_thr_lo) || fabs(qd) >= 0.999999999;
                            }
                            wanted = wanted && pass;
                            if (__builtin_amdgcn_ballot_w64(wanted) == 0ull) continue;       // wave-uniform
                        }
                        float out[4];
                        constexpr bool kDefer = EXACT_DIV && !MASKED && kNormed;      // (see the plain tiling's epilogue)
                        constexpr bool kDeferM = EXACT_DIV && kMaskedNormed;
                        bool redo[4] = {false, false, false, false}, sat[4] = {false, false, false, false};
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            out[k] = MASKED ? finish_lean_masked<(METHOD < 0 ? 0 : METHOD), EXACT_DIV, kDeferM>(a32[k], pp1[k], psum2[k],
                                                                                                              prsq[k], T, &redo[k])
                                            : finish_fast<(METHOD < 0 ? 0 : METHOD), EXACT_DIV, CH, kDefer>(a32[k], ps1[k], pp1[k],
                                                                                                          psum2[k], psq[k], prsq[k],
                                                                                                          T, &redo[k], &sat[k]);
                        if constexpr (kDefer) {
                            if (__builtin_amdgcn_ballot_w64(redo[0] || redo[1] || redo[2] || redo[3]) != 0ull) {   // wave-uniform
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    float e = finish_fast<(METHOD < 0 ? 0 : METHOD), EXACT_DIV, CH>(a32[k], ps1[k], pp1[k], psum2[k],
                                                                                                  psq[k], prsq[k], T);
                                    asm volatile("" : "+v"(e));
                                    out[k] = redo[k] ? e : out[k];
                                }
                            }
                            if (__builtin_amdgcn_ballot_w64(sat[0] || sat[1] || sat[2] || sat[3]) != 0ull) {     // wave-uniform
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    float e = finish_saturated<(METHOD < 0 ? 0 : METHOD), CH>(a32[k], ps1[k], pp1[k], psum2[k], psq[k], T);
                                    asm volatile("" : "+v"(e));
                                    out[k] = sat[k] ? e : out[k];
                                }
                            }
                        }
                        if constexpr (kDeferM) {
                            if (__builtin_amdgcn_ballot_w64(redo[0] || redo[1] || redo[2] || redo[3]) != 0ull) {   // wave-uniform
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    float e = finish_lean_masked<(METHOD < 0 ? 0 : METHOD), EXACT_DIV>(a32[k], pp1[k], psum2[k], prsq[k], T);
                                    asm volatile("" : "+v"(e));
                                    out[k] = redo[k] ? e : out[k];
                                }
                            }
                        }
                        const bool ones = !MASKED && T.all_ones != 0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) out[k] = ones ? 1.0f : out[k];
                        if (pretest && !p.hits_only) {          // seg_skip (see the plain tiling's epilogue)
                            const float below = METHOD == MTM_TM_SQDIFF_NORMED ? INFINITY : -INFINITY;
#pragma unroll
                            for (int k = 0; k < 4; ++k) out[k] = wanted ? out[k] : below;
                        }
                        const float hi4 = fmaxf(fmaxf(out[0], out[1]), fmaxf(out[2], out[3]));
                        const float lo4 = fminf(fminf(out[0], out[1]), fminf(out[2], out[3]));
                        const bool above = wanted && (p.cand_min ? -lo4 : hi4) > p.cand_thr;
                        if constexpr (EXT) {
                            ext_update(out, yy, T.ext_hi, &ext_slot[t], wanted);
                        } else if (p.cand_on) {
                            if (__builtin_amdgcn_ballot_w64(above) != 0ull) emit_at(out, t, yy, above ? 15u : 0u);
                        }
                        if (!p.hits_only) {
                            if (col_on) store4(maps + T.map_off + (size_t)yy * T.map_pitch + xq, out);
                            if constexpr (!EXT) if (p.seg_flags != nullptr) {      // segment flags (see the plain tiling's epilogue)
                                if (__builtin_amdgcn_ballot_w64(above) != 0ull)
                                    p.seg_flags[(size_t)T.flag_base + (size_t)yy * p.flag_rstride + (x0 >> 8)] = 1;
                            }
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (EXT && lane < p.rm_nt) {            // one global atomic per template this wave improved
            const unsigned long long key = ext_slot[lane];
            if (key && lane < p.n_list) atomicMax(&p.ext_best[2 * tlist[lane] + p.cand_min], key);
        }
    } else if constexpr (METHOD == kMfRaw) {
        // ---- raw mode: transpose through LDS and store the int32 accumulators, 4 pixels per lane
        __syncthreads();              // every wave is done reading the image tile: the buffers alias it
        const bool lane_on = y < p.oh && xq < p.ow;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
#pragma unroll 1
            for (int round = 0; round < 2; ++round) {
                if ((q >> 1) == round) put(acc[mb]);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                {   // (the only divergent code between the transposition stages are the stores themselves)
#pragma unroll
                    for (int s8 = 0; s8 < 8; ++s8) {
                        const int li = tg * MB * 16 + mb * 16 + 8 * round + s8;
                        if (li >= p.n_list) break;                          // wave-uniform
                        const v4i a4 = *reinterpret_cast<const v4i*>(&epi[s8 * kMfEpiPitch + rd_off]);
                        int* orow = p.raw_out + (size_t)li * p.raw_map + (size_t)y * p.raw_pitch + xq;
                        if (lane_on && xq + 3 < p.ow) {
                            *reinterpret_cast<v4i*>(orow) = a4;
                        } else {
                            const int a32[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (lane_on && xq + i < p.ow) orow[i] = a32[i];
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    } else if constexpr (METHOD == kMfU16) {
        // ---- uint16 epilogue.  The work item holds, for templates 16 tg .. 16 tg + 15: a_hh = u16_hh (I_hi x T_hi),
        // a_mid = acc[0] (I_hi x T_lo + I_lo x T_hi) and a_ll = acc[1] (I_lo x T_lo).  With R_xy = a_xy + 128 S1_x + K_y
        // (K_y = 128 sum(T_y) - 16384 A, S1_x the window sum of byte plane x):
        //   sum I*T = 65536 R_hh + 256 (R_hl + R_lh) + R_ll
        //           = 65536 a_hh + 256 a_mid + a_ll  +  32896 S1  +  257 (256 K_hi + K_lo)
        // because 256 S1_hi + S1_lo = S1, the window sum of the 16-bit image the statistics pass provides - the byte
        // planes need no sums of their own.  Every term is an integer < 2^53: exact in float64 in any order.  Then the
        // common normalisation (run-time method).  Four stages of 4 templates: the lanes with q == stage put the three
        // sets of their 4 templates into the wave's buffer (rows 0-3 / 4-7 / 8-11).
        __syncthreads();              // every wave is done reading the image tile: the buffers alias it
        const bool lane_on = y < p.oh && xq < p.ow;
        const int method = p.method;
        const bool need_sum2 = method == MTM_TM_SQDIFF || method == MTM_TM_SQDIFF_NORMED;
        const bool normed = method == MTM_TM_SQDIFF_NORMED || method == MTM_TM_CCORR_NORMED || method == MTM_TM_CCOEFF_NORMED;
        const int n_here = min(16, p.n_list - tg * 16);
        // ---- hits-only screen (TM_CCORR_NORMED / TM_CCOEFF_NORMED): the per-lane bound of the 8-bit tilings (see the
        // row-multiplexed epilogue) on the three partial sums.  Lane (j, q) holds templates 4 q + e at the 16 outputs
        // 16 j + c of the wave's row; 65536 a_hh + 256 a_mid + a_ll is formed in float32 (three conversions, two fused
        // multiply-adds: at most 4 x 2^-24 x 66049 x 16384 h w = 258.1 h w away from the integer, whatever the partial
        // sums are), its maximum over c per template bounds the numerator over the block together with the range of S1
        // over the block, and the smallest sqrt statistic of the block bounds the denominator (StatPlanes::blk, written
        // by stats_u16_kernel).  A wave none of whose lanes can hold a candidate skips the float64 epilogue: 0.53 of
        // this kernel's 3.2 ms at 4K x 32 templates.
        if (p.hits_only && p.screen_l1 && st.blk != nullptr && (EXT || p.cand_thr_lo >= 0.0) &&
            (method == MTM_TM_CCORR_NORMED || method == MTM_TM_CCOEFF_NORMED)) {
            float vmax[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const v4i hh = u16_hh[c], md = acc[0][c], ll = acc[MB - 1][c];
                vmax[0] = fmaxf(vmax[0], fmaf(65536.0f, (float)hh.x, fmaf(256.0f, (float)md.x, (float)ll.x)));
                vmax[1] = fmaxf(vmax[1], fmaf(65536.0f, (float)hh.y, fmaf(256.0f, (float)md.y, (float)ll.y)));
                vmax[2] = fmaxf(vmax[2], fmaf(65536.0f, (float)hh.z, fmaf(256.0f, (float)md.z, (float)ll.z)));
                vmax[3] = fmaxf(vmax[3], fmaf(65536.0f, (float)hh.w, fmaf(256.0f, (float)md.w, (float)ll.w)));
            }
            const int bj = min((x0 >> 4) + j, st.blk_pitch - 1);
            const double* bp = st.blk + ((size_t)min(y, p.oh - 1) * st.blk_pitch + bj) * 4;
            const double2 lohi = *reinterpret_cast<const double2*>(bp);
            const double third = bp[2];
            const double slack = 258.1 * (double)p.h * (double)p.w;
            bool pass1 = false;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int lt = 4 * q + e;
                const bool live = lt < n_here && y < p.oh;
                const MfTemplConst& T = tcl[min(lt, 15)];
                const double thr_lo_e = EXT ? T.ext_thr_lo : p.cand_thr_lo;
                const double hi = EXT ? fmin(thr_lo_e, 0.999999) - 1e-6 : p.screen_hi;
                const double m = method == MTM_TM_CCOEFF_NORMED ? 32896.0 - T.mean[0] : 32896.0;
                const double bound = (((double)vmax[e] + slack) + T.mfma_k) + fmax(m * lohi.x, m * lohi.y);
                const bool could = T.all_ones != 0 || bound > hi * T.templ_norm * fmax(third, p.sq_floor);
                pass1 = pass1 || (live && (thr_lo_e < 0.0 || could));
            }
            if (__builtin_amdgcn_ballot_w64(pass1) == 0ull) continue;      // wave-uniform; no work-group barrier below
        }
        // the window statistics of this lane's four outputs - behind the screen: 40 registers that would otherwise be
        // live across it next to the 192 accumulator registers (they went to scratch memory in every work item: 63 MB of
        // HBM writes per 4K launch), and plane reads that the waves the screen sends away never need
        // Round 5: NO divergent region from here to the stores.  With `if (lane_on)` around the statistics loads and the
        // normalisation this epilogue returned wrong scores in row segments that are only partly inside the map: pixels
        // 4..6 of every 16-pixel block, templates of stages 1..3, depending on what earlier launches of the process had
        // done (profiles/r05_flake/diag.txt).  The lanes that WRITE the accumulators into the transposition
        // buffer (q == stage) are not the lanes of the pixels, 192 accumulator registers are live and ~300 bytes of them
        // spill: a spill / reload pair under two different execution masks is the suspected mechanism.  Lanes outside the
        // map now compute on clamped coordinates like everyone else; only the stores and the candidate list look at lane_on.
        double us1[4] = {0.0, 0.0, 0.0, 0.0}, up1[4], usum2[4] = {0.0, 0.0, 0.0, 0.0}, usq[4] = {0.0, 0.0, 0.0, 0.0}, ursq[4];
        {
            const size_t sidx = (size_t)min(y, p.oh - 1) * st.pitch + min(xq, st.pitch - 4);       // pitch is a multiple of 4: the 4 values exist
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const double2 a = *reinterpret_cast<const double2*>(st.t[0] + sidx + 2 * hh);
                us1[2 * hh] = a.x, us1[2 * hh + 1] = a.y;
                if (need_sum2) {
                    const double2 d = *reinterpret_cast<const double2*>(st.sum2 + sidx + 2 * hh);
                    usum2[2 * hh] = d.x, usum2[2 * hh + 1] = d.y;
                }
                if (normed) {
                    const double2 d = *reinterpret_cast<const double2*>(st.sq + sidx + 2 * hh);
                    usq[2 * hh] = d.x, usq[2 * hh + 1] = d.y;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            up1[i] = 32896.0 * us1[i];
            ursq[i] = (normed && usq[i] > 0.0) ? 1.0 / usq[i] : 0.0;
        }
        unsigned long long* ext_slot = reinterpret_cast<unsigned long long*>(smem + p.ext_off) + wave * 32;
        if (EXT && lane < 32) ext_slot[lane] = 0ull;   // ordered before the first update by the fences below
        // (this kernel's transposition buffer has 12 rows per wave - kMfU16EpiBytesPerWave - so that the three sets of a
        // stage's four templates go through it together)
        int* epi3 = reinterpret_cast<int*>(smem + wave * kMfU16EpiBytesPerWave);
#pragma unroll 1
        for (int stage = 0; stage < 4; ++stage) {
            if (4 * stage >= n_here) break;                                   // wave-uniform
            if (q == stage) {
                int* dst = &epi3[16 * j];
                const int rot = mf_epi_rot(j);
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const int col = (c + rot) & 15;
                    dst[0 * kMfEpiPitch + col] = u16_hh[c].x;
                    dst[1 * kMfEpiPitch + col] = u16_hh[c].y;
                    dst[2 * kMfEpiPitch + col] = u16_hh[c].z;
                    dst[3 * kMfEpiPitch + col] = u16_hh[c].w;
                    dst[4 * kMfEpiPitch + col] = acc[0][c].x;
                    dst[5 * kMfEpiPitch + col] = acc[0][c].y;
                    dst[6 * kMfEpiPitch + col] = acc[0][c].z;
                    dst[7 * kMfEpiPitch + col] = acc[0][c].w;
                    dst[8 * kMfEpiPitch + col] = acc[MB - 1][c].x;
                    dst[9 * kMfEpiPitch + col] = acc[MB - 1][c].y;
                    dst[10 * kMfEpiPitch + col] = acc[MB - 1][c].z;
                    dst[11 * kMfEpiPitch + col] = acc[MB - 1][c].w;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            {
#pragma unroll 1
                for (int e = 0; e < 4; ++e) {
                    const int lt = 4 * stage + e, li = tg * 16 + lt;
                    if (lt >= n_here) break;                                     // wave-uniform
                    if (p.only_li >= 0 && li != p.only_li) continue;
                    const MfTemplConst T = tcl[lt];
                    const v4i hh4 = *reinterpret_cast<const v4i*>(&epi3[e * kMfEpiPitch + rd_off]);
                    const v4i md4 = *reinterpret_cast<const v4i*>(&epi3[(4 + e) * kMfEpiPitch + rd_off]);
                    const v4i ll4 = *reinterpret_cast<const v4i*>(&epi3[(8 + e) * kMfEpiPitch + rd_off]);
                    const int a_hh[4] = {hh4.x, hh4.y, hh4.z, hh4.w}, a_md[4] = {md4.x, md4.y, md4.z, md4.w};
                    const int a_ll[4] = {ll4.x, ll4.y, ll4.z, ll4.w};
                    float out[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const double a = fma(65536.0, (double)a_hh[i], fma(256.0, (double)a_md[i], (double)a_ll[i]));
                        const double corr = a + (up1[i] + T.mfma_k);
                        out[i] = finish_rt<EXACT_DIV>(method, corr, us1[i], usum2[i], usq[i], ursq[i], T);
                    }
                    {
                        const bool ones = T.all_ones != 0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) out[i] = ones ? 1.0f : out[i];
                    }
                    if constexpr (EXT) {
                        ext_update(out, y, T.ext_hi, &ext_slot[lt], lane_on);
                    } else if (p.cand_on) {
                        const float hi = fmaxf(fmaxf(out[0], out[1]), fmaxf(out[2], out[3]));
                        const float lo = fminf(fminf(out[0], out[1]), fminf(out[2], out[3]));
                        const bool above = lane_on && (p.cand_min ? -lo : hi) > p.cand_thr;
                        if (__builtin_amdgcn_ballot_w64(above) != 0ull) emit_at(out, li, y, above ? 15u : 0u);     // wave-uniform entry
                    }
                    if (lane_on && !p.hits_only) store4(maps + T.map_off + (size_t)y * T.map_pitch + xq, out);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (EXT && lane < 16) {        // one global atomic per template this wave improved
            const unsigned long long key = ext_slot[lane];
            const int li = tg * 16 + lane;
            if (key && li < p.n_list) atomicMax(&p.ext_best[2 * tlist[li] + p.cand_min], key);
        }
    } else if constexpr (C1) {
        // ---- hits-only screen, straight from the accumulator registers (no LDS transposition): in this
        // mode nothing is stored, and almost no work item holds a candidate.  In the MFMA C/D layout lane
        // (j, q) owns the 16 consecutive pixels 16 j + c and the templates 16 mb + 4 q + e.  Two levels:
        //   1. a rigorous bound per lane and template.  A candidate needs num = acc + K + m S1 > thr * templ_norm * sqrt
        //      at its pixel, so none of the lane's 16 pixels can be one unless
        //         max(acc) + K + max(m S1_min, m S1_max)  >  thr * templ_norm * sqrt_min
        //      - one integer v_max per accumulator register, the ranges of the 16 pixels' statistics shared by the four
        //      lane groups of a column block, a handful of float64 operations per template.  (sqrt_min: a flat window has
        //      sqrt = 0 and scores 0, never a candidate of a non-negative threshold; a window that is not flat has
        //      sqrt >= 1 / sqrt(w h), the sums being integers - sq_floor.)
        //   2. only if some lane's bound says "possible" (or a saturated / constant-template output is): the running
        //      extremes of u = (acc + K + S1 (128 - mean)) / sqrt per pixel (reciprocal by v_rcp_f64, ~2^-26), compared
        //      once with the threshold scaled by the template norm, lowered by 1e-6 - round 2's screen, 5 float64
        //      operations per accumulator register.
        // Only if level 2 still sees a possible candidate does the wave run the full epilogue below, which
        // repeats the exact test.  Results are therefore those of the full epilogue.
        bool wave_has_work = true;
        if constexpr (CH == 1 && !MASKED && (METHOD == MTM_TM_CCORR_NORMED || METHOD == MTM_TM_CCOEFF_NORMED)) {
            // (quotients <= -1 saturate to -1 or 0, which can only be candidates below a negative threshold:
            // such calls skip the screen instead of tracking the minima as well)
            if ((p.hits_only || (!EXT && p.seg_skip)) && (EXT || p.cand_thr_lo >= 0.0)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const uint8_t* sw0 = smem + p.st_off + wave * (R2 ? MB * 4 * 1024 : mf_stat_bytes_per_wave(1));
                bool pass1 = false;
                if (p.screen_l1) {
                    // v_min_f64 / v_max_f64 as they are (fmin / fmax on loaded values cost a canonicalising v_max each)
                    auto min64 = [](double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
                    auto max64 = [](double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
                    // value of lane ^ off (ds_bpermute; the address from this scope's lane index - __shfl_xor takes it from a
                    // lane id the compiler computes ahead of the item loop and spills across the K loop)
                    auto xlane = [&](double v, int off) {
                        const int addr = (lane ^ off) << 2;
                        const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
                        const int hi2 = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
                        return __hiloint2double(hi2, lo);
                    };
                    double s1lo[R2 ? MB : 1], s1hi[R2 ? MB : 1], sqlo[R2 ? MB : 1];
                    if constexpr (R2) {
                        // the ranges of this lane's column block, one row at a time, as stats_u8_kernel wrote them
#pragma unroll
                        for (int r = 0; r < MB; ++r) {
                            const uint8_t* sb = smem + p.st_off + wave * (((MB + 1) / 2) * 1024) + r * 512 + j * 32;
                            const double2 lohi = *reinterpret_cast<const double2*>(sb);
                            s1lo[r] = lohi.x;
                            s1hi[r] = lohi.y;
                            sqlo[r] = *reinterpret_cast<const double*>(sb + 16);
                        }
                    } else {
#pragma unroll
                    for (int r = 0; r < 1; ++r) {
                        // this lane's share of its column block: pixels 16 j + 4 q .. + 3; the block's other three lane
                        // groups (lanes j + 16, + 32, + 48 around) hold the rest
                        const uint8_t* sw = sw0 + (4 * j + q) * 16;
                        constexpr int kSqPlane = 4;
                        const double2 sa = *reinterpret_cast<const double2*>(sw + 0 * 1024);
                        const double2 sb = *reinterpret_cast<const double2*>(sw + 1 * 1024);
                        const double2 qa = *reinterpret_cast<const double2*>(sw + kSqPlane * 1024);
                        const double2 qb = *reinterpret_cast<const double2*>(sw + (kSqPlane + 1) * 1024);
                        s1lo[r] = min64(min64(sa.x, sa.y), min64(sb.x, sb.y));
                        s1hi[r] = max64(max64(sa.x, sa.y), max64(sb.x, sb.y));
                        sqlo[r] = min64(min64(qa.x, qa.y), min64(qb.x, qb.y));
                    }
#pragma unroll
                    for (int off = 16; off <= 32; off <<= 1) {
                        s1lo[0] = min64(s1lo[0], xlane(s1lo[0], off));
                        s1hi[0] = max64(s1hi[0], xlane(s1hi[0], off));
                        sqlo[0] = min64(sqlo[0], xlane(sqlo[0], off));
                    }
                    }
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) {
                        int amax[4] = {INT_MIN, INT_MIN, INT_MIN, INT_MIN};
#pragma unroll
                        for (int c = 0; c < 16; ++c) {
                            const v4i a = acc[mb][c];
                            amax[0] = max(amax[0], a.x);
                            amax[1] = max(amax[1], a.y);
                            amax[2] = max(amax[2], a.z);
                            amax[3] = max(amax[3], a.w);
                        }
                        const int r = R2 ? mb : 0;
                        const double sq_eff = fmax(sqlo[r], p.sq_floor);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int lt = (R2 ? 0 : 16 * mb) + 4 * q + e;
                            const MfTemplConst& T = tcl[lt];        // (beyond the list: constants that cannot pass, see above)
                            const double m = METHOD == MTM_TM_CCOEFF_NORMED ? T.m128[0] : 128.0;
                            const double thr_lo_e = EXT ? T.ext_thr_lo : p.cand_thr_lo;
                            const double hi = EXT ? fmin(thr_lo_e, 0.999999) - 1e-6 : p.screen_hi;
                            const double bound = ((double)amax[e] + T.mfma_k) + fmax(m * s1lo[r], m * s1hi[r]);
                            pass1 = pass1 || T.all_ones != 0 || thr_lo_e < 0.0 || bound > hi * T.templ_norm * sq_eff;
                        }
                    }
                } else {
                    pass1 = true;
                }
                wave_has_work = __builtin_amdgcn_ballot_w64(pass1) != 0ull;
                if (wave_has_work) {
                bool pass = false;
                // one MFMA group (4 templates of this lane) at a time: 4 x (K, 128 - mean, running extremes)
                // stay in registers next to the 64 * MB accumulators
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const uint8_t* sw = sw0;                                 // (multi-row variants: group mb = the wave's row mb,
                    constexpr int kSqPlane = 4;                              //  per-pixel statistics straight from memory)
                    const size_t grow = (size_t)min(y + (R2 ? mb : 0), p.oh - 1) * st.pitch;
                    double kk[4], mm[4], umax[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int lt = (R2 ? 0 : 16 * mb) + 4 * q + e;
                        const MfTemplConst& T = tcl[lt];
                        kk[e] = T.mfma_k;
                        mm[e] = METHOD == MTM_TM_CCOEFF_NORMED ? T.m128[0] : 128.0;
                        pass = pass || T.all_ones != 0;
                        umax[e] = -INFINITY;
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int L = 4 * j + g;
                        double2 sa, sb, qa, qb;
                        if constexpr (R2) {
                            const size_t gi = grow + min(x0 + 4 * L, st.pitch - 4);
                            sa = *reinterpret_cast<const double2*>(st.t[0] + gi);
                            sb = *reinterpret_cast<const double2*>(st.t[0] + gi + 2);
                            qa = *reinterpret_cast<const double2*>(st.sq + gi);
                            qb = *reinterpret_cast<const double2*>(st.sq + gi + 2);
                        } else {
                            sa = *reinterpret_cast<const double2*>(sw + 0 * 1024 + L * 16);
                            sb = *reinterpret_cast<const double2*>(sw + 1 * 1024 + L * 16);
                            qa = *reinterpret_cast<const double2*>(sw + kSqPlane * 1024 + L * 16);
                            qb = *reinterpret_cast<const double2*>(sw + (kSqPlane + 1) * 1024 + L * 16);
                        }
                        const double s1g[4] = {sa.x, sa.y, sb.x, sb.y}, sqg[4] = {qa.x, qa.y, qb.x, qb.y};
#pragma unroll
                        for (int c4 = 0; c4 < 4; ++c4) {
                            const double rq = sqg[c4] > 0.0 ? __builtin_amdgcn_rcp(sqg[c4]) : 0.0;
                            const v4i a = acc[mb][4 * g + c4];
                            const int av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const double u = fma(s1g[c4], mm[e], (double)av[e] + kk[e]) * rq;
                                umax[e] = fmax(umax[e], u);
                            }
                        }
                        // (level 2 is the rare path: one group of statistics in flight at a time - with all sixteen loads of a
                        // row hoisted to the top the allocator ran out of registers next to the 128 accumulators)
                        if constexpr (R2) __builtin_amdgcn_sched_barrier(0);
                    }
                    // quotient q = u / templ_norm: candidate if q > threshold; q >= 1 saturates (above any
                    // threshold < 1, tested anyway)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int lt = (R2 ? 0 : 16 * mb) + 4 * q + e;
                        const double tn = tcl[lt].templ_norm;
                        // extremum mode: the template's own running best is the threshold (negative or none
                        // yet: no screen for this template)
                        const double thr_lo_e = EXT ? tcl[lt].ext_thr_lo : p.cand_thr_lo;
                        const double hi = EXT ? fmin(thr_lo_e, 0.999999) - 1e-6 : p.screen_hi;
                        pass = pass || thr_lo_e < 0.0 || umax[e] > hi * tn;
                    }
                }
                wave_has_work = __builtin_amdgcn_ballot_w64(pass) != 0ull;
                }   // level 2
            }
        }
        // ---- single channel, method fixed at compile time.  The statistics of the lane's pixels
        // come from the LDS prefetch; the template loop is software pipelined (constants and
        // accumulators of the next template are requested before the current one is normalised)
        // and branch-free apart from wave-uniform tests.
        double ps1[4][CH], pp1[4], psum2[4], psq[4], prsq[4];
        // statistics of the lane's four pixels: from the LDS prefetch into registers (R2: per row, inside the loop)
        auto load_stats = [&](int r2_row) {
            const uint8_t* sl = smem + p.st_off + wave * mf_stat_bytes_per_wave(CH) + lane * 16;
            constexpr int kSq = 2 * CH + 2;
            // multi-row variants: this path is the rare one behind the screen - its statistics come from memory
            const size_t gi = (size_t)min(y + r2_row, p.oh - 1) * st.pitch + min(xq, st.pitch - 4);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                for (int cc = 0; cc < CH; ++cc) {
                    const double2 a = R2 ? *reinterpret_cast<const double2*>(st.t[cc] + gi + 2 * hh)
                                         : *reinterpret_cast<const double2*>(sl + (2 * cc + hh) * 1024);
                    ps1[2 * hh][cc] = a.x;
                    ps1[2 * hh + 1][cc] = a.y;
                }
                double2 b = make_double2(0.0, 0.0), d = make_double2(0.0, 0.0);
                if (kNeedSum2) b = *reinterpret_cast<const double2*>(sl + (2 * CH + hh) * 1024);
                if (kNormed) d = R2 ? *reinterpret_cast<const double2*>(st.sq + gi + 2 * hh)
                                    : *reinterpret_cast<const double2*>(sl + (kSq + hh) * 1024);
                psum2[2 * hh] = b.x;
                psum2[2 * hh + 1] = b.y;
                psq[2 * hh] = d.x;
                psq[2 * hh + 1] = d.y;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double s1all = ps1[i][0];           // bias term: 128 * sum over channels of S1 (exact integers)
#pragma unroll
                for (int cc = 1; cc < CH; ++cc) s1all += ps1[i][cc];
                pp1[i] = 128.0 * s1all;
                // (1 / sq once per pixel, shared by the work item's templates: the reciprocal path's factor and, in IEEE-division
                // builds, quotient_as_float's; the hits-only pre-test of those builds - kExactNoRcp below - needs none)
                prsq[i] = (kNormed && psq[i] > 0.0) ? 1.0 / psq[i] : 0.0;
                if (kMaskedNormed) prsq[i] = 1.0 / sqrt(psum2[i]);
            }
        };
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (!R2) load_stats(0);
        __syncthreads();              // every wave is done reading the image tile: the buffers alias it
        if (!wave_has_work) continue;                       // wave-uniform; no work-group barrier below
        const bool col_on = xq < p.ow;
        unsigned long long* ext_slot = reinterpret_cast<unsigned long long*>(smem + p.ext_off) + wave * 32;
        if (EXT && lane < 32) ext_slot[lane] = 0ull;   // ordered before the first update by the fences below
        // Round 6: NO divergent region between put() and the stores, and no ballot / continue under a divergent mask.  Lanes
        // outside the map run the same code on clamped coordinates (the statistics above are loaded that way); `lane_on` only
        // gates what leaves the lane - stores, candidate records, extremum keys, segment flags.  Why: the lanes that WRITE the
        // transposition buffer (put) are not the lanes of the pixels, so every lane's accumulators must survive whatever the
        // epilogue does, and this toolchain has placed accumulator spills ahead of the exec restore of an `if (lane_on)`
        // region (profiles/r06_flake/: the lanes outside the map then reload other launches' scratch memory).
        // tools/spill_exec_scan.py checks every build for that placement; this form keeps the opportunity for it small.
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int yrow = y + (R2 ? mb : 0);
            const bool lane_on = yrow < p.oh && col_on;
            if constexpr (R2) load_stats(mb);
#pragma unroll 1
            for (int round = 0; round < 2; ++round) {
                if ((q >> 1) == round) put(acc[mb]);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // LDS writes above, reads below
                __builtin_amdgcn_wave_barrier();
                {
                    const int lt0 = (R2 ? 0 : mb * 16) + 8 * round;       // template inside this work item
                    v4i an = *reinterpret_cast<const v4i*>(&epi[rd_off]);
#pragma unroll 1
                    for (int s8 = 0; s8 < 8; ++s8) {
                        // template = 4*q_src + e with q_src = 2*round + (s8 >> 2), e = s8 & 3
                        // (C/D layout of the 16x16 MFMA: row = 4*(lane>>4) + reg)
                        // (round 6: the constants are read where they are used - a second MfTemplConst in flight, requested one
                        // template ahead, cost ~20 registers next to the 128 accumulators and was what spilled)
                        const MfTemplConst& T = tcl[lt0 + s8];
                        const v4i a4 = an;
                        if (s8 < 7) an = *reinterpret_cast<const v4i*>(&epi[(s8 + 1) * kMfEpiPitch + rd_off]);
                        const int li = tg * kTG + lt0 + s8;
                        if (li >= p.n_list || (p.only_li >= 0 && li != p.only_li)) continue;   // wave-uniform
                        const int a32[4] = {a4.x, a4.y, a4.z, a4.w};
                        // wanted: this lane's four outputs may leave it (inside the map; behind the pre-test: can pass)
                        bool wanted = lane_on;
                        constexpr bool kPreTest = kNormed;
                        const bool pretest = kPreTest && (p.hits_only || (!MASKED && !EXT && p.seg_skip));
                        if (pretest) {
                            // Only outputs that can reach the threshold need the full normalisation (hits-only: nothing
                            // is stored; seg_skip: a value certainly below the threshold needs no value): q is the
                            // float64 quotient of the reciprocal path (within 2 ulp(double) of num / t), compared with a
                            // threshold lowered by 8 float32 ulps.  |q| >= 1 (saturation cases) and constant templates
                            // always take the exact path below.
                            bool pass = T.all_ones != 0;
                            const double rt = T.rtempl_norm;
                            constexpr bool kExactNoRcp = EXACT_DIV && !MASKED;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                double num = (double)a32[i] + T.mfma_k;
#pragma unroll
                                for (int cc = 0; cc < CH; ++cc)
                                    num = fma(ps1[i][cc], METHOD == MTM_TM_CCOEFF_NORMED ? T.m128[cc] : 128.0, num);
                                if (METHOD == MTM_TM_SQDIFF_NORMED) num = fmax(psum2[i] - 2.0 * num + T.templ_sum2, 0.0);
                                if constexpr (kExactNoRcp) {
                                    // the same test without a quotient: q > thr <=> num > thr * t for t > 0 (minima: -q > thr
                                    // <=> num < -thr * t); a flat window (t == 0) scores the rules' constant, 0 or 1
                                    const double tt = psq[i] * T.templ_norm;
                                    const double thr_lo = EXT ? T.ext_thr_lo : p.cand_thr_lo;
                                    const bool over = METHOD == MTM_TM_SQDIFF_NORMED ? num < -thr_lo * tt : num > thr_lo * tt;
                                    const bool flat_over = (METHOD == MTM_TM_SQDIFF_NORMED ? -1.0 : 0.0) > thr_lo;
                                    pass = pass || (tt > 0.0 ? (over || fabs(num) >= 0.999999999 * tt) : flat_over);
                                } else {
                                    const double qd = num * (prsq[i] * rt);
                                    const double quality = METHOD == MTM_TM_SQDIFF_NORMED ? -qd : qd;
                                    pass = pass || quality > (EXT ? T.ext_thr_lo : p.cand_thr_lo) || fabs(qd) >= 0.999999999;
                                }
                            }
                            wanted = wanted && pass;
                            // hits-only: nothing to do for a (template, segment) without a possible candidate.  seg_skip: such
                            // a row segment stays unwritten and unflagged (the peak pass reads flagged segments only and takes
                            // their unflagged neighbours as "below the threshold").  Wave-uniform either way.
                            if (__builtin_amdgcn_ballot_w64(wanted) == 0ull) continue;
                        }
                        float out[4];
                        constexpr bool kDefer = EXACT_DIV && !MASKED && kNormed;
                        constexpr bool kDeferM = EXACT_DIV && kMaskedNormed;
                        bool redo[4] = {false, false, false, false}, sat[4] = {false, false, false, false};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (MASKED)
                                out[i] = finish_lean_masked<METHOD, EXACT_DIV, kDeferM>(a32[i], pp1[i], psum2[i], prsq[i], T, &redo[i]);
                            else
                                out[i] = finish_fast<METHOD, EXACT_DIV, CH, kDefer>(a32[i], ps1[i], pp1[i], psum2[i], psq[i],
                                                                                    prsq[i], T, &redo[i], &sat[i]);
                            // (IEEE division: one quotient after the other - four interleaved division sequences need ~40
                            // registers more than the epilogue has next to 128 accumulators, and spilled)
                            if constexpr (EXACT_DIV && !MASKED && !kDefer) __builtin_amdgcn_sched_barrier(0);
                        }
                        if constexpr (kDefer) {
                            // 1.2e-7 of the outputs: the quotient next to a float32 rounding boundary, through the division
                            if (__builtin_amdgcn_ballot_w64(redo[0] || redo[1] || redo[2] || redo[3]) != 0ull) {   // wave-uniform
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    float e = finish_fast<METHOD, EXACT_DIV, CH>(a32[i], ps1[i], pp1[i], psum2[i], psq[i], prsq[i], T);
                                    asm volatile("" : "+v"(e));       // (keeps the division inside the branch)
                                    out[i] = redo[i] ? e : out[i];
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            }
                            // flat windows, saturated quotients: the rules' constants
                            if (__builtin_amdgcn_ballot_w64(sat[0] || sat[1] || sat[2] || sat[3]) != 0ull) {     // wave-uniform
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    float e = finish_saturated<METHOD, CH>(a32[i], ps1[i], pp1[i], psum2[i], psq[i], T);
                                    asm volatile("" : "+v"(e));
                                    out[i] = sat[i] ? e : out[i];
                                }
                            }
                        }
                        if constexpr (kDeferM) {
                            // masked normalised methods: num / sqrt(tms c2) without the square root and the division
                            if (__builtin_amdgcn_ballot_w64(redo[0] || redo[1] || redo[2] || redo[3]) != 0ull) {   // wave-uniform
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    float e = finish_lean_masked<METHOD, EXACT_DIV>(a32[i], pp1[i], psum2[i], prsq[i], T);
                                    asm volatile("" : "+v"(e));
                                    out[i] = redo[i] ? e : out[i];
                                }
                            }
                        }
                        if (!MASKED) {
                            const bool ones = T.all_ones != 0;
#pragma unroll
                            for (int i = 0; i < 4; ++i) out[i] = ones ? 1.0f : out[i];
                        }
                        if (pretest && !p.hits_only) {
                            // seg_skip: in a segment that is written, outputs that cannot pass get "below the threshold"
                            const float below = METHOD == MTM_TM_SQDIFF_NORMED ? INFINITY : -INFINITY;
#pragma unroll
                            for (int i = 0; i < 4; ++i) out[i] = wanted ? out[i] : below;
                        }
                        // does one of the four reach the threshold?  (emit_at() repeats the exact per-pixel test)
                        const float hi4 = fmaxf(fmaxf(out[0], out[1]), fmaxf(out[2], out[3]));
                        const float lo4 = fminf(fminf(out[0], out[1]), fminf(out[2], out[3]));
                        const bool above = wanted && (p.cand_min ? -lo4 : hi4) > p.cand_thr;
                        if constexpr (EXT) {
                            ext_update(out, yrow, T.ext_hi, &ext_slot[lt0 + s8], wanted);
                        } else if (p.cand_on) {
                            if (__builtin_amdgcn_ballot_w64(above) != 0ull) emit_at(out, li, yrow, above ? 15u : 0u);
                        }
                        if (!p.hits_only) {
                            if (lane_on) store4(maps + T.map_off + (size_t)yrow * T.map_pitch + xq, out);
                            if constexpr (!EXT) if (p.seg_flags != nullptr) {
                                // segment flags (dense images): the peak pass only visits row segments in which some output
                                // passes the threshold
                                if (__builtin_amdgcn_ballot_w64(above) != 0ull)
                                    p.seg_flags[(size_t)T.flag_base + (size_t)yrow * p.flag_rstride + (x0 >> 8)] = 1;
                            }
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // reads above, next stage's writes below
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (EXT && lane < kTG) {       // one global atomic per template this wave improved
            const unsigned long long key = ext_slot[lane];
            const int li = tg * kTG + lane;
            if (key && li < p.n_list) atomicMax(&p.ext_best[2 * tlist[li] + p.cand_min], key);
        }
    } else {
        // ---- generic path: any channel count, run-time method
        __syncthreads();              // every wave is done reading the image tile: the buffers alias it
#pragma unroll 1
        for (int stage = 0; stage < 2 * MB; ++stage) {
            const int mb = stage >> 1, round = stage & 1;
            if ((q >> 1) == round) {
                if (mb == 0) put(acc[0]);
                else put(acc[MB - 1]);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            {   // (no divergent region: lanes outside the map compute on clamped coordinates - see the single-channel path)
                const bool lane_on = y < p.oh && xq < p.ow;
#pragma unroll 1
                for (int s8 = 0; s8 < 8; ++s8) {
                    const int lt = mb * 16 + 8 * round + s8;
                    const int li = tg * MB * 16 + lt;
                    if (li >= p.n_list) break;                          // wave-uniform
                    if (p.only_li >= 0 && li != p.only_li) continue;
                    const MfTemplConst T = tcl[lt];
                    const v4i a4 = *reinterpret_cast<const v4i*>(&epi[s8 * kMfEpiPitch + rd_off]);
                    const int a32[4] = {a4.x, a4.y, a4.z, a4.w};
                    float out[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int x = min(xq + i, p.ow - 1);
                        const size_t sidx = (size_t)min(y, p.oh - 1) * st.pitch + x;
                        double tv[kMaxChans] = {0.0, 0.0, 0.0, 0.0};
                        double s1 = 0.0;
#pragma unroll
                        for (int cc = 0; cc < kMaxChans; ++cc)
                            if (cc < p.chans) {
                                tv[cc] = st.t[cc][sidx];
                                s1 += tv[cc];
                            }
                        const double corr = ((double)a32[i] + 128.0 * s1) + T.mfma_k;
                        out[i] = finish_vals<-1>(p.method, corr, tv, st.sum2[sidx], st.sq[sidx], T, p.chans);
                    }
                    if (p.cand_on) emit_at(out, li, y, lane_on ? 15u : 0u);
                    if (lane_on) store4(maps + T.map_off + (size_t)y * T.map_pitch + xq, out);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (cs_on) {
        // ---- the wave's staged candidates -> the global list: one atomic, records copied by all lanes (6 dwords each)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int n_st = __builtin_amdgcn_readfirstlane(cs_cnt[0]);
        if (n_st > 0) {
            unsigned long long base = 0ull;
            if (lane == 0) base = atomicAdd(p.cand_counter, (unsigned long long)n_st);
            const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base);
            const uint32_t bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
            base = ((unsigned long long)bhi << 32) | blo;
                for (int r = lane; r < n_st; r += 64)
                    mf_put_cand(p, base + (unsigned long long)r, cs_rec[r]);
        }
    }
    }   // epilogue scope
    } while (false);   // work item
    if (p.clk_out != nullptr && blockIdx.x == (gridDim.x >> 1) && threadIdx.x == 0) {
        const unsigned long long* clk0 = reinterpret_cast<const unsigned long long*>(s_item + 4);
        const unsigned long long dt = __builtin_readcyclecounter() - clk0[0], dr = __builtin_amdgcn_s_memrealtime() - clk0[1];
        if (dr > 0) *p.clk_out = (float)((double)dt * 100.0 / (double)dr);
    }
}

}  // namespace mtm
