// float32 images on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16).
//
// The reference casts every non-uint8 input to float32 and lets cv2 correlate it with a float64 DFT
// (MTM/__init__.py:71-74 -> cv2.matchTemplate, :92).  The exact float64 VALU kernel (ncc_f64_kernel) reproduces that
// to rounding but runs at the fp64 vector rate - 50 ms at 4K x 32 templates against 0.7 ms for uint8.  This kernel
// computes the same sliding dot products on the matrix cores to ~1e-5 of the normalised score (north_star's
// tolerance is 1e-4):
//
//   * every float is split into two bfloat16 pieces, v = v0 + v1 (v0 = RNE(v), v1 = RNE(v - v0): 16 significant
//     bits), and  I*T ~ I0*T0 + I0*T1 + I1*T0  - three bf16 MFMAs into one float32 accumulator; the dropped terms are
//     <= 2^-17 |I||T| each;
//   * what makes 16 bits enough is CENTRING, which is free: templates are packed as Tc = T - mean(T), so
//     sum I*T = sum I*Tc + mean(T)*S1 with S1 the (float64) window sum the statistics pass provides, and because
//     sum(Tc) = 0 ANY constant may be subtracted from the image inside a window without changing sum I*Tc.  A
//     work-group subtracts the mean of a sample of its own image tile while it stages the tile, so the pieces
//     carry the local contrast, not the local brightness;
//   * float32 accumulation of 128 MFMA results per 64 x 64 template (the K = 32 sums inside an MFMA are exact
//     products added in wide precision): ~1e-6 relative.
//
// GEMM mapping as in ncc_mfma_kernel with 2-byte elements: one MFMA = 16 templates x 16 pixels x 32 taps; the 16
// pixels of an MFMA are 8 apart (x_j = x0 + 8 j + c, phase c = 0..7), all 8 phases of a lane read the same two
// aligned 16-byte LDS chunks of a piece plane, shifted by c elements: even c are whole-dword shifts, odd c one
// v_alignbyte_b32 by 2 bytes.  A wave owns 128 consecutive pixels of one output row for 16 * MB templates
// (32 * MB accumulator VGPRs); a work-group = 4 waves = 4 rows sharing the two piece tiles.
// Unmasked templates only (float masks keep the float64 kernel); w <= 256.
#pragma once
#include "mtm_device_util.hip.h"
#include "mtm_bf16_params.h"
#include <type_traits>

namespace mtm {

// round-to-nearest-even bfloat16 bits of a finite float
__device__ __forceinline__ uint32_t bf16_rne(float v) {
    const uint32_t b = __float_as_uint(v);
    return (b + 0x7FFFu + ((b >> 16) & 1u)) >> 16;
}
// finish_unmasked on values already in registers (same arithmetic, same order)
__device__ __forceinline__ float bf_finish(int method, double corr, const double (&t)[kMaxChans], double sum2, double sq,
                                           const BfTemplConst& T, int chans) {
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
        if (an < tt) num = num / tt;
        else if (an < tt * 1.125) num = (num > 0.0) ? 1.0 : -1.0;
        else num = (method == MTM_TM_SQDIFF_NORMED) ? 1.0 : 0.0;
    }
    return (float)num;
}

// bf_finish + what the error bound of the refined routes applies to: for a normalised method the ratio num / t BEFORE
// the saturation rules (*r_out), or *exact_out = true when the rules yield a constant whatever the numerator is (flat
// window, constant template: t == 0).  Same operations as bf_finish, same results.
__device__ __forceinline__ float bf_finish_r(int method, double corr, const double (&t)[kMaxChans], double sum2, double sq,
                                             const BfTemplConst& T, int chans, double* r_out, bool* exact_out) {
    *r_out = 0.0;
    *exact_out = false;
    if (T.all_ones) {
        *exact_out = true;
        return 1.0f;
    }
    if (method == MTM_TM_CCORR) {
        *r_out = corr;
        return (float)corr;
    }
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
    if (!normed) *r_out = num;                  // raw sums: the value itself (the bound is in its units)
    if (normed) {
        const double tt = sq * T.templ_norm;
        const double an = fabs(num);
        const double r = num / tt;
        if (tt > 0.0) *r_out = r;
        else *exact_out = true;
        if (an < tt) num = r;
        else if (an < tt * 1.125) num = (num > 0.0) ? 1.0 : -1.0;
        else num = (method == MTM_TM_SQDIFF_NORMED) ? 1.0 : 0.0;
    }
    return (float)num;
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), in order
template <int N, class F>
__device__ __forceinline__ void bf_static_for(F& f) {
    if constexpr (N > 0) {
        bf_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// One K step: 32 taps x 8 phases x MB template groups x NP piece products (3: I0 T0 + I0 T1 + I1 T0; 1: I0 T0 alone - the
// one-product screen of round 6, below).  h0/h1 (l0/l1): the lane's two aligned 16-byte chunks of the first (second) piece
// plane; a0 / a1: the packed template pieces.
template <int MB, int NP>
__device__ __forceinline__ void bf_step(v4f (&acc)[MB][8], const v4i_b h0, const v4i_b h1, const v4i_b l0, const v4i_b l1,
                                        const v4i_b (&a0)[MB], const v4i_b (&a1)[MB]) {
    // (round 6: the lane's 32 bytes of a piece plane as ONE eight-register value and the phases' operands as sub-ranges of
    // it.  MFMA operand tuples start at even registers, so the odd-offset windows need copies: 16 of the 27 vector
    // instructions of a one-product step are moves.  This form did NOT remove them - the compiler canonicalises the
    // shuffles back into the per-element form and emits the same ISA; only a generated asm step as the int8 kernel's,
    // which places the copies in the MFMAs' shadow, would.  Kept: it states what the operands are.)
    typedef int v8i_b __attribute__((ext_vector_type(8)));
    const v8i_b W = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    const v8i_b V = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    v8i_b EW = W, EV = V;
#pragma unroll
    for (int m = 0; m < 7; ++m) {
        EW[m] = (int)__builtin_amdgcn_alignbyte((uint32_t)W[m + 1], (uint32_t)W[m], 2);
        if constexpr (NP == 3) EV[m] = (int)__builtin_amdgcn_alignbyte((uint32_t)V[m + 1], (uint32_t)V[m], 2);
    }
    auto window = [](const v8i_b& x, auto k_c) {
        constexpr int k = decltype(k_c)::value;
        return __builtin_shufflevector(x, x, k, k + 1, k + 2, k + 3);
    };
    auto phase = [&](auto ph_c) {
        constexpr int ph = decltype(ph_c)::value;
        constexpr int k = ph >> 1;
        const v4i_b bh = (ph & 1) ? window(EW, std::integral_constant<int, k>{}) : window(W, std::integral_constant<int, k>{});
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            v4f a = acc[mb][ph];
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a0[mb]), __builtin_bit_cast(v8bf, bh), a, 0, 0, 0);
            if constexpr (NP == 3) {
                const v4i_b bl = (ph & 1) ? window(EV, std::integral_constant<int, k>{}) : window(V, std::integral_constant<int, k>{});
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a1[mb]), __builtin_bit_cast(v8bf, bh), a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a0[mb]), __builtin_bit_cast(v8bf, bl), a, 0, 0, 0);
            }
            acc[mb][ph] = a;
        }
    };
    bf_static_for<8>(phase);
}

// NP = 1 (round 6): the ONE-PRODUCT SCREEN.  Only the leading bfloat16 piece of both operands is correlated - a third of
// the matrix-core work, half of the LDS and template traffic - and the result is good to ~2^-7 of
// sqrt(sum (I - mu)^2 sum (T - mean)^2) instead of ~2^-15 (bf16_rig_eps).  That is useless as a score and entirely sufficient
// as a screen: the refined hits-only routes (Bf16Params::rig / ext_raw) list every output whose UPPER bound passes the
// threshold (or reaches the running best) and the float64 chain decides on those, so a wider bound only lengthens the
// list - on images whose maps are sparse above the threshold by a few records.  The host launches this instantiation only
// where nothing but the list leaves the kernel (hits-only; PUBLISHED maps are never written from it - the masked classes'
// raw launches write scratch maps that mtm_maskf32.hip.h bounds with this launch's eps) and falls back to NP = 3 when the
// list overflows (mtm_api.hip).
template <int MB, int NP>
__global__ __launch_bounds__(256, 2) void ncc_bf16_kernel(Bf16Params p, const TemplDev* __restrict__ td,
                                                          const int* __restrict__ tlist,
                                                          const uint8_t* __restrict__ apack, StatPlanes st,
                                                          float* __restrict__ maps) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_bf[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const int per_xcd = (p.n_work + 7) >> 3;
    const int wid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (wid >= p.n_work) return;
    const int tg = wid % p.ntg;
    const int rest = wid / p.ntg;
    const int seg = rest % p.nseg, yb = rest / p.nseg + p.yb0;
    const int x0 = seg * kBfSeg, y0 = yb * kBfRows;

    // (Round 6 measured a phase offset between the two work-groups of a CU - the one in the odd wave slot of the first
    // generation sleeping half a tile period, so that one wave of a SIMD multiplies while the other normalises: 2.03-2.05
    // against 1.95-1.96 ms at 4K x 32, one piece product, alternating on one box; removed.  The same start stagger had been
    // measured on ncc_mfma_kernel in rounds 3 and 4 with the same result.)
    v4f acc[MB][8];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[mb][c] = v4f{0.f, 0.f, 0.f, 0.f};

    const int row_bytes = p.lds_cols * 2;
    const int tile_rows_max = p.chunk_h + kBfRows - 1;
    uint8_t* thi = smem_bf;
    uint8_t* tlo = smem_bf + (size_t)tile_rows_max * row_bytes;
    float* s_mu = reinterpret_cast<float*>(smem_bf + 2 * (size_t)tile_rows_max * row_bytes);
    BfTemplConst* tcl = reinterpret_cast<BfTemplConst*>(smem_bf + 2 * (size_t)tile_rows_max * row_bytes + 16);
    const uint8_t* apack_g = apack + (long long)tg * MB * p.group_bytes + (size_t)lane * 16;
    unsigned long long* ext_slot = reinterpret_cast<unsigned long long*>(tcl + 32) + wave * 32;
    if (lane < 32) ext_slot[lane] = 0ull;       // wave-private; ordered before the epilogue by the staging barriers

    // per-template constants -> LDS (ordered before the epilogue by the staging barriers)
    if (threadIdx.x < 16 * MB) {
        const int li = tg * MB * 16 + threadIdx.x;
        if (li < p.n_list) {
            const int tglob = tlist[li];
            const TemplDev& T = td[tglob];
            BfTemplConst k;
#pragma unroll
            for (int cc = 0; cc < kMaxChans; ++cc) {
                k.mean[cc] = T.mean[cc];
                k.centre[cc] = T.centre[cc];
            }
            k.templ_norm = T.templ_norm;
            k.templ_sum2 = T.templ_sum2;
            k.t2c = T.centred_sum2 * 1.000001;        // (a float64 variance times the area: its own rounding covered)
            {
                const bool nrm = p.method == MTM_TM_SQDIFF_NORMED || p.method == MTM_TM_CCORR_NORMED || p.method == MTM_TM_CCOEFF_NORMED;
                const double esc = (p.method == MTM_TM_SQDIFF_NORMED || p.method == MTM_TM_SQDIFF) ? 2.0 : 1.0;
                k.bfac = nrm ? (T.templ_norm > 0.0 ? esc * sqrt(k.t2c) / T.templ_norm * 1.000001 : 0.0)
                             : esc * sqrt(k.t2c) * 1.000001;            // raw sums (rig == 2): the bound in the sum's own units
            }
            k.map_off = T.map_off;
            k.map_pitch = T.map_pitch;
            k.all_ones = T.all_ones;
            k.tglob = tglob;
            k.pad_ = 0;
            tcl[threadIdx.x] = k;
        }
    }

    for (int c = 0; c < p.chans; ++c) {
        const float* plane = p.img + c * p.plane;
        // The constant subtracted from this channel's pixels: the mean of an 8 x 8 sample grid over the work item's whole
        // image patch.  ANY constant is exact - but it must be ONE constant for all rows of the template (the taps
        // of a channel sum to zero only over the whole template, not over a chunk of its rows); the mean is the
        // accurate one.
        __syncthreads();                        // previous channel's tile fully consumed (s_mu is rewritten)
        if (wave == 0) {
            // (samples are clamped into the image: the zero padding beyond it would drag the constant away from the
            // pixels that matter, and the pieces would spend their bits on the difference)
            const int sr = min(y0 + (lane >> 3) * (p.h + kBfRows - 2) / 7, p.rows - 1);
            const int sc = min(x0 + (lane & 7) * (p.lds_cols - 1) / 7, p.cols - 1);
            float v = plane[(size_t)sr * p.pitch + sc];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0) {
                s_mu[c] = v * (1.0f / 64.0f);      // (one slot per channel: the epilogue's error bound reads them back)
                if (p.mu_out != nullptr && tg == 0 && c == 0) p.mu_out[(size_t)yb * p.nseg + seg] = v * (1.0f / 64.0f);
            }
        }
        __syncthreads();
        const float mu = s_mu[c];
        for (int cy0 = 0; cy0 < p.h; cy0 += p.chunk_h) {
            const int ch = min(p.chunk_h, p.h - cy0);
            const int trows = ch + kBfRows - 1;
            const float* grow = plane + (size_t)(y0 + cy0) * p.pitch + x0;
            __syncthreads();                    // previous tile fully consumed
            // stage: two adjacent pixels per thread and step -> one dword per piece plane; (row, pair) advance
            // without divisions, eight requests in flight per thread
            {
                const int pairs = p.lds_cols >> 1;
                const int rstep = 256 / pairs, cstep = 256 - rstep * pairs;
                int r = threadIdx.x / pairs, cp = threadIdx.x - r * pairs;
                while (r < trows) {
                    constexpr int kInFlight = 8;        // (round 6: 8 instead of 4 requests per thread - the tile is a chain of load latencies)
                    float2 v[kInFlight];
                    int rr[kInFlight], cc[kInFlight];
#pragma unroll
                    for (int u = 0; u < kInFlight; ++u) {
                        rr[u] = r;
                        cc[u] = cp;
                        if (r < trows) v[u] = *reinterpret_cast<const float2*>(grow + (size_t)r * p.pitch + 2 * cp);
                        r += rstep;
                        cp += cstep;
                        if (cp >= pairs) {
                            cp -= pairs;
                            ++r;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kInFlight; ++u) {
                        if (rr[u] >= trows) break;
                        const float a = v[u].x - mu, b = v[u].y - mu;
                        const uint32_t a0 = bf16_rne(a), b0 = bf16_rne(b);
                        *reinterpret_cast<uint32_t*>(thi + (size_t)rr[u] * row_bytes + 4 * cc[u]) = a0 | (b0 << 16);
                        if constexpr (NP == 3) {
                            const uint32_t a1 = bf16_rne(a - bf16_to_float(a0)), b1 = bf16_rne(b - bf16_to_float(b0));
                            *reinterpret_cast<uint32_t*>(tlo + (size_t)rr[u] * row_bytes + 4 * cc[u]) = a1 | (b1 << 16);
                        }
                    }
                }
            }
            __syncthreads();
            // K loop: template rows of this chunk x 32-tap blocks, software pipelined with two register sets: the
            // operands of the next step (4 LDS chunks + 2 MB packed template rows) are requested before the 24 MB
            // MFMAs of the current one issue; the loop body is branch-free and requests up to two steps past the
            // chunk (pack arena slack / inside the LDS allocation; never used).
            const uint8_t* aptr = apack_g + ((size_t)(c * p.h + cy0) * p.nkb) * 1024;
            const int nsteps = ch * p.nkb;
            const int row_adv = row_bytes - (p.nkb - 1) * 64;
            int loff = wave * row_bytes + (j + q) * 16;
            int kb_i = 0;
            v4i_b hA0, hA1, lA0, lA1, hB0, hB1, lB0, lB1, aA0[MB], aA1[MB], aB0[MB], aB1[MB];
            if constexpr (NP != 3) {            // (never loaded, never multiplied: defined values for the compiler's sake)
                lA0 = lA1 = lB0 = lB1 = v4i_b{0, 0, 0, 0};
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) aA1[mb] = aB1[mb] = v4i_b{0, 0, 0, 0};
            }
#define MTM_BF_LOAD(H0, H1, L0, L1, A0, A1)                                                    \
            H0 = *reinterpret_cast<const v4i_b*>(thi + loff);                                  \
            H1 = *reinterpret_cast<const v4i_b*>(thi + loff + 16);                             \
            if constexpr (NP == 3) {                                                           \
                L0 = *reinterpret_cast<const v4i_b*>(tlo + loff);                              \
                L1 = *reinterpret_cast<const v4i_b*>(tlo + loff + 16);                         \
            }                                                                                  \
            _Pragma("unroll") for (int mb = 0; mb < MB; ++mb) {                                \
                A0[mb] = *reinterpret_cast<const v4i_b*>(aptr + mb * p.group_bytes);           \
                if constexpr (NP == 3)                                                         \
                    A1[mb] = *reinterpret_cast<const v4i_b*>(aptr + p.piece_bytes + mb * p.group_bytes); \
            }
#define MTM_BF_ADVANCE()                                        \
            {                                                   \
                aptr += 1024;                                   \
                const bool wrap_ = kb_i + 1 == p.nkb;           \
                loff += wrap_ ? row_adv : 64;                   \
                kb_i = wrap_ ? 0 : kb_i + 1;                    \
            }
            if constexpr (NP == 1) {
                // One piece product: a step is 8 MB MFMAs, ~270 cycles of the matrix pipe (twice that while the SIMD's other
                // wave is in its K loop too) - less than the latency of the packed template rows, which come from memory
                // (L2) every step.  With the one-step look-ahead of the three-product loop the waves sat in s_waitcnt for a
                // third of their lives (PMC, 4K x 32: SQ_WAIT_ANY 44 k of a wave's 134 k cycles, matrix pipe 49 % busy).
                // Here the two register sets hold the template rows of kStage = 4 steps each: a stage's rows are requested
                // while the previous stage multiplies, four steps ahead of their use; the image chunks from LDS stay one step
                // ahead.  Requests beyond the chunk's last step repeat that step's address (nothing is read past the packs),
                // steps beyond it are skipped (wave-uniform).  (A four-set rotation with single steps was tried first: the
                // compiler re-rotated it into a one-step look-ahead with register copies.)
                constexpr int kStage = 4;
                const uint8_t* abase = aptr;
                v4i_b sA[kStage][MB], sB[kStage][MB], hq0a, hq0b, hq1a, hq1b;
                const v4i_b zero4 = v4i_b{0, 0, 0, 0};
#define MTM_BF1_LDSTAGE(S, STEP0)                                                                           \
                _Pragma("unroll") for (int i_ = 0; i_ < kStage; ++i_) {                                     \
                    const uint8_t* ap_ = abase + (size_t)min((STEP0) + i_, nsteps - 1) * 1024;              \
                    _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                       \
                        S[i_][mb] = *reinterpret_cast<const v4i_b*>(ap_ + mb * p.group_bytes);              \
                }
#define MTM_BF1_LDH(HA, HB)                                                                                 \
                HA = *reinterpret_cast<const v4i_b*>(thi + loff);                                           \
                HB = *reinterpret_cast<const v4i_b*>(thi + loff + 16);
#define MTM_BF1_ADV()                                                                                       \
                {                                                                                           \
                    const bool wrap_ = kb_i + 1 == p.nkb;                                                   \
                    loff += wrap_ ? row_adv : 64;                                                           \
                    kb_i = wrap_ ? 0 : kb_i + 1;                                                            \
                }
                // the four steps of a stage: image chunks alternate between the two (hq*) sets, the next step's are
                // requested before this step's MFMAs (one step past the chunk at its end: inside the LDS allocation)
#define MTM_BF1_RUNSTAGE(S, STEP0)                                                                          \
                {                                                                                           \
                    const int left_ = nsteps - (STEP0);                                                     \
                    MTM_BF1_ADV()                                                                           \
                    MTM_BF1_LDH(hq1a, hq1b)                                                                 \
                    __builtin_amdgcn_sched_barrier(0);                                                      \
                    bf_step<MB, 1>(acc, hq0a, hq0b, zero4, zero4, S[0], S[0]);                              \
                    __builtin_amdgcn_sched_barrier(0);                                                      \
                    if (left_ > 1) {                                                                        \
                        MTM_BF1_ADV()                                                                       \
                        MTM_BF1_LDH(hq0a, hq0b)                                                             \
                        __builtin_amdgcn_sched_barrier(0);                                                  \
                        bf_step<MB, 1>(acc, hq1a, hq1b, zero4, zero4, S[1], S[1]);                          \
                        __builtin_amdgcn_sched_barrier(0);                                                  \
                    }                                                                                       \
                    if (left_ > 2) {                                                                        \
                        MTM_BF1_ADV()                                                                       \
                        MTM_BF1_LDH(hq1a, hq1b)                                                             \
                        __builtin_amdgcn_sched_barrier(0);                                                  \
                        bf_step<MB, 1>(acc, hq0a, hq0b, zero4, zero4, S[2], S[2]);                          \
                        __builtin_amdgcn_sched_barrier(0);                                                  \
                    }                                                                                       \
                    if (left_ > 3) {                                                                        \
                        MTM_BF1_ADV()                                                                       \
                        MTM_BF1_LDH(hq0a, hq0b)                                                             \
                        __builtin_amdgcn_sched_barrier(0);                                                  \
                        bf_step<MB, 1>(acc, hq1a, hq1b, zero4, zero4, S[3], S[3]);                          \
                        __builtin_amdgcn_sched_barrier(0);                                                  \
                    }                                                                                       \
                }
                MTM_BF1_LDSTAGE(sA, 0)
                MTM_BF1_LDH(hq0a, hq0b)
                for (int ks = 0; ks < nsteps; ks += 2 * kStage) {
                    MTM_BF1_LDSTAGE(sB, ks + kStage)
                    __builtin_amdgcn_sched_barrier(0);
                    MTM_BF1_RUNSTAGE(sA, ks)
                    if (ks + kStage < nsteps) {
                        MTM_BF1_LDSTAGE(sA, ks + 2 * kStage)
                        __builtin_amdgcn_sched_barrier(0);
                        MTM_BF1_RUNSTAGE(sB, ks + kStage)
                    }
                }
#undef MTM_BF1_LDSTAGE
#undef MTM_BF1_LDH
#undef MTM_BF1_ADV
#undef MTM_BF1_RUNSTAGE
            } else {
            MTM_BF_LOAD(hA0, hA1, lA0, lA1, aA0, aA1)
            int ks = 0;
            for (; ks + 2 <= nsteps; ks += 2) {
                MTM_BF_ADVANCE()
                MTM_BF_LOAD(hB0, hB1, lB0, lB1, aB0, aB1)
                __builtin_amdgcn_sched_barrier(0);
                bf_step<MB, NP>(acc, hA0, hA1, lA0, lA1, aA0, aA1);
                __builtin_amdgcn_sched_barrier(0);
                MTM_BF_ADVANCE()
                MTM_BF_LOAD(hA0, hA1, lA0, lA1, aA0, aA1)
                __builtin_amdgcn_sched_barrier(0);
                bf_step<MB, NP>(acc, hB0, hB1, lB0, lB1, aB0, aB1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ks < nsteps) bf_step<MB, NP>(acc, hA0, hA1, lA0, lA1, aA0, aA1);
            }   // NP == 3
#undef MTM_BF_LOAD
#undef MTM_BF_ADVANCE
        }
    }

    // ---- epilogue: lane (j, q) holds pixels x0 + 8 j + 0..7 of row y for templates 16 mb + 4 q + e.  Statistics of
    // four pixels at a time in registers (they do not depend on the template), constants from LDS.
    const int y = y0 + wave;
    const int xq = x0 + 8 * j;
    const bool lane_on = y < p.oh && xq < p.ow;
    const int method = p.method;
    const bool normed = method == MTM_TM_SQDIFF_NORMED || method == MTM_TM_CCORR_NORMED || method == MTM_TM_CCOEFF_NORMED;
    // listing decisions by the per-output error bound (Bf16Params::rig): 1 = normalised methods (bound of the ratio),
    // 2 = raw sums with a threshold (bound of the sum itself: eps sqrt(sum (I - mu)^2 sum (T - mean)^2), twice for TM_SQDIFF)
    const bool rig_n = p.rig == 1 && normed, rig_r = p.rig == 2 && !normed;
    const bool rig = rig_n || rig_r;
    const bool need_sum2 = method == MTM_TM_SQDIFF || method == MTM_TM_SQDIFF_NORMED || (p.ext_on && p.ext_raw) || rig;
    // the quotient-free listing screen of the hits-only route (epilogue_half)
    const bool fast_screen = rig_n && p.cand_on && !p.ext_on && p.hits_only && p.rig_flag == nullptr;
    const double fs_thr = (double)p.rig_thr;
    const double fs_d = p.cand_min ? 2.4e-6 + 1e-9 : 6e-7 * fmax(1.0, fabs(fs_thr)) + 1e-9;
    // Round 6: the same screen for the refined global extremum of the maxima methods (N_object == 1: cv2.minMaxLoc).  An
    // output matters only if its UPPER bound reaches its template's best LOWER bound; what other work items have published
    // so far is read here, once per wave (the final best can only be higher), and an output whose bound stays below it by the
    // screen's allowance is neither a key nor a listing - its accumulator registers are set to NaN, which the listing pass
    // (list_half) never lists.  Templates without a published best, or with a negative one, are not screened.
    const bool ext_fast = rig_n && p.ext_on && p.ext_raw && !p.cand_min && p.ext_margin > 0.0f;
    double ext_thr[MB][4];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ext_thr[mb][e] = -1.0;
            const int lt = mb * 16 + 4 * q + e, li = tg * MB * 16 + lt;
            if (ext_fast && li < p.n_list) {
                const unsigned long long gk = __hip_atomic_load(&p.ext_best[2 * tlist[li] + p.cand_min], __ATOMIC_RELAXED,
                                                                __HIP_MEMORY_SCOPE_AGENT);
                if (gk) ext_thr[mb][e] = (double)mf_order_float((uint32_t)(gk >> 32));      // (maxima: the key holds the quality itself)
            }
        }
    // (the two halves of a lane's eight pixels as a generic lambda over a compile-time constant: every accumulator index
    // below must be one, or the accumulators leave the register file)
    auto epilogue_half = [&](auto half_c) {
        constexpr int half = decltype(half_c)::value;
        double ts[4][kMaxChans], s2[4], sq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int x = min(xq + 4 * half + i, p.ow - 1);
            const size_t sidx = (size_t)y * st.pitch + x;
#pragma unroll
            for (int cc = 0; cc < kMaxChans; ++cc) ts[i][cc] = cc < p.chans ? st.t[cc][sidx] : 0.0;
            s2[i] = need_sum2 ? st.sum2[sidx] : 0.0;
            sq[i] = normed ? st.sq[sidx] : 0.0;
        }
        // rig: the pixel's share of the bound, rig_eps * sqrt(sum (I - mu)^2) / sq (0 where the window is flat: the
        // normalisation rules return a constant there)
        double bp[4] = {0.0, 0.0, 0.0, 0.0};
        double fs_E[4] = {0.0, 0.0, 0.0, 0.0}, fs_P[4] = {0.0, 0.0, 0.0, 0.0};      // the quotient-free listing screen's per-pixel terms
        bool fs_wide = false;
        if (rig) {
            const double area = (double)p.h * (double)p.w;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double s2c = s2[i], mu2 = 0.0;
#pragma unroll
                for (int cc = 0; cc < kMaxChans; ++cc)
                    if (cc < p.chans) {
                        const double m = (double)s_mu[cc];
                        s2c += m * (area * m - 2.0 * ts[i][cc]);
                        mu2 += m * m;
                    }
                // (+ the cancellation in s2c itself: three terms of the size of s2 and area mu^2, summed over the channels)
                s2c = fmax(s2c, 0.0) * 1.000001 + 1e-12 * (fabs(s2[i]) + area * mu2);
                const double ee = (double)p.rig_eps * sqrt(s2c);
                bp[i] = rig_r ? ee : sq[i] > 0.0 ? ee / sq[i] : 0.0;
                if (fast_screen || ext_fast) {
                    fs_E[i] = ee * 1.0000001;            // (never below bp sq as the exact test multiplies it out)
                    fs_P[i] = (p.cand_min ? fs_d - fs_thr : fs_thr - fs_d) * sq[i];
                    fs_wide = fs_wide || !(sq[i] > 0.0) || !(bp[i] <= 0.2);
                }
            }
        }
        // one template of the lane (group mb, accumulator element e) - instantiated 4 MB times by bf_static_for: as a
        // `#pragma unroll` loop the body grew past the unroller's size limit in round 6, the loop stayed a loop and the
        // accumulators it indexes went to scratch memory
        auto one_template = [&](auto idx_c) {
            constexpr int mb = decltype(idx_c)::value / 4, e = decltype(idx_c)::value % 4;
            {
                const int lt = mb * 16 + 4 * q + e, li = tg * MB * 16 + lt;
                if (li >= p.n_list || (p.only_li >= 0 && li != p.only_li)) return;
                const BfTemplConst& T = tcl[lt];
                bool skip = false;
                if (ext_fast && ext_thr[mb][e] >= 0.0 && !fs_wide && T.all_ones == 0 && T.templ_norm > 0.0) {
                    // (the listing screen below with this template's published best as the threshold: an output with
                    // r + B <= best - d has an upper bound below the best lower bound - by d, twice the exact path's rounding
                    // allowance at |r| <= 1; r < -1: r + B + 3e-7 |r| < 0 <= best.  The exact extremum and every tie with it have
                    // upper bounds >= their scores >= every lower bound: never screened.)
                    const double tn = T.templ_norm, G = T.bfac * tn, tnthr = tn * (ext_thr[mb][e] - (6e-7 + 1e-9));
                    const double cm0 = method == MTM_TM_CCOEFF_NORMED ? T.centre[0] - T.mean[0] : T.centre[0];
                    bool anyp = false;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        double num = (double)acc[mb][4 * half + i][e];
                        if (p.chans == 1) {
                            num = fma(cm0, ts[i][0], num);
                        } else {
                            for (int cc = 0; cc < p.chans; ++cc)
                                num = fma(method == MTM_TM_CCOEFF_NORMED ? T.centre[cc] - T.mean[cc] : T.centre[cc], ts[i][cc], num);
                        }
                        anyp = anyp || fma(G, fs_E[i], num) > tnthr * sq[i];
                    }
                    if (!anyp) {
                        skip = true;
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[mb][4 * half + i][e] = __builtin_nanf("");
                    }
                }
                if (fast_screen) {
                    // Hits-only listing by the bound (normalised methods): most outputs are nowhere near the threshold, and
                    // the listing test  r + B > thr  (r = num / t, t = sq tn, B = bp bfac the output's bound) is, multiplied
                    // by t > 0,   num + G E > tn P   with G = bfac tn per template and E = eps sqrt(sum (I - mu)^2),
                    // P = (thr - d) sq per pixel (fs_E, fs_P above; minima: num - G E < tn P, P = (-thr + d) sq): a
                    // conversion, two multiply-adds, a multiplication and a comparison per output instead of a division
                    // sequence, the saturation analysis and the bound arithmetic - with one piece product in the K loop the
                    // exact form of this epilogue took as many cycles as the K loop itself (PMC: 2.1 G VALU-active against
                    // 2.1 G MFMA-busy cycles per launch at 4K x 32).  d is twice the rounding allowance 3e-7 max(1, |r|) of
                    // the exact test below wherever |r| can matter, which also dwarfs the float64 roundings of this
                    // restatement: a pixel whose bound is wide (bp > 0.2: a low-contrast window beside a step - also every
                    // flat window and anything not finite) sends all its outputs to the exact test (fs_wide), so here
                    // B <= 0.41 and a failing output has r <= thr - d, |r| <= max(1, |thr|) or r < -1 where r + B + 3e-7 |r|
                    // < 0 <= thr (negative thresholds list everything: list_all); minima: r >= 0, and r (1 - 3e-7) >=
                    // B - thr by the same d.  Constant templates take the exact test as well.
                    const double tn = T.templ_norm, G = T.bfac * tn;
                    bool anyp = p.list_all != 0 || T.all_ones != 0 || !(tn > 0.0) || fs_wide;
                    double cm[kMaxChans];
#pragma unroll
                    for (int cc = 0; cc < kMaxChans; ++cc)
                        cm[cc] = method == MTM_TM_CCOEFF_NORMED ? T.centre[cc] - T.mean[cc] : T.centre[cc];
                    const double ts2 = T.templ_sum2;
                    // (the bound's sign folded into G: one fma and one comparison direction per output; the channel loop behind a
                    // wave-uniform branch - as selects it was 44 v_cndmask per template)
                    const double Gs = p.cand_min ? -G : G;
                    double numv[4];
                    if (p.chans == 1) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) numv[i] = fma(cm[0], ts[i][0], (double)acc[mb][4 * half + i][e]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            double num = (double)acc[mb][4 * half + i][e];
                            for (int cc = 0; cc < p.chans; ++cc) num = fma(cm[cc], ts[i][cc], num);
                            numv[i] = num;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        double num = numv[i];
                        if (method == MTM_TM_SQDIFF_NORMED) num = fmax(fma(-2.0, num, s2[i] + ts2), 0.0);
                        const double lhs = fma(Gs, fs_E[i], num), rhs = tn * fs_P[i];
                        anyp = anyp || (p.cand_min ? lhs < rhs : lhs > rhs);
                    }
                    skip = !anyp;
                }
                if (!skip) {
                float out[4];
                double qv[4], Mv[4];        // rig: quality before the saturation rules, bound of its error
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    double corr = (double)acc[mb][4 * half + i][e];
#pragma unroll
                    for (int cc = 0; cc < kMaxChans; ++cc)
                        if (cc < p.chans) corr += T.centre[cc] * ts[i][cc];
                    double r;
                    bool ex;
                    out[i] = bf_finish_r(method, corr, ts[i], s2[i], sq[i], T, p.chans, &r, &ex);
                    if (ex) r = (double)out[i];
                    qv[i] = p.cand_min ? -r : r;
                    // (+ the rounding of the exact score to float32 and of this arithmetic)
                    Mv[i] = ex ? 0.0 : bp[i] * T.bfac + 3e-7 * fmax(1.0, fabs(r));
                }
                const int xb = xq + 4 * half;
                if (rig_n && p.rig_flag != nullptr) {
                    // map mode: the scan that follows works with tolerances of rig_cap - an output that could pass the
                    // threshold with a larger bound than that takes the call to the float64 kernel
                    bool wide = false;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        wide = wide || (xb + i < p.ow && Mv[i] > (double)p.rig_cap && (p.list_all || qv[i] + Mv[i] > (double)p.rig_thr));
                    if (wide) *p.rig_flag = 1u;
                }
                float key_v[4] = {out[0], out[1], out[2], out[3]};      // what the published key is built from
                if (p.ext_on && p.ext_raw && rig_n) {
                    // normalised methods, refined: bounds of the exact QUALITY instead of scores.  Maxima: the exact score is
                    // sat(r) <= min(r, 1), and 0 where |r| >= 1.125; minima (TM_SQDIFF_NORMED): -clamp(r, 0, 1), monotone.
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        double up = qv[i] + Mv[i], lo = qv[i] - Mv[i];
                        if (!p.cand_min) {
                            lo = fmin(lo, 1.0);
                            if (up >= 1.125) lo = fmin(lo, 0.0);
                            up = fmin(up, 1.0);
                        } else {
                            up = -fmin(fmax(-up, 0.0), 1.0);
                            lo = -fmin(fmax(-lo, 0.0), 1.0);
                        }
                        acc[mb][4 * half + i][e] = (float)up + fabsf((float)up) * 1.2e-7f;      // (never rounded below the bound)
                        const float lq = (float)lo - fabsf((float)lo) * 1.2e-7f;
                        key_v[i] = p.cand_min ? -lq : lq;
                    }
                } else if (p.ext_on && p.ext_raw) {
                    // raw-sum methods, refined: bounds instead of scores (see Bf16Params::ext_raw).  The accumulator
                    // registers take the upper bound of the quality, the key the lower bound (as a score).
                    const double escale = (method == MTM_TM_SQDIFF ? 2.0 : 1.0) * (double)p.ext_eps;
                    const double area = (double)p.h * (double)p.w;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        double s2c = s2[i];
#pragma unroll
                        for (int cc = 0; cc < kMaxChans; ++cc)
                            if (cc < p.chans) {
                                const double m = (double)s_mu[cc];
                                s2c += m * (area * m - 2.0 * ts[i][cc]);
                            }
                        const double q = p.cand_min ? -(double)out[i] : (double)out[i];
                        // + the rounding of the score to float32 and of the bounds below
                        const double E = escale * sqrt(fmax(s2c, 0.0) * T.t2c) * 1.000001 + fabs(q) * 2.4e-7 + 1e-30;
                        acc[mb][4 * half + i][e] = (float)(q + E);
                        const float lq = (float)(q - E);
                        key_v[i] = p.cand_min ? -lq : lq;
                    }
                } else if (p.ext_on && p.ext_margin > 0.0f) {       // refined extremum mode: the outputs are looked at again below
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[mb][4 * half + i][e] = out[i];
                }
                if (p.ext_on) {
                    // cv2.minMaxLoc: the first index wins ties, NaN never wins
                    unsigned long long bestk = 0ull;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float v = key_v[i];
                        if (xb + i < p.ow && v == v) {
                            const uint32_t o = mf_float_order(v);
                            const unsigned long long key = ((unsigned long long)(p.cand_min ? ~o : o) << 32) |
                                                           (unsigned long long)(0xFFFFFFFFu - (uint32_t)(y * p.ow + xb + i));
                            bestk = key > bestk ? key : bestk;
                        }
                    }
                    if (bestk) atomicMax(&ext_slot[lt], bestk);
                } else if (p.cand_on && rig) {
                    // refined route: everything whose exact score could pass the threshold (rig_thr: the exact one)
                    bool pass[4], any = false;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        pass[i] = xb + i < p.ow && (p.list_all != 0 || qv[i] + Mv[i] > (double)p.rig_thr);
                        any = any || pass[i];
                    }
                    if (any) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            mtm_hit hrec;
                            hrec.templ_idx = T.tglob;
                            hrec.x = xb + i;
                            hrec.y = y;
                            hrec.w = p.w;
                            hrec.h = p.h;
                            hrec.score = out[i];
                            cand_append(pass[i], p.cand_counter, p.cand_cap, p.cand_hits, hrec);
                        }
                    }
                } else if (p.cand_on) {
                    // any-of-4 first (rare on sparse maps); the append itself takes one atomic per wave
                    const float hi = fmaxf(fmaxf(out[0], out[1]), fmaxf(out[2], out[3]));
                    const float lo = fminf(fminf(out[0], out[1]), fminf(out[2], out[3]));
                    if ((p.cand_min ? -lo : hi) > p.cand_thr) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float v = p.cand_min ? -out[i] : out[i];
                            mtm_hit hrec;
                            hrec.templ_idx = T.tglob;
                            hrec.x = xb + i;
                            hrec.y = y;
                            hrec.w = p.w;
                            hrec.h = p.h;
                            hrec.score = out[i];
                            cand_append(xb + i < p.ow && v > p.cand_thr, p.cand_counter, p.cand_cap, p.cand_hits, hrec);
                        }
                    }
                }
                if (!p.hits_only) {
                    float* orow = maps + T.map_off + (size_t)y * T.map_pitch + xb;
                    if (xb + 3 < p.ow) {
                        *reinterpret_cast<float4*>(orow) = make_float4(out[0], out[1], out[2], out[3]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (xb + i < p.ow) orow[i] = out[i];
                    }
                }
                }   // !skip
            }
        };
        bf_static_for<4 * MB>(one_template);
    };
    if (lane_on) {
        epilogue_half(std::integral_constant<int, 0>{});
        if (xq + 4 < p.ow) epilogue_half(std::integral_constant<int, 1>{});
    }
    if (p.ext_on) {                 // one global atomic per template this wave improved
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const bool refined = p.ext_margin > 0.0f;
        if (lane < 16 * MB) {
            const unsigned long long key = ext_slot[lane];
            const int li = tg * MB * 16 + lane;
            unsigned long long g = 0ull;
            if (li < p.n_list) {
                unsigned long long* addr = &p.ext_best[2 * tlist[li] + p.cand_min];
                if (key) {
                    const unsigned long long old = atomicMax(addr, key);
                    g = old > key ? old : key;
                } else if (refined) {
                    g = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (refined) ext_slot[lane] = g;        // the best of this template published so far, this wave's included
        }
        if (refined) {
            // Refined extremum mode (mtm_refine.hip.h): the scores above are within ~1e-5 of the exact ones, so the exact
            // extremum of a template is among the outputs within the margin of the best approximate one.  Every output of
            // this wave within the margin of the best published so far is appended to the candidate list; the exact
            // extremum is then taken over the re-scored list.  (The list stays short: a wave lists something only while
            // its outputs are within the margin of everything that finished before it.)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            auto list_half = [&](auto half_c) {
                    constexpr int half = decltype(half_c)::value;
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int lt = mb * 16 + 4 * q + e, li = tg * MB * 16 + lt;
                            if (li >= p.n_list) continue;
                            const unsigned long long gk = ext_slot[lt];
                            float lo = -INFINITY;
                            if (gk) {
                                const uint32_t hiw = (uint32_t)(gk >> 32);
                                const float sc = mf_order_float(p.cand_min ? ~hiw : hiw);
                                const float ql = p.cand_min ? -sc : sc;
                                lo = p.ext_raw ? nextafterf(ql, -INFINITY) : ql - p.ext_margin * fmaxf(1.0f, fabsf(ql));
                            }
                            const int xb = xq + 4 * half;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float v = acc[mb][4 * half + i][e];
                                // (raw-sum methods: the register holds the upper bound of the QUALITY already)
                                const float ql = p.ext_raw ? v : (p.cand_min ? -v : v);
                                mtm_hit hrec;
                                hrec.templ_idx = tcl[lt].tglob;
                                hrec.x = xb + i;
                                hrec.y = y;
                                hrec.w = p.w;
                                hrec.h = p.h;
                                hrec.score = v;
                                cand_append(xb + i < p.ow && v == v && ql > lo, p.cand_counter, p.cand_cap, p.cand_hits, hrec);
                            }
                        }
                    }
            };
            if (lane_on) {
                list_half(std::integral_constant<int, 0>{});
                if (xq + 4 < p.ow) list_half(std::integral_constant<int, 1>{});
            }
        }
    }
}

}  // namespace mtm
