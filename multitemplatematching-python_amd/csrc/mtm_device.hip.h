// Device kernels of libmtm_hip.so (gfx950 / CDNA4 only).  Included once, by mtm_hip.hip.
//
// Score-map arithmetic: the sliding dot products are exact (uint32/uint64 integers for uint8
// pixels, float64 FMA chains for float32 pixels); the normalisation epilogue is evaluated in
// float64 in the operation order of OpenCV's common_matchTemplate (the arithmetic behind the
// cv2.matchTemplate call at reference MTM/__init__.py:92) as restated in oracle/mtm_oracle.py, and
// the result is stored as float32.  This TU is compiled with -ffp-contract=off: every FMA below is
// explicit.
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>

#include "mtm_kernels.h"
#include "../../include/mtm_hip.h"
#include "mtm_device_util.hip.h"

namespace mtm {

// ---------------------------------------------------------------------------------------------
// image layout conversion: interleaved rows -> planar, padded, (u8 +) f32
// ---------------------------------------------------------------------------------------------
// u8b = the same planes with every byte ^ 0x80 (int8 view, value - 128): operand of the MFMA kernel,
// which stages its tiles by LDS-DMA and therefore cannot convert on the way.
__global__ void planarize_u8_kernel(const uint8_t* __restrict__ raw, int rows, int cols, int chans,
                                    uint8_t* __restrict__ u8, uint8_t* __restrict__ u8b, int u8_pitch,
                                    long long u8_plane, float* __restrict__ f32, int f32_pitch,
                                    long long f32_plane, int x_begin) {
    const int x = x_begin + blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= cols || y >= rows) return;
    const uint8_t* p = raw + ((size_t)y * cols + x) * chans;
    for (int c = 0; c < chans; ++c) {
        const uint8_t v = p[c];
        u8[c * u8_plane + (size_t)y * u8_pitch + x] = v;
        u8b[c * u8_plane + (size_t)y * u8_pitch + x] = v ^ 0x80;
        if (f32 != nullptr) f32[c * f32_plane + (size_t)y * f32_pitch + x] = (float)v;
    }
}

// Single-channel fast path: 16 pixels per thread (one 16-byte load, two 16-byte and four 16-byte
// stores).  cols16 = cols / 16 full groups; the tail columns go through planarize_u8_kernel.
__global__ __launch_bounds__(256) void planarize_u8_c1_kernel(const uint8_t* __restrict__ raw, int rows, int cols,
                                                              int cols16, uint8_t* __restrict__ u8,
                                                              uint8_t* __restrict__ u8b, int u8_pitch,
                                                              float* __restrict__ f32, int f32_pitch) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (g >= cols16 || y >= rows) return;
    const uint8_t* src = raw + (size_t)y * cols + 16 * (size_t)g;
    uint32_t w[4];
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const uint4 v = *reinterpret_cast<const uint4*>(src);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            w[k] = (uint32_t)src[4 * k] | ((uint32_t)src[4 * k + 1] << 8) | ((uint32_t)src[4 * k + 2] << 16) |
                   ((uint32_t)src[4 * k + 3] << 24);
    }
    const size_t o = (size_t)y * u8_pitch + 16 * (size_t)g;          // pitch is a multiple of 64: 16-byte aligned
    *reinterpret_cast<uint4*>(u8 + o) = make_uint4(w[0], w[1], w[2], w[3]);
    *reinterpret_cast<uint4*>(u8b + o) = make_uint4(w[0] ^ 0x80808080u, w[1] ^ 0x80808080u, w[2] ^ 0x80808080u,
                                                    w[3] ^ 0x80808080u);
    if (f32 == nullptr) return;         // banded uploads: no consumer of the float32 plane in that call (u8_to_f32_kernel later)
    float* fo = f32 + (size_t)y * f32_pitch + 16 * (size_t)g;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        *reinterpret_cast<float4*>(fo + 4 * k) = make_float4((float)(w[k] & 255u), (float)((w[k] >> 8) & 255u),
                                                             (float)((w[k] >> 16) & 255u), (float)(w[k] >> 24));
}

// The float32 plane of a uint8 image from its padded uint8 plane (4 pixels per thread), when a later call needs it
// (float64 / naive kernels, the generic statistics) after an upload that skipped it.
__global__ __launch_bounds__(256) void u8_to_f32_kernel(const uint8_t* __restrict__ u8, float* __restrict__ f32, size_t n4) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n4) return;
    const uint32_t w = reinterpret_cast<const uint32_t*>(u8)[g];
    reinterpret_cast<float4*>(f32)[g] = make_float4((float)(w & 255u), (float)((w >> 8) & 255u), (float)((w >> 16) & 255u),
                                                    (float)(w >> 24));
}

// Integer-factor area downscale fused with the layout conversion (reference use:
// cv2.resize(image, smallDim, interpolation=cv2.INTER_AREA) before matching,
// tutorials/Tutorial3-SpeedingUp.ipynb:395).  Output pixel = mean of an f x f block; uint8 rounding as
// OpenCV's integer-factor INTER_AREA path: f == 2 -> (sum + 2) >> 2, otherwise
// rint((float)sum * (1.f / (f*f))) (float32 product, ties to even).
__global__ void planarize_u8_down_kernel(const uint8_t* __restrict__ raw, int src_cols, int chans, int f,
                                         int rows, int cols, uint8_t* __restrict__ u8,
                                         uint8_t* __restrict__ u8b, int u8_pitch, long long u8_plane,
                                         float* __restrict__ f32, int f32_pitch, long long f32_plane) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= cols || y >= rows) return;
    const float scale = 1.0f / (float)(f * f);
    for (int c = 0; c < chans; ++c) {
        unsigned sum = 0;
        for (int dy = 0; dy < f; ++dy) {
            const uint8_t* p = raw + ((size_t)(y * f + dy) * src_cols + (size_t)x * f) * chans + c;
            for (int dx = 0; dx < f; ++dx) sum += p[(size_t)dx * chans];
        }
        const unsigned r = (f == 2) ? ((sum + 2u) >> 2) : (unsigned)rintf((float)sum * scale);
        const uint8_t v = (uint8_t)(r > 255u ? 255u : r);
        u8[c * u8_plane + (size_t)y * u8_pitch + x] = v;
        u8b[c * u8_plane + (size_t)y * u8_pitch + x] = v ^ 0x80;
        f32[c * f32_plane + (size_t)y * f32_pitch + x] = (float)v;
    }
}

// float32: block sum accumulated in float32 in row-major order, then * (1.f / (f*f)).
__global__ void planarize_f32_down_kernel(const float* __restrict__ raw, int src_cols, int chans, int f, int rows,
                                          int cols, float* __restrict__ f32, int f32_pitch, long long f32_plane) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= cols || y >= rows) return;
    const float scale = 1.0f / (float)(f * f);
    for (int c = 0; c < chans; ++c) {
        float sum = 0.0f;
        for (int dy = 0; dy < f; ++dy) {
            const float* p = raw + ((size_t)(y * f + dy) * src_cols + (size_t)x * f) * chans + c;
            for (int dx = 0; dx < f; ++dx) sum += p[(size_t)dx * chans];
        }
        f32[c * f32_plane + (size_t)y * f32_pitch + x] = sum * scale;
    }
}

// uint16 images (single channel): exact integer matching through the int8 matrix cores needs the two
// bytes of every pixel as separate int8 planes (hi ^ 0x80, lo ^ 0x80; see the kMfU16 pass of ncc_mfma_kernel).  Also
// writes the unbiased high-byte plane (window sums of the high bytes) and the float32 plane (window
// statistics, float64 fallback kernel).  f > 1: area-downscaled first, rounding as for uint8 with the
// uint16 saturation.  `hi_lo` may be null (multi-channel uint16 images only take the float64 kernel).
__global__ void planarize_u16_kernel(const uint16_t* __restrict__ raw, int src_cols, int chans, int f, int rows, int cols,
                                     uint8_t* __restrict__ hi, uint8_t* __restrict__ hib, uint8_t* __restrict__ lob,
                                     int u8_pitch, float* __restrict__ f32, int f32_pitch, long long f32_plane) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= cols || y >= rows) return;
    const float scale = 1.0f / (float)(f * f);
    for (int c = 0; c < chans; ++c) {
        unsigned v;
        if (f == 1) {
            v = raw[((size_t)y * src_cols + x) * chans + c];
        } else {
            unsigned sum = 0;
            for (int dy = 0; dy < f; ++dy) {
                const uint16_t* p = raw + ((size_t)(y * f + dy) * src_cols + (size_t)x * f) * chans + c;
                for (int dx = 0; dx < f; ++dx) sum += p[(size_t)dx * chans];
            }
            v = (f == 2) ? ((sum + 2u) >> 2) : (unsigned)rintf((float)sum * scale);
            v = v > 65535u ? 65535u : v;
        }
        f32[c * f32_plane + (size_t)y * f32_pitch + x] = (float)v;
        if (hi != nullptr && c == 0) {
            const size_t o = (size_t)y * u8_pitch + x;
            hi[o] = (uint8_t)(v >> 8);
            hib[o] = (uint8_t)((v >> 8) ^ 0x80u);
            lob[o] = (uint8_t)((v & 255u) ^ 0x80u);
        }
    }
}

__global__ void planarize_f32_kernel(const float* __restrict__ raw, int rows, int cols, int chans,
                                     float* __restrict__ f32, int f32_pitch, long long f32_plane) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= cols || y >= rows) return;
    const float* p = raw + ((size_t)y * cols + x) * chans;
    for (int c = 0; c < chans; ++c) f32[c * f32_plane + (size_t)y * f32_pitch + x] = p[c];
}

// ---------------------------------------------------------------------------------------------
// window statistics: separable box sums (exact integers for uint8 sources)
//   pass 1: hs1[c][y][x] = sum_{dx<w} I_c[y][x+dx],  hs2 likewise for I^2      (all image rows)
//   pass 2: vertical sums over h rows + the per-pixel, template-independent part of the
//           normalisation (window sums per channel, sum of squares, sqrt(diff2) with the
//           flat-window guard).
// ---------------------------------------------------------------------------------------------
constexpr int kHsumSeg = 16;

template <typename AccT>
__global__ void hsum_kernel(const float* __restrict__ img, int pitch, long long plane, int rows,
                            int w, int ow, AccT* __restrict__ hs1, AccT* __restrict__ hs2,
                            int hs_pitch, long long hs_plane) {
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * kHsumSeg;
    const int y = blockIdx.y;
    const int c = blockIdx.z;
    if (x0 >= ow || y >= rows) return;
    const float* row = img + c * plane + (size_t)y * pitch;
    AccT s1 = 0, s2 = 0;
    for (int dx = 0; dx < w; ++dx) {
        const AccT v = (AccT)row[x0 + dx];
        s1 += v;
        s2 += v * v;
    }
    AccT* o1 = hs1 + c * hs_plane + (size_t)y * hs_pitch;
    AccT* o2 = hs2 + c * hs_plane + (size_t)y * hs_pitch;
    for (int k = 0; k < kHsumSeg; ++k) {
        const int x = x0 + k;
        if (x >= ow) break;
        o1[x] = s1;
        o2[x] = s2;
        const AccT vn = (AccT)row[x + w];   // padded image: always readable
        const AccT vo = (AccT)row[x];
        s1 += vn - vo;                      // uint32: modular arithmetic, exact
        s2 += vn * vn - vo * vo;
    }
}

// Inclusive prefix sum over the 64 lanes of a wave with DPP (no LDS, no ds_bpermute): the classic
// row_shr 1/2/3, row_shr 4 (banks 1-3), row_shr 8 (banks 2-3), row_bcast 15 (rows 1,3), row_bcast 31
// (rows 2,3) sequence.  Lanes without a source keep 0 (the `old` operand).
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t x) {
    uint32_t s = x + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);   // row_shr:1
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);                // row_shr:2
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x113, 0xf, 0xf, false);                // row_shr:3
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x114, 0xf, 0xe, false);                // row_shr:4
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x118, 0xf, 0xc, false);                // row_shr:8
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x142, 0xa, 0xf, false);                // row_bcast:15
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x143, 0xc, 0xf, false);                // row_bcast:31
    return s;
}

// Horizontal box sums of one uint8 image row per work-group, through inclusive prefix sums held in
// LDS (uint32, exact): fully coalesced global reads and writes.  Element i of the row is owned by
// thread i % 256 in round i / 256; each round is a 256-wide block scan (wave shuffles + one LDS
// exchange) plus the carry of the previous rounds.  Used for uint8 images up to 8191 columns; the
// generic hsum_kernel above covers the rest.
__global__ __launch_bounds__(256) void hsum_u8_kernel(const uint8_t* __restrict__ img, int pitch, long long plane,
                                                      int cols, int w, int ow, uint32_t* __restrict__ hs1,
                                                      uint32_t* __restrict__ hs2, int hs_pitch, long long hs_plane) {
    extern __shared__ uint32_t pre[];            // P1[cols + 1], P2[cols + 1]
    __shared__ uint32_t wsum[2][4];
    uint32_t* P1 = pre;
    uint32_t* P2 = pre + cols + 1;
    const int y = blockIdx.x, c = blockIdx.y;
    const uint8_t* row = img + c * plane + (size_t)y * pitch;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        P1[0] = 0;
        P2[0] = 0;
    }
    uint32_t carry1 = 0, carry2 = 0;
    for (int base = 0; base < cols; base += 256) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < cols ? row[i] : 0u;
        const uint32_t a = wave_inclusive_scan_u32(v), b = wave_inclusive_scan_u32(v * v);
        if (lane == 63) {
            wsum[0][wave] = a;
            wsum[1][wave] = b;
        }
        __syncthreads();
        uint32_t oa = carry1, ob = carry2, ta = 0, tb = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < wave) {
                oa += wsum[0][k];
                ob += wsum[1][k];
            }
            ta += wsum[0][k];
            tb += wsum[1][k];
        }
        if (i < cols) {
            P1[i + 1] = a + oa;
            P2[i + 1] = b + ob;
        }
        carry1 += ta;
        carry2 += tb;
        __syncthreads();
    }
    uint32_t* o1 = hs1 + c * hs_plane + (size_t)y * hs_pitch;
    uint32_t* o2 = hs2 + c * hs_plane + (size_t)y * hs_pitch;
    for (int x = threadIdx.x; x < ow; x += 256) {
        o1[x] = P1[x + w] - P1[x];
        o2[x] = P2[x + w] - P2[x];
    }
}

// ---------------------------------------------------------------------------------------------
// Fused window statistics for single-channel uint8 images: one kernel, no intermediate planes.
// A work-group owns a strip of `owg` output columns (owg + w - 1 <= 1024 image columns) x
// kStatBand4 output rows; a thread owns FOUR adjacent image columns: one aligned dword load per
// image row, four 8-byte statistics per plane and output row (two 16-byte stores).  Column sums over
// the template height (C1 = sum I, C2 = sum I^2 per image column) are kept in registers and slid down
// one row at a time (two dword loads per output row, requested one iteration ahead); the window
// sums are differences of the exclusive prefix scan of the column sums over the strip, held in LDS
// (uint32, exact: differences are taken modulo 2^32 and the true window sums fit).  One block scan
// (thread-local prefix, DPP wave scan, one LDS exchange) and two barriers serve 4 x 256 columns.
// The launcher uses it for w <= 768 and w * h * 255^2 < 2^32; everything else takes hsum_* +
// vsum_stats_kernel.
// ---------------------------------------------------------------------------------------------
#ifndef MTM_STAT_BAND4
#define MTM_STAT_BAND4 8
#endif
constexpr int kStatBand4 = MTM_STAT_BAND4;    // stats_u8_kernel: output rows per work-group
constexpr int kStatStrip = 1024;               // image columns per work-group (4 per thread)

// output columns per work-group for a template width (multiple of 16: strips start dword-aligned, and the 16-pixel
// column blocks whose statistic ranges the kernel can write - `blk` - never straddle two strips)
inline int stats_u8_owg(int w) { return (kStatStrip + 1 - w) & ~15; }

__global__ __launch_bounds__(256) void stats_u8_kernel(const uint8_t* __restrict__ img, int pitch, int h, int w,
                                                       int oh, int ow, int owg, double inv_area, int num_type,
                                                       int want_sq, int want_t, int want_sum2, double* __restrict__ t0,
                                                       double* __restrict__ sum2, double* __restrict__ sq,
                                                       int st_pitch, double* __restrict__ rsq = nullptr,
                                                       int yb_off = 0, double* __restrict__ blk = nullptr,
                                                       int blk_pitch = 0) {
    __shared__ __attribute__((aligned(16))) uint32_t E1[kStatStrip + 4], E2[kStatStrip + 4];   // exclusive prefixes
    __shared__ uint32_t wsum[2][4];
    const int x0 = blockIdx.x * owg, y0 = ((int)blockIdx.y + yb_off) * kStatBand4;   // yb_off: banded launches
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int L = owg + w - 1;                       // image columns of this strip (<= kStatStrip)
    // the image is padded by kPadCols columns only: quads further right (beyond every valid window) read 0
    const bool ld = 4 * t < L && x0 + 4 * t + 3 < pitch;
    const uint8_t* base = img + (size_t)y0 * pitch + x0 + 4 * t;
    uint32_t c1[4] = {0, 0, 0, 0}, c2[4] = {0, 0, 0, 0};
    auto unpack = [](uint32_t v, uint32_t (&b)[4]) {
        b[0] = v & 255u;
        b[1] = (v >> 8) & 255u;
        b[2] = (v >> 16) & 255u;
        b[3] = v >> 24;
    };
    // 8 rows per batch: the loads of a batch are all in flight before the first add needs one
    for (int r0 = 0; r0 < h; r0 += 8) {
        uint32_t v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            v[i] = ld ? *reinterpret_cast<const uint32_t*>(base + (size_t)min(r0 + i, h - 1) * pitch) : 0u;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (r0 + i < h) {
                uint32_t b[4];
                unpack(v[i], b);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    c1[k] += b[k];
                    c2[k] += b[k] * b[k];
                }
            }
    }
    const int y1 = min(y0 + kStatBand4, oh);
    const int xg = x0 + 4 * t;                       // first of this thread's four output columns
    const bool out_on = 4 * t < owg && xg < st_pitch;   // st_pitch is a multiple of 4: xg + 3 < st_pitch too
    for (int y = y0; y < y1; ++y) {
        // request the two image rows of the slide at the end of this iteration now: their latency
        // hides behind the scan and the float64 statistics
        uint32_t vn = 0, vo = 0;
        if (y + 1 < y1 && ld) {
            vn = *reinterpret_cast<const uint32_t*>(base + (size_t)(y - y0 + h) * pitch);
            vo = *reinterpret_cast<const uint32_t*>(base + (size_t)(y - y0) * pitch);
        }
        // block-wide exclusive scan of the column sums (thread-local prefix, wave scan, cross-wave)
        const uint32_t a = c1[0] + c1[1] + c1[2] + c1[3], b = c2[0] + c2[1] + c2[2] + c2[3];
        const uint32_t sa = wave_inclusive_scan_u32(a), sb = wave_inclusive_scan_u32(b);
        if (lane == 63) {
            wsum[0][wave] = sa;
            wsum[1][wave] = sb;
        }
        __syncthreads();                 // also: previous row's E reads are done
        uint32_t oa = sa - a, ob = sb - b;          // exclusive offset of this thread's first column
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < wave) {
                oa += wsum[0][k];
                ob += wsum[1][k];
            }
        const uint32_t e1[4] = {oa, oa + c1[0], oa + c1[0] + c1[1], oa + c1[0] + c1[1] + c1[2]};
        const uint32_t e2[4] = {ob, ob + c2[0], ob + c2[0] + c2[1], ob + c2[0] + c2[1] + c2[2]};
        *reinterpret_cast<uint4*>(&E1[4 * t]) = make_uint4(e1[0], e1[1], e1[2], e1[3]);
        *reinterpret_cast<uint4*>(&E2[4 * t]) = make_uint4(e2[0], e2[1], e2[2], e2[3]);
        if (t == 255) {                  // E[kStatStrip]: read when the strip is full width
            E1[kStatStrip] = oa + a;
            E2[kStatStrip] = ob + b;
        }
        __syncthreads();
        double blk_s1[4] = {0.0, 0.0, 0.0, 0.0}, blk_sq[4] = {0.0, 0.0, 0.0, 0.0};
        if (out_on) {
            double tt[4], ws2[4], sqv[4], rs[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t s1 = E1[4 * t + k + w] - e1[k], s2 = E2[4 * t + k + w] - e2[k];
                tt[k] = (double)s1;
                ws2[k] = (double)s2;
                double wnd_mean2 = 0.0;
                if (num_type == 1) wnd_mean2 = (tt[k] * tt[k]) * inv_area;
                const double diff2 = fmax(ws2[k] - wnd_mean2, 0.0);
                const bool small = diff2 <= fmin(0.5, (10.0 * (double)FLT_EPSILON) * ws2[k]);
#ifdef MTM_PROBE_STAT_NO_SQRT   /* timing experiment (wrong results) */
                sqv[k] = small ? 0.0 : diff2;
#else
                sqv[k] = small ? 0.0 : sqrt(diff2);
#endif
                rs[k] = sqv[k] > 0.0 ? 1.0 / sqv[k] : 0.0;
                blk_s1[k] = tt[k];
                blk_sq[k] = sqv[k];
            }
            const size_t o = (size_t)y * st_pitch + xg;
            if (want_t) {
                *reinterpret_cast<double2*>(t0 + o) = make_double2(tt[0], tt[1]);
                *reinterpret_cast<double2*>(t0 + o + 2) = make_double2(tt[2], tt[3]);
            }
            if (want_sum2) {
                *reinterpret_cast<double2*>(sum2 + o) = make_double2(ws2[0], ws2[1]);
                *reinterpret_cast<double2*>(sum2 + o + 2) = make_double2(ws2[2], ws2[3]);
            }
            if (want_sq) {
                *reinterpret_cast<double2*>(sq + o) = make_double2(sqv[0], sqv[1]);
                *reinterpret_cast<double2*>(sq + o + 2) = make_double2(sqv[2], sqv[3]);
                if (rsq != nullptr) {                // row-multiplexed MFMA classes
                    *reinterpret_cast<double2*>(rsq + o) = make_double2(rs[0], rs[1]);
                    *reinterpret_cast<double2*>(rsq + o + 2) = make_double2(rs[2], rs[3]);
                }
            }
        }
        if (blk != nullptr) {
            // ranges over the 16-pixel column block this thread's quad of threads covers (the hits-only screen of the
            // multi-row MFMA variants bounds a lane's 16 outputs with them): S1 min / max and the smallest sqrt over the
            // block's output columns (x < ow); a block without any gets sqrt = +inf - no candidate can pass that
            double lo = INFINITY, hi = 0.0, sm = INFINITY;
            if (out_on) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (xg + k < ow) {
                        lo = fmin(lo, blk_s1[k]);
                        hi = fmax(hi, blk_s1[k]);
                        sm = fmin(sm, blk_sq[k]);
                    }
            }
#pragma unroll
            for (int off = 1; off <= 2; off <<= 1) {
                lo = fmin(lo, __shfl_xor(lo, off));
                hi = fmax(hi, __shfl_xor(hi, off));
                sm = fmin(sm, __shfl_xor(sm, off));
            }
            if ((t & 3) == 0 && 4 * t < owg && (xg >> 4) < blk_pitch) {
                double* o = blk + ((size_t)y * blk_pitch + (xg >> 4)) * 4;
                *reinterpret_cast<double2*>(o) = make_double2(lo == INFINITY ? 0.0 : lo, hi);
                *reinterpret_cast<double2*>(o + 2) = make_double2(sm, 0.0);
            }
        }
        // slide the column sums one row down (zeros on the last row: nothing changes)
        uint32_t bn[4], bo[4];
        unpack(vn, bn);
        unpack(vo, bo);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            c1[k] += bn[k] - bo[k];
            c2[k] += bn[k] * bn[k] - bo[k] * bo[k];
        }
    }
}

// The same for CH interleaved-to-planar channels (RGB): per-channel window sums S1_c, the sum of squares
// over all channels and the guarded sqrt of  sum_c S2_c - (sum_c S1_c^2) / A  (operation order of
// vsum_stats_kernel, so both routes round alike; the squares of the channels are added as integers
// before the scan, which is exact).  CH + 1 scans behind ONE pair of barriers per output row.  The
// launcher requires CH * w * h * 255^2 < 2^32.
template <int CH>
__global__ __launch_bounds__(256) void stats_u8_mc_kernel(const uint8_t* __restrict__ img, int pitch, long long plane,
                                                          int h, int w, int oh, int ow, int owg, double inv_area,
                                                          int num_type, int want_sq, int want_t, int want_sum2,
                                                          double* __restrict__ t0, long long t_plane,
                                                          double* __restrict__ sum2, double* __restrict__ sq,
                                                          int st_pitch) {
    __shared__ __attribute__((aligned(16))) uint32_t E[CH + 1][kStatStrip + 4];     // exclusive prefixes: S1_c, S2
    __shared__ uint32_t wsum[CH + 1][4];
    const int x0 = blockIdx.x * owg, y0 = blockIdx.y * kStatBand4;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int L = owg + w - 1;
    const bool ld = 4 * t < L && x0 + 4 * t + 3 < pitch;
    const uint8_t* base = img + (size_t)y0 * pitch + x0 + 4 * t;
    uint32_t cs[CH + 1][4];                          // column sums: S1 of each channel, S2 of all channels
#pragma unroll
    for (int c = 0; c <= CH; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) cs[c][k] = 0u;
    auto unpack = [](uint32_t v, uint32_t (&b)[4]) {
        b[0] = v & 255u;
        b[1] = (v >> 8) & 255u;
        b[2] = (v >> 16) & 255u;
        b[3] = v >> 24;
    };
    for (int r0 = 0; r0 < h; r0 += 4) {
        uint32_t v[4][CH];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < CH; ++c)
                v[i][c] = ld ? *reinterpret_cast<const uint32_t*>(base + c * plane + (size_t)min(r0 + i, h - 1) * pitch) : 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (r0 + i < h) {
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    uint32_t b[4];
                    unpack(v[i][c], b);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        cs[c][k] += b[k];
                        cs[CH][k] += b[k] * b[k];
                    }
                }
            }
    }
    const int y1 = min(y0 + kStatBand4, oh);
    const int xg = x0 + 4 * t;
    const bool out_on = 4 * t < owg && xg < st_pitch;
    for (int y = y0; y < y1; ++y) {
        uint32_t vn[CH], vo[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            vn[c] = vo[c] = 0u;
            if (y + 1 < y1 && ld) {
                vn[c] = *reinterpret_cast<const uint32_t*>(base + c * plane + (size_t)(y - y0 + h) * pitch);
                vo[c] = *reinterpret_cast<const uint32_t*>(base + c * plane + (size_t)(y - y0) * pitch);
            }
        }
        uint32_t tot[CH + 1], sc[CH + 1];
#pragma unroll
        for (int c = 0; c <= CH; ++c) {
            tot[c] = cs[c][0] + cs[c][1] + cs[c][2] + cs[c][3];
            sc[c] = wave_inclusive_scan_u32(tot[c]);
            if (lane == 63) wsum[c][wave] = sc[c];
        }
        __syncthreads();                 // also: previous row's E reads are done
        uint32_t e[CH + 1][4];
#pragma unroll
        for (int c = 0; c <= CH; ++c) {
            uint32_t off = sc[c] - tot[c];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < wave) off += wsum[c][k];
            e[c][0] = off;
            e[c][1] = off + cs[c][0];
            e[c][2] = e[c][1] + cs[c][1];
            e[c][3] = e[c][2] + cs[c][2];
            *reinterpret_cast<uint4*>(&E[c][4 * t]) = make_uint4(e[c][0], e[c][1], e[c][2], e[c][3]);
            if (t == 255) E[c][kStatStrip] = off + tot[c];
        }
        __syncthreads();
        if (out_on) {
            const size_t o = (size_t)y * st_pitch + xg;
            double mean2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                double tt[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    tt[k] = (double)(E[c][4 * t + k + w] - e[c][k]);
                    if (num_type == 1) mean2[k] += tt[k] * tt[k];
                }
                if (want_t) {
                    *reinterpret_cast<double2*>(t0 + c * t_plane + o) = make_double2(tt[0], tt[1]);
                    *reinterpret_cast<double2*>(t0 + c * t_plane + o + 2) = make_double2(tt[2], tt[3]);
                }
            }
            double ws2[4], sqv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ws2[k] = (double)(E[CH][4 * t + k + w] - e[CH][k]);
                const double wnd_mean2 = mean2[k] * inv_area;
                const double diff2 = fmax(ws2[k] - wnd_mean2, 0.0);
                const bool small = diff2 <= fmin(0.5, (10.0 * (double)FLT_EPSILON) * ws2[k]);
                sqv[k] = small ? 0.0 : sqrt(diff2);
            }
            if (want_sum2) {
                *reinterpret_cast<double2*>(sum2 + o) = make_double2(ws2[0], ws2[1]);
                *reinterpret_cast<double2*>(sum2 + o + 2) = make_double2(ws2[2], ws2[3]);
            }
            if (want_sq) {
                *reinterpret_cast<double2*>(sq + o) = make_double2(sqv[0], sqv[1]);
                *reinterpret_cast<double2*>(sq + o + 2) = make_double2(sqv[2], sqv[3]);
            }
        }
        // slide the column sums one row down (zeros on the last row: nothing changes)
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            uint32_t bn[4], bo[4];
            unpack(vn[c], bn);
            unpack(vo[c], bo);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                cs[c][k] += bn[k] - bo[k];
                cs[CH][k] += bn[k] * bn[k] - bo[k] * bo[k];
            }
        }
    }
}

#ifndef MTM_VSUM_BAND
#define MTM_VSUM_BAND 32
#endif
constexpr int kVsumBand = MTM_VSUM_BAND;

template <typename AccT, typename SumT>
__global__ void vsum_stats_kernel(const AccT* __restrict__ hs1, const AccT* __restrict__ hs2,
                                  int hs_pitch, long long hs_plane, int chans, int h, int oh, int ow,
                                  double inv_area, int num_type, int want_sq, int want_t,
                                  double* __restrict__ t0, double* __restrict__ t1,
                                  double* __restrict__ t2, double* __restrict__ t3,
                                  double* __restrict__ sum2, double* __restrict__ sq, int pitch) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y0 = blockIdx.y * kVsumBand;
    if (x >= ow || y0 >= oh) return;
    double* tp[kMaxChans] = {t0, t1, t2, t3};
    SumT s1[kMaxChans], s2[kMaxChans];
    // all channel loops are unrolled over kMaxChans with a guard: no dynamically indexed
    // private arrays (they would go to scratch)
#pragma unroll
    for (int c = 0; c < kMaxChans; ++c) {
        s1[c] = 0;
        s2[c] = 0;
        if (c < chans) {
            SumT a = 0, b = 0;
            const AccT* p1 = hs1 + c * hs_plane + (size_t)y0 * hs_pitch + x;
            const AccT* p2 = hs2 + c * hs_plane + (size_t)y0 * hs_pitch + x;
            for (int dy = 0; dy < h; ++dy) {
                a += (SumT)p1[(size_t)dy * hs_pitch];
                b += (SumT)p2[(size_t)dy * hs_pitch];
            }
            s1[c] = a;
            s2[c] = b;
        }
    }
    const int y1 = min(y0 + kVsumBand, oh);
    for (int y = y0; y < y1; ++y) {
        double wnd_mean2 = 0.0, wnd_sum2 = 0.0;
#pragma unroll
        for (int c = 0; c < kMaxChans; ++c) {
            if (c < chans) {
                const double t = (double)s1[c];
                if (num_type == 1) wnd_mean2 += t * t;
                if (want_t) tp[c][(size_t)y * pitch + x] = t;
                wnd_sum2 += (double)s2[c];
            }
        }
        wnd_mean2 *= inv_area;
        sum2[(size_t)y * pitch + x] = wnd_sum2;
        if (want_sq) {
            const double diff2 = fmax(wnd_sum2 - wnd_mean2, 0.0);
            const bool small = diff2 <= fmin(0.5, (10.0 * (double)FLT_EPSILON) * wnd_sum2);
            sq[(size_t)y * pitch + x] = small ? 0.0 : sqrt(diff2);
        }
        if (y + 1 < y1) {
#pragma unroll
            for (int c = 0; c < kMaxChans; ++c) {
                if (c < chans) {
                    const size_t o = c * hs_plane + x;
                    s1[c] += (SumT)hs1[o + (size_t)(y + h) * hs_pitch] - (SumT)hs1[o + (size_t)y * hs_pitch];
                    s2[c] += (SumT)hs2[o + (size_t)(y + h) * hs_pitch] - (SumT)hs2[o + (size_t)y * hs_pitch];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Masked templates: sum I^2 * M over every window on the matrix cores.  I^2 is a 16-bit number; its two
// bytes are image planes of their own (square_planes_kernel), the binary mask is the "template" of a
// row-multiplexed raw correlation (one template, 16 output rows per MFMA), and masksq_combine_kernel
// puts the two byte-plane results together:  c2 = 256 (a_h + 128 S1_h + K) + (a_l + 128 S1_l + K),
// K = 128 sum(M) - 16384 A, S1_h / S1_l the window sums of the byte planes (S1_l = S2 - 256 S1_h with
// S2 the plain window sum of squares).  All integers < 2^53: exact.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void square_planes_kernel(const uint8_t* __restrict__ u8, size_t n16,
                                                            uint8_t* __restrict__ sh, uint8_t* __restrict__ shb,
                                                            uint8_t* __restrict__ slb) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n16) return;
    const uint4 v = reinterpret_cast<const uint4*>(u8)[g];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        hi[k] = lo[k] = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t px = (w[k] >> (8 * b)) & 255u, sq = px * px;
            hi[k] |= (sq >> 8) << (8 * b);
            lo[k] |= (sq & 255u) << (8 * b);
        }
    }
    reinterpret_cast<uint4*>(sh)[g] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    reinterpret_cast<uint4*>(shb)[g] = make_uint4(hi[0] ^ 0x80808080u, hi[1] ^ 0x80808080u, hi[2] ^ 0x80808080u,
                                                  hi[3] ^ 0x80808080u);
    reinterpret_cast<uint4*>(slb)[g] = make_uint4(lo[0] ^ 0x80808080u, lo[1] ^ 0x80808080u, lo[2] ^ 0x80808080u,
                                                  lo[3] ^ 0x80808080u);
}

__global__ __launch_bounds__(256) void masksq_combine_kernel(const int* __restrict__ raw_h, const int* __restrict__ raw_l,
                                                             int raw_pitch, const double* __restrict__ s1h,
                                                             double* __restrict__ sum2, int st_pitch, double km,
                                                             int oh, int ow) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= ow || y >= oh) return;
    const size_t o = (size_t)y * st_pitch + x, r = (size_t)y * raw_pitch + x;
    const double h1 = s1h[o], l1 = sum2[o] - 256.0 * h1;     // sum2 holds the plain window sum of squares here
    const double ch = (double)raw_h[r] + 128.0 * h1 + km, cl = (double)raw_l[r] + 128.0 * l1 + km;
    sum2[o] = 256.0 * ch + cl;
}

// ---------------------------------------------------------------------------------------------
// uint16 images and templates on the int8 matrix cores, exactly.
//   I = 256 Ih + Il, T = 256 Th + Tl  (bytes)  =>
//   sum I*T = 65536 R_hh + 256 (R_hl + R_lh) + R_ll,   R_xy = sum I_x * T_y   (uint8 x uint8)
// Two launches of ncc_mfma_kernel over the image's byte planes, each against [T_hi | T_lo] of 16 templates per work
// item: the high-byte pass stores its biased accumulators a_hh, a_hl (RAW mode), the low-byte pass (kMfU16) reads
// them back in its epilogue, rebuilds R_xy = a_xy + 128 S1_x + 128 sum(T_y) - 16384 A (S1_x the window sum of byte
// plane x; S1_lo = S1 - 256 S1_hi), combines in float64 (all terms are integers < 2^53: exact) and normalises like
// every other kernel.  See mtm_mfma.hip.h.
// ---------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------
// Large uint8 templates on the int8 matrix cores.  The int32 accumulator of ncc_mfma_kernel holds
// |sum (I-128)(T-128)| <= 16384 * w * h * C only for w*h*C <= 131071 (and its LDS tile wants w <= 256), so a larger
// template is cut into slabs - row ranges x column blocks x channels - each within those limits.  Every slab is a
// template of its own correlated (RAW mode: biased int32 accumulators) against the image shifted by the slab's
// offset; the slabs of a template add up to its full biased correlation, and
//   sum I*T = sum_slabs a_s + 128 * S1 + 128 * sum(T) - 16384 * w * h * C
// with S1 the window sum over the WHOLE template window (the slabs' window sums add up to it) - exact integers,
// summed here in float64 (< 2^53) and normalised by finish_unmasked like every other kernel.
// raw layout: [slab][template (list position)][oh][pitch] int32.
// ---------------------------------------------------------------------------------------------
struct SlabParams {
    mtm_hit* cand_hits;
    unsigned long long* cand_counter;
    unsigned long long cand_cap;
    float cand_thr;
    int cand_min, cand_on, hits_only;
    int w, h, chans;
    const int* raw;
    long long raw_slab;        // ints per slab block: n_list * raw_map
    long long raw_map;         // ints per template map: oh * pitch
    int n_slabs;
    int oh, ow, pitch;
    int n_list;
    int method;
};

__global__ __launch_bounds__(256) void slab_combine_kernel(SlabParams p, const TemplDev* __restrict__ td,
                                                           const int* __restrict__ tlist, StatPlanes st,
                                                           float* __restrict__ maps, int only_li) {
    const int li = blockIdx.z;
    if (only_li >= 0 && li != only_li) return;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= p.ow || y >= p.oh) return;
    const TemplDev T = td[tlist[li]];
    const size_t o = (size_t)li * p.raw_map + (size_t)y * p.pitch + x;
    long long a = 0;
    for (int k = 0; k < p.n_slabs; ++k) a += (long long)p.raw[(size_t)k * p.raw_slab + o];
    const size_t sidx = (size_t)y * st.pitch + x;
    double s1 = 0.0;
    for (int c = 0; c < p.chans; ++c) s1 += st.t[c][sidx];
    const double corr = ((double)a + 128.0 * s1) + T.mfma_k;
    const float out = finish_unmasked(p.method, corr, st, sidx, T, p.chans);
    if (p.cand_on) {
        mtm_hit hrec;
        hrec.templ_idx = tlist[li];
        hrec.x = x;
        hrec.y = y;
        hrec.w = p.w;
        hrec.h = p.h;
        hrec.score = out;
        cand_append((p.cand_min ? -out : out) > p.cand_thr, p.cand_counter, p.cand_cap, p.cand_hits, hrec);
    }
    if (!p.hits_only) maps[T.map_off + (size_t)y * T.map_pitch + x] = out;
}

// ---------------------------------------------------------------------------------------------
// NAIVE score-map kernel: one thread per output pixel, float64 FMA chain over the window.
// Generic (uint8 or float32 pixels, masks, any size); it is the in-library cross-check for the
// tiled kernels and the fallback for shapes they do not take.
// ---------------------------------------------------------------------------------------------
__global__ void ncc_naive_kernel(ImageDev img, const TemplDev* __restrict__ td,
                                 const int* __restrict__ tlist, const double* __restrict__ weights,
                                 StatPlanes st, int method, int masked, float* __restrict__ maps) {
    const int t = tlist[blockIdx.z];
    const TemplDev T = td[t];
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= T.ow || y >= T.oh) return;
    const int h = T.rows, w = T.cols;
    double c1 = 0.0, c2 = 0.0;
    for (int c = 0; c < img.chans; ++c) {
        const float* ip = img.f32 + c * img.f32_plane + (size_t)y * img.f32_pitch + x;
        const double* k1 = weights + T.k1_off + (size_t)c * h * w;
        const double* k2 = masked ? (weights + T.k2_off + (size_t)c * h * w) : nullptr;
        double a1 = 0.0, a2 = 0.0;
        for (int dy = 0; dy < h; ++dy) {
            const float* r = ip + (size_t)dy * img.f32_pitch;
            for (int dx = 0; dx < w; ++dx) {
                const double v = (double)r[dx];
                a1 = fma(v, k1[dy * w + dx], a1);
                if (masked) a2 = fma(v * v, k2[dy * w + dx], a2);
            }
        }
        c1 += a1;
        c2 += a2;
    }
    float out;
    if (masked) out = finish_masked(method, c1, c2, T);
    else out = finish_unmasked(method, c1, st, (size_t)y * st.pitch + x, T, img.chans);
    maps[T.map_off + (size_t)y * T.map_pitch + x] = out;
}

// ---------------------------------------------------------------------------------------------
// TILED float64 score-map kernel for float32 pixels and for masked templates.
// Block = 32x8 threads; every thread owns 4 consecutive outputs of one row; the image tile is
// staged in LDS as float32, the template weights (float64) are wave-uniform scalar loads.
// ---------------------------------------------------------------------------------------------
constexpr int kF64ChunkH = 16, kF64ChunkW = 32;
constexpr int kF64BX = 128, kF64BY = 8;
constexpr int kF64LdsPitch = kF64BX + kF64ChunkW + 4;   // floats, multiple of 4

template <bool MASKED>
__global__ __launch_bounds__(256) void ncc_f64_kernel(ImageDev img, const TemplDev* __restrict__ td,
                                                      const int* __restrict__ tlist,
                                                      const double* __restrict__ weights,
                                                      StatPlanes st, int method,
                                                      float* __restrict__ maps, int ntx) {
    __shared__ __attribute__((aligned(16))) float tile[(kF64BY + kF64ChunkH - 1) * kF64LdsPitch];
    const int t = tlist[blockIdx.y];
    const TemplDev T = td[t];
    const int h = T.rows, w = T.cols;
    const int txi = blockIdx.x % ntx, tyi = blockIdx.x / ntx;
    const int tx0 = txi * kF64BX, ty0 = tyi * kF64BY;
    if (tx0 >= T.ow || ty0 >= T.oh) return;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    double acc1[4] = {0, 0, 0, 0}, acc2[4] = {0, 0, 0, 0};
    double tot1[4] = {0, 0, 0, 0}, tot2[4] = {0, 0, 0, 0};

    for (int c = 0; c < img.chans; ++c) {
        const float* plane = img.f32 + c * img.f32_plane;
        const double* k1 = weights + T.k1_off + (size_t)c * h * w;
        const double* k2 = MASKED ? (weights + T.k2_off + (size_t)c * h * w) : nullptr;
        for (int cy0 = 0; cy0 < h; cy0 += kF64ChunkH) {
            const int ch = min(kF64ChunkH, h - cy0);
            for (int cx0 = 0; cx0 < w; cx0 += kF64ChunkW) {
                const int cw = min(kF64ChunkW, w - cx0);
                __syncthreads();
                // stage (ch + BY - 1) rows x (BX + 32 + 4) floats, as float4
                const int nrow = ch + kF64BY - 1;
                constexpr int q4 = kF64LdsPitch / 4;
                for (int idx = threadIdx.x; idx < nrow * q4; idx += 256) {
                    const int r = idx / q4, q = idx - r * q4;
                    const float4 v = *reinterpret_cast<const float4*>(
                        plane + (size_t)(ty0 + cy0 + r) * img.f32_pitch + tx0 + cx0 + 4 * q);
                    *reinterpret_cast<float4*>(&tile[r * kF64LdsPitch + 4 * q]) = v;
                }
                __syncthreads();
                for (int dy = 0; dy < ch; ++dy) {
                    const float* lrow = &tile[(ly + dy) * kF64LdsPitch + 4 * lx];
                    const double* kr1 = k1 + (size_t)(cy0 + dy) * w + cx0;
                    const double* kr2 = MASKED ? (k2 + (size_t)(cy0 + dy) * w + cx0) : nullptr;
                    float4 cur = *reinterpret_cast<const float4*>(lrow);
                    for (int dx4 = 0; dx4 < cw; dx4 += 4) {
                        const float4 nxt = *reinterpret_cast<const float4*>(lrow + dx4 + 4);
                        const double v[8] = {(double)cur.x, (double)cur.y, (double)cur.z, (double)cur.w,
                                             (double)nxt.x, (double)nxt.y, (double)nxt.z, (double)nxt.w};
                        double v2[8];
                        if (MASKED) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) v2[i] = v[i] * v[i];
                        }
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            if (dx4 + s < cw) {
                                const double ka = kr1[dx4 + s];
#pragma unroll
                                for (int k = 0; k < 4; ++k) acc1[k] = fma(v[k + s], ka, acc1[k]);
                                if (MASKED) {
                                    const double kb = kr2[dx4 + s];
#pragma unroll
                                    for (int k = 0; k < 4; ++k) acc2[k] = fma(v2[k + s], kb, acc2[k]);
                                }
                            }
                        }
                        cur = nxt;
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tot1[k] += acc1[k]; acc1[k] = 0.0;
            tot2[k] += acc2[k]; acc2[k] = 0.0;
        }
    }
    const int y = ty0 + ly;
    if (y >= T.oh) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = tx0 + 4 * lx + k;
        if (x >= T.ow) continue;
        float out;
        if (MASKED) out = finish_masked(method, tot1[k], tot2[k], T);
        else out = finish_unmasked(method, tot1[k], st, (size_t)y * st.pitch + x, T, img.chans);
        maps[T.map_off + (size_t)y * T.map_pitch + x] = out;
    }
}

// ---------------------------------------------------------------------------------------------
// DOT4 score-map kernel: the uint8 hot path on the vector ALU.
//
//   * work-group = 256 threads as 32 (x) x 8 (y); a thread owns PX consecutive output columns x
//     PY consecutive output rows for NT templates: PX*PY*NT uint32 accumulators in registers;
//     output tile = (32*PX) x (8*PY) pixels.
//   * the image tile (tile + template chunk halo) is staged once in LDS as dwords and reused for
//     all NT templates and all PY rows; a lane walks a tile row one dword at a time and forms
//     its byte-shifted windows with v_alignbyte_b32 (3 per dword).
//   * template rows are wave-uniform: they are read with scalar loads (s_load_dword*) straight
//     into SGPRs and used as the scalar operand of v_dot4_u32_u8: no VGPRs, no LDS bandwidth.
//   * templates larger than 64x64 are processed in 64x64 chunks (image tile re-staged per chunk),
//     so LDS use is bounded (<= 31 KB) for any template size.
//   * sums are exact: a chunk's partial sum is < 2^32; WIDE folds it into uint64 totals.
//   * epilogue in float64 (finish_unmasked), float32 store.
//
// MASKSQ variant: the packed "template" is a binary mask (bytes 0xFF / 0x00) and the inner operation is
// dot4(window & mask, window): the masked sum of squares  sum I^2 * M  that OpenCV's matchTemplateMask
// needs, exact in uint32; written as float64 into the statistics plane `sumsq_out`.
//
// Packed template layout (host: pack_template_dot4): per template, per channel, per chunk
// (cy, cx): (kDotChunk + 2*kDotPadRows) rows of kDotChunk bytes, zero filled, template row dy of
// the chunk at packed row dy + kDotPadRows: rows that fall outside the chunk multiply by zero, so
// the PY-row register blocking needs no conditionals.
// ---------------------------------------------------------------------------------------------
constexpr int kDotChunk = 64;
constexpr int kDotPadRows = 3;                                   // supports PY <= 4
constexpr int kDotPackRows = kDotChunk + 2 * kDotPadRows;
constexpr int kDotChunkBytes = kDotPackRows * kDotChunk;

struct DotParams {
    const uint8_t* img;     // planar padded u8
    int pitch;              // bytes
    long long plane;
    int chans;
    int h, w;               // template size of this class
    int oh, ow;
    int ncy, ncx;           // chunk grid of the packed templates
    int n_list;             // templates in this launch
    int ntx, nty;           // output tile grid
    int nchunks;            // ceil(n_list / NT)
    int n_work;             // ntx * nty * nchunks
    int method;
    double* sumsq_out;      // MASKSQ: destination plane (pitch = st.pitch) of sum I^2 * M
};

template <int PX, int PY, int NT, bool WIDE, bool MASKSQ = false>
__global__ __launch_bounds__(256) void ncc_dot4_kernel(DotParams p, const TemplDev* __restrict__ td,
                                                       const int* __restrict__ tlist,
                                                       const uint8_t* __restrict__ packs,
                                                       StatPlanes st, float* __restrict__ maps) {
    constexpr int BX = 32 * PX, BY = 8 * PY;
    constexpr int PXD = PX / 4;
    constexpr int LP = BX / 4 + kDotChunk / 4 + 1;          // LDS row pitch in dwords
    constexpr int LROWS = kDotChunk + BY - 1;
    constexpr int EPAD = 256 + 32 / PX;                     // epilogue LDS pitch: conflict-free
    constexpr int ELDS = PX * PY * EPAD * (WIDE ? 2 : 1);
    constexpr int LDS_DW = (LROWS * LP > ELDS) ? LROWS * LP : ELDS;
    __shared__ uint32_t tile[LDS_DW];

    // XCD-aware work mapping: block b runs on XCD b % 8; give each XCD a contiguous range of work
    // items (tile-major, template-chunk-minor) so the template chunks of one image tile hit the
    // same L2.
    const int per_xcd = (p.n_work + 7) >> 3;
    const int wid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (wid >= p.n_work) return;
    const int chunk = wid % p.nchunks;
    const int tile_id = wid / p.nchunks;
    const int txi = tile_id % p.ntx, tyi = tile_id / p.ntx;
    const int tx0 = txi * BX, ty0 = tyi * BY;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;

    // wave-uniform template indices / packed bases of this chunk (tail entries repeat the last)
    int tidx[NT];
    const uint32_t* tbase[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int li = min(chunk * NT + t, p.n_list - 1);
        tidx[t] = tlist[li];
        tbase[t] = reinterpret_cast<const uint32_t*>(packs + td[tidx[t]].pack_off);
    }

    uint32_t acc[NT][PY][PX];
    unsigned long long tot[WIDE ? NT : 1][WIDE ? PY : 1][WIDE ? PX : 1];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < PY; ++r)
#pragma unroll
            for (int k = 0; k < PX; ++k) {
                acc[t][r][k] = 0u;
                if (WIDE) tot[t][r][k] = 0ull;
            }

    const int w4 = (p.w + 3) & ~3;
    for (int c = 0; c < p.chans; ++c) {
        const uint8_t* plane = p.img + c * p.plane;
        for (int cyi = 0; cyi < p.ncy; ++cyi) {
            const int cy0 = cyi * kDotChunk;
            const int ch = min(kDotChunk, p.h - cy0);
            for (int cxi = 0; cxi < p.ncx; ++cxi) {
                const int cx0 = cxi * kDotChunk;
                const int cw4 = min(kDotChunk, w4 - cx0) >> 2;      // dwords per template row
                const int chunk_dw = ((c * p.ncy + cyi) * p.ncx + cxi) * (kDotChunkBytes / 4);
                // ---- stage the image tile: (ch + BY - 1) rows x (BX/4 + cw4 + 1) dwords
                __syncthreads();
                {
                    const int nrow = ch + BY - 1;
                    const int ncol = BX / 4 + cw4 + 1;
                    const int col = threadIdx.x & 63, r0 = threadIdx.x >> 6;
                    for (int cc = col; cc < ncol; cc += 64) {
                        const uint8_t* g = plane + (size_t)(ty0 + cy0) * p.pitch + tx0 + cx0 + 4 * cc;
                        for (int r = r0; r < nrow; r += 4)
                            tile[r * LP + cc] = *reinterpret_cast<const uint32_t*>(g + (size_t)r * p.pitch);
                    }
                }
                __syncthreads();
                // ---- accumulate: walk the tile rows this thread's PY output rows touch
                const int nj = ch + PY - 1;
                for (int j = 0; j < nj; ++j) {
                    const uint32_t* lrow = &tile[(ly * PY + j) * LP + lx * PXD];
                    uint32_t d[PXD + 1];
#pragma unroll
                    for (int q = 0; q < PXD; ++q) d[q] = lrow[q];
                    // packed template row of output row r at tile row j: (j - r) + kDotPadRows
                    const int prow0 = chunk_dw + (j + kDotPadRows) * (kDotChunk / 4);
#pragma unroll 4
                    for (int s = 0; s < cw4; ++s) {
                        d[PXD] = lrow[s + PXD];
                        uint32_t win[PX];
#pragma unroll
                        for (int q = 0; q < PXD; ++q) {
                            win[4 * q + 0] = d[q];
                            win[4 * q + 1] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], 1);
                            win[4 * q + 2] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], 2);
                            win[4 * q + 3] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], 3);
                        }
#pragma unroll
                        for (int r = 0; r < PY; ++r) {
#pragma unroll
                            for (int t = 0; t < NT; ++t) {
                                const uint32_t tw = tbase[t][prow0 - r * (kDotChunk / 4) + s];
#pragma unroll
                                for (int k = 0; k < PX; ++k)
                                    acc[t][r][k] = MASKSQ ? __builtin_amdgcn_udot4(win[k] & tw, win[k], acc[t][r][k], false)
                                                          : __builtin_amdgcn_udot4(win[k], tw, acc[t][r][k], false);
                            }
                        }
#pragma unroll
                        for (int q = 0; q < PXD; ++q) d[q] = d[q + 1];
                    }
                }
                if (WIDE) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int r = 0; r < PY; ++r)
#pragma unroll
                            for (int k = 0; k < PX; ++k) {
                                tot[t][r][k] += acc[t][r][k];
                                acc[t][r][k] = 0u;
                            }
                }
            }
        }
    }

    // ---- epilogue.  The accumulators of one template at a time go through LDS (transposed), so
    // that the float64 normalisation runs as ONE rolled loop per template (small code) in which
    // consecutive lanes own consecutive output columns: coalesced statistics loads and map stores.
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        __syncthreads();        // the tile (or the previous template's values) is no longer read
#pragma unroll
        for (int r = 0; r < PY; ++r)
#pragma unroll
            for (int k = 0; k < PX; ++k) {
                const int slot = (r * PX + k) * EPAD + threadIdx.x;
                if (WIDE) {
                    tile[slot] = (uint32_t)tot[t][r][k];
                    tile[PX * PY * EPAD + slot] = (uint32_t)(tot[t][r][k] >> 32);
                } else {
                    tile[slot] = acc[t][r][k];
                }
            }
        __syncthreads();
        if (chunk * NT + t >= p.n_list) continue;      // wave-uniform
        const TemplDev T = td[tidx[t]];
        float* mbase = maps + T.map_off;
        for (int i = 0; i < PX * PY; ++i) {
            const int idx = i * 256 + threadIdx.x;
            const int cc = idx % BX, rr = idx / BX;                 // pixel inside the tile
            const int slot = ((rr % PY) * PX + (cc % PX)) * EPAD + (rr / PY) * 32 + (cc / PX);
            const int x = tx0 + cc, y = ty0 + rr;
            if (x < p.ow && y < p.oh) {
                double corr;
                if (WIDE) corr = (double)(((unsigned long long)tile[PX * PY * EPAD + slot] << 32) | tile[slot]);
                else corr = (double)tile[slot];
                if (MASKSQ)      // "template" = binary mask bytes (0xFF / 0): sum over the window of I^2 * M
                    p.sumsq_out[(size_t)y * st.pitch + x] = corr;
                else
                    mbase[(size_t)y * T.map_pitch + x] =
                        finish_unmasked(p.method, corr, st, (size_t)y * st.pitch + x, T, p.chans);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// peak extraction: skimage.feature.peak_local_max(map, threshold_abs=thr, exclude_border=False)
// on a 2-D map (reference MTM/__init__.py:45): pixel == max of its 3x3 neighbourhood and
// pixel > thr.  mode_min evaluates it on the negated map (MTM/__init__.py:51-53).  Candidates are
// appended to a global hit buffer; `nontrivial[t]` records that some pixel differs from its local
// max (skimage returns no peak at all for a map where none does).
// ---------------------------------------------------------------------------------------------
// One wave owns a strip of 256 columns x kPkRows rows and walks it top to bottom: per row one
// coalesced float4 load per lane (4 pixels), the horizontal neighbours come from the adjacent
// lanes by shuffle (strip edges: two scalar loads), three rows of horizontal 3-maxima stay in
// registers.  A work-group is 4 such strips stacked vertically.
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

__global__ __launch_bounds__(256) void peaks_kernel(const float* __restrict__ maps,
                                                    const TemplDev* __restrict__ td,
                                                    const int* __restrict__ tlist, int mode_min,
                                                    float thr, int border, mtm_hit* __restrict__ hits,
                                                    unsigned long long cap,
                                                    unsigned long long* __restrict__ counter,
                                                    int* __restrict__ nontrivial) {
    const int t = tlist[blockIdx.z];
    const TemplDev T = td[t];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xs = blockIdx.x * kPkCols;
    const int y0 = (blockIdx.y * 4 + wave) * kPkRows;
    int nontriv = 0;
    if (xs < T.ow && y0 < T.oh) {
        const float* m = maps + T.map_off;
        const float padv = (border == MTM_BORDER_CONSTANT) ? 0.0f : -INFINITY;
        const float thr2 = mode_min ? -thr : thr;
        const int xb = xs + 4 * lane;
        // hm_*[k] = max over columns xb+k-1 .. xb+k+1 of one row; c_* = the row's own values
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
                if (x < T.ow) {
                    const float v = vb[k];
                    const float mx = fmaxf(fmaxf(hm_a[k], hm_b[k]), hm_c[k]);
                    if (!(v == mx)) {
                        nontriv = 1;
                    } else if (v > thr2) {
                        const unsigned long long slot = atomicAdd(counter, 1ull);
                        if (slot < cap) {
                            mtm_hit hrec;
                            hrec.templ_idx = t;
                            hrec.x = x;
                            hrec.y = y;
                            hrec.w = T.cols;
                            hrec.h = T.rows;
                            hrec.score = mode_min ? -v : v;
                            hits[slot] = hrec;
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                vb[k] = vc[k];
                hm_a[k] = hm_b[k];
                hm_b[k] = hm_c[k];
            }
        }
    }
    if (__syncthreads_or(nontriv) && threadIdx.x == 0) nontrivial[t] = 1;
}

// Second half of the fused peak extraction: the score-map kernel has appended every pixel above the
// threshold to `cands`; a candidate is a peak iff it equals the maximum of its 3x3 neighbourhood
// (same border rule and minima handling as peaks_kernel).  tcount[t] counts the peaks of template t:
// tcount[t] == oh*ow means every pixel equals its local maximum, i.e. skimage's "trivial image".
__global__ __launch_bounds__(256) void verify_peaks_kernel(const float* __restrict__ maps,
                                                           const TemplDev* __restrict__ td, int mode_min,
                                                           int border, const mtm_hit* __restrict__ cands,
                                                           const unsigned long long* __restrict__ cand_count,
                                                           unsigned long long cand_cap, mtm_hit* __restrict__ hits,
                                                           unsigned long long hit_cap,
                                                           unsigned long long* __restrict__ hit_count,
                                                           int* __restrict__ tcount, float thr_q) {
    const unsigned long long n = min(*cand_count, cand_cap);
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const mtm_hit c = cands[i];
    const TemplDev T = td[c.templ_idx];
    const float* m = maps + T.map_off;
    const float padv = (border == MTM_BORDER_CONSTANT) ? 0.0f : -INFINITY;
    const float v = mode_min ? -c.score : c.score;
    float mx = v;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = c.y + dy, xx = c.x + dx;
            float nv = padv;
            if (yy >= 0 && yy < T.oh && xx >= 0 && xx < T.ow) {
                nv = m[(size_t)yy * T.map_pitch + xx];
                if (mode_min) nv = -nv;
            }
            mx = fmaxf(mx, nv);
        }
    if (v == mx && v > thr_q) {       // (v > thr_q: always true for the integer kernels' lists; the float32 screen lists with a margin)
        const unsigned long long slot = atomicAdd(hit_count, 1ull);
        if (slot < hit_cap) hits[slot] = c;
        atomicAdd(&tcount[c.templ_idx], 1);
    }
}

// Hits-only mode (no score maps in memory): the same test on the candidate list alone.  Every pixel
// above the threshold IS a candidate, so a neighbour that is not in the list is <= threshold < v and
// cannot beat the candidate; neighbours that are in the list are found through an open-addressing
// hash table keyed by (template, y, x) built by cand_hash_insert_kernel.
__device__ __forceinline__ unsigned long long cand_key(int t, int y, int x) {
    return ((unsigned long long)(t + 1) << 42) | ((unsigned long long)y << 21) | (unsigned long long)x;
}
__device__ __forceinline__ unsigned cand_slot(unsigned long long k, unsigned mask) {
    return (unsigned)((k * 0x9E3779B97F4A7C15ull) >> 32) & mask;
}

__global__ __launch_bounds__(256) void cand_hash_insert_kernel(const mtm_hit* __restrict__ cands,
                                                               const unsigned long long* __restrict__ cand_count,
                                                               unsigned long long cand_cap,
                                                               unsigned long long* __restrict__ keys,
                                                               int* __restrict__ vals, unsigned mask) {
    const unsigned long long n = min(*cand_count, cand_cap);
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const mtm_hit c = cands[i];
    const unsigned long long k = cand_key(c.templ_idx, c.y, c.x);
    for (unsigned s = cand_slot(k, mask);; s = (s + 1) & mask) {
        const unsigned long long prev = atomicCAS(&keys[s], 0ull, k);
        if (prev == 0ull || prev == k) {
            vals[s] = (int)i;
            return;
        }
    }
}

__global__ __launch_bounds__(256) void verify_hash_kernel(const TemplDev* __restrict__ td, int mode_min, int border,
                                                          const mtm_hit* __restrict__ cands,
                                                          const unsigned long long* __restrict__ cand_count,
                                                          unsigned long long cand_cap,
                                                          const unsigned long long* __restrict__ keys,
                                                          const int* __restrict__ vals, unsigned mask,
                                                          mtm_hit* __restrict__ hits, unsigned long long hit_cap,
                                                          unsigned long long* __restrict__ hit_count,
                                                          int* __restrict__ tcount, float thr_q) {
    const unsigned long long n = min(*cand_count, cand_cap);
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const mtm_hit c = cands[i];
    const TemplDev T = td[c.templ_idx];
    const float padv = (border == MTM_BORDER_CONSTANT) ? 0.0f : -INFINITY;
    const float v = mode_min ? -c.score : c.score;
    float mx = v;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            if (dy == 0 && dx == 0) continue;
            const int yy = c.y + dy, xx = c.x + dx;
            if (yy < 0 || yy >= T.oh || xx < 0 || xx >= T.ow) {
                mx = fmaxf(mx, padv);
                continue;
            }
            const unsigned long long k = cand_key(c.templ_idx, yy, xx);
            for (unsigned s = cand_slot(k, mask);; s = (s + 1) & mask) {
                const unsigned long long have = keys[s];
                if (have == 0ull) break;                  // not a candidate: <= threshold < v
                if (have == k) {
                    const float nv = cands[vals[s]].score;
                    mx = fmaxf(mx, mode_min ? -nv : nv);
                    break;
                }
            }
        }
    if (v == mx && v > thr_q) {       // (v > thr_q: always true for the integer kernels' lists; the float32 screen lists with a margin)
        const unsigned long long slot = atomicAdd(hit_count, 1ull);
        if (slot < hit_cap) hits[slot] = c;
        atomicAdd(&tcount[c.templ_idx], 1);
    }
}

// ---------------------------------------------------------------------------------------------
// global extremum: cv2.minMaxLoc (reference MTM/__init__.py:226): first occurrence in row-major
// order wins ties.  One packed 64-bit key per (template, min|max): high word = order-preserving
// image of the float, low word = ~index, combined with atomicMax.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t float_order(float v) {
    if (v == 0.0f) v = 0.0f;     // -0 -> +0
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ __launch_bounds__(256) void extremum_kernel(const float* __restrict__ maps,
                                                       const TemplDev* __restrict__ td, int n_blocks_per_map,
                                                       unsigned long long* __restrict__ best) {
    const int t = blockIdx.y;
    const TemplDev T = td[t];
    const long long n = (long long)T.oh * T.ow;
    const float* m = maps + T.map_off;
    unsigned long long kmax = 0ull, kmin = 0ull;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)n_blocks_per_map * 256) {
        const int y = (int)(i / T.ow), x = (int)(i - (long long)y * T.ow);
        const float v = m[(size_t)y * T.map_pitch + x];
        if (v != v) continue;   // NaN never wins
        const uint32_t o = float_order(v);
        const uint32_t ri = 0xFFFFFFFFu - (uint32_t)i;
        const unsigned long long a = ((unsigned long long)o << 32) | ri;
        const unsigned long long b = ((unsigned long long)(~o) << 32) | ri;
        kmax = a > kmax ? a : kmax;
        kmin = b > kmin ? b : kmin;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long a = __shfl_down(kmax, off);
        const unsigned long long b = __shfl_down(kmin, off);
        kmax = a > kmax ? a : kmax;
        kmin = b > kmin ? b : kmin;
    }
    if ((threadIdx.x & 63) == 0) {
        if (kmax) atomicMax(&best[2 * t], kmax);
        if (kmin) atomicMax(&best[2 * t + 1], kmin);
    }
}

}  // namespace mtm
