// Templates on the device: sources, augmentation, statistics and the A-operand packs of ncc_mfma_kernel.
//
// The reference's caller augments templates on the host and appends the copies to listTemplates
// (tutorials/Tutorial2-Template_Augmentation.ipynb:313: np.rot90; multi-scale copies for BASELINE configs[4]).
// Here a template as matched - a "unit" - is a VIEW of a source image kept in a device arena: a base template
// as handed over, or an area-resized copy a kernel made of it, read through one of the eight axis permutations /
// reflections (rot90 x flips).  The packs the score kernel consumes are gathered straight from those views, so
// rotated / flipped / resized copies never exist on the host and never cross PCIe.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "mtm_kernels.h"
#include "mtm_templates_params.h"

namespace mtm {


__device__ __forceinline__ unsigned unit_px(const uint8_t* __restrict__ arena, const UnitSrc& u, int c, int y, int x) {
    const int sy = u.ay * y + u.by * x + u.cy, sx = u.ax * y + u.bx * x + u.cx;
    return arena[u.off + ((long long)c * u.sh + sy) * u.sw + sx];
}
__device__ __forceinline__ unsigned unit_mask(const uint8_t* __restrict__ arena, const UnitSrc& u, int c, int y, int x) {
    const int sy = u.ay * y + u.by * x + u.cy, sx = u.ax * y + u.bx * x + u.cx;
    return arena[u.moff + ((long long)c * u.sh + sy) * u.sw + sx] > 0 ? 1u : 0u;     // CV_8U masks are binary masks
}

// ---------------------------------------------------------------------------------------------
// Area resize of a source plane set to dh x dw, exact rational arithmetic: output pixel (i, j) integrates the
// source over [i*sh/dh, (i+1)*sh/dh) x [j*sw/dw, (j+1)*sw/dw); overlap lengths are integers in units of
// 1/dh (1/dw) source pixels, the weighted sum is an integer < 2^63 and the result is
// (2*num + den) / (2*den) with den = sh*sw, i.e. round-half-up of the exact mean.  Same arithmetic as
// MTM.augment.resize_area (numpy, int64) - the two agree byte for byte.
// ---------------------------------------------------------------------------------------------
__global__ void resize_area_kernel(const uint8_t* __restrict__ src, int sh, int sw, uint8_t* __restrict__ dst, int dh,
                                   int dw, int planes) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, p = blockIdx.z;
    if (j >= dw || i >= dh || p >= planes) return;
    const uint8_t* s = src + (size_t)p * sh * sw;
    const long long ylo = (long long)i * sh, yhi = (long long)(i + 1) * sh;        // in units of 1/dh source rows
    const long long xlo = (long long)j * sw, xhi = (long long)(j + 1) * sw;        // in units of 1/dw source cols
    long long num = 0;
    for (long long y = ylo / dh; y < sh && y * dh < yhi; ++y) {
        const long long wy = min(yhi, (y + 1) * dh) - max(ylo, y * dh);
        if (wy <= 0) continue;
        long long row = 0;
        for (long long x = xlo / dw; x < sw && x * dw < xhi; ++x) {
            const long long wx = min(xhi, (x + 1) * dw) - max(xlo, x * dw);
            if (wx > 0) row += wx * (long long)s[y * sw + x];
        }
        num += wy * row;
    }
    const long long den = (long long)sh * sw;
    const long long v = (2 * num + den) / (2 * den);
    dst[(size_t)p * dh * dw + (size_t)i * dw + j] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// Integer-factor area downscale with OpenCV's INTER_AREA rounding for uint8 (MTM.augment.downscale,
// planarize_u8_down_kernel): factor 2 -> (sum + 2) >> 2, else rint((float)sum * (1.f / f^2)); remainder rows /
// columns dropped.
__global__ void downscale_int_kernel(const uint8_t* __restrict__ src, int sh, int sw, uint8_t* __restrict__ dst, int f,
                                     int planes) {
    const int dh = sh / f, dw = sw / f;
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, p = blockIdx.z;
    if (j >= dw || i >= dh || p >= planes) return;
    const uint8_t* s = src + (size_t)p * sh * sw;
    unsigned sum = 0;
    for (int dy = 0; dy < f; ++dy)
        for (int dx = 0; dx < f; ++dx) sum += s[(size_t)(i * f + dy) * sw + (size_t)j * f + dx];
    const float scale = 1.0f / (float)(f * f);
    const unsigned r = (f == 2) ? ((sum + 2u) >> 2) : (unsigned)rintf((float)sum * scale);
    dst[(size_t)p * dh * dw + (size_t)i * dw + j] = (uint8_t)(r > 255u ? 255u : r);
}


__global__ __launch_bounds__(256) void source_sums_kernel(const uint8_t* __restrict__ arena,
                                                          const SourceDesc* __restrict__ srcs,
                                                          unsigned long long* __restrict__ out) {
    const SourceDesc d = srcs[blockIdx.x];
    __shared__ unsigned long long red[4][kSumsPerSource];
    const long long n = (long long)d.sh * d.sw;
    unsigned long long acc[kSumsPerSource];
#pragma unroll
    for (int k = 0; k < kSumsPerSource; ++k) acc[k] = 0ull;
    for (int c = 0; c < d.chans; ++c)
        for (long long i = threadIdx.x; i < n; i += 256) {
            const unsigned long long v = arena[d.off + c * n + i];
            const unsigned long long m = d.moff >= 0 ? (arena[d.moff + c * n + i] > 0 ? 1ull : 0ull) : 0ull;
            acc[4 * c + 0] += v;
            acc[4 * c + 1] += v * v;
            acc[4 * c + 2] += v * m;
            acc[4 * c + 3] += v * v * m;            // (v*m)^2 = v^2 * m for m in {0, 1}
            if (c == 0) acc[4 * kMaxChans] += m;
        }
#pragma unroll
    for (int k = 0; k < kSumsPerSource; ++k) {
        unsigned long long v = acc[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < kSumsPerSource)
        out[(size_t)blockIdx.x * kSumsPerSource + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}


__global__ __launch_bounds__(256) void pack_units_kernel(PackParams p, const uint8_t* __restrict__ arena,
                                                         const UnitSrc* __restrict__ units, const int* __restrict__ tl,
                                                         uint8_t* __restrict__ out) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= p.n_chunks) return;
    const int lane = (int)(k & 63), i = lane & 15, q = lane >> 4;
    long long r = k >> 6;
    int li = -1, dy = -1, ch = 0;
    int b = 0, qtap = q;                 // 16-tap segment of the row: dx = 64 b + 16 qtap + byte
    if (p.nseg > 0) {
        const int bb = (int)(r % p.kblocks);
        r /= p.kblocks;
        const int sidx = 4 * bb + q, srow = sidx / p.nseg;
        qtap = sidx - srow * p.nseg;
        if (p.mode == 0) {
            dy = srow;
            ch = (int)(r % p.chans);
            li = 16 * (int)(r / p.chans) + i;
        } else {                         // row-multiplexed: [ch][MFMA group g][step]; stream row r: template row r - g R - rho
            const int g = (int)(r & 1);
            ch = (int)(r >> 1);
            const int t = i % p.nt, rho = i / p.nt;
            dy = srow - g * p.R - rho;
            li = t;
        }
    } else {
    b = (int)(r % p.nb);
    r /= p.nb;
    if (p.mode == 0) {
        dy = (int)(r % p.h);
        r /= p.h;
        ch = (int)(r % p.chans);
        const int g = (int)(r / p.chans);
        li = 16 * g + i;
    } else {
        const int steps = p.h + 3 * p.R - 1;
        const int sp = (int)(r % steps);
        ch = (int)(r / steps);
        const int t = p.mode == 2 ? 0 : i % p.nt, rho = p.mode == 2 ? i : i / p.nt;
        dy = sp - p.R - rho;
        li = t;
    }
    }
    uint32_t wds[4] = {0u, 0u, 0u, 0u};
    if (li >= 0 && li < p.n && dy >= 0 && dy < p.hv) {
        const UnitSrc u = units[tl[li]];
#pragma unroll
        for (int byte = 0; byte < 16; ++byte) {
            const int dx = 64 * b + 16 * qtap + byte;
            if (dx < p.w) {
                unsigned v;
                if (p.mode == 2) v = unit_mask(arena, u, 0, dy, dx);
                else {
                    v = unit_px(arena, u, ch, dy, dx);
                    if (p.masked) v *= unit_mask(arena, u, ch, dy, dx);
                }
                // (the mask of the sum I^2 M pass is its own int8 value, 0 or 1 - see masksq_combine_kernel; pixels are biased)
                wds[byte >> 2] |= ((p.mode == 2 ? v : (v ^ 0x80u)) & 255u) << (8 * (byte & 3));
            }
        }
    }
    *reinterpret_cast<uint4*>(out + k * 16) = make_uint4(wds[0], wds[1], wds[2], wds[3]);
}

// The pixels (and binary mask) of a unit as contiguous planar bytes: for the kernels that still take host-packed
// weights (float64 / dot4 fallbacks): the launcher copies them back and packs on the host.
__global__ void gather_unit_kernel(const uint8_t* __restrict__ arena, UnitSrc u, uint8_t* __restrict__ px,
                                   uint8_t* __restrict__ mask) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, c = blockIdx.z;
    if (x >= u.w || y >= u.h) return;
    const size_t o = ((size_t)c * u.h + y) * u.w + x;
    px[o] = (uint8_t)unit_px(arena, u, c, y, x);
    if (mask != nullptr) mask[o] = u.moff >= 0 ? (uint8_t)unit_mask(arena, u, c, y, x) : 0;
}

}  // namespace mtm
