// Image layout kernels: interleaved rows -> planar, padded device planes (uint8 + its int8 view, float32; downscaling, uint16 byte planes).  Launched by mtm_context.hip only.
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


}  // namespace mtm
