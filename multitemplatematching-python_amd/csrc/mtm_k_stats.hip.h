// Window statistics kernels (exact box sums and the template-independent part of the normalisation per output pixel) and the sum I^2 M helpers of masked classes.  Launched by mtm_launch.hip only.
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>

#include "mtm_kernels.h"
#include "../../include/mtm_hip.h"
#include "mtm_device_util.hip.h"

namespace mtm {

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

// hsum_kernel<double> with the row segment staged through LDS (round 6).  The generic kernel's threads start 16 floats = 64 B
// apart, so every one of its loads touches 64 cache lines: 270 us for a 4K float32 plane, next to a 2 ms score kernel once the
// float32 screen runs one piece product.  Here the work-group copies its stretch of the row into LDS with coalesced loads
// and every thread then performs EXACTLY the generic kernel's operations in the generic kernel's order on it (the float64
// sums depend on the order - the statistics planes are bit for bit what they were).  LDS index i -> i + i / 16: the threads'
// stride of 16 floats would put all lanes of a wave on four banks.  Same grid as hsum_kernel (blockDim 256); needs
// 4 * (256 * kHsumSeg + w + its padding) bytes of dynamic LDS; the launcher keeps the generic kernel for w > 1024.
__host__ __device__ __forceinline__ int hsum_lds_idx(int i) { return i + (i >> 4); }
inline size_t hsum_lds_bytes(int w) { return sizeof(float) * (size_t)(hsum_lds_idx(256 * kHsumSeg + w) + 2); }

__global__ __launch_bounds__(256) void hsum_lds_kernel(const float* __restrict__ img, int pitch, long long plane, int rows,
                                                       int w, int ow, double* __restrict__ hs1, double* __restrict__ hs2,
                                                       int hs_pitch, long long hs_plane, int y_off) {
    extern __shared__ float hsum_row[];
    const int xb = blockIdx.x * 256 * kHsumSeg;           // first output column of the work-group
    const int y = (int)blockIdx.y + y_off, c = blockIdx.z;      // y_off: banded uploads - the rows that have just arrived
    if (y >= rows) return;
    const float* row = img + c * plane + (size_t)y * pitch + xb;
    // outputs xb .. xb + n_out - 1 read row elements 0 .. n_out + w - 1 (the padded image keeps x + w readable, as in the
    // generic kernel); clamp to the row pitch all the same
    const int n_out = min(256 * kHsumSeg, ow - xb);
    const int n_in = min(n_out + w, pitch - xb);
    for (int i = threadIdx.x; i < n_in; i += 256) hsum_row[hsum_lds_idx(i)] = row[i];
    __syncthreads();
    const int x0 = threadIdx.x * kHsumSeg;
    if (x0 >= n_out) return;
    double s1 = 0, s2 = 0;
    for (int dx = 0; dx < w; ++dx) {
        const double v = (double)hsum_row[hsum_lds_idx(x0 + dx)];
        s1 += v;
        s2 += v * v;
    }
    double* o1 = hs1 + c * hs_plane + (size_t)y * hs_pitch + xb;
    double* o2 = hs2 + c * hs_plane + (size_t)y * hs_pitch + xb;
    for (int k = 0; k < kHsumSeg; ++k) {
        const int x = x0 + k;
        if (x >= n_out) break;
        o1[x] = s1;
        o2[x] = s2;
        const double vn = (double)hsum_row[hsum_lds_idx(x + w)];
        const double vo = (double)hsum_row[hsum_lds_idx(x)];
        s1 += vn - vo;
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
#ifndef MTM_STAT_PRO_BATCH
#define MTM_STAT_PRO_BATCH 32
#endif
constexpr int kStatProBatch = MTM_STAT_PRO_BATCH;   // stats_u8_kernel: image rows of the column-sum prologue requested together

// output columns per work-group for a template width (multiple of 16: strips start dword-aligned, and the 16-pixel
// column blocks whose statistic ranges the kernel can write - `blk` - never straddle two strips)
inline int stats_u8_owg(int w) { return (kStatStrip + 1 - w) & ~15; }

// layout conversion fused into a banded statistics launch (stats_u8_kernel); u8b == nullptr: none
struct StatLayout {
    uint8_t* u8 = nullptr;
    uint8_t* u8b = nullptr;
    int pitch = 0;          // of the planes
    int r0 = 0, r1 = 0;     // image rows to convert
    unsigned long long* zero16 = nullptr;   // non-null: 16 bytes the launch clears (the candidate list's header: the first
                                            // kernel of a banded call does it instead of a fill command between two calls)
};

__global__ __launch_bounds__(256) void stats_u8_kernel(const uint8_t* __restrict__ img, int pitch, int h, int w,
                                                       int oh, int ow, int owg, double inv_area, int num_type,
                                                       int want_sq, int want_t, int want_sum2, double* __restrict__ t0,
                                                       double* __restrict__ sum2, double* __restrict__ sq,
                                                       int st_pitch, double* __restrict__ rsq = nullptr,
                                                       int yb_off = 0, double* __restrict__ blk = nullptr,
                                                       int blk_pitch = 0, StatLayout lay = StatLayout{}) {
    __shared__ __attribute__((aligned(16))) uint32_t E1[kStatStrip + 4], E2[kStatStrip + 4];   // exclusive prefixes
    __shared__ uint32_t wsum[2][4];
    const int x0 = blockIdx.x * owg, y0 = ((int)blockIdx.y + yb_off) * kStatBand4;   // yb_off: banded launches
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (lay.zero16 != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && t == 0) {
        lay.zero16[0] = 0ull;
        lay.zero16[1] = 0ull;
    }
    if (lay.u8b != nullptr) {
        // Round 5: the layout conversion of a band's rows rides on its statistics launch (`img` is then the RAW upload buffer,
        // `pitch` its row length - a multiple of 4).  The band's image rows lay.r0 .. lay.r1 - 1 are split evenly over the
        // launch's row blocks, the columns over its strips (the last strip takes the rest of the row): dword in, the
        // padded uint8 plane and its int8 view (byte ^ 0x80, what the LDS-DMA of the score kernel reads) out.
        const int nby = (int)gridDim.y, q = (lay.r1 - lay.r0 + nby - 1) / nby;
        const int ra = lay.r0 + (int)blockIdx.y * q, rb = min(lay.r1, ra + q);
        const int xe = blockIdx.x + 1 == gridDim.x ? pitch : min(pitch, x0 + owg);
        const int x = x0 + 4 * t;
        if (x < xe) {
            for (int r = ra; r < rb; r += 8) {
                uint32_t v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    v[i] = r + i < rb ? *reinterpret_cast<const uint32_t*>(img + (size_t)(r + i) * pitch + x) : 0u;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (r + i < rb) {
                        const size_t o = (size_t)(r + i) * lay.pitch + x;
                        *reinterpret_cast<uint32_t*>(lay.u8 + o) = v[i];
                        *reinterpret_cast<uint32_t*>(lay.u8b + o) = v[i] ^ 0x80808080u;
                    }
            }
        }
    }
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
    // kStatProBatch rows per batch: the loads of a batch are all in flight before the first add needs one.  (Round 5: 32
    // instead of 8 - the kernel is a chain of memory latencies, a 64-row window was eight of them before the first output row.)
    for (int r0 = 0; r0 < h; r0 += kStatProBatch) {
        uint32_t v[kStatProBatch];
#pragma unroll
        for (int i = 0; i < kStatProBatch; ++i)
            v[i] = ld ? *reinterpret_cast<const uint32_t*>(base + (size_t)min(r0 + i, h - 1) * pitch) : 0u;
#pragma unroll
        for (int i = 0; i < kStatProBatch; ++i)
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
                sqv[k] = small ? 0.0 : sqrt(diff2);
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

// The same for single-channel uint16 images, read as the two biased byte planes the MFMA kernel takes (hib, lob:
// byte ^ 0x80).  Window sums of I fit uint32 when w * h * 65535 < 2^32 (the launcher's condition; differences modulo
// 2^32 as above); the squares need uint64 prefixes - one more wave scan on the high halves' carries is avoided by
// scanning the 64-bit values with shuffles.  Operation order of the float64 statistics as in vsum_stats_kernel.
__device__ __forceinline__ unsigned long long wave_inclusive_scan_u64(unsigned long long x, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long y = __shfl_up(x, off);
        if (lane >= off) x += y;
    }
    return x;
}

__global__ __launch_bounds__(256) void stats_u16_kernel(const uint8_t* __restrict__ hib, const uint8_t* __restrict__ lob,
                                                        int pitch, int h, int w, int oh, int ow, int owg, double inv_area,
                                                        int num_type, int want_sq, int want_t, int want_sum2,
                                                        double* __restrict__ t0, double* __restrict__ sum2,
                                                        double* __restrict__ sq, int st_pitch,
                                                        double* __restrict__ blk = nullptr, int blk_pitch = 0, int yb_off = 0) {
    __shared__ __attribute__((aligned(16))) uint32_t E1[kStatStrip + 4];
    __shared__ __attribute__((aligned(16))) unsigned long long E2[kStatStrip + 4];
    __shared__ uint32_t wsum1[4];
    __shared__ unsigned long long wsum2[4];
    const int x0 = blockIdx.x * owg, y0 = ((int)blockIdx.y + yb_off) * kStatBand4;   // yb_off: banded launches
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int L = owg + w - 1;
    const bool ld = 4 * t < L && x0 + 4 * t + 3 < pitch;
    const size_t base = (size_t)y0 * pitch + x0 + 4 * t;
    uint32_t c1[4] = {0, 0, 0, 0};
    unsigned long long c2[4] = {0, 0, 0, 0};
    auto px = [](uint32_t vh, uint32_t vl, uint32_t (&b)[4]) {      // four pixels from the two biased byte quads
        vh ^= 0x80808080u;
        vl ^= 0x80808080u;
#pragma unroll
        for (int k = 0; k < 4; ++k) b[k] = (((vh >> (8 * k)) & 255u) << 8) | ((vl >> (8 * k)) & 255u);
    };
    const uint32_t zero = 0x80808080u;                              // biased zero
    for (int r0 = 0; r0 < h; r0 += 4) {
        uint32_t vh[4], vl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t o = base + (size_t)min(r0 + i, h - 1) * pitch;
            vh[i] = ld ? *reinterpret_cast<const uint32_t*>(hib + o) : zero;
            vl[i] = ld ? *reinterpret_cast<const uint32_t*>(lob + o) : zero;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (r0 + i < h) {
                uint32_t b[4];
                px(vh[i], vl[i], b);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    c1[k] += b[k];
                    c2[k] += (unsigned long long)(b[k] * b[k]);
                }
            }
    }
    const int y1 = min(y0 + kStatBand4, oh);
    const int xg = x0 + 4 * t;
    const bool out_on = 4 * t < owg && xg < st_pitch;
    for (int y = y0; y < y1; ++y) {
        uint32_t nh = zero, nl = zero, oh_ = zero, ol = zero;
        if (y + 1 < y1 && ld) {
            const size_t on = base + (size_t)(y - y0 + h) * pitch, oo = base + (size_t)(y - y0) * pitch;
            nh = *reinterpret_cast<const uint32_t*>(hib + on);
            nl = *reinterpret_cast<const uint32_t*>(lob + on);
            oh_ = *reinterpret_cast<const uint32_t*>(hib + oo);
            ol = *reinterpret_cast<const uint32_t*>(lob + oo);
        }
        const uint32_t a = c1[0] + c1[1] + c1[2] + c1[3];
        const unsigned long long b = c2[0] + c2[1] + c2[2] + c2[3];
        const uint32_t sa = wave_inclusive_scan_u32(a);
        const unsigned long long sb = wave_inclusive_scan_u64(b, lane);
        if (lane == 63) {
            wsum1[wave] = sa;
            wsum2[wave] = sb;
        }
        __syncthreads();
        uint32_t oa = sa - a;
        unsigned long long ob = sb - b;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < wave) {
                oa += wsum1[k];
                ob += wsum2[k];
            }
        const uint32_t e1[4] = {oa, oa + c1[0], oa + c1[0] + c1[1], oa + c1[0] + c1[1] + c1[2]};
        const unsigned long long e2[4] = {ob, ob + c2[0], ob + c2[0] + c2[1], ob + c2[0] + c2[1] + c2[2]};
        *reinterpret_cast<uint4*>(&E1[4 * t]) = make_uint4(e1[0], e1[1], e1[2], e1[3]);
#pragma unroll
        for (int k = 0; k < 4; ++k) E2[4 * t + k] = e2[k];
        if (t == 255) {
            E1[kStatStrip] = oa + a;
            E2[kStatStrip] = ob + b;
        }
        __syncthreads();
        double blk_s1[4] = {0.0, 0.0, 0.0, 0.0}, blk_sq[4] = {0.0, 0.0, 0.0, 0.0};
        if (out_on) {
            double tt[4], ws2[4], sqv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t s1 = E1[4 * t + k + w] - e1[k];
                const unsigned long long s2 = E2[4 * t + k + w] - e2[k];
                tt[k] = (double)s1;
                ws2[k] = (double)s2;
                double wnd_mean2 = 0.0;
                if (num_type == 1) wnd_mean2 = (tt[k] * tt[k]) * inv_area;
                const double diff2 = fmax(ws2[k] - wnd_mean2, 0.0);
                const bool small = diff2 <= fmin(0.5, (10.0 * (double)FLT_EPSILON) * ws2[k]);
                sqv[k] = small ? 0.0 : sqrt(diff2);
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
            }
        }
        if (blk != nullptr) {            // ranges over 16-pixel column blocks, as stats_u8_kernel writes them
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
        uint32_t bn[4], bo[4];
        px(nh, nl, bn);
        px(oh_, ol, bo);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            c1[k] += bn[k] - bo[k];
            c2[k] += (unsigned long long)(bn[k] * bn[k]) - (unsigned long long)(bo[k] * bo[k]);
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
                                  double* __restrict__ sum2, double* __restrict__ sq, int pitch, int yb_off) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y0 = ((int)blockIdx.y + yb_off) * kVsumBand;      // yb_off: banded uploads (a band of kVsumBand rows restarts its column sums: whole bands only)
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
            // 8 rows per batch, all sixteen loads in flight before the first add (same order of additions: a tall window
            // - 400 rows in the reference's benchmark shape - is a chain of 400 dependent memory latencies otherwise)
            for (int d0 = 0; d0 < h; d0 += 8) {
                AccT v1[8], v2[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const size_t o = (size_t)min(d0 + i, h - 1) * hs_pitch;
                    v1[i] = p1[o];
                    v2[i] = p2[o];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (d0 + i < h) {
                        a += (SumT)v1[i];
                        b += (SumT)v2[i];
                    }
            }
            s1[c] = a;
            s2[c] = b;
        }
    }
    const int y1 = min(y0 + kVsumBand, oh);
    for (int y = y0; y < y1; ++y) {
        // the four values of the slide at the end of this iteration are requested now: their latency hides behind the
        // float64 statistics and the stores
        AccT n1[kMaxChans], o1[kMaxChans], n2[kMaxChans], o2[kMaxChans];
#pragma unroll
        for (int c = 0; c < kMaxChans; ++c) {
            n1[c] = o1[c] = n2[c] = o2[c] = 0;
            if (c < chans && y + 1 < y1) {
                const size_t o = c * hs_plane + x;
                n1[c] = hs1[o + (size_t)(y + h) * hs_pitch];
                o1[c] = hs1[o + (size_t)y * hs_pitch];
                n2[c] = hs2[o + (size_t)(y + h) * hs_pitch];
                o2[c] = hs2[o + (size_t)y * hs_pitch];
            }
        }
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
                    s1[c] += (SumT)n1[c] - (SumT)o1[c];
                    s2[c] += (SumT)n2[c] - (SumT)o2[c];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Masked templates: sum I^2 * M over every window on the matrix cores.  I^2 is a 16-bit number; its two
// bytes are image planes of their own (square_planes_kernel, biased to int8 like every image plane), the binary
// mask is the "template" of a row-multiplexed raw correlation (one template, 16 output rows per MFMA) - as the
// int8 values 0 / 1 it already is, NOT biased: with a_x = sum (I_x - 128) * M the window sums drop out,
//   sum I_x * M = a_x + 128 * sum(M),        c2 = 256 * sum I_h * M + sum I_l * M = 256 a_h + a_l + 257 * 128 * sum(M)
// (round 2 biased the mask too and needed a window-sum pass over the high-byte plane per class to undo it).
// All integers < 2^53: exact.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void square_planes_kernel(const uint8_t* __restrict__ u8, size_t n16,
                                                            uint8_t* __restrict__ shb, uint8_t* __restrict__ slb) {
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
    reinterpret_cast<uint4*>(shb)[g] = make_uint4(hi[0] ^ 0x80808080u, hi[1] ^ 0x80808080u, hi[2] ^ 0x80808080u,
                                                  hi[3] ^ 0x80808080u);
    reinterpret_cast<uint4*>(slb)[g] = make_uint4(lo[0] ^ 0x80808080u, lo[1] ^ 0x80808080u, lo[2] ^ 0x80808080u,
                                                  lo[3] ^ 0x80808080u);
}

__global__ __launch_bounds__(256) void masksq_combine_kernel(const int* __restrict__ raw_h, const int* __restrict__ raw_l,
                                                             int raw_pitch, double* __restrict__ sum2, int st_pitch,
                                                             double km257, int oh, int ow, double* __restrict__ blk,
                                                             int blk_pitch) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    double c2 = INFINITY;
    if (x < ow && y < oh) {
        const size_t o = (size_t)y * st_pitch + x, r = (size_t)y * raw_pitch + x;
        c2 = fma(256.0, (double)raw_h[r], (double)raw_l[r]) + km257;
        sum2[o] = c2;
    }
    if (blk != nullptr) {
        // smallest c2 of every 16-pixel column block (third slot of the block record; the row-multiplexed tiling's
        // hits-only screen bounds sqrt(tms c2) from below with it); +inf for a block right of the last output column
#pragma unroll
        for (int off = 1; off <= 8; off <<= 1) c2 = fmin(c2, __shfl_xor(c2, off));
        if ((threadIdx.x & 15) == 0 && y < oh && (x >> 4) < blk_pitch) blk[((size_t)y * blk_pitch + (x >> 4)) * 4 + 2] = c2;
    }
}


}  // namespace mtm
