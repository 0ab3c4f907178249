// Score-map kernels besides the matrix-core ones: slab combination, the naive cross-check, the tiled float64 kernel, the dot4 (VALU) kernel.  Launched by mtm_launch.hip only.
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>

#include "mtm_kernels.h"
#include "../../include/mtm_hip.h"
#include "mtm_device_util.hip.h"
#include "mtm_score_params.h"

namespace mtm {

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

__global__ __launch_bounds__(256) void slab_combine_kernel(SlabParams p, const TemplDev* __restrict__ td,
                                                           const int* __restrict__ tlist, StatPlanes st,
                                                           float* __restrict__ maps, int only_li) {
    const int li = blockIdx.z;
    if (only_li >= 0 && li != only_li) return;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const bool on = x < p.ow && y < p.oh;
    if (!on && !p.ext_on) return;
    const TemplDev T = td[tlist[li]];
    float out = NAN;
    if (on) {
        const size_t o = (size_t)li * p.raw_map + (size_t)y * p.pitch + x;
        long long a = 0;
        for (int k = 0; k < p.n_slabs; ++k) a += (long long)p.raw[(size_t)k * p.raw_slab + o];
        const size_t sidx = (size_t)y * st.pitch + x;
        double s1 = 0.0;
        for (int c = 0; c < p.chans; ++c) s1 += st.t[c][sidx];
        const double corr = ((double)a + 128.0 * s1) + T.mfma_k;
        out = finish_unmasked(p.method, corr, st, sidx, T, p.chans);
    }
    if (p.ext_on) {
        // cv2.minMaxLoc inside the combine pass: the key of extremum_kernel (NaN never wins, first occurrence in row-major
        // order wins ties), one atomic per wave
        unsigned long long key = 0ull;
        if (on && out == out) {
            const uint32_t ord = mf_float_order(out);
            key = ((unsigned long long)(p.cand_min ? ~ord : ord) << 32) | (0xFFFFFFFFu - (uint32_t)(y * p.ow + x));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long other = __shfl_down(key, off);
            key = other > key ? other : key;
        }
        if ((threadIdx.x & 63) == 0 && key) atomicMax(&p.ext_best[2 * tlist[li] + p.cand_min], key);
        return;
    }
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


}  // namespace mtm
