// Peak extraction kernels: full 3x3 scan, verification of kernel candidates (against maps / by hash), global extremum.  Launched by mtm_api.hip only.
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>

#include "mtm_kernels.h"
#include "../../include/mtm_hip.h"
#include "mtm_device_util.hip.h"

namespace mtm {

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
        // rows are requested three ahead of their use (round 4: the loop used to load row y + 1 and consume it at once -
        // one exposed memory latency per row and wave; 0.38 ms for the 1 GB of maps of 4K x 32 templates)
        const float padr = mode_min ? -padv : padv;          // the pad value as it would sit in memory
        PeakRow r0 = peaks_fetch_row(m, T.map_pitch, T.oh, T.ow, y0 - 1, xb, lane, padr);
        PeakRow r1 = peaks_fetch_row(m, T.map_pitch, T.oh, T.ow, y0, xb, lane, padr);
        PeakRow r2 = peaks_fetch_row(m, T.map_pitch, T.oh, T.ow, y0 + 1, xb, lane, padr);
        PeakRow r3 = peaks_fetch_row(m, T.map_pitch, T.oh, T.ow, y0 + 2, xb, lane, padr);
        PeakRow r4 = peaks_fetch_row(m, T.map_pitch, T.oh, T.ow, y0 + 3, xb, lane, padr);
        peaks_finish_row(r0, lane, mode_min, va, hl, hr);
        hmax(va, hl, hr, hm_a);
        peaks_finish_row(r1, lane, mode_min, vb, hl, hr);
        hmax(vb, hl, hr, hm_b);
        const int y1 = min(y0 + kPkRows, T.oh);
        for (int y = y0; y < y1; ++y) {
            const PeakRow rn = peaks_fetch_row(m, T.map_pitch, T.oh, T.ow, y + 4, xb, lane, padr);
            peaks_finish_row(r2, lane, mode_min, vc, hl, hr);
            r2 = r3;
            r3 = r4;
            r4 = rn;
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

// The same scan over the FLAGGED row segments only (MfmaParams::seg_flags): the score kernel set a flag per row segment -
// the 256 outputs of one of its waves, which is one row of a strip here - in which some output passes the threshold; a
// row segment without the flag holds no peak (a peak is above the threshold).  A wave reads the 32 flags of its strip in
// one load, and for every flagged row fetches the three rows it needs (6.7 % of the rows on a photograph-like 4K image
// at threshold 0.5; the full scan reads 1 GB of maps for 4K x 32 templates).  The maps themselves are complete.
// nontrivial[t]: byte 0 as peaks_kernel, over the flagged rows; byte 1 = some segment flagged, byte 2 = some segment not
// flagged.  Both together mean a pixel above the threshold and one below it exist, i.e. the map is not constant; if every
// segment is flagged, byte 0 saw every pixel.  The host reads  nontrivial = byte 0 or (byte 1 and byte 2).
// Peaks go to a list per (template, strip column) - region blockIdx.z * gridDim.x + blockIdx.x of `hits_t`, `cap_t` records
// each, its own counter - staged per wave in LDS and appended with one atomic per 64 records; compact_hits_kernel then
// builds the one list the host reads (same-address atomics again: 480 counters share what one counter would queue up).
constexpr int kPkStage = 64;        // records staged per wave
// rows per wave: flagged rows cluster (bright regions), and a wave works through its flagged rows one memory latency after
// the other - with the 32-row strips of the full scan the pass took as long as the fully flagged strips did, four
// generations of them: 0.30 ms; the rows of the next flagged row are requested before the current one is evaluated
constexpr int kPkSparseRows = 8;
__global__ __launch_bounds__(256) void peaks_sparse_kernel(const float* __restrict__ maps, const TemplDev* __restrict__ td,
                                                           const int* __restrict__ tlist, int mode_min, float thr, int border,
                                                           mtm_hit* __restrict__ hits_t, unsigned long long cap_t,
                                                           unsigned long long* __restrict__ counts_t,
                                                           int* __restrict__ nontrivial, const uint8_t* __restrict__ seg_flags,
                                                           int flag_tstride, int flag_rstride, int holes) {
    __shared__ mtm_hit stage[4][kPkStage];
    const int t = tlist[blockIdx.z];
    const TemplDev T = td[t];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sx = blockIdx.x, xs = sx * kPkCols;
    const int y0 = (blockIdx.y * 4 + wave) * kPkSparseRows;
    if (!(xs < T.ow && y0 < T.oh)) return;               // (whole waves; no work-group barrier below)
    const float* m = maps + T.map_off;
    const int y1 = min(y0 + kPkSparseRows, T.oh);
    static_assert(kPkSparseRows <= 64, "one flag per lane");
    const bool mine = lane < y1 - y0 && seg_flags[(size_t)t * flag_tstride + (size_t)(y0 + lane) * flag_rstride + sx] != 0;
    unsigned long long todo = __builtin_amdgcn_ballot_w64(mine);
    const int n_rows = y1 - y0;
    int bits = (todo ? 2 : 0) | (__popcll(todo) < n_rows ? 4 : 0);
    const float padv = (border == MTM_BORDER_CONSTANT) ? 0.0f : -INFINITY;
    const float padr = mode_min ? -padv : padv;           // the pad value as it would sit in memory
    const float thr2 = mode_min ? -thr : thr;
    const int xb = xs + 4 * lane;
    const unsigned list = blockIdx.z * gridDim.x + blockIdx.x;
    mtm_hit* region = hits_t + (size_t)list * cap_t;
    mtm_hit* st = stage[wave];
    int n_st = 0;                                         // staged records (wave-uniform)
    auto flush = [&]() {
        if (n_st == 0) return;
        unsigned long long base = 0ull;
        if (lane == 0) base = atomicAdd(&counts_t[list], (unsigned long long)n_st);
        const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
        base = ((unsigned long long)bhi << 32) | blo;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (lane < n_st && base + (unsigned long long)lane < cap_t) region[base + (unsigned long long)lane] = st[lane];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // the records are read: their slots may be rewritten
        n_st = 0;
    };
    auto hmax = [](const float (&v)[4], float hl, float hr, float (&h)[4]) {
        h[0] = fmaxf(fmaxf(hl, v[0]), v[1]);
        h[1] = fmaxf(fmaxf(v[0], v[1]), v[2]);
        h[2] = fmaxf(fmaxf(v[1], v[2]), v[3]);
        h[3] = fmaxf(fmaxf(v[2], v[3]), hr);
    };
    int nontriv = 0;
    // holes (MfmaParams::seg_skip, round 5): only flagged segments - and segments in which some output came close to the
    // threshold - were written at all; what an unflagged segment holds is stale memory.  Its outputs are all at or below
    // the threshold, so as NEIGHBOURS of a flagged segment's pixels they count as "below": never read.
    const float below_raw = mode_min ? INFINITY : -INFINITY;
    const int n_sx = (T.ow + kPkCols - 1) / kPkCols;
    auto fetch = [&](int y) {
        if (!holes || y < 0 || y >= T.oh) return peaks_fetch_row(m, T.map_pitch, T.oh, T.ow, y, xb, lane, padr);
        const uint8_t* fr = seg_flags + (size_t)t * flag_tstride + (size_t)y * flag_rstride;
        PeakRow r;
        if (fr[sx] != 0) {
            r = peaks_fetch_row(m, T.map_pitch, T.oh, T.ow, y, xb, lane, padr);
        } else {
            r.q = make_float4(below_raw, below_raw, below_raw, below_raw);
            r.el = r.er = padr;
            // (columns beyond the map keep the pad value, as peaks_fetch_row leaves them)
            if (xb >= T.ow) r.q.x = padr;
            if (xb + 1 >= T.ow) r.q.y = padr;
            if (xb + 2 >= T.ow) r.q.z = padr;
            if (xb + 3 >= T.ow) r.q.w = padr;
            if (lane == 0 && xb - 1 >= 0) r.el = m[(size_t)y * T.map_pitch + xb - 1];
            if (lane == 63 && xb + 4 < T.ow) r.er = m[(size_t)y * T.map_pitch + xb + 4];
        }
        if (lane == 0 && sx > 0 && fr[sx - 1] == 0) r.el = below_raw;
        if (lane == 63 && sx + 1 < n_sx && fr[sx + 1] == 0 && xb + 4 < T.ow) r.er = below_raw;
        return r;
    };
    PeakRow na{}, nb{}, nc{};                             // the rows of the next flagged row, requested one iteration ahead
    int y_next = -1;
    if (todo) {
        y_next = y0 + (int)__builtin_ctzll(todo);
        todo &= todo - 1;
        na = fetch(y_next - 1);
        nb = fetch(y_next);
        nc = fetch(y_next + 1);
    }
    while (y_next >= 0) {                                 // wave-uniform
        const int y = y_next;
        const PeakRow ra = na, rb = nb, rc = nc;
        y_next = -1;
        if (todo) {
            y_next = y0 + (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            na = fetch(y_next - 1);
            nb = fetch(y_next);
            nc = fetch(y_next + 1);
        }
        float va[4], vb[4], vc[4], hl, hr, hm_a[4], hm_b[4], hm_c[4];
        peaks_finish_row(ra, lane, mode_min != 0, va, hl, hr);
        hmax(va, hl, hr, hm_a);
        peaks_finish_row(rb, lane, mode_min != 0, vb, hl, hr);
        hmax(vb, hl, hr, hm_b);
        peaks_finish_row(rc, lane, mode_min != 0, vc, hl, hr);
        hmax(vc, hl, hr, hm_c);
        unsigned pk = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (xb + k < T.ow) {
                const float v = vb[k];
                const float mx = fmaxf(fmaxf(hm_a[k], hm_b[k]), hm_c[k]);
                if (!(v == mx)) nontriv = 1;
                else if (v > thr2) pk |= 1u << k;
            }
        }
        const unsigned long long b0 = __builtin_amdgcn_ballot_w64((pk & 1u) != 0), b1 = __builtin_amdgcn_ballot_w64((pk & 2u) != 0);
        const unsigned long long b2 = __builtin_amdgcn_ballot_w64((pk & 4u) != 0), b3 = __builtin_amdgcn_ballot_w64((pk & 8u) != 0);
        const int n0 = __popcll(b0), n1 = __popcll(b1), n2 = __popcll(b2), n3 = __popcll(b3);
        const int total = n0 + n1 + n2 + n3;
        if (total == 0) continue;
        if (n_st + total > kPkStage) flush();
        const unsigned long long bb[4] = {b0, b1, b2, b3};
        const int pre[4] = {0, n0, n0 + n1, n0 + n1 + n2};
        unsigned long long gbase = 0ull;                  // a row with more peaks than the buffer holds: straight to the region
        const bool direct = total > kPkStage;
        if (direct) {
            if (lane == 0) gbase = atomicAdd(&counts_t[list], (unsigned long long)total);
            const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)gbase), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(gbase >> 32));
            gbase = ((unsigned long long)bhi << 32) | blo;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if ((pk >> k) & 1u) {
                const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bb[k] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bb[k], 0u));
                mtm_hit hrec;
                hrec.templ_idx = t;
                hrec.x = xb + k;
                hrec.y = y;
                hrec.w = T.cols;
                hrec.h = T.rows;
                hrec.score = mode_min ? -vb[k] : vb[k];
                if (direct) {
                    const unsigned long long slot = gbase + (unsigned long long)(pre[k] + below);
                    if (slot < cap_t) region[slot] = hrec;
                } else {
                    st[n_st + pre[k] + below] = hrec;
                }
            }
        if (!direct) n_st += total;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    flush();
    if (__builtin_amdgcn_ballot_w64(nontriv != 0) != 0ull) bits |= 1;
    // plain byte stores (one byte of nontrivial[t] per bit), not atomicOr: a returning atomic on an address every wave of the
    // launch goes to costs ~0.25 us a piece on this chip - they queue up at the memory side - and made the pass as slow as
    // the number of its waves (0.30 ms with 32 k waves, 1.1 ms with 126 k)
    if (lane == 0) {
        uint8_t* nb8 = reinterpret_cast<uint8_t*>(&nontrivial[t]);
        if (bits & 1) nb8[0] = 1;
        if (bits & 2) nb8[1] = 1;
        if (bits & 4) nb8[2] = 1;
    }
}

// The per-template lists of peaks_sparse_kernel -> the one list (and count) the host fetches.  Block z copies region z behind
// the regions before it; a region that overflowed makes the reported count exceed `hit_cap` (the host grows the lists and
// repeats the pass, as it does for the single list).
__global__ __launch_bounds__(256) void compact_hits_kernel(const mtm_hit* __restrict__ hits_t, unsigned long long cap_t,
                                                           const unsigned long long* __restrict__ counts_t, int n_lists,
                                                           mtm_hit* __restrict__ hits, unsigned long long hit_cap,
                                                           unsigned long long* __restrict__ counter) {
    const int z = blockIdx.x;
    // (every work-group sums the counters itself, 256 at a time: hundreds of lists, one dependent load each otherwise)
    __shared__ unsigned long long s_off[256], s_tot[256];
    __shared__ int s_over[256];
    unsigned long long off = 0ull, total = 0ull;
    int over = 0;
    for (int i = threadIdx.x; i < n_lists; i += 256) {
        const unsigned long long ci = counts_t[i];
        if (i < z) off += min(ci, cap_t);
        // (a list that overflowed counts 8 times - the lists are an eighth of the capacity the host then grows)
        total += ci > cap_t ? 8ull * ci : ci;
        over |= ci > cap_t ? 1 : 0;
    }
    s_off[threadIdx.x] = off;
    s_tot[threadIdx.x] = total;
    s_over[threadIdx.x] = over;
    __syncthreads();
    for (int step = 128; step > 0; step >>= 1) {
        if ((int)threadIdx.x < step) {
            s_off[threadIdx.x] += s_off[threadIdx.x + step];
            s_tot[threadIdx.x] += s_tot[threadIdx.x + step];
            s_over[threadIdx.x] |= s_over[threadIdx.x + step];
        }
        __syncthreads();
    }
    off = s_off[0];
    total = s_tot[0];
    over = s_over[0];
    if (z == 0 && threadIdx.x == 0) counter[0] = over ? max(total, hit_cap + 1ull) : total;
    const unsigned long long n = min(counts_t[z], cap_t);
    const mtm_hit* src = hits_t + (size_t)z * cap_t;
    for (unsigned long long r = threadIdx.x; r < n; r += blockDim.x)
        if (off + r < hit_cap) hits[off + r] = src[r];
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


// The candidate list of a hits-only call goes to the host's page-locked landing buffer from a kernel (one small
// work-group behind the score launch) instead of a device-to-host copy command: a queued kernel starts the moment its
// predecessor retires, a copy command 15 us later (rocprofv3 timelines, profiles/r03_tl2) - and only the records that
// exist cross PCIe.  src: [count (8 B), clock (8 B)][records]; dst: mapped host memory, same layout.
__global__ __launch_bounds__(256) void fetch_cands_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                          unsigned long long max_records) {
    const unsigned long long count = *reinterpret_cast<const unsigned long long*>(src);
    const unsigned long long nrec = count < max_records ? count : max_records;
    const unsigned n16 = (unsigned)((16ull + sizeof(mtm_hit) * nrec + 15ull) / 16ull);
    for (unsigned i = threadIdx.x; i < n16; i += 256) dst[i] = src[i];
    __threadfence_system();
}

}  // namespace mtm
