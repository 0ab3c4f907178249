// Non-maxima suppression, the device's share (reference MTM/NMS.py:53-84: cv2.dnn.NMSBoxes over the hits of all templates),
// for calls whose peak pass leaves thousands of hits on the device (dense images).  Greedy NMS keeps a hit iff no EARLIER
// hit (mtm_nms_core.h: nms_earlier) that is itself kept overlaps it by more than the threshold.  Two facts need no
// sequential pass: a hit that no earlier hit overlaps at all ("champion": the best of its neighbourhood) is kept, and a
// hit that a champion overlaps is suppressed - and a suppressed hit never suppresses anything, so the host's greedy pass
// over the list WITHOUT those gives the same result as over the whole list.  On a photograph-like image most hits sit in
// clusters around a champion: of 16,773 peaks a few thousand cross PCIe and reach the host's sort and NMS.
// Hits only interact within a box side of each other: every hit looks at the hits of the 3 x 3 grid cells around its
// corner (cell = the largest box side), the hits sorted by cell.  (A first version decided everything on the device, every thread waiting for the
// earlier hits it depends on: correct, but in dense fields a hit has hundreds of earlier neighbours and the waiting
// scans took 1 ms - twice the host's pass.)  Launched by mtm_api.hip only.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "mtm_nms_core.h"

namespace mtm {

struct NmsParams {
    const mtm_hit* hits;     // the peak list (device), any order
    // its length lives on the device (the peak pass's counter): the launches are queued right behind that pass, without a
    // round trip to the host in between (118 us of idle GPU in the timeline of a dense call), and do nothing unless
    // n_min <= *n_ptr <= n_max
    const unsigned long long* n_ptr;
    unsigned n_min, n_max;
    int ascending;           // difference methods: scores become 1 - score
    float thr_score;         // NMSBoxes score_threshold (already transformed)
    float thr_overlap;       // NMSBoxes nms_threshold (>= 0)
    int cell, gw, gh;        // grid: cell side, cells per row / column (one empty ring included)
    // the candidates sorted by grid cell (a counting sort: nms_count_kernel, nms_offsets_kernel, nms_scatter_kernel) - a
    // cell's hits are one contiguous run, read with independent loads (linked lists cost a memory latency per element)
    unsigned* cell_cnt;      // gw * gh + 1: hits per cell, then - in place - the exclusive prefix (cell_cnt[c] .. cell_cnt[c + 1])
    unsigned* rank;          // n: position inside its cell (0xFFFFFFFF: not a candidate)
    mtm_hit* sorted;         // n_cand records, cell by cell
    int* status;             // per sorted position: 0 undecided, 1 champion (kept)
    mtm_hit* out;            // n slots: champions from the front, the hits the host still has to decide about from the back
    unsigned long long* out_count;     // [champions, undecided]
};

constexpr int kNmsUndecided = 0, kNmsKept = 1;

__device__ __forceinline__ unsigned nms_n(const NmsParams& p) {
    const unsigned long long c = *p.n_ptr;
    return (c < (unsigned long long)p.n_min || c > (unsigned long long)p.n_max) ? 0u : (unsigned)c;
}

__device__ __forceinline__ int nms_cell_of(const NmsParams& p, const mtm_hit& h) {
    const int cx = min(max(h.x / p.cell, 0), p.gw - 3) + 1, cy = min(max(h.y / p.cell, 0), p.gh - 3) + 1;
    return cy * p.gw + cx;
}

__global__ __launch_bounds__(256) void nms_count_kernel(NmsParams p) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nms_n(p)) return;
    const mtm_hit h = p.hits[i];
    const bool cand = nms_score(h, p.ascending) > p.thr_score;          // (false for NaN)
    p.rank[i] = cand ? atomicAdd(&p.cell_cnt[nms_cell_of(p, h)], 1u) : 0xFFFFFFFFu;
}

// exclusive prefix over the cells, in place (one work-group; cell_cnt[n_cells] becomes the number of candidates)
__global__ __launch_bounds__(1024) void nms_offsets_kernel(NmsParams p) {
    __shared__ unsigned part[1024];
    const int n_cells = p.gw * p.gh, t = threadIdx.x;
    const int per = (n_cells + 1023) / 1024, c0 = t * per, c1 = min(c0 + per, n_cells);
    unsigned s = 0;
    for (int c = c0; c < c1; ++c) s += p.cell_cnt[c];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {           // inclusive scan of the per-thread sums
        const unsigned v = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    unsigned run = t ? part[t - 1] : 0u;
    for (int c = c0; c < c1; ++c) {
        const unsigned v = p.cell_cnt[c];
        p.cell_cnt[c] = run;
        run += v;
    }
    if (t == 1023) p.cell_cnt[n_cells] = part[1023];
}

__global__ __launch_bounds__(256) void nms_scatter_kernel(NmsParams p) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nms_n(p)) return;
    const unsigned r = p.rank[i];
    if (r == 0xFFFFFFFFu) return;
    const mtm_hit h = p.hits[i];
    const unsigned pos = p.cell_cnt[nms_cell_of(p, h)] + r;
    p.sorted[pos] = h;
    p.status[pos] = kNmsUndecided;
}

// Both neighbourhood passes below: EIGHT lanes per candidate, each looking at every eighth hit of the 3 x 3 cells (round 6).
// With a thread per candidate the passes were chains of ~75 dependent-latency loop turns in dense fields (89 + 42 us for
// 16,773 hits on an otherwise idle chip); "is there an earlier hit that overlaps" / "is there a champion that overlaps" do not
// depend on the order the neighbours are looked at.  A wave owns eight candidates per turn of a wave-uniform loop.
constexpr int kNmsSub = 8;

// champions: candidates that no earlier candidate overlaps by more than the threshold
__global__ __launch_bounds__(256) void nms_champion_kernel(NmsParams p) {
    const unsigned n_cand = p.cell_cnt[p.gw * p.gh];
    const unsigned gid = blockIdx.x * 256 + threadIdx.x, n_waves = gridDim.x * 4;
    const int lane = threadIdx.x & 63, sub = lane & (kNmsSub - 1), grp = lane & ~(kNmsSub - 1);
    for (unsigned first = (gid >> 6) * (64 / kNmsSub); first < n_cand; first += n_waves * (64 / kNmsSub)) {     // wave-uniform
        const unsigned i = first + (unsigned)(lane / kNmsSub);
        const bool valid = i < n_cand;
        bool beaten = false;
        if (valid) {
            const mtm_hit a = p.sorted[i];
            const int c = nms_cell_of(p, a);
            for (int dy = -1; dy <= 1 && !beaten; ++dy) {
                // the three cells of a grid row are neighbours in the sorted list too: one run
                const unsigned j0 = p.cell_cnt[c + dy * p.gw - 1], j1 = p.cell_cnt[c + dy * p.gw + 2];
                for (unsigned j = j0 + (unsigned)sub; j < j1; j += kNmsSub) {
                    const mtm_hit b = p.sorted[j];
                    if (b.x >= a.x + a.w || a.x >= b.x + b.w || b.y >= a.y + a.h || a.y >= b.y + b.h) continue;     // disjoint
                    if (j == i) continue;
                    if (!nms_earlier(b, a, p.ascending)) continue;
                    if (nms_rect_overlap(a, b) <= p.thr_overlap) continue;
                    beaten = true;
                    break;
                }
            }
        }
        const unsigned long long any = (__builtin_amdgcn_ballot_w64(beaten) >> grp) & ((1ull << kNmsSub) - 1ull);
        if (valid && sub == 0 && any == 0ull) p.status[i] = kNmsKept;
    }
}

// candidates a champion overlaps are out; everything else (champions and undecided hits) goes to the host's list.  A
// work-group owns 32 candidates per turn (four waves of eight, eight lanes each) and reserves their slots in `out` with ONE
// atomic per side: the waves' counts meet in LDS (a reservation per wave was 4200 same-address atomics for 16,773 hits -
// 14 us of the pass).
__global__ __launch_bounds__(256) void nms_prune_kernel(NmsParams p) {
    __shared__ unsigned s_cnt[2][4];
    __shared__ unsigned long long s_base[2];
    const unsigned n_cand = p.cell_cnt[p.gw * p.gh];
    const unsigned n = nms_n(p);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane & (kNmsSub - 1), grp = lane & ~(kNmsSub - 1);
    constexpr unsigned kPerWave = 64 / kNmsSub, kPerBlock = 4 * kPerWave;
    for (unsigned first = blockIdx.x * kPerBlock; first < n_cand; first += gridDim.x * kPerBlock) {             // block-uniform
        const unsigned i = first + (unsigned)wave * kPerWave + (unsigned)(lane / kNmsSub);
        const bool valid = i < n_cand;
        mtm_hit a{};
        bool champ = false, out = false;
        if (valid) {
            a = p.sorted[i];
            champ = p.status[i] == kNmsKept;
            if (!champ) {
                const int c = nms_cell_of(p, a);
                for (int dy = -1; dy <= 1 && !out; ++dy) {
                    const unsigned j0 = p.cell_cnt[c + dy * p.gw - 1], j1 = p.cell_cnt[c + dy * p.gw + 2];
                    for (unsigned j = j0 + (unsigned)sub; j < j1; j += kNmsSub) {
                        if (p.status[j] != kNmsKept) continue;               // champions only (they precede whatever they overlap)
                        if (nms_rect_overlap(a, p.sorted[j]) <= p.thr_overlap) continue;
                        out = true;
                        break;
                    }
                }
            }
        }
        const unsigned long long any = (__builtin_amdgcn_ballot_w64(out) >> grp) & ((1ull << kNmsSub) - 1ull);
        const bool keep = valid && sub == 0 && any == 0ull;
        // one slot per surviving hit: champions from the front of `out`, undecided hits from its back (the host inserts the
        // champions into its grid without testing them)
        const unsigned long long act0 = __builtin_amdgcn_ballot_w64(keep && champ), act1 = __builtin_amdgcn_ballot_w64(keep && !champ);
        if (lane == 0) {
            s_cnt[0][wave] = (unsigned)__popcll(act0);
            s_cnt[1][wave] = (unsigned)__popcll(act1);
        }
        __syncthreads();
        if (threadIdx.x < 2) {
            const unsigned tot = s_cnt[threadIdx.x][0] + s_cnt[threadIdx.x][1] + s_cnt[threadIdx.x][2] + s_cnt[threadIdx.x][3];
            s_base[threadIdx.x] = tot ? atomicAdd(p.out_count + threadIdx.x, (unsigned long long)tot) : 0ull;
        }
        __syncthreads();
        if (keep) {
            const int side = champ ? 0 : 1;
            const unsigned long long act = champ ? act0 : act1;
            unsigned before = 0;
            for (int w = 0; w < wave; ++w) before += s_cnt[side][w];
            const unsigned below = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
            const unsigned long long slot = s_base[side] + before + below;
            p.out[side == 0 ? slot : (unsigned long long)n - 1ull - slot] = a;
        }
        __syncthreads();                                  // s_cnt / s_base are rewritten in the next turn
    }
}

}  // namespace mtm
