// Host-visible part of the float32 (bf16-piece) score-map kernel: tile constants, launch parameters, LDS size.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>

#include "mtm_kernels.h"
#include "../../include/mtm_hip.h"

namespace mtm {


typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i_b __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

constexpr int kBfSeg = 128;          // output pixels per wave (8 phases x 16 columns)
constexpr int kBfRows = 4;           // output rows per work-group
constexpr int kBfMaxW = 256;

struct Bf16Params {
    const float* img;        // planar padded float32 image
    int pitch;               // floats per row
    long long plane;         // floats per plane
    int chans;
    int rows, cols;          // image size (the planes are zero padded beyond it)
    int h, w, oh, ow;
    int nkb;                 // 32-tap blocks per template row
    int chunk_h;             // template rows per LDS tile
    int lds_cols;            // elements per tile row: 128 + 32 * nkb
    int n_list;
    int nseg, nyb, ntg, n_work;
    int method;
    long long group_bytes;   // bytes of one 16-template pack of ONE piece: chans * h * nkb * 1024
    long long piece_bytes;   // bytes between the T0 packs and the T1 packs
    int only_li;
    // fused peak candidates / hits-only, as in the other score kernels
    mtm_hit* cand_hits;
    unsigned long long* cand_counter;
    unsigned long long cand_cap;
    float cand_thr;
    int cand_min, cand_on, hits_only;
    // fused global extremum (mtm_find_matches, MTM_PEAKS_GLOBAL): nothing is stored, every wave keeps the best
    // (ordered score, ~index) key per template in LDS and merges it into ext_best[2 * template + cand_min]
    int ext_on;
    float ext_margin;        // > 0: refined extremum mode - outputs within this margin of the running best are listed (cand_hits)
    unsigned long long* ext_best;
    // Refined extremum mode of the raw-sum methods (TM_SQDIFF / TM_CCORR / TM_CCOEFF; round 4).  Their extremum can be a
    // difference of large terms (an exact copy has TM_SQDIFF 0), so no margin relative to the score is safe.  Instead every
    // output carries a rigorous error bound E = ext_eps * sqrt(sum (I - mu)^2 * sum (T - centre)^2) (Cauchy-Schwarz over the
    // dropped piece products and the float32 accumulation; twice that for TM_SQDIFF), the kernel publishes the best LOWER
    // bound q - E of a template and lists every output whose UPPER bound q + E reaches it: the exact extremum - and every
    // exact tie with it - is always listed, whatever the order the waves finish in.
    int ext_raw;
    float ext_eps;
};

// Per-template constants of a work item, staged in LDS once (the epilogue reads them as LDS broadcasts).
struct BfTemplConst {
    double mean[kMaxChans];
    double centre[kMaxChans];
    double templ_norm, templ_sum2;
    double t2c;              // sum over channels of sum (T - centre)^2 (error bound of the refined raw-sum extremum)
    long long map_off;
    int map_pitch, all_ones, tglob, pad_;
};

__host__ __device__ inline float bf16_to_float(uint32_t h) {
    const uint32_t b = h << 16;
    float f;
    __builtin_memcpy(&f, &b, 4);
    return f;
}

// LDS: [piece tile 0][piece tile 1][16 B: the subtracted constant][32 BfTemplConst][4 waves x 32 extremum keys]; the K loop requests operands up
// to two steps past a chunk (never used): those reads stay inside this allocation.
__host__ __device__ inline size_t bf16_lds_bytes(int chunk_h, int lds_cols) {
    return 2 * (size_t)(chunk_h + kBfRows - 1) * lds_cols * 2 + 16 + 32 * sizeof(BfTemplConst) +
           (size_t)kBfRows * 32 * sizeof(unsigned long long);
}

// ncc_bf16_kernel<MB> (defined in mtm_bf16.hip), MB = 1 or 2 groups of 16 templates per wave
using Bf16Fn = void (*)(Bf16Params, const TemplDev*, const int*, const uint8_t*, StatPlanes, float*);
Bf16Fn bf16_kernel(int mb);

}  // namespace mtm
