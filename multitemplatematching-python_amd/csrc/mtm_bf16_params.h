// Host-visible part of the float32 (bf16-piece) score-map kernel: tile constants, launch parameters, LDS size.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
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
    int yb0;                 // first row block of this launch (banded float32 uploads, round 6: a launch per band of rows); nyb counts from it
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
    // Round 5: the same bound for the NORMALISED methods, per output, in score units - what the listing decisions of the
    // refined routes rest on instead of round 3's empirical margins (1e-4 around the threshold, 5e-5 around a 3x3 maximum).
    //   |approximate ratio - exact ratio|  <=  M = rig_eps * sqrt(sum (I - mu)^2) / sq * (escale sqrt(t2c) / templ_norm)
    // (sq, templ_norm: the method's own denominators; escale = 2 for TM_SQDIFF_NORMED).  sum (I - mu)^2 / sq^2 is
    // 1 + A (window mean - mu)^2 / (A var): a low-contrast window beside a brightness step - where the tile constant mu is
    // far from the window's own mean - gets the large margin it needs, a textured one ~rig_eps.  An output is listed
    // (candidate list, running best) whenever its UPPER bound passes; exact re-scoring decides.
    int rig;                 // 1: listing by that bound (cand_thr / the running best are then compared WITHOUT a margin);
                             // 2: the raw-sum methods with a threshold - the bound of the sum itself (round 4's E, above)
    float rig_eps;
    int list_all;            // the threshold lies beyond the score range's clamp value on the far side (maxima: < 0, minima:
                             // > 1): a saturated exact score passes whatever the approximate one says - list everything
    // map mode (route 2: approximate maps + refine_scan_kernel with tolerances rig_cap / 2 rig_cap): an output that could
    // pass the threshold while its bound exceeds rig_cap sets *rig_flag - the host then takes the float64 kernel
    float rig_cap;
    float rig_thr;           // the exact quality threshold (score_threshold, negated for minima)
    unsigned int* rig_flag;
    // Round 6 (masked float32 templates, mtm_maskf32.hip.h): the constant a work item subtracted from its tile, per (row
    // block, segment) - mu_out[yb * nseg + seg], single-channel launches - so that a later pass can restate this launch's
    // error bound eps * sqrt(sum (I - mu)^2 ...) with the very constant that was used.  nullptr: not wanted
    float* mu_out;
};

// Per-template constants of a work item, staged in LDS once (the epilogue reads them as LDS broadcasts).
struct BfTemplConst {
    double mean[kMaxChans];
    double centre[kMaxChans];
    double templ_norm, templ_sum2;
    double t2c;              // sum over channels of sum (T - centre)^2 (error bound of the refined raw-sum extremum)
    double bfac;             // normalised methods: escale * sqrt(t2c) / templ_norm (0: constant template, scores exact)
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

// ncc_bf16_kernel<MB, NP> (defined in mtm_bf16.hip), MB = 1 or 2 groups of 16 templates per wave, NP = 3 piece products
// (scores to ~1e-5) or 1 (the one-product screen of the hits-only refined routes)
using Bf16Fn = void (*)(Bf16Params, const TemplDev*, const int*, const uint8_t*, StatPlanes, float*);
Bf16Fn bf16_kernel(int mb, int np);

}  // namespace mtm
