// Device-side data structures shared by the kernels and the host-side units.
#pragma once
#include <cstdint>

namespace mtm {

constexpr int kMaxChans = 4;

// float32 refinement (mtm_refine.hip.h): margins of the bf16 screen around the exact decisions
constexpr float kRefineThrMargin = 1e-4f;   // candidates: approximate quality > threshold - margin * max(1, |threshold|)
constexpr int kSlabStreams = 8;             // side streams of a slab class whose launches cannot fill the chip one at a time

// Padding of the planar device image so that tile staging never needs bounds checks:
// every kernel may read up to kPadCols bytes right of / kPadRows rows below the image.
constexpr int kPadCols = 512;
constexpr int kPadRows = 160;

// Per-template constants (device copy).  Mirrors mtm::TemplStats plus placement.
struct TemplDev {
    double mean[kMaxChans];
    double templ_norm;
    double templ_sum2;
    double templ2_mask2_sum;
    double centred_sum2;    // sum over channels of sum (T - channel mean)^2 (all methods)
    double mfma_k;          // 128*sum(T) - 16384*w*h*C: bias correction of the int8 MFMA path
    double centre[kMaxChans];   // float32 templates on the bf16 matrix cores: the per-channel mean the packed template
                                // was centred by (sum I*T = sum I*(T - centre) + centre * S1)
    long long map_off;      // float offset of this template's score map in the map arena
    long long k1_off;       // double offset of K1 (T, or T*M^2) in the weight arena, planar [C][h][w]
    long long k2_off;       // double offset of K2 (M^2) or -1
    long long pack_off;     // byte offset of the dot4-packed template in the pack arena, or -1
    int all_ones;
    int rows, cols;         // h, w
    int map_pitch;          // floats per score-map row on the device (multiple of 4)
    int oh, ow;             // score-map size
    int cls;                // index of the template's size class
};

// Window statistics planes of one size class (all double, pitch = stat_pitch elements).
struct StatPlanes {
    const double* t[kMaxChans];   // window sums per channel (numType == 1 only)
    const double* sum2;           // sum over channels of window sum of squares
    const double* sq;             // sqrt(diff2), or 0 where the window is flat (normed only)
    int pitch;
    // ranges of the statistics over every 16-pixel column block of an output row (stats_u8_kernel; the multi-row MFMA
    // variants' hits-only screen): [S1 min, S1 max, sqrt min, -] per block, blk_pitch blocks per row; sqrt min = +inf
    // for a block right of the last output column
    int blk_pitch;
    const double* blk;
};

struct ImageDev {
    const uint8_t* u8;     // planar, padded (u8 images only)
    const float* f32;      // planar, padded
    int rows, cols, chans;
    int u8_pitch;          // bytes per row
    int f32_pitch;         // floats per row
    long long u8_plane;    // bytes per plane
    long long f32_plane;   // floats per plane
};

}  // namespace mtm
