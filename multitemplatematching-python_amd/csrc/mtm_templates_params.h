// Host-visible structures of the template kernels (mtm_templates.hip.h): unit views, source descriptors, pack parameters.
#pragma once
#include <cstdint>

#include "mtm_kernels.h"

namespace mtm {

// A unit's pixels: unit(c, y, x) = source(c, ay*y + by*x + cy, ax*y + bx*x + cx).  Coefficients in {-1, 0, 1}
// (plus the offsets): identity, np.rot90 k = 1..3, np.fliplr, np.flipud and their compositions.
struct UnitSrc {
    long long off;       // byte offset of channel plane 0 of the source in the arena ([C][sh][sw] uint8)
    long long moff;      // the mask source (same geometry), or -1
    int sh, sw;          // source rows / cols
    int ay, by, cy;      // source row  = ay * y + by * x + cy
    int ax, bx, cx;      // source col  = ax * y + bx * x + cx
    int h, w;            // unit rows / cols
    int chans;
    int pad_;
};

// ---------------------------------------------------------------------------------------------
// Exact sums of a source (what cv::meanStdDev / matchTemplateMask need; invariant under the eight views):
// out[src][0 .. 4C+1] = per channel {sum v, sum v^2, sum v*m, sum (v*m)^2}, then {sum m (channel 0), unused}.
// One work-group per source; uint64 accumulation (v <= 255, <= 2^31 pixels).
// ---------------------------------------------------------------------------------------------
struct SourceDesc {
    long long off, moff;     // pixel / mask planes in the arena (moff = -1: no mask)
    int sh, sw, chans, pad_;
};
constexpr int kSumsPerSource = 4 * kMaxChans + 2;

// ---------------------------------------------------------------------------------------------
// A-operand packs of ncc_mfma_kernel, gathered from the units (layouts: mtm_placement.hip pack_class_* comments).
// One thread per 16-byte lane chunk.  `tl` = the class's template list (unit indices).
//   mode 0 (plain) : group g = li / 16: [ch][dy][b][lane = 16 q + i][16]; lane (i, q) = taps 64 b + 16 q .. + 15
//                    of template li = 16 g + i (bytes T ^ 0x80; taps beyond w / templates beyond n = 0)
//   mode 1 (RM)    : [ch][sp = 0 .. h + 3R - 2][b][lane][16]; A row i = template i % nt, row offset i / nt:
//                    template row dy = sp - R - i / nt (0 outside 0 .. h - 1)
//   mode 2 (mask)  : mode 1 with nt = 1, R = 16 and the byte = (mask > 0) - 0 or 1, NOT biased - from unit tl[0]
// `masked`: bytes are T * M (M binary).
// ---------------------------------------------------------------------------------------------
struct PackParams {
    int mode, h, w, nb, chans, n;        // h = rows per group in the pack, n = templates of the class
    int hv;                              // valid template rows (rows hv .. h - 1 of a group stay zero: two-row variant)
    int nt, R;                           // RM
    long long group_bytes, cstride;      // plain: bytes per 16-template group; RM: bytes per channel
    int masked;
    int nseg;                            // > 0: packed K (MfmaParams::kp_nseg): [ch][block b][lane][16], lane group q of block b =
                                         // segment 4 b + q of the row stream (row (4 b + q) / nseg, taps 16 ((4 b + q) % nseg) ..)
    long long n_chunks;
    int kblocks, pad_;                   // packed K: blocks per channel
};

}  // namespace mtm
