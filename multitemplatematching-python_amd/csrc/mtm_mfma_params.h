// Host-visible part of the MFMA score-map kernel: tile constants, the launch parameter block and the kernel
// function type.  The kernel template itself (mtm_mfma.hip.h) is only instantiated in the mtm_mfma_*.hip units;
// the launcher picks an instantiation through mfma_kernel().
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "mtm_kernels.h"
#include "../../include/mtm_hip.h"

namespace mtm {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kMfSeg = 256;          // output pixels per wave (16 phases x 16 columns)
constexpr int kMfRows = 4;           // output rows per work-group (one per wave)
constexpr int kMfChunkH = 64;        // template rows per LDS tile
constexpr int kMfChunkR2 = 72;       // K steps per LDS tile of the two-row variant (h + 1 steps: 64-row templates in one tile)
// Epilogue buffer of one wave: [8 templates][kMfEpiPitch] int32.  Pixel column xl = 16 j + c is
// stored at physical column 16 j + ((c + 4 * ((j >> 1) & 3)) & 15): 4-pixel groups stay contiguous
// and 16-byte aligned (one ds_read_b128 per lane and template, conflict-free), while the
// phase-strided ds_write_b32 of the MFMA C/D layout (lane = pixel column j) are at most 2-way
// conflicting, which costs nothing for 4-byte LDS stores.  Pitch = 4 (mod 8) dwords.
constexpr int kMfEpiPitch = 260;
constexpr int kMfEpiBytesPerWave = 8 * kMfEpiPitch * 4;
constexpr int kMfU16EpiBytesPerWave = 12 * kMfEpiPitch * 4;     // uint16 kernel: three accumulator sets of 4 templates per stage
// Window statistics of a wave's 256 pixels, prefetched to LDS by LDS-DMA while the image tile is
// staged: [3 planes: S1, S2, sqrt][2 halves][64 lanes][2 doubles]; lane L owns pixels 4L..4L+3.
constexpr int kMfStatPlaneBytes = 2 * 1024;                // one statistics plane of a wave's 256 pixels
constexpr int kMfStatBytesPerWave = 3 * kMfStatPlaneBytes;   // single channel: S1, S2, sqrt
__host__ __device__ constexpr int mf_stat_bytes_per_wave(int ch) { return (ch + 2) * kMfStatPlaneBytes; }
// Work-group scratch words behind the 32 per-template constants: [0] work item, [1] late start, [2] candidate list
// full, [4..7] the start values of the shader-clock probe (two 64-bit counters)
constexpr int kMfItemBytes = 32;
// Wave-private staging of peak candidates in LDS (hits-only / candidate mode): [count, pad x 3][kMfCandStage records]
constexpr int kMfCandStage = 128;
constexpr int kMfCandStageBytes = 16 + kMfCandStage * 24;
// METHOD value of the raw mode: the biased int8 accumulators are stored as they are (uint16 images:
// the first of two byte-plane passes, see kMfU16; sum I^2 M of masked classes; slabs).
constexpr int kMfRaw = 6;
// METHOD value of the second byte-plane pass of uint16 images: the low-byte plane against [T_hi | T_lo] of 16
// templates per work item; the epilogue reads the first pass's raw accumulators (high-byte plane, kMfRaw) of the
// same templates and finishes the exact 16-bit correlation there - no second set of raw maps, no combine kernel.
constexpr int kMfU16 = 7;

struct MfmaParams {
    const uint8_t* img;      // planar padded image, bytes already biased to int8 (^ 0x80)
    int pitch;
    long long plane;
    int chans;
    int h, w;
    int oh, ow;
    int nb;                  // 64-tap blocks per template row: ceil(w / 64)
    int n_list;              // templates of this class (list order = pack order)
    int nseg, nyb, ntg;      // work grid: x segments, row blocks, template groups (of 16*MB)
    int yb0;                 // first row block of this launch (banded launches: the image arrives in row bands)
    int n_work;
    int method;
    int lds_pitch;           // bytes per LDS tile row: (16 + 4*nb + 1) * 16
    int cpr, cpr_rstep, cpr_dstep;   // 16-byte chunks per tile row; 256 / cpr and 256 % cpr (tile DMA)
    int cpr_magic, cpr_pad_;         // floor(65536 / cpr) + 1: i / cpr == (i * cpr_magic) >> 16 for i < 256
    long long group_bytes;   // bytes of one 16-template A pack: chans * h * nb * 1024
    int only_li;             // >= 0: store only the template at this list position (mtm_score_map)
    int tc_off;              // byte offset in LDS of the per-template constants (after tile/epilogue)
    int st_off;              // byte offset in LDS of the prefetched window statistics (4 waves)
    int pad0_[4];
    mtm_hit* cand_hits;
    unsigned long long* cand_counter;
    unsigned long long cand_cap;
    float cand_thr;
    int cand_min;
    int cand_on;
    // Row-multiplexed mode (template parameter RM; classes of <= 16 templates): the 16 A rows of an
    // MFMA are rm_nt templates x rm_R consecutive output rows (A row i = template i % rm_nt, row
    // i / rm_nt), a wave owns 2 * rm_R output rows (MFMA group 1 = the next rm_R rows, its A operand is
    // the pack rm_R steps earlier: group_bytes = -rm_R * nb * 1024) and walks rm_steps = h + 2 rm_R - 1
    // image rows.  rm_R = 16 / rm_nt.
    int rm_R, rm_nt, rm_log2nt, rm_steps;
    long long rm_cstride;    // bytes between the packs of two channels (RM with 3 channels)
    const double* rm_rsq;    // 1 / sqrt plane of the class (0 for flat windows), pitch = st.pitch
    // slabs of one large-template class in ONE launch (raw row-multiplexed mode, slabs of equal height and block count):
    // work items [k nseg nyb, (k + 1) nseg nyb) belong to slab k = (channel * slab_nrb + row block) * slab_ncb + column
    // block, whose image window starts slab_rh rows / slab_cw columns per block further in, whose A pack lies
    // slab_ap_step bytes and whose raw maps slab_raw_step ints behind the previous slab's (0 / 1: a launch per slab)
    int n_slab, slab_ncb, slab_nrb, slab_cw, slab_rh;
    long long slab_ap_step, slab_raw_step;
    int* raw_out;            // METHOD == kMfRaw: int32 accumulators of list position li at raw_out + li * raw_map
    long long raw_map;       //   (+ y * raw_pitch + x); kMfU16: the raw maps of the first pass (read)
    int raw_pitch;
    double cand_thr_lo;      // cand_thr minus 8 float32 ulps (hits-only pre-test in float64)
    double screen_hi;        // hits-only screen: min(cand_thr_lo, 0.999999) - 1e-6 (extremum mode: per template, from its running best)
    double sq_floor;         // hits-only screen: lower bound of the sqrt statistic of a window that is not flat, 0.99 / sqrt(w h)
    int screen_l1, screen_pad_;   // 1: the screen starts with the per-lane bound (MTM_SCREEN_L1=0: round 2's screen alone)
    int hits_only;           // 1: candidates only, the score maps are not written (mtm_find_matches
                             // without map consumers); needs cand_on
    // fused global extremum (template parameter EXT; mtm_find_matches, MTM_PEAKS_GLOBAL, plain single-channel classes): nothing is
    // stored, every wave keeps the best (ordered score, ~index) key per template in LDS (ext_off: 4 waves x 32
    // keys) and merges it into ext_best[2 * template + cand_min] with one atomicMax per improved template.  The
    // best seen so far, re-read when a work item starts, is that template's threshold for the pre-tests.
    int ext_off;
    unsigned long long* ext_best;
    int dbg_unused_;
    // Packed K (kp_nseg > 0; plain and row-multiplexed tilings of unmasked uint8 classes whose width is not a multiple
    // of 64): the K dimension is the STREAM of 16-tap segments of the template rows (kp_nseg = ceil(w / 16) per row),
    // four consecutive segments per MFMA wherever the rows end - lane group q of step b holds segment 4 b + q, i.e.
    // image row (4 b + q) / nseg, taps 16 ((4 b + q) % nseg) .. + 15.  A 41-wide template then walks 3 segments per
    // row instead of a whole 64-tap block (the other 23 taps multiplied zeros), a 65-wide one 5 instead of 8.
    int kp_nseg;
    int kp_blocks;           // plain tiling: MFMA steps per channel = ceil(h * nseg / 4) (row-multiplexed: rm_cstride / 1024)
    // METHOD == kMfU16: byte sums of the templates ([0 .. npad) high bytes, [npad .. 2 npad) low bytes, class list
    // order), template area
    const double* u16_tsum;
    int u16_npad, u16_pad_;
    double u16_area;
    // Row-multiplexed raw mode as the sum I^2 M pass of a masked class in one launch: the two byte planes of I^2 are the
    // launch's channels (chans = 2), the accumulators are scaled by 256 between them and the epilogue writes
    // c2 = acc + sq_k as float64 into st.sum2 plus its 16-pixel block minima into st.blk (no raw maps, no combine kernel).
    int cs_off;              // byte offset in LDS of the candidate staging buffers (4 waves x kMfCandStageBytes); 0 = none
    int sq_fused;
    int pad_rm_edges_;
    // segment flags (dense images, maps in memory, the lean single- / three-channel epilogue): when one of the 256 outputs
    // of a wave's row of a template passes the candidate test, the byte
    // seg_flags[flag_base(template) + row * flag_rstride + segment] is set - the peak pass visits only those row segments
    // ("everything else is <= threshold" is all a 3x3 maximum above the threshold needs to know).
    // flag_base = template index * flag_tstride.  nullptr: off
    uint8_t* seg_flags;
    int flag_tstride, flag_rstride;
    // Round 5: with the segment flags on (dense images, maps not published), unmasked normalised classes finish only the
    // outputs that can pass the threshold (the hits-only pre-test) and store -inf (minima: +inf) for the others: a value that
    // is below the threshold can neither be a peak nor beat one, whatever it is exactly
    int seg_skip;
    double sq_k;             // 257 * 128 * sum(M)
    float* clk_out;          // non-null: the work-group in the middle of the grid stores the shader clock it ran at, in
                             // MHz (s_memtime ticks - shader cycles - per s_memrealtime tick of the 100 MHz reference)
    // Round 5: the first cand_pin_n slots of the candidate list are ALSO written into page-locked host memory by the
    // wave that fills them (a handful of 24-byte stores over PCIe per call) - the host reads them when the launch has
    // ended, without a fetch kernel / copy command and its kernel boundary behind the score launch.  0 = off.
    mtm_hit* cand_pin;
    unsigned long long cand_pin_n;
};

// one record of the candidate list: device list (slots below the capacity) + the host-visible window
__device__ __forceinline__ void mf_put_cand(const MfmaParams& p, unsigned long long slot, const mtm_hit& h) {
    if (slot < p.cand_cap) p.cand_hits[slot] = h;
    if (slot < p.cand_pin_n) p.cand_pin[slot] = h;
}

// Per-template constants staged in LDS once per work-group (the epilogue reads them with LDS
// broadcasts instead of dependent scalar loads per template).
struct MfTemplConst {
    double mean[kMaxChans];
    double templ_norm, templ_sum2, mfma_k;
    double rtempl_norm;      // 1 / templ_norm (0 when templ_norm == 0)
    double m128[kMaxChans];  // 128 - mean[c]: CCOEFF numerator straight from the biased accumulator
    double tms, rsqrt_tms;   // masked templates: sum((T*M)^2) and its inverse square root
    long long map_off;
    int map_pitch, all_ones;
    double ext_thr_lo;       // ext_on: quality (score, or -score for minima) of the best output seen so far,
    unsigned ext_hi;         //   lowered by 1e-6 relative; high word of its key (0: none yet)
    int flag_base;           // first byte of this template's row-segment flags (MfmaParams::seg_flags)
};

// One instantiation of ncc_mfma_kernel (defined in the mtm_mfma_*.hip units).
using MfmaFn = void (*)(MfmaParams, const TemplDev*, const int*, const uint8_t*, StatPlanes, float*);

// Template arguments of ncc_mfma_kernel as run-time values.
struct MfmaSel {
    int mb = 2;             // MFMA groups per wave (1: classes of <= 16 templates on the plain tiling)
    int method = 0;         // -1: generic epilogue (any channel count, run-time method); 0..5; kMfRaw; kMfU16
    bool exact_div = false; // IEEE division in the epilogue (MTM_OPT_EXACT_DIV)
    bool masked = false;    // binary uint8 mask (methods 0..3, one channel)
    bool rm = false;        // row-multiplexed tiling (<= 16 templates)
    int ch = 1;             // channels handled by the lean epilogue: 1 or 3
    bool ext = false;       // fused global extremum (N_object == 1)
    bool r2 = false;        // multi-row tiling: mb (2 or 3) consecutive output rows x 16 templates per wave
    bool kp = false;        // packed K
};
// The instantiation for `s`, or nullptr if that combination is not built.
MfmaFn mfma_kernel(const MfmaSel& s);
// per translation unit (mtm_mfma_plain / _rm / _ext / _kp / _rows .hip); mfma_kernel() dispatches between them
MfmaFn mfma_kernel_plain(const MfmaSel& s);
MfmaFn mfma_kernel_rm(const MfmaSel& s);
MfmaFn mfma_kernel_ext(const MfmaSel& s);
MfmaFn mfma_kernel_kp(const MfmaSel& s);
MfmaFn mfma_kernel_rows(const MfmaSel& s);

}  // namespace mtm
