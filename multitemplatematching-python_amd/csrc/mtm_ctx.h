// Internal header of libmtm_hip.so's GPU side: the context, the size classes of a template set, and the functions the
// translation units share.  Units: mtm_context.hip (context, options, image upload), mtm_placement.hip (template sets ->
// size classes, packs, constants), mtm_launch.hip (window statistics and score-map launches), mtm_api.hip
// (mtm_find_matches and friends: peak extraction, hit lists), mtm_comm.hip (RCCL hit exchange).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "mtm_kernels.h"
#include "mtm_mfma_params.h"
#include "mtm_bf16_params.h"
#include "mtm_score_params.h"
#include "mtm_templates_params.h"
#include "mtm_internal.h"

struct ncclComm;

namespace mtmi {

using namespace mtm;

#define HIPC(expr)                                                                          \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                   \
            return MTM_E_HIP;                                                               \
        }                                                                                   \
    } while (0)

#define MTMC(expr)                                                                          \
    do {                                                                                    \
        int r_ = (expr);                                                                    \
        if (r_ != MTM_OK) return r_;                                                        \
    } while (0)
// between mtm_find_matches_async and mtm_find_matches_wait the context belongs to that call
#define MTM_NOT_IN_FLIGHT(c, who)                                                                         \
    do {                                                                                                  \
        if ((c)->fm_in_flight) {                                                                          \
            set_error(std::string(who) + ": a mtm_find_matches_async call is in flight (collect it with " \
                      "mtm_find_matches_wait first)");                                                    \
            return MTM_E_INVALID;                                                                         \
        }                                                                                                 \
    } while (0)


inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

inline size_t elem_size(int dtype) { return dtype == MTM_U8 ? 1 : dtype == MTM_U16 ? 2 : 4; }

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    // Grows to at least `bytes` (contents are NOT preserved).  The new block is allocated before the old one
    // is released: a failed allocation leaves the buffer as it was.
    int ensure(size_t bytes) {
        if (bytes <= cap) return MTM_OK;
        const size_t want = round_up(bytes + bytes / 8, 256);
        void* fresh = nullptr;
        HIPC(hipMalloc(&fresh, want));
        if (p) (void)hipFree(p);
        p = fresh;
        cap = want;
        return MTM_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct HostTempl {
    int rows = 0, cols = 0, chans = 0, dtype = 0;
    bool masked = false;
    std::vector<double> px;     // planar [C][h][w]
    std::vector<double> mask;   // planar weights (binarised for uint8 masks) or empty
    TemplStats st;
    int cls = -1;
    // uint8 template sets: the pixels live on the device as a view of a source in the arena (mtm_templates.hip.h);
    // px / mask above stay empty until a kernel that needs host-packed weights asks for them (ensure_host_pixels)
    bool on_device = false;
    UnitSrc src{};
    double sum_t = 0.0;                 // sum over all channels of T (masked: T * M): bias term of the MFMA path
    double mask_ones = 0.0;             // set mask pixels (channel 0)
    unsigned long long mask_key = 0;    // identifies the (transformed) mask: equal keys = equal masks
};

struct SizeClass {
    int h = 0, w = 0;
    bool masked = false;
    bool all_u8 = true;
    bool all_u16 = true;
    bool all_f32 = true;
    bool bf16_ok = false;       // float32 class on the bf16 matrix cores (ncc_bf16_kernel)
    bool mfma16_ok = false;     // uint16 class on the int8 MFMA path (byte-plane decomposition)
    int rm_nt = 0, rm_R = 0;    // > 0: row-multiplexed MFMA mode (<= 16 templates: nt x R = 16 A rows)
    int kp_nseg = 0;            // > 0: packed K (MfmaParams::kp_nseg): ceil(w / 16) segments per template row, 4 per MFMA step
    // large templates (w > 256 or w*h*C > 131071) on the MFMA kernel: cut into slabs (slab_combine_kernel)
    struct Slab {
        int r0, r1, c0, c1, ch;
        long long apack_off;    // this slab's packs in the apack arena
        int tlist_off;          // its view list (unit-table indices) in the device tlist
    };
    std::vector<Slab> slabs;
    int slab_nt = 0, slab_R = 0;    // > 0: row-multiplexed raw launches (<= 16 templates); 0: plain raw launches
    int r2 = 0;                 // multi-row MFMA variant (> 16 templates, w <= 64, one channel, methods 2..5): 2 or 3 consecutive
                                // output rows x 16 templates per wave; packs of h + r2 - 1 rows per 16-template group (the
                                // extra rows zero); 0 = off
    long long mask_rm_off = -1; // masked class: row-multiplexed pack (1 "template" = the binary mask, R = 16) in apacks
    double mask_ones = 0.0;     // number of set mask pixels
    int n_pad = 0;              // members rounded up to a multiple of 16 (uint16 packs)
    long long tsum_off = -1;    // doubles: [sum(T_hi) per member][sum(T_lo) per member] in the tsum arena
    std::vector<int> members;
    int tlist_off = 0;          // offset into the device tlist array
    bool mfma_ok = false;       // packed for ncc_mfma_kernel
    bool masked_int = false;    // masked class on the integer path: binary uint8 mask shared by all members
    unsigned long long mask_hash = 0;
    long long mask_pack_off = -1;   // dot4 pack of the mask bytes (0xFF / 0) in the pack arena (MTM_ROW_MUX=0 / MTM_FUSE_STATS=0 only)
    // float32 class with masks on the bf16 matrix cores as a screen (mtm_maskf32.hip.h, round 6): bf16 packs of U = T M^2 and
    // V = M^2 in the apack arena (the class itself stays a float64-kernel class: weights, fallback, exact re-scoring)
    bool mask_bf16 = false;
    long long mbf_off_u = -1, mbf_off_v = -1, mbf_group_bytes = 0;
    long long apack_off = 0;    // byte offset of this class's A packs in the apack arena
    long long group_bytes = 0;
};

}  // namespace mtmi

// What mtm_find_matches knows after its asynchronous half (everything up to and including the kernels and the
// first fetch are queued on the stream) and needs in its synchronising half.  mtm_find_matches_async /
// mtm_find_matches_wait keep one of these in the context between the two calls.
namespace mtmi {
struct FmState {
    int mode = 0;
    float thr = 0.0f;
    bool mode_min = false, fused = false, prefetched = false;
    bool banded_u8 = false;     // the call's image came in row bands (uint8): the next such call clears the candidate header itself
    bool pin_direct = false;    // the score kernel wrote the head of the candidate list into the pinned window itself (no fetch)
    bool pp_mode = false;       // float32 refinement by map scan: the candidate buffer holds potential peaks whose
                                // neighbourhoods in the maps are exact - decisions by verify_peaks_kernel, never from the list alone
    int n = 0;
    int64_t cand_cap = 0;
    unsigned hash_mask = 0;
};
}  // namespace mtmi

struct mtm_ctx {
    using DevBuf = mtmi::DevBuf;
    using HostTempl = mtmi::HostTempl;
    using SizeClass = mtmi::SizeClass;
    using FmState = mtmi::FmState;
    using TemplDev = mtm::TemplDev;
    using UnitSrc = mtm::UnitSrc;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ncc_ev;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> sq_ev;       // event pairs of the sum I^2 M passes (timing.masked_stat_ms)
    int cand_pinned = 1;        // MTM_CAND_PINNED: the score kernel writes the first records of its candidate list into the pinned
                                // landing buffer itself (MfmaParams::cand_pin) - no fetch kernel behind the score launch (0: round 4)
    bool cand_pin_now = false;  // ... in the launches being queued
    size_t cand_pin_n = 0;
    bool single_band_now = false;
    bool zero_pending = false;  // ... and has not done so yet
    int fuse_layout = 1;        // MTM_FUSE_LAYOUT: banded uploads convert a band's rows inside its statistics launch (0: planarize kernel)
    int lay_r0 = 0, lay_r1 = 0; // ... the rows the statistics launch being queued converts (run_score_banded -> launch_stats)
    int masksq_fused = 1;       // MTM_MASKSQ_FUSED: sum I^2 M of a masked class as ONE launch over both byte planes of I^2 that
                                // writes the sum2 plane itself (0: round 3's two raw launches + masksq_combine_kernel)

    // image
    bool have_image = false;
    int rows = 0, cols = 0, chans = 0, dtype = 0;
    int u8_pitch = 0, f32_pitch = 0, rows_alloc = 0;
    // Device copies of an image: raw (as uploaded), planar padded u8 / int8-biased u8 / f32.  Two slots:
    // `cur` is what the kernels read; the other one receives the next image of a stream
    // (mtm_find_matches_next) on copy_stream while the kernels run.
    struct ImageSlot {
        DevBuf raw, u8, u8b, f32;
        long long geom = -1;        // (rows, cols, chans, dtype) the padding was initialised for
        bool f32_valid = true;      // false after a banded uint8 upload: the float32 plane was skipped (ensure_f32_plane)
    } slot[2];
    int cur = 0;
    DevBuf sq_planes;           // two planes: [high byte of I^2 ^ 0x80][low byte ^ 0x80] of the current uint8 image
    bool sq_valid = false;
    hipStream_t copy_stream = nullptr;
    // mtm_find_matches_image_nms: MTM.matchTemplates' non-maxima suppression as part of the call (the host's nms_boxes on
    // the fetched list).  Thousands of peaks on the device (the flagged-segment route of dense images) are pruned there
    // first (mtm_k_nms.hip.h; MTM_NMS_DEVICE=0: never): what a neighbourhood's best hit suppresses never crosses PCIe.
    struct NmsRequest {
        bool on = false;
        double score_threshold = 0.0, max_overlap = 0.0;
        long long n_object = -1;
    } nms_req;
    long long nms_raw_count = -1;           // >= 0: the device pruned this call's peak list; the count before that
    long long nms_sure = 0;                 // ... and its first nms_sure hits are kept for certain (the neighbourhoods' best)
    long long nms_device_min = 4096;        // fewer peaks than this: the host is as fast
    DevBuf nms_buf;
    // segment flags (MTM_SPARSE_MAPS, default 1): the route of a call on dense maps (candidate list overflowed recently)
    // when every class runs the lean MFMA epilogue - maps in memory, one flag per row segment that holds something above the
    // threshold, peaks_sparse_kernel over the flagged segments instead of the full scan (MfmaParams::seg_flags)
    int sparse_maps = 1;
    bool sparse_now = false;
    DevBuf seg_flags, hits_t;               // (hits_t: per-template peak lists of peaks_sparse_kernel + their counters)
    int flag_tstride = 0, flag_rstride = 0;
    hipEvent_t next_ready = nullptr;
    // mtm_find_matches_image: the image arrives in row bands on copy_stream (copy, layout conversion, window
    // statistics of the rows that became computable); the score kernel of a band waits for its event
    hipStream_t stats_stream = nullptr;     // non-null while a banded call queues its statistics launches
    int screen_l1 = 1;                      // MTM_SCREEN_L1: the hits-only screen starts with the per-lane bound (0: round 2's screen alone)
    int f32_mfma = 1;                       // MTM_F32_MFMA / MTM_OPT_F32_MFMA: unmasked float32 classes on the bf16 matrix cores:
                                            // 0 = float64 kernel, 1 = bf16 screen + exact float64 re-scoring of everything
                                            // that could be a peak (hit lists of the float64 kernel), 2 = bf16 scores as they are,
                                            // 3 = as 1 without the one-product tier (every screen with three piece products),
                                            // 4 = (diagnostic) as 2 with ONE piece product: the screen's raw scores, what
                                            // tests/test_gpu_parity.py measures its bound on - never a result
    // Round 6: the hits-only refined routes screen with ONE piece product first (ncc_bf16_kernel<MB, 1>, a third of the
    // matrix-core work, bound 2^-7 instead of 2^-15 of the norms' product); a list that overflows repeats the launch with
    // three products, and the next np1_backoff calls start there
    int bf16_np_now = 3;                    // this call's piece products of the hits-only screens (fm_begin)
    int np1_backoff = 0, np1_backoff_len = 16;
    int seg_skip = 1;                       // MTM_SEG_SKIP: dense route - outputs that cannot pass the threshold are not finished (MfmaParams::seg_skip)
    bool seg_skip_used = false;             // this call: some map holds such placeholders (the maps are not published)
    bool raw_rig_now = false;               // this call: a raw-sum method with a threshold, listed by the bound of the sum (route 1 only)
    float rig_thr = 0.0f;                   // the exact quality threshold of the call (the lists' own cand_thr carries a margin in map mode)
    float scan_thr = 0.0f;                  // map mode: the threshold of refine_scan_kernel (rig_thr lowered by rig_cap)
    float rig_cap = 0.0f;                   // map mode: the bound up to which the scan's tolerances hold (else: float64 kernel)
    // float32 refinement (mtm_refine.hip.h), state of the current mtm_find_matches
    bool refine_now = false;                // bf16 classes of this call are refined
    bool refine_scan_now = false;           // ... by map scan + ring re-scoring (maps in memory) instead of kernel candidates
    bool f32_exact_now = false;             // bf16 classes run the float64 kernel in this call (refinement lists overflowed)
    int templ_on_device = 1;                // MTM_TEMPL_ON_DEVICE: uint8 template sets live on the device (views + device packing); 0 = packed on
                                            // the host (the independent restatement test_template_sets_on_device compares with)
    int mfma_r2 = 1;                        // MTM_MFMA_R2: 1 = two-row variant of the MFMA kernel where it applies, 0 = off
    // lanes of a multi-class call: size classes are independent (their own statistics, launches, scratch); consecutive
    // classes go to alternating lanes - a stream plus the per-class scratch buffers - so that the statistics / combine
    // kernels and the tail of one class run under the score kernel of the next.  Lane 0 is the context's own stream and
    // buffers.
    struct Lane {
        DevBuf stats, stats_rsq, stats_blk, hs1, hs2, raw16, slab_raw, stats_hi, mask_td, sched;
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        // the lane's own side streams for a slab class (two slab classes on two lanes must neither share fork / join
        // events nor serialise on one set of side streams)
        std::vector<hipStream_t> slab_streams;
        std::vector<hipEvent_t> slab_done;
        hipEvent_t slab_fork = nullptr;
    };
    std::vector<Lane> lanes;                // lanes 1 .. n - 1
    bool slab_fork_early = false;           // run_score_classes recorded slab_fork ahead of the class's statistics pass
    hipEvent_t lane_fork = nullptr;
    hipEvent_t f32_built = nullptr;         // multi-lane calls: the float32 plane rebuilt after a banded upload (run_score_classes)
    int class_lanes = 2;                    // MTM_CLASS_LANES (1: classes one after another on the main stream)
    // side streams of a slab class: its raw launches are independent and (few templates, small images) far too small to
    // fill the chip one at a time
    std::vector<hipStream_t> slab_streams;
    std::vector<hipEvent_t> slab_done;
    hipEvent_t slab_fork = nullptr;
    std::vector<hipEvent_t> band_ev;
    int banded_cls = -1;                    // the size class a banded call runs under the upload (banded_ok)
    std::vector<double> upload_bands{0.25, 1.0};   // cumulative row fractions (MTM_UPLOAD_BANDS)
    // MTM_HOST_TRACE=1: host time stamps at the phases of a fused call, averaged and printed when the context is destroyed
    bool host_trace = false;
    double trace_acc[24] = {0};
    long long trace_n[24] = {0};
    long long trace_calls = 0;
    double trace_t0 = 0.0;

    // templates
    bool have_templ = false;
    bool placed = false;
    int method = MTM_TM_CCOEFF_NORMED;
    std::vector<HostTempl> templs;
    std::vector<SizeClass> classes;
    std::vector<TemplDev> td_host;
    std::vector<int> tlist_host;
    std::vector<int> list2d;        // templates with a 2-D score map
    int list2d_off = 0;
    size_t maps_floats = 0;
    std::vector<UnitSrc> usrc_host;                 // the unit views (device copy: usrc_dev), slab views appended at placement
    size_t usrc_units = 0;                          // entries that are units (the rest are slab views)
    std::vector<uint8_t> tstage;                    // host image of the source arena's prefix (set_templates_device)
    bool stage_pending = false;                     // copies from tstage / usrc_host may be in flight on `stream`
    bool place_pending = false;                     // copies from td_host / tlist_host may be in flight on `stream`
    DevBuf slab_raw;                                // raw int32 maps of the slabs
    // masked float32 classes screened on the bf16 matrix cores: the U / V template tables, the approximate c1 / c2 maps,
    // J = I^2, the window sums of I and J, the launches' tile constants, the list of outputs to re-score exactly
    DevBuf td_u, td_v, mbf_maps, f32_sq, mbf_stats, mbf_mu, mbf_list, mbf_best;
    bool mbf_thr_on = false;                        // this call: local extrema against mbf_thr, or (mbf_global) N_object == 1 (find_matches_impl)
    bool mbf_global = false;
    float mbf_thr = 0.0f;
    bool mbf_used = false;                          // this call: some class's maps hold "below the threshold" placeholders
    bool f32_sq_valid = false;                      // f32_sq holds the square of the current float32 plane
    DevBuf tsrc, usrc_dev, tsums_dev, tgather;      // template source arena, unit views, source sums, gather scratch
    DevBuf td, tlist, weights, packs, apacks, maps, hs1, hs2, stats, hits, counters, sched, cands, mask_td, chash, raw16, stats_hi, tsum, stats_rsq, stats_blk;

    // options
    int opt_kernel = MTM_KERNEL_AUTO;
    int opt_border = MTM_BORDER_NEAREST;   // scikit-image >= 0.19 (maximum_filter mode='nearest'); MTM_PEAK_BORDER=constant: <= 0.18
    // Capacity (records) of the candidate list and of the hit list.  2^18: measured in round 4 against 2^20 on a
    // photograph-like 4K image x 32 templates at threshold 0.5 (~1e6 outputs above the threshold): a list that holds them
    // all keeps the call in hits-only mode, but a million emissions cost the score kernel +0.64 ms (1.32 against 0.68 ms,
    // one atomic per wave and item) - more than map mode + the candidate test of the dense route; up to ~2.6e5
    // candidates the list is the cheaper way, and that is where it overflows.
    int64_t hit_cap = 1 << 18;
    int dot_variant = 0;
    int fuse_stats = 1;        // MTM_FUSE_STATS: single-kernel window statistics (0 = the two-pass kernels, which large windows and float32
                               // images use anyway: test_uint16_fused_window_statistics_bit_for_bit compares the two)
    int row_mux = 1;           // MTM_ROW_MUX: row-multiplexed MFMA tiling for classes of <= 16 templates (0 = the plain tiling:
                               // test_fused_global_extremum runs both)
    double band_min_fill = 1.0;   // MTM_BAND_MIN_FILL: a band is only worth a launch of its own if its work items fill the resident
                               // work-group slots this many times (0 lets the band tests use small images)
    int n_cus = 0;
    // candidate emission of the current launch sequence (set by mtm_find_matches)
    bool cand_on = false;
    bool cand_min = false;
    float cand_thr = 0.f;
    int hits_only = 1;         // MTM_OPT_HITS_ONLY: mtm_find_matches does not materialise the score maps when
                               // every class runs the single-channel MFMA kernel (candidates + hash verify)
    int backoff_len = 16;      // length of the next back-off period: doubles with every overflow in a row (<= 1024), reset by a
                               // call whose candidates fitted
    int fuse_backoff = 0;      // calls left without fused candidates (map mode + full peak pass: maps known to be dense)
    bool hits_only_now = false;
    bool maps_valid = false;   // the map arena holds every score map of the last mtm_find_matches (mtm_last_score_map)
    FmState fm;                         // mtm_find_matches_async -> mtm_find_matches_wait
    bool fm_in_flight = false;
    const void* cands_zeroed = nullptr;   // candidate buffer whose counter was cleared after the previous call's fetch
    bool ext_now = false;      // this call: global extrema come out of the MFMA epilogue (no maps, no extremum_kernel)
    int exact_div = 1;         // MTM_OPT_EXACT_DIV: 1 (default since round 5) = IEEE division in the MFMA epilogue, bit-identical to the
                               // oracle; 0 = correctly rounded reciprocals (<= 1 ulp(float32) on ~1e-8 of the outputs); 2 = strict: also
                               // the fused extremum of masked classes (reciprocal-only kernels) goes through maps + extremum_kernel
    int auto_kernel = MTM_KERNEL_MFMA;   // what MTM_KERNEL_AUTO resolves to for uint8 classes (dot4 when not eligible)

    mtm_timing timing{};
    std::vector<mtm_hit> last_hits;     // result of the last mtm_find_matches (for mtm_last_hits)
    void* pin_small = nullptr;          // 64 page-locked bytes: landing place of small device-to-host copies (device NMS counters)
    void* pinned = nullptr;             // pinned host buffer the candidate records land in
    size_t pinned_cap = 0;
    void* comm_pin = nullptr;           // pinned staging of the hit exchange: [my slot | gathered slots]
    size_t comm_pin_cap = 0;
    std::vector<uint8_t> templ_blob;    // bytes of the templates of the last mtm_set_templates (unchanged-input test)

    // RCCL
    void* rccl_lib = nullptr;
    ncclComm* comm = nullptr;           // ncclComm_t
    int n_ranks = 1, rank = 0;
    long long comm_slot_hits = 512;
    double comm_timeout_s = 300.0;      // deadline of one hit exchange (MTM_COMM_TIMEOUT_S; 0 = none)
    std::vector<unsigned long long> vh_keys;   // host verification of the candidate list: open-addressing table
    std::vector<int> vh_vals;
    DevBuf comm_send, comm_recv;
    std::vector<long long> comm_last_counts;   // per-rank counts of the last exchange (mtm_comm_last_gather)
    size_t comm_last_slot = 0;                 //   and its slot size in bytes; the slots are still in comm_pin
};

namespace mtmi {

inline double host_now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// phase `k` of the current call reached (k = 0: the call starts)
inline void host_trace(mtm_ctx* c, int k) {
    if (!c->host_trace) return;
    const double t = host_now_us();
    if (k == 0) {
        c->trace_t0 = t;
        ++c->trace_calls;
    }
    if (c->trace_calls <= 8) return;        // (the first calls create streams, events and buffers: not the steady state)
    c->trace_acc[k] += t - c->trace_t0;
    ++c->trace_n[k];
}

inline ImageDev image_dev(const mtm_ctx* c) {
    ImageDev d;
    d.u8 = c->slot[c->cur].u8.as<uint8_t>();
    d.f32 = c->slot[c->cur].f32.as<float>();
    d.rows = c->rows;
    d.cols = c->cols;
    d.chans = c->chans;
    d.u8_pitch = c->u8_pitch;
    d.f32_pitch = c->f32_pitch;
    d.u8_plane = (long long)c->u8_pitch * c->rows_alloc;
    d.f32_plane = (long long)c->f32_pitch * c->rows_alloc;
    return d;
}

// ---- geometry of the operand packs (shared by the placement and the launches)
constexpr int kMfmaMaxW = 256;          // widest template of one MFMA launch (wider: slabs)
inline long long mfma_group_bytes(int h, int w, int chans) { return (long long)chans * h * ((w + 63) / 64) * 1024; }
// packed K: MFMA steps (1 KiB of A operand each) of `rows` stream rows of nseg 16-tap segments
inline int kp_blocks(int rows, int nseg) { return (rows * nseg + 3) / 4; }
inline int mfma_groups_alloc(int n) { return (((n + 15) / 16) + 1) & ~1; }     // multiple of MB = 2
// Row-multiplexed packs (<= 16 uint8 templates, one channel, no mask): steps sp' = 0 .. h + 3R - 2, A row
// i = (template i % nt, row offset i / nt) holds template row sp' - R - i / nt (zero outside 0..h-1).
// MFMA group 0 of step s reads pack step s + R, group 1 (the wave's next R output rows) pack step s.
inline long long rm_pack_bytes(int h, int w, int R) { return (long long)(h + 3 * R - 1) * ((w + 63) / 64) * 1024; }
// bytes of one channel of a class's row-multiplexed pack.  Packed K: the two MFMA groups have packs of their own
// (group g, image row r of the h + 2R - 1 a wave walks: template row r - g R - rho), kp_blocks steps each - the shifted
// reuse of one pack (classic layout) would need R rows to be a whole number of 4-segment steps.
inline long long class_rm_pack_bytes(const SizeClass& sc) {
    return sc.kp_nseg ? 2LL * kp_blocks(sc.h + 2 * sc.rm_R - 1, sc.kp_nseg) * 1024 : rm_pack_bytes(sc.h, sc.w, sc.rm_R);
}
inline int bf16_nkb(int w) { return (w + 31) / 32; }
inline long long bf16_group_bytes(int h, int w, int chans) { return (long long)chans * h * bf16_nkb(w) * 1024; }
// dynamic LDS of one ncc_mfma_kernel work-group: image tile (aliased by the epilogue buffers), per-template constants,
// work-group scratch words, prefetched statistics
inline size_t mfma_lds_bytes(int tile_rows, int nb, size_t stat_bytes) {
    const size_t lds_pitch = (size_t)(16 + 4 * nb + 1) * 16;
    const size_t lds_main = (std::max<size_t>((size_t)tile_rows * lds_pitch, (size_t)kMfRows * kMfEpiBytesPerWave) + 15) & ~(size_t)15;
    const size_t st_off = (lds_main + sizeof(MfTemplConst) * 32 + kMfItemBytes + 15) & ~(size_t)15;
    return st_off + stat_bytes;
}

// the image of a fused "upload + search" call (mtm_find_matches_image)
struct ImageArgs {
    const void* px;
    int rows, cols, chans, dtype;
    int64_t stride;
};
// Geometry of the planar device copies of an image (after an optional integer downscale).
struct SlotGeom {
    int rows, cols, rows_alloc, pitch;
    size_t u8_bytes;
};

// Error of a bf16-piece correlation relative to sqrt(sum (I - mu)^2 * sum (T - centre)^2) (Cauchy-Schwarz): 2^-15 covers the
// dropped piece products and the two 16-bit representations (3 * 2^-18 + the float32 rounding of I - mu), the rest the
// float32 accumulation - three MFMAs per 32-tap block, counted as TWO roundings of 2^-24 each (the matrix core's own
// 32-term sum is not documented as a single rounding; tests/test_gpu_parity.py::test_float32_error_bound_holds measures
// the whole bound against the float64 kernel on adversarial and random data).
// Round 6 (advisor): the whole is doubled.  The "two roundings per MFMA" is an assumption about undocumented hardware; the
// measured worst case is 0.16 of the undoubled bound, so the factor costs a few more exact re-scores and buys a guarantee that
// survives an ASIC or compiler whose accumulation is a little worse than assumed.
// (Round 6, second look at the first term: bfloat16 carries 8 significant bits, unit roundoff 2^-8, so two pieces represent a
// float to 2^-16 and the three neglected terms - I's residual, T's residual, I1 T1 - are 3 * 2^-16 = 4.6e-5 of |I||T| per tap,
// not the "3 * 2^-18" the 2^-15 above was written for; the doubled total, 6.1e-5 + ..., still covers it.)
// np == 1 (the one-product screen, ncc_bf16_kernel<MB, 1>): only I0 T0 is summed.  |I T - I0 T0| <= |I - I0||T| + |I0||T - T0|
// <= (2^-8 + 2^-8 (1 + 2^-8)) |I||T| = 2^-7 (1 + 2^-9) |I||T| per tap, Cauchy-Schwarz over the taps; ONE MFMA per 32-tap block,
// two roundings each, doubled like the rest for the undocumented accumulation.
inline bool f32_refined(const mtm_ctx* c) { return c->f32_mfma == 1 || c->f32_mfma == 3; }
inline float bf16_rig_eps(int chans, int h, int nkb, int np = 3) {
    if (np == 1) return (float)(0.0078125 * 1.002 + 2.0 * (2.0 * 1.0 * (double)chans * h * nkb * 5.97e-8));
    return (float)(2.0 * (3.0518e-5 + 2.0 * 3.0 * (double)chans * h * nkb * 5.97e-8));
}
// ... and what the refined raw-sum extremum (Bf16Params::ext_eps) works with
inline float bf16_ext_eps(int chans, int h, int nkb, int np = 3) {
    if (np == 1) return bf16_rig_eps(chans, h, nkb, 1);
    return (float)(3.0518e-5 + 3.0 * (double)chans * h * nkb * 5.97e-8);
}

// ---- mtm_context.hip
int prepare_slot(mtm_ctx* c, mtm_ctx::ImageSlot& sl, int src_rows, int src_cols, int chans, int dtype, hipStream_t stream,
                 int factor, SlotGeom* out);
int upload_rows_u8c1(mtm_ctx::ImageSlot& sl, const SlotGeom& g, const void* src, int64_t src_stride, int r0, int r1,
                     hipStream_t stream, bool skip_f32, hipEvent_t copy_done = nullptr, hipEvent_t before_kernels = nullptr,
                     bool convert = true);
int upload_rows_u16c1(mtm_ctx::ImageSlot& sl, const SlotGeom& g, const void* src, int64_t src_stride, int r0, int r1,
                      hipStream_t stream, hipEvent_t copy_done = nullptr, hipEvent_t before_kernels = nullptr);
int upload_image(mtm_ctx* c, mtm_ctx::ImageSlot& sl, const void* src, int64_t src_stride, int src_rows, int src_cols,
                 int chans, int dtype, hipStream_t stream, int factor = 1);
void adopt_image(mtm_ctx* c, int rows, int cols, int chans, int dtype);
int check_image_args(const void* px, int rows, int cols, int chans, int dtype, int64_t row_stride_bytes, const char* who);
int ensure_f32_plane(mtm_ctx* c);
int upload_rows_f32c1(mtm_ctx::ImageSlot& sl, const SlotGeom& g, const void* src, int64_t src_stride, int r0, int r1, hipStream_t stream);
int ensure_copy_stream(mtm_ctx* c);
int ensure_lanes(mtm_ctx* c, int n);
// ---- mtm_comm.hip
int comm_allgather_hits_flagged(mtm_ctx* c, const mtm_hit* local, int64_t n_local, int32_t my_flag, mtm_hit* out,
                                int64_t capacity, int64_t* counts_out, int64_t* n_out, std::vector<int32_t>* flags_out);
// ---- mtm_placement.hip
int place_templates(mtm_ctx* c);
// ---- mtm_launch.hip
int resolved_kernel(const mtm_ctx* c, const SizeClass& sc);
bool dot_variant_ok(int64_t v);
int launch_stats(mtm_ctx* c, const SizeClass& sc, StatPlanes* out, int sb0 = 0, int sb1 = -1);
int launch_ncc(mtm_ctx* c, const SizeClass& sc, int list_off, int n_list, const StatPlanes& st, int only_li = -1,
               int yb0 = 0, int yb1 = -1);
int ensure_maps(mtm_ctx* c);
int run_score_all(mtm_ctx* c);
bool banded_ok(mtm_ctx* c, const ImageArgs& a);
int run_score_banded(mtm_ctx* c, const ImageArgs& a);
int collect_ncc_time(mtm_ctx* c);

}  // namespace mtmi
