// libmtm_hip.so - context, launch logic and the GPU entry points of include/mtm_hip.h.
// gfx950 (MI355X / CDNA4) only; built by multitemplatematching-python_amd/build.py with
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "mtm_device.hip.h"
#include "mtm_mfma_params.h"
#include "mtm_templates.hip.h"
#include "mtm_bf16_params.h"
#include "mtm_refine.hip.h"
#include "mtm_internal.h"

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

namespace {

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
    long long mask_pack_off = -1;   // dot4 pack of the mask bytes (0xFF / 0) in the pack arena
    long long apack_off = 0;    // byte offset of this class's A packs in the apack arena
    long long group_bytes = 0;
};

// dot4 kernel variants
struct DotVariant {
    int px, py, nt;
    bool wide;
    void (*fn)(DotParams, const TemplDev*, const int*, const uint8_t*, StatPlanes, float*);
};
#define DOTV(PX, PY, NT, W) {PX, PY, NT, W, ncc_dot4_kernel<PX, PY, NT, W>}
const DotVariant kDotVariants[] = {
    DOTV(4, 4, 4, false),   // 0: default
    DOTV(4, 4, 2, false),   // 1
    DOTV(8, 2, 4, false),   // 2
    DOTV(8, 4, 2, false),   // 3
    DOTV(4, 2, 4, false),   // 4
    DOTV(8, 2, 2, false),   // 5
    DOTV(4, 2, 8, false),   // 6
    DOTV(4, 2, 2, true),    // 7: uint64 totals for templates with C*w*h*255^2 >= 2^32
};
constexpr int kDotWideVariant = 7;
constexpr int kNumDotVariants = sizeof(kDotVariants) / sizeof(kDotVariants[0]);

}  // namespace

// What mtm_find_matches knows after its asynchronous half (everything up to and including the kernels and the
// first fetch are queued on the stream) and needs in its synchronising half.  mtm_find_matches_async /
// mtm_find_matches_wait keep one of these in the context between the two calls.
struct FmState {
    int mode = 0;
    float thr = 0.0f;
    bool mode_min = false, fused = false, prefetched = false;
    bool pp_mode = false;       // float32 refinement by map scan: the candidate buffer holds potential peaks whose
                                // neighbourhoods in the maps are exact - decisions by verify_peaks_kernel, never from the list alone
    int n = 0;
    int64_t cand_cap = 0;
    unsigned hash_mask = 0;
};

struct mtm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ncc_ev;

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
    DevBuf sq_planes;           // [high byte of I^2][the same ^ 0x80][low byte ^ 0x80] of the current uint8 image
    bool sq_valid = false;
    hipStream_t copy_stream = nullptr;
    hipEvent_t next_ready = nullptr;
    // mtm_find_matches_image: the image arrives in row bands on copy_stream (copy, layout conversion, window
    // statistics of the rows that became computable); the score kernel of a band waits for its event
    hipStream_t stats_stream = nullptr;     // non-null while a banded call queues its statistics launches
    hipStream_t stream2 = nullptr;          // second compute stream: the score launches of consecutive bands alternate
                                            // between `stream` and this one, so the tail of one launch (its last
                                            // work-groups draining) is filled by the next launch instead of idling
    hipStream_t ncc_stream = nullptr;       // non-null: launch_ncc queues the MFMA kernel (and its timing events) here
    int screen_l1 = 1;                      // MTM_SCREEN_L1: the hits-only screen starts with the per-lane bound (0: round 2's screen alone)
    int kpack = 1;                          // MTM_KPACK: packed K for template widths that are not multiples of 64
    int skip_f32 = 1;                       // MTM_SKIP_F32: banded uploads leave the float32 plane out (rebuilt on demand)
    int f32_mfma = 1;                       // MTM_F32_MFMA / MTM_OPT_F32_MFMA: unmasked float32 classes on the bf16 matrix cores:
                                            // 0 = float64 kernel, 1 = bf16 screen + exact float64 re-scoring of everything
                                            // that could be a peak (hit lists of the float64 kernel), 2 = bf16 scores as they are
    // float32 refinement (mtm_refine.hip.h), state of the current mtm_find_matches
    bool refine_now = false;                // bf16 classes of this call are refined
    bool refine_scan_now = false;           // ... by map scan + ring re-scoring (maps in memory) instead of kernel candidates
    bool f32_exact_now = false;             // bf16 classes run the float64 kernel in this call (refinement lists overflowed)
    int mfma_r2 = 1;                        // MTM_MFMA_R2: 1 = two-row variant of the MFMA kernel where it applies, 3 = three rows, 0 = off
    int dual_stream = 0;                    // MTM_DUAL_STREAM=1: score launches of consecutive bands alternate between two
                                            // streams (their tails overlap; per-launch durations then overlap too)
    int templ_on_device = 1;                // MTM_TEMPL_ON_DEVICE: uint8 template sets live on the device (views + device packing)
    int copy_prio = 1;                      // MTM_COPY_PRIO: 1 = the copy stream gets the highest stream priority
    hipEvent_t stream2_done = nullptr;
    std::vector<hipEvent_t> band_ev;
    std::vector<double> upload_bands{0.25, 1.0};   // cumulative row fractions (MTM_UPLOAD_BANDS)

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
    int slab_mfma = 1;                              // MTM_SLAB_MFMA: large templates as slabs on the MFMA kernel
    DevBuf slab_raw;                                // raw int32 maps of the slabs
    DevBuf tsrc, usrc_dev, tsums_dev, tgather;      // template source arena, unit views, source sums, gather scratch
    DevBuf td, tlist, weights, packs, apacks, maps, hs1, hs2, stats, hits, counters, sched, cands, mask_td, chash, raw16, stats_hi, tsum, stats_rsq, stats_blk;

    // options
    int opt_kernel = MTM_KERNEL_AUTO;
    int opt_border = MTM_BORDER_NEAREST;   // scikit-image >= 0.19 (maximum_filter mode='nearest'); MTM_PEAK_BORDER=constant: <= 0.18
    int64_t hit_cap = 1 << 18;
    int dot_variant = 0;
    int mfma_dbg = 0;
    int fuse_stats = 1;        // MTM_FUSE_STATS: single-kernel window statistics (uint8, one channel)
    int n_cus = 0;
    std::map<std::pair<const void*, size_t>, int> occupancy_cache;
    int fuse_peaks = 1;        // MTM_FUSE_PEAKS: candidates from the MFMA epilogue + verify kernel
    // candidate emission of the current launch sequence (set by mtm_find_matches)
    bool cand_on = false;
    bool cand_min = false;
    float cand_thr = 0.f;
    int row_mux = 1;           // MTM_ROW_MUX: row-multiplexed MFMA mode for classes of <= 16 templates
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
    int exact_div = 0;         // MTM_OPT_EXACT_DIV: IEEE division in the MFMA epilogue (bit-exact mode)
    int mfma_persistent = 0;   // 1: persistent grid + atomic work counter (measured slightly slower)
    int mfma_stagger = -1;     // < 0: automatic
    int mfma_stagger_mode = 0;
    int mfma_per_cu = 2;
    int mfma_stagger_np = 0;   // > 0: stagger the first wave of blocks of a non-persistent launch by this many s_sleep(127)
    int auto_kernel = MTM_KERNEL_MFMA;   // what MTM_KERNEL_AUTO resolves to for uint8 classes (dot4 when not eligible)

    mtm_timing timing{};
    std::vector<mtm_hit> last_hits;     // result of the last mtm_find_matches (for mtm_last_hits)
    void* pinned = nullptr;             // pinned host buffer the candidate records land in
    size_t pinned_cap = 0;
    void* comm_pin = nullptr;           // pinned staging of the hit exchange: [my slot | gathered slots]
    size_t comm_pin_cap = 0;
    std::vector<uint8_t> templ_blob;    // bytes of the templates of the last mtm_set_templates (unchanged-input test)

    // RCCL
    void* rccl_lib = nullptr;
    ncclComm_t comm = nullptr;
    int n_ranks = 1, rank = 0;
    long long comm_slot_hits = 512;
    double comm_timeout_s = 300.0;      // deadline of one hit exchange (MTM_COMM_TIMEOUT_S; 0 = none)
    std::vector<unsigned long long> vh_keys;   // host verification of the candidate list: open-addressing table
    std::vector<int> vh_vals;
    DevBuf comm_send, comm_recv;
    std::vector<long long> comm_last_counts;   // per-rank counts of the last exchange (mtm_comm_last_gather)
    size_t comm_last_slot = 0;                 //   and its slot size in bytes; the slots are still in comm_pin
};

namespace {

ImageDev image_dev(const mtm_ctx* c) {
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

// Packs one uint8 template for ncc_dot4_kernel (layout documented in mtm_device.hip.h).
size_t dot_pack_bytes(int h, int w, int chans) {
    const int w4 = (w + 3) & ~3;
    const int ncy = (h + kDotChunk - 1) / kDotChunk, ncx = (w4 + kDotChunk - 1) / kDotChunk;
    return (size_t)chans * ncy * ncx * kDotChunkBytes;
}

void pack_template_dot4(const HostTempl& t, uint8_t* out) {
    const int h = t.rows, w = t.cols;
    const int w4 = (w + 3) & ~3;
    const int ncy = (h + kDotChunk - 1) / kDotChunk, ncx = (w4 + kDotChunk - 1) / kDotChunk;
    std::memset(out, 0, dot_pack_bytes(h, w, t.chans));
    for (int c = 0; c < t.chans; ++c)
        for (int cy = 0; cy < ncy; ++cy)
            for (int cx = 0; cx < ncx; ++cx) {
                uint8_t* chunk = out + ((size_t)(c * ncy + cy) * ncx + cx) * kDotChunkBytes;
                const int ch = std::min(kDotChunk, h - cy * kDotChunk);
                const int cw = std::min(kDotChunk, w - cx * kDotChunk);
                for (int dy = 0; dy < ch; ++dy)
                    for (int dx = 0; dx < cw; ++dx)
                        chunk[(dy + kDotPadRows) * kDotChunk + dx] =
                            (uint8_t)t.px[((size_t)c * h + cy * kDotChunk + dy) * w + cx * kDotChunk + dx];
            }
}

// A-operand packs of ncc_mfma_kernel for one size class: per group of 16 templates (list order),
// [channel][template row][64-tap block][lane = 16*q + i][16 bytes]: lane (i, q) holds taps
// 64*b + 16*q .. +15 of template i, biased to int8 (T ^ 0x80); taps beyond the template width and
// templates beyond the list are 0 (the signed zero), so they add nothing.
constexpr int kMfmaMaxW = 256;
// Slab layout of a large unmasked uint8 class (empty: not slab-able).  Column blocks are multiples of 64 taps wide
// (the LDS-DMA tile staging wants 16-byte aligned image offsets): 128 for the row-multiplexed raw launches of
// <= 16 templates (their LDS tile is 6R rows taller), 256 otherwise; row ranges keep rows * width <= 131071.
std::vector<SizeClass::Slab> slab_layout(const mtm_ctx* c, const SizeClass& sc) {
    std::vector<SizeClass::Slab> out;
    if (!c->slab_mfma || sc.masked || !sc.all_u8 || c->dtype != MTM_U8) return out;
    for (int m : sc.members)
        if (!c->templs[(size_t)m].on_device) return out;
    const int cw = sc.members.size() <= 16 ? 128 : 256;
    const int rows_max = 131071 / std::min(cw, ((sc.w + 63) / 64) * 64);
    const int nrb = (sc.h + rows_max - 1) / rows_max, rh = (sc.h + nrb - 1) / nrb;
    for (int ch = 0; ch < c->chans; ++ch)
        for (int r0 = 0; r0 < sc.h; r0 += rh)
            for (int c0 = 0; c0 < sc.w; c0 += cw)
                out.push_back(SizeClass::Slab{r0, std::min(sc.h, r0 + rh), c0, std::min(sc.w, c0 + cw), ch, 0, 0});
    if (out.size() > 24) out.clear();            // absurdly large: the VALU kernel takes it
    return out;
}

bool mfma_class_ok(const mtm_ctx* c, const SizeClass& sc) {
    if (c->dtype == MTM_U8 && sc.all_u8 && !sc.masked && (sc.w > kMfmaMaxW || (long long)c->chans * sc.w * sc.h > 131071))
        return !slab_layout(c, sc).empty();
    if (!(c->dtype == MTM_U8 && sc.all_u8 && sc.w <= kMfmaMaxW && (long long)c->chans * sc.w * sc.h <= 131071))
        return false;
    // masked: single channel, sum I^2*M must fit the uint32 dot4 accumulator (w*h*255^2 < 2^32)
    if (sc.masked) return c->chans == 1 && (long long)sc.w * sc.h <= 66051;
    return true;
}
// dynamic LDS of one ncc_mfma_kernel work-group: image tile (aliased by the epilogue buffers), per-template constants,
// work-group scratch words, prefetched statistics
size_t mfma_lds_bytes(int tile_rows, int nb, size_t stat_bytes) {
    const size_t lds_pitch = (size_t)(16 + 4 * nb + 1) * 16;
    const size_t lds_main = (std::max<size_t>((size_t)tile_rows * lds_pitch, (size_t)kMfRows * kMfEpiBytesPerWave) + 15) & ~(size_t)15;
    const size_t st_off = (lds_main + sizeof(MfTemplConst) * 32 + kMfItemBytes + 15) & ~(size_t)15;
    return st_off + stat_bytes;
}

long long mfma_group_bytes(int h, int w, int chans) { return (long long)chans * h * ((w + 63) / 64) * 1024; }
// packed K: MFMA steps (1 KiB of A operand each) of `rows` stream rows of nseg 16-tap segments
inline int kp_blocks(int rows, int nseg) { return (rows * nseg + 3) / 4; }
int mfma_groups_alloc(int n) { return (((n + 15) / 16) + 1) & ~1; }     // multiple of MB = 2

// Row-multiplexed packs (<= 16 uint8 templates, one channel, no mask): steps sp' = 0 .. h + 3R - 2, A row
// i = (template i % nt, row offset i / nt) holds template row sp' - R - i / nt (zero outside 0..h-1).
// MFMA group 0 of step s reads pack step s + R, group 1 (the wave's next R output rows) pack step s.
long long rm_pack_bytes(int h, int w, int R) { return (long long)(h + 3 * R - 1) * ((w + 63) / 64) * 1024; }
// bytes of one channel of a class's row-multiplexed pack.  Packed K: the two MFMA groups have packs of their own
// (group g, image row r of the h + 2R - 1 a wave walks: template row r - g R - rho), kp_blocks steps each - the shifted
// reuse of one pack (classic layout) would need R rows to be a whole number of 4-segment steps.
long long class_rm_pack_bytes(const SizeClass& sc) {
    return sc.kp_nseg ? 2LL * kp_blocks(sc.h + 2 * sc.rm_R - 1, sc.kp_nseg) * 1024 : rm_pack_bytes(sc.h, sc.w, sc.rm_R);
}

// the binary mask of a masked class as the single "template" (nt = 1, R = 16) of the sum I^2 M pass
void pack_mask_rm(const mtm_ctx* c, const SizeClass& sc, uint8_t* out) {
    const int h = sc.h, w = sc.w, nb = (w + 63) / 64, R = 16;
    std::memset(out, 0, (size_t)rm_pack_bytes(h, w, R));
    const HostTempl& ht = c->templs[sc.members[0]];
    for (int sp = 0; sp < h + 3 * R - 1; ++sp)
        for (int i = 0; i < 16; ++i) {
            const int dy = sp - R - i;
            if (dy < 0 || dy >= h) continue;
            for (int dx = 0; dx < w; ++dx) {
                const int b = dx / 64, q = (dx % 64) / 16, byte = dx % 16;
                const uint8_t v = ht.mask[(size_t)dy * w + dx] > 0.0 ? 1 : 0;
                out[(((size_t)sp * nb + b) * 64 + (16 * q + i)) * 16 + byte] = v ^ 0x80;
            }
        }
}

void pack_class_rm(const mtm_ctx* c, const SizeClass& sc, uint8_t* out) {
    const int h = sc.h, w = sc.w, nb = (w + 63) / 64, R = sc.rm_R, nt = sc.rm_nt, nseg = sc.kp_nseg;
    const int chans = sc.masked ? 1 : c->chans;                 // one pack per channel, class_rm_pack_bytes apart
    const size_t cstride = (size_t)class_rm_pack_bytes(sc);
    std::memset(out, 0, cstride * chans);
    if (nseg) {                                  // packed K: [ch][group g][step][lane][16], stream row r: dy = r - g R - rho
        const size_t gbytes = cstride / 2;
        for (int ch = 0; ch < chans; ++ch)
            for (int g = 0; g < 2; ++g)
                for (int r = 0; r < h + 2 * R - 1; ++r)
                    for (int i = 0; i < 16; ++i) {
                        const int t = i % nt, rho = i / nt, dy = r - g * R - rho;
                        if (t >= (int)sc.members.size() || dy < 0 || dy >= h) continue;
                        const HostTempl& ht = c->templs[sc.members[(size_t)t]];
                        for (int dx = 0; dx < w; ++dx) {
                            const size_t k = ((size_t)ch * h + dy) * w + dx;
                            const int sidx = r * nseg + dx / 16;
                            const uint8_t v = (uint8_t)(ht.masked ? ht.px[k] * ht.mask[k] : ht.px[k]);   // masked: T*M
                            out[ch * cstride + g * gbytes + (((size_t)(sidx / 4) * 64) + (16 * (sidx % 4) + i)) * 16 + dx % 16] =
                                v ^ 0x80;
                        }
                    }
        return;
    }
    for (int ch = 0; ch < chans; ++ch)
        for (int sp = 0; sp < h + 3 * R - 1; ++sp)
            for (int i = 0; i < 16; ++i) {
                const int t = i % nt, rho = i / nt, dy = sp - R - rho;
                if (t >= (int)sc.members.size() || dy < 0 || dy >= h) continue;
                const HostTempl& ht = c->templs[sc.members[(size_t)t]];
                for (int dx = 0; dx < w; ++dx) {
                    const size_t k = ((size_t)ch * h + dy) * w + dx;
                    const uint8_t v = (uint8_t)(ht.masked ? ht.px[k] * ht.mask[k] : ht.px[k]);   // masked: T*M, M in {0,1}
                    const size_t blk = (size_t)sp * nb + dx / 64;
                    const int q = (dx % 64) / 16;
                    out[ch * cstride + ((blk * 64) + (16 * q + i)) * 16 + dx % 16] = v ^ 0x80;
                }
            }
}

// uint16 image + uint16 templates, one channel, no mask: four uint8 byte-plane correlations on the int8
// MFMA kernel (a raw pass over the high bytes, a finishing pass over the low bytes).  Same int32 accumulator bound
// as the uint8 path.
bool mfma16_class_ok(const mtm_ctx* c, const SizeClass& sc) {
    return c->dtype == MTM_U16 && sc.all_u16 && c->chans == 1 && !sc.masked && sc.w <= kMfmaMaxW &&
           (long long)sc.w * sc.h <= 131071;
}

// float32 image + float32 templates, no mask: two bfloat16 pieces per value on the bf16 matrix cores - for the
// NORMALISED methods, whose outputs are O(1) and stay within ~1e-5 of the float64 result.  The raw sums (TM_SQDIFF,
// TM_CCORR, TM_CCOEFF) are as accurate relative to the sums they are built from (~1e-7), but they can cancel to
// values far smaller than those sums (an exact copy: SQDIFF = 0), where no relative bound holds: float64 kernel.
bool bf16_class_ok(const mtm_ctx* c, const SizeClass& sc) {
    const bool normed = c->method == MTM_TM_SQDIFF_NORMED || c->method == MTM_TM_CCORR_NORMED || c->method == MTM_TM_CCOEFF_NORMED;
    // (1-D and 1x1 score maps go through scipy's find_peaks on the host, which has no refinement: float64 kernel)
    return c->f32_mfma && normed && c->dtype == MTM_F32 && sc.all_f32 && !sc.masked && sc.w <= kBfMaxW &&
           c->rows > sc.h && c->cols > sc.w;
}
inline int bf16_nkb(int w) { return (w + 31) / 32; }
long long bf16_group_bytes(int h, int w, int chans) { return (long long)chans * h * bf16_nkb(w) * 1024; }

// A packs of a float32 class for ncc_bf16_kernel: [piece 0 | piece 1][group of 16][ch][dy][32-tap block][lane = 16 q + i]
// [8 bf16]: lane (i, q) holds taps 32 kb + 8 q .. + 7 of template i, centred by its channel mean and split
// v = v0 + v1 (bfloat16, round to nearest even).  centre[] receives the means (TemplDev::centre).
void pack_class_bf16(const mtm_ctx* c, const SizeClass& sc, uint8_t* out, std::vector<TemplDev>& td_host) {
    const int h = sc.h, w = sc.w, nkb = bf16_nkb(w), chans = c->chans;
    const long long gb = bf16_group_bytes(h, w, chans);
    const int groups = mfma_groups_alloc((int)sc.members.size());
    const long long piece = gb * groups;
    std::memset(out, 0, (size_t)(2 * piece));
    auto rne = [](float v) {
        uint32_t b;
        std::memcpy(&b, &v, 4);
        return (uint16_t)((b + 0x7FFFu + ((b >> 16) & 1u)) >> 16);
    };
    for (size_t li = 0; li < sc.members.size(); ++li) {
        const HostTempl& t = c->templs[sc.members[li]];
        TemplDev& d = td_host[(size_t)sc.members[li]];
        uint8_t* g = out + (li / 16) * gb;
        const int i = (int)(li % 16);
        const size_t plane = (size_t)h * w;
        for (int ch = 0; ch < chans; ++ch) {
            double mean = 0.0;
            for (size_t k = 0; k < plane; ++k) mean += t.px[ch * plane + k];
            mean /= (double)plane;
            d.centre[ch] = mean;
            for (int dy = 0; dy < h; ++dy)
                for (int dx = 0; dx < w; ++dx) {
                    const float v = (float)(t.px[ch * plane + (size_t)dy * w + dx] - mean);
                    const uint16_t v0 = rne(v);
                    const uint16_t v1 = rne(v - bf16_to_float(v0));
                    const int kb = dx / 32, q = (dx % 32) / 8, e = dx % 8;
                    const size_t o = ((((size_t)ch * h + dy) * nkb + kb) * 64 + (16 * q + i)) * 16 + 2 * e;
                    std::memcpy(g + o, &v0, 2);
                    std::memcpy(g + piece + o, &v1, 2);
                }
        }
    }
}

// A packs of a uint16 class: 2 * n_pad pseudo-templates in 16-template groups [high bytes of members 16 g .. 16 g + 15]
// [low bytes of the same members] - one work item of ncc_mfma_kernel (32 pseudo-templates) holds both byte planes of
// 16 templates.  Same lane order as pack_class_mfma.  Also the byte sums of every member (bias terms of the
// combination): tsum[li] high bytes, tsum[n_pad + li] low bytes.
void pack_class_mfma16(const mtm_ctx* c, const SizeClass& sc, uint8_t* out, double* tsum) {
    const int h = sc.h, w = sc.w, nb = (w + 63) / 64, n_pad = sc.n_pad, nseg = sc.kp_nseg;
    const long long gb = sc.group_bytes;
    std::memset(out, 0, (size_t)gb * (2 * n_pad / 16));
    for (int k = 0; k < 2 * n_pad; ++k) tsum[k] = 0.0;
    for (size_t li = 0; li < sc.members.size(); ++li) {
        const HostTempl& t = c->templs[sc.members[li]];
        for (int part = 0; part < 2; ++part) {
            uint8_t* g = out + (2 * (li / 16) + (size_t)part) * gb;
            const int i = (int)(li % 16);
            double sum = 0.0;
            for (int dy = 0; dy < h; ++dy)
                for (int dx = 0; dx < w; ++dx) {
                    const unsigned v16 = (unsigned)t.px[(size_t)dy * w + dx];
                    const uint8_t v = part == 0 ? (uint8_t)(v16 >> 8) : (uint8_t)(v16 & 255u);
                    sum += v;
                    const int b = dx / 64, q = (dx % 64) / 16, byte = dx % 16;
                    if (nseg) {                                  // packed K: segment dy * nseg + dx / 16 of the row stream
                        const int sidx = dy * nseg + dx / 16;
                        g[(((size_t)(sidx / 4)) * 64 + (16 * (sidx % 4) + i)) * 16 + byte] = v ^ 0x80;
                        continue;
                    }
                    g[(((size_t)dy * nb + b) * 64 + (16 * q + i)) * 16 + byte] = v ^ 0x80;
                }
            tsum[(size_t)part * n_pad + li] = sum;
        }
    }
}

void pack_class_mfma(const mtm_ctx* c, const SizeClass& sc, uint8_t* out) {
    const int h = sc.h, w = sc.w, nb = (w + 63) / 64, chans = c->chans, nseg = sc.kp_nseg;
    // multi-row variant: groups of h + r2 - 1 rows (one channel), the extra rows stay zero; packed K: kp_blocks steps per channel
    const long long gb = (sc.r2 || nseg) ? sc.group_bytes : mfma_group_bytes(h, w, chans);
    std::memset(out, 0, (size_t)gb * (sc.r2 ? ((int)sc.members.size() + 15) / 16 : mfma_groups_alloc((int)sc.members.size())));
    for (size_t li = 0; li < sc.members.size(); ++li) {
        const HostTempl& t = c->templs[sc.members[li]];
        uint8_t* g = out + (li / 16) * gb;
        const int i = (int)(li % 16);
        for (int ch = 0; ch < chans; ++ch)
            for (int dy = 0; dy < h; ++dy)
                for (int dx = 0; dx < w; ++dx) {
                    const int b = dx / 64, q = (dx % 64) / 16, byte = dx % 16;
                    const size_t k = ((size_t)ch * h + dy) * w + dx;
                    const uint8_t v = (uint8_t)(t.masked ? t.px[k] * t.mask[k] : t.px[k]);   // masked: T*M, M in {0,1}
                    if (nseg) {                                  // packed K: segment dy * nseg + dx / 16 of the row stream
                        const int sidx = dy * nseg + dx / 16;
                        g[((((size_t)ch * kp_blocks(h, nseg) + sidx / 4) * 64) + (16 * (sidx % 4) + i)) * 16 + byte] = v ^ 0x80;
                        continue;
                    }
                    g[((((size_t)ch * (sc.r2 ? h + sc.r2 - 1 : h) + dy) * nb + b) * 64 + (16 * q + i)) * 16 + byte] = v ^ 0x80;
                }
    }
}

int resolved_kernel(const mtm_ctx* c, const SizeClass& sc);

// Pixels (and binary mask) of a device-resident uint8 template as the planar float64 arrays the host-side packers
// take: one gather kernel + one copy.  Only the fallback kernels (float64, dot4) ever need it.
int ensure_host_pixels(mtm_ctx* c, int i) {
    HostTempl& t = c->templs[(size_t)i];
    if (!t.on_device || !t.px.empty()) return MTM_OK;
    const size_t n = (size_t)t.chans * t.rows * t.cols;
    MTMC(c->tgather.ensure(2 * n));
    uint8_t* dpx = c->tgather.as<uint8_t>();
    uint8_t* dmk = t.masked ? dpx + n : nullptr;
    hipLaunchKernelGGL(gather_unit_kernel, dim3((t.cols + 63) / 64, t.rows, t.chans), dim3(64), 0, c->stream,
                       c->tsrc.as<uint8_t>(), t.src, dpx, dmk);
    HIPC(hipGetLastError());
    std::vector<uint8_t> host(2 * n);
    HIPC(hipMemcpyAsync(host.data(), dpx, (t.masked ? 2 : 1) * n, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    t.px.resize(n);
    for (size_t k = 0; k < n; ++k) t.px[k] = (double)host[k];
    if (t.masked) {
        t.mask.resize(n);
        for (size_t k = 0; k < n; ++k) t.mask[k] = host[n + k] ? 1.0 : 0.0;
    }
    return MTM_OK;
}

// A packs of one MFMA class gathered on the device from the unit views (pack_units_kernel); layouts as the host
// packers above.  The class's template list must already be in c->tlist.
int pack_class_on_device(mtm_ctx* c, const SizeClass& sc) {
    const uint8_t* arena = c->tsrc.as<uint8_t>();
    const UnitSrc* units = c->usrc_dev.as<UnitSrc>();
    const int* tl = c->tlist.as<int>() + sc.tlist_off;
    PackParams p{};
    p.h = sc.h;
    p.hv = sc.h;
    p.w = sc.w;
    p.nb = (sc.w + 63) / 64;
    p.n = (int)sc.members.size();
    auto launch = [&](long long off) {
        hipLaunchKernelGGL(pack_units_kernel, dim3((unsigned)((p.n_chunks + 255) / 256)), dim3(256), 0, c->stream, p, arena,
                           units, tl, c->apacks.as<uint8_t>() + off);
    };
    if (sc.mask_rm_off >= 0) {            // the binary mask as the single "template" of the sum I^2 M pass
        p.mode = 2;
        p.chans = 1;
        p.nt = 1;
        p.R = 16;
        p.masked = 0;
        p.cstride = rm_pack_bytes(sc.h, sc.w, 16);
        p.n_chunks = p.cstride / 16;
        launch(sc.mask_rm_off);
    }
    if (!sc.slabs.empty()) {               // every slab is a class of its own: its view list, its dimensions
        for (const auto& sl : sc.slabs) {
            p.h = p.hv = sl.r1 - sl.r0;
            p.w = sl.c1 - sl.c0;
            p.nb = (p.w + 63) / 64;
            p.chans = 1;
            p.masked = 0;
            tl = c->tlist.as<int>() + sl.tlist_off;
            if (sc.slab_R > 0) {
                p.mode = 1;
                p.nt = sc.slab_nt;
                p.R = sc.slab_R;
                p.cstride = rm_pack_bytes(p.h, p.w, sc.slab_R);
                p.n_chunks = p.cstride / 16;
            } else {
                p.mode = 0;
                p.group_bytes = mfma_group_bytes(p.h, p.w, 1);
                p.n_chunks = p.group_bytes * mfma_groups_alloc(p.n) / 16;
            }
            launch(sl.apack_off);
        }
        HIPC(hipGetLastError());
        return MTM_OK;
    }
    p.masked = sc.masked ? 1 : 0;
    p.nseg = sc.kp_nseg;
    if (sc.rm_R > 0) {
        p.mode = 1;
        p.chans = sc.masked ? 1 : c->chans;
        p.nt = sc.rm_nt;
        p.R = sc.rm_R;
        p.cstride = class_rm_pack_bytes(sc);
        p.kblocks = (int)(p.cstride / 2048);             // packed K: steps per MFMA group ([ch][group][step])
        p.n_chunks = p.cstride * p.chans / 16;
    } else if (sc.r2) {
        p.mode = 0;
        p.chans = 1;
        p.h = sc.h + sc.r2 - 1;              // rows per group in the pack; rows >= h are zero (hv = valid template rows)
        p.group_bytes = sc.group_bytes;
        p.n_chunks = p.group_bytes * ((p.n + 15) / 16) / 16;
    } else {
        p.mode = 0;
        p.chans = c->chans;
        p.group_bytes = sc.group_bytes;                  // (packed K: chans * kp_blocks steps)
        p.kblocks = sc.kp_nseg ? kp_blocks(sc.h, sc.kp_nseg) : 0;
        p.n_chunks = p.group_bytes * mfma_groups_alloc(p.n) / 16;
    }
    launch(sc.apack_off);
    HIPC(hipGetLastError());
    return MTM_OK;
}

int place_templates(mtm_ctx* c) {
    if (!c->have_image || !c->have_templ) {
        set_error("set the image and the templates first");
        return MTM_E_STATE;
    }
    if (c->placed) return MTM_OK;
    const int n = (int)c->templs.size();
    // Everything is derived into locals and committed at the end: a failure half-way (an allocation, a copy)
    // leaves the context exactly as it was - templates set, not placed.
    std::vector<SizeClass> classes = c->classes;
    std::vector<TemplDev> td_host((size_t)n, TemplDev{});
    std::vector<int> tlist_host, list2d;
    size_t map_off = 0, w_off = 0, p_off = 0;
    const bool img_u8 = c->dtype == MTM_U8;
    // Which kernel will run each class decides what has to be packed: int8 A-packs for the MFMA kernel,
    // dot4 packs for the VALU kernel, float64 weights for the float64 / naive kernels.  (Changing
    // MTM_OPT_KERNEL re-places.)
    std::vector<int> class_kernel(classes.size(), MTM_KERNEL_AUTO);
    for (size_t k = 0; k < classes.size(); ++k) {
        classes[k].mfma_ok = mfma_class_ok(c, classes[k]);
        classes[k].mfma16_ok = mfma16_class_ok(c, classes[k]);
        classes[k].bf16_ok = bf16_class_ok(c, classes[k]);
        classes[k].n_pad = (int)round_up(classes[k].members.size(), 16);
        class_kernel[k] = resolved_kernel(c, classes[k]);
        // row-multiplexed mode: uint8 class of <= 16 templates (one channel, masked or not, or unmasked RGB) whose
        // window statistics the fused kernels produce (the single-channel one also writes the 1/sqrt plane)
        SizeClass& sc = classes[k];
        sc.rm_nt = sc.rm_R = 0;
        const size_t n_cls = sc.members.size();
        if (c->row_mux && class_kernel[k] == MTM_KERNEL_MFMA && n_cls <= 16 && c->fuse_stats &&
            (c->chans == 1 || (c->chans == 3 && !sc.masked)) &&
            (double)c->chans * sc.w * sc.h * 65025.0 < 4294967296.0) {
            int nt = 1;
            while (nt < (int)n_cls) nt <<= 1;
            // two work-groups per CU need <= ~76 KB of LDS each: wide templates take fewer rows per MFMA group
            const size_t lds_pitch = (size_t)(16 + 4 * ((sc.w + 63) / 64) + 1) * 16;
            auto tile_bytes = [&](int R) { return (size_t)(std::min(sc.h + 2 * R - 1, kMfChunkH) + 6 * R) * lds_pitch; };
            while (nt < 16 && tile_bytes(16 / nt) > 72 * 1024) nt <<= 1;
            sc.rm_nt = nt;
            sc.rm_R = 16 / nt;
        }
        sc.slabs.clear();
        sc.slab_nt = sc.slab_R = 0;
        const bool big = sc.w > kMfmaMaxW || (long long)c->chans * sc.w * sc.h > 131071;
        if (class_kernel[k] == MTM_KERNEL_MFMA && big) {
            sc.slabs = slab_layout(c, sc);
            sc.rm_nt = sc.rm_R = 0;
            if (n_cls <= 16) {                     // row-multiplexed raw launches: nt templates x R rows per MFMA group
                int nt = 1;
                while (nt < (int)n_cls) nt <<= 1;
                int hs = 0, ws = 0;
                for (const auto& sl : sc.slabs) {
                    hs = std::max(hs, sl.r1 - sl.r0);
                    ws = std::max(ws, sl.c1 - sl.c0);
                }
                const size_t lds_pitch = (size_t)(16 + 4 * ((ws + 63) / 64) + 1) * 16;
                auto tile_bytes = [&](int R) { return (size_t)(std::min(hs + 2 * R - 1, kMfChunkH) + 6 * R) * lds_pitch; };
                while (nt < 16 && tile_bytes(16 / nt) > 72 * 1024) nt <<= 1;
                sc.slab_nt = nt;
                sc.slab_R = 16 / nt;
            }
        }
        sc.r2 = (c->mfma_r2 && class_kernel[k] == MTM_KERNEL_MFMA && sc.rm_R == 0 && sc.slabs.empty() && n_cls > 16 &&
                 sc.w <= 64 && c->chans == 1 && !sc.masked && c->method >= MTM_TM_CCORR && c->fuse_stats) ? 2 : 0;
        // MTM_MFMA_R2=3: three rows per wave where they fit - one LDS tile for all h + 2 steps, and two work-groups per CU
        // (<= 80 KB each, the fused extremum's keys included).  48 MFMAs then share the operand shifts of a step instead of
        // 32: +5 % on the K step in isolation (tools/ubench/step3), 3 % fewer cycles in the kernel - and 0.5 % less time,
        // because the chip is at its power budget on random operands and answers with a 2.4 % lower clock
        // (tools/ubench/power: the pure MFMA stream itself runs at 2.0 instead of 2.4 GHz on such data); its larger work
        // items also quantise worse on the short launches of a banded upload.  Not the default.
        if (sc.r2 && c->mfma_r2 == 3 && sc.h + 2 <= kMfChunkR2) {
            const size_t lds3 = mfma_lds_bytes(sc.h + 2 + (kMfRows - 1) * 3, (sc.w + 63) / 64, (size_t)kMfRows * 2 * 1024 + 1024);
            if (lds3 <= 80 * 1024) sc.r2 = 3;
        }
        // packed K: uint8 classes (one channel, masked or not; RGB) on the plain or row-multiplexed tiling whose width
        // leaves part of the last 64-tap block empty.  Replaces the two-row variant where both apply (that one saves template loads,
        // this one whole MFMA steps).
        sc.kp_nseg = 0;
        {
            const int nseg = (sc.w + 15) / 16;
            const bool normed = c->method == MTM_TM_SQDIFF_NORMED || c->method == MTM_TM_CCORR_NORMED ||
                                c->method == MTM_TM_CCOEFF_NORMED;       // the instantiated variants (ncc_mfma_kernel<.., KP>)
            if (c->kpack && class_kernel[k] == MTM_KERNEL_MFMA && sc.slabs.empty() && nseg % 4 != 0 && normed &&
                (c->chans == 1 || (c->chans == 3 && !sc.masked)) && (!sc.masked || c->method != MTM_TM_CCOEFF_NORMED)) {
                sc.kp_nseg = nseg;
                sc.r2 = false;
            }
        }
    }
    for (int i = 0; i < n; ++i) {
        const HostTempl& t = c->templs[i];
        const int kern = class_kernel[(size_t)t.cls];
        // float64 weights: the float64 / naive kernels, and the exact re-scoring behind the bf16 kernel
        const bool want_f64 = kern == MTM_KERNEL_AUTO || kern == MTM_KERNEL_NAIVE || kern == MTM_KERNEL_MFMA_F32;
        if (t.chans != c->chans) {
            set_error("template " + std::to_string(i) + " has a different channel count than the image");
            return MTM_E_INVALID;
        }
        if (t.rows > c->rows || t.cols > c->cols) {
            set_error("template " + std::to_string(i) + " is larger than the image");
            return MTM_E_INVALID;
        }
        TemplDev& d = td_host[i];
        for (int k = 0; k < kMaxChans; ++k) d.mean[k] = t.st.mean[k];
        d.templ_norm = t.st.templ_norm;
        d.templ_sum2 = t.st.templ_sum2;
        d.templ2_mask2_sum = t.st.templ2_mask2_sum;
        d.all_ones = t.st.all_ones;
        {
            double sum_t = t.sum_t;  // exact: integers
            if (t.dtype == MTM_U8 && !t.on_device)
                for (size_t k = 0; k < t.px.size(); ++k) sum_t += t.masked ? t.px[k] * t.mask[k] : t.px[k];
            d.mfma_k = 128.0 * sum_t - 16384.0 * (double)t.rows * (double)t.cols * (double)t.chans;
        }
        d.rows = t.rows;
        d.cols = t.cols;
        d.cls = t.cls;
        d.oh = c->rows - t.rows + 1;
        d.ow = c->cols - t.cols + 1;
        d.map_pitch = (int)round_up((size_t)d.ow, 4);
        d.map_off = (long long)map_off;
        map_off += (size_t)d.map_pitch * d.oh;
        const size_t plane = (size_t)t.chans * t.rows * t.cols;
        d.k1_off = d.k2_off = -1;
        if (want_f64) {
            d.k1_off = (long long)w_off;
            w_off += plane;
            if (t.masked) {
                d.k2_off = (long long)w_off;
                w_off += plane;
            }
        }
        if (kern == MTM_KERNEL_DOT4 && img_u8 && t.dtype == MTM_U8 && !t.masked) {
            d.pack_off = (long long)p_off;
            p_off += dot_pack_bytes(t.rows, t.cols, t.chans);
        } else {
            d.pack_off = -1;
        }
    }
    // masked classes on the integer path: one dot4 pack of the (shared, binary) mask bytes per class
    for (size_t k = 0; k < classes.size(); ++k) {
        SizeClass& sc = classes[k];
        sc.masked_int = sc.masked && class_kernel[k] == MTM_KERNEL_MFMA;
        sc.mask_pack_off = -1;
        if (sc.masked_int && !(c->row_mux && c->fuse_stats)) {   // dot4 route of sum I^2 M (otherwise: matrix cores)
            sc.mask_pack_off = (long long)p_off;
            p_off += dot_pack_bytes(sc.h, sc.w, 1);
        }
    }
    // templates that live on the device but are matched by a kernel with host-packed operands: fetch their pixels
    for (int i = 0; i < n; ++i) {
        const int kern = class_kernel[(size_t)c->templs[i].cls];
        const bool mfma_dev = kern == MTM_KERNEL_MFMA && !(classes[(size_t)c->templs[i].cls].masked_int &&
                                                           classes[(size_t)c->templs[i].cls].mask_pack_off >= 0);
        if (c->templs[i].on_device && !mfma_dev) MTMC(ensure_host_pixels(c, i));
    }
    // weights (float64): K1 = T (or T*M^2), K2 = M^2
    std::vector<double> wts(w_off);
    std::vector<uint8_t> packs(p_off);
    for (int i = 0; i < n; ++i) {
        const HostTempl& t = c->templs[i];
        const TemplDev& d = td_host[i];
        const size_t plane = (size_t)t.chans * t.rows * t.cols;
        if (d.k1_off >= 0 && t.masked) {
            for (size_t k = 0; k < plane; ++k) {
                const double m2 = t.mask[k] * t.mask[k];
                wts[d.k1_off + k] = t.px[k] * m2;
                wts[d.k2_off + k] = m2;
            }
        } else if (d.k1_off >= 0) {
            std::copy(t.px.begin(), t.px.end(), wts.begin() + d.k1_off);
        }
        if (d.pack_off >= 0) pack_template_dot4(t, packs.data() + d.pack_off);
    }
    for (const SizeClass& sc : classes)
        if (sc.masked_int && sc.mask_pack_off >= 0) {
            HostTempl mk = c->templs[sc.members[0]];
            for (size_t k = 0; k < mk.px.size(); ++k) mk.px[k] = mk.mask[k] > 0.0 ? 255.0 : 0.0;
            pack_template_dot4(mk, packs.data() + sc.mask_pack_off);
        }
    // int8 MFMA packs, per eligible class
    size_t a_off = 0;
    for (size_t k = 0; k < classes.size(); ++k) {
        SizeClass& sc = classes[k];
        if (class_kernel[k] != MTM_KERNEL_MFMA) continue;
        sc.mask_rm_off = -1;
        if (sc.masked && c->row_mux && c->fuse_stats) {
            sc.mask_rm_off = (long long)a_off;
            a_off += (size_t)rm_pack_bytes(sc.h, sc.w, 16);
            const HostTempl& m0 = c->templs[sc.members[0]];
            sc.mask_ones = m0.on_device ? m0.mask_ones : 0.0;
            if (!m0.on_device)
                for (double m : m0.mask) sc.mask_ones += m > 0.0 ? 1.0 : 0.0;
        }
        if (sc.rm_R > 0) {
            sc.group_bytes = sc.kp_nseg ? class_rm_pack_bytes(sc) / 2          // packed K: a pack per MFMA group
                                        : -(long long)sc.rm_R * ((sc.w + 63) / 64) * 1024;
            sc.apack_off = (long long)a_off;
            a_off += (size_t)class_rm_pack_bytes(sc) * (sc.masked ? 1 : c->chans);
            continue;
        }
        if (!sc.slabs.empty()) {              // one pack per slab (a template of its own, one channel)
            sc.apack_off = (long long)a_off;
            for (auto& sl : sc.slabs) {
                sl.apack_off = (long long)a_off;
                const int hs = sl.r1 - sl.r0, ws = sl.c1 - sl.c0;
                a_off += sc.slab_R > 0 ? (size_t)rm_pack_bytes(hs, ws, sc.slab_R)
                                       : (size_t)mfma_group_bytes(hs, ws, 1) * mfma_groups_alloc((int)sc.members.size());
            }
            continue;
        }
        if (sc.r2) {
            sc.group_bytes = mfma_group_bytes(sc.h + sc.r2 - 1, sc.w, 1);        // h + r2 - 1 rows, the extra ones zero
            sc.apack_off = (long long)a_off;
            a_off += (size_t)sc.group_bytes * (((int)sc.members.size() + 15) / 16);
            continue;
        }
        sc.group_bytes = sc.kp_nseg ? (long long)c->chans * kp_blocks(sc.h, sc.kp_nseg) * 1024
                                    : mfma_group_bytes(sc.h, sc.w, c->chans);
        sc.apack_off = (long long)a_off;
        a_off += (size_t)sc.group_bytes * mfma_groups_alloc((int)sc.members.size());
    }
    for (size_t k = 0; k < classes.size(); ++k) {
        SizeClass& sc = classes[k];
        if (class_kernel[k] != MTM_KERNEL_MFMA_F32) continue;
        sc.group_bytes = bf16_group_bytes(sc.h, sc.w, c->chans);
        sc.apack_off = (long long)a_off;
        a_off += (size_t)(2 * sc.group_bytes * mfma_groups_alloc((int)sc.members.size()));
    }
    size_t ts_off = 0;
    for (size_t k = 0; k < classes.size(); ++k) {
        SizeClass& sc = classes[k];
        sc.tsum_off = -1;
        if (class_kernel[k] != MTM_KERNEL_MFMA16) continue;
        {   // packed K for the two byte-plane passes (widths that are not multiples of 64), as for uint8 classes
            const int nseg = (sc.w + 15) / 16;
            sc.kp_nseg = (c->kpack && nseg % 4 != 0) ? nseg : 0;
        }
        sc.group_bytes = sc.kp_nseg ? (long long)kp_blocks(sc.h, sc.kp_nseg) * 1024 : mfma_group_bytes(sc.h, sc.w, 1);
        sc.apack_off = (long long)a_off;
        a_off += (size_t)sc.group_bytes * (2 * sc.n_pad / 16);
        sc.tsum_off = (long long)ts_off;
        ts_off += 2 * (size_t)sc.n_pad;
    }
    // classes whose members all live on the device are packed there (after the uploads below)
    std::vector<char> dev_pack(classes.size(), 0);
    bool any_host_pack = false;
    for (size_t k = 0; k < classes.size(); ++k) {
        if (class_kernel[k] != MTM_KERNEL_MFMA && class_kernel[k] != MTM_KERNEL_MFMA16 && class_kernel[k] != MTM_KERNEL_MFMA_F32)
            continue;
        bool all_dev = class_kernel[k] == MTM_KERNEL_MFMA;
        for (int m : classes[k].members) all_dev = all_dev && c->templs[(size_t)m].on_device;
        dev_pack[k] = all_dev ? 1 : 0;
        any_host_pack = any_host_pack || !all_dev;
    }
    std::vector<uint8_t> apacks(any_host_pack ? a_off : 0);
    std::vector<double> tsums(ts_off);
    for (size_t k = 0; k < classes.size(); ++k) {
        if (dev_pack[k]) continue;
        if (class_kernel[k] == MTM_KERNEL_MFMA)
            for (int m : classes[k].members) MTMC(ensure_host_pixels(c, m));
        if (class_kernel[k] == MTM_KERNEL_MFMA && classes[k].mask_rm_off >= 0)
            pack_mask_rm(c, classes[k], apacks.data() + classes[k].mask_rm_off);
        if (class_kernel[k] == MTM_KERNEL_MFMA && classes[k].rm_R > 0)
            pack_class_rm(c, classes[k], apacks.data() + classes[k].apack_off);
        else if (class_kernel[k] == MTM_KERNEL_MFMA)
            pack_class_mfma(c, classes[k], apacks.data() + classes[k].apack_off);
        if (class_kernel[k] == MTM_KERNEL_MFMA16)
            pack_class_mfma16(c, classes[k], apacks.data() + classes[k].apack_off, tsums.data() + classes[k].tsum_off);
        if (class_kernel[k] == MTM_KERNEL_MFMA_F32) pack_class_bf16(c, classes[k], apacks.data() + classes[k].apack_off, td_host);
    }
    // template lists: one per class, then the list of templates with a 2-D score map
    for (SizeClass& sc : classes) {
        sc.tlist_off = (int)tlist_host.size();
        tlist_host.insert(tlist_host.end(), sc.members.begin(), sc.members.end());
    }
    for (int i = 0; i < n; ++i)
        if (td_host[i].oh > 1 && td_host[i].ow > 1) list2d.push_back(i);
    const int list2d_off = (int)tlist_host.size();
    tlist_host.insert(tlist_host.end(), list2d.begin(), list2d.end());
    // slab views: windows into the units of a large-template class, appended to the unit table (a view of a view:
    // the offsets move, the orientation stays)
    std::vector<UnitSrc> units_all(c->usrc_host.begin(), c->usrc_host.begin() + (long)std::min(c->usrc_units, c->usrc_host.size()));
    for (SizeClass& sc : classes)
        for (auto& sl : sc.slabs) {
            sl.tlist_off = (int)tlist_host.size();
            for (int m : sc.members) {
                UnitSrc v = c->templs[(size_t)m].src;
                v.off += (long long)sl.ch * v.sh * v.sw;
                v.moff = -1;
                v.cy += v.ay * sl.r0 + v.by * sl.c0;
                v.cx += v.ax * sl.r0 + v.bx * sl.c0;
                v.h = sl.r1 - sl.r0;
                v.w = sl.c1 - sl.c0;
                v.chans = 1;
                tlist_host.push_back((int)units_all.size());
                units_all.push_back(v);
            }
        }
    if (units_all.size() > c->usrc_units) {
        MTMC(c->usrc_dev.ensure(sizeof(UnitSrc) * units_all.size()));
        HIPC(hipMemcpyAsync(c->usrc_dev.p, units_all.data(), sizeof(UnitSrc) * units_all.size(), hipMemcpyHostToDevice, c->stream));
    }

    MTMC(c->td.ensure(sizeof(TemplDev) * n));
    MTMC(c->tlist.ensure(sizeof(int) * std::max<size_t>(1, tlist_host.size())));
    MTMC(c->weights.ensure(sizeof(double) * std::max<size_t>(1, w_off)));
    MTMC(c->packs.ensure(std::max<size_t>(4, p_off)));
    MTMC(c->apacks.ensure(std::max<size_t>(16, a_off) + 16384));     // the K loop requests up to two steps past a pack
    // host-packed classes (device-packed regions of the same arena are written afterwards, in stream order, by
    // pack_units_kernel below)
    if (a_off && any_host_pack) HIPC(hipMemcpyAsync(c->apacks.p, apacks.data(), a_off, hipMemcpyHostToDevice, c->stream));
    // the score-map arena (4 bytes per pixel and template) is only allocated when something writes maps:
    // mtm_find_matches in hits-only mode never does (ensure_maps, called by the launch paths)
    HIPC(hipMemcpyAsync(c->td.p, td_host.data(), sizeof(TemplDev) * n, hipMemcpyHostToDevice, c->stream));
    if (!tlist_host.empty())
        HIPC(hipMemcpyAsync(c->tlist.p, tlist_host.data(), sizeof(int) * tlist_host.size(),
                            hipMemcpyHostToDevice, c->stream));
    if (w_off) HIPC(hipMemcpyAsync(c->weights.p, wts.data(), sizeof(double) * w_off, hipMemcpyHostToDevice, c->stream));
    if (p_off) HIPC(hipMemcpyAsync(c->packs.p, packs.data(), p_off, hipMemcpyHostToDevice, c->stream));
    MTMC(c->tsum.ensure(sizeof(double) * std::max<size_t>(2, ts_off)));
    if (ts_off) HIPC(hipMemcpyAsync(c->tsum.p, tsums.data(), sizeof(double) * ts_off, hipMemcpyHostToDevice, c->stream));
    // device-side packing: gathers the A operands straight from the unit views (the template list is in place)
    for (size_t k = 0; k < classes.size(); ++k)
        if (dev_pack[k]) MTMC(pack_class_on_device(c, classes[k]));
    HIPC(hipStreamSynchronize(c->stream));   // host staging vectors go out of scope
    if (units_all.size() > c->usrc_units) c->usrc_host.swap(units_all);
    c->classes.swap(classes);
    c->td_host.swap(td_host);
    c->tlist_host.swap(tlist_host);
    c->list2d.swap(list2d);
    c->list2d_off = list2d_off;
    c->maps_floats = map_off;
    c->placed = true;
    return MTM_OK;
}

int resolved_kernel(const mtm_ctx* c, const SizeClass& sc);

// The float32 plane of the current image, if the upload skipped it (banded uint8 uploads do: single channel).
int ensure_f32_plane(mtm_ctx* c) {
    mtm_ctx::ImageSlot& sl = c->slot[c->cur];
    if (sl.f32_valid) return MTM_OK;
    const size_t n4 = (size_t)c->u8_pitch * c->rows_alloc / 4;         // pitch is a multiple of 64
    hipLaunchKernelGGL(u8_to_f32_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, c->stream, sl.u8.as<uint8_t>(),
                       sl.f32.as<float>(), n4);
    HIPC(hipGetLastError());
    sl.f32_valid = true;
    return MTM_OK;
}

// raw-mode instantiations of ncc_mfma_kernel (biased int32 accumulators stored as they are)
MfmaFn mfma_raw_fn(bool row_mux, bool packed_k) {
    MfmaSel s;
    s.method = kMfRaw;
    s.rm = row_mux;
    s.kp = packed_k;
    return mfma_kernel(s);
}

// Window statistics of one size class (two kernels), into c->stats.  Returns the plane table.
// `sb0`, `sb1`: range of kStatBand4-row output blocks to compute (banded image upload; fused single-channel
// kernel only), sb1 < 0 = all.
int launch_stats(mtm_ctx* c, const SizeClass& sc, StatPlanes* out, int sb0 = 0, int sb1 = -1) {
    const int h = sc.h, w = sc.w;
    const int oh = c->rows - h + 1, ow = c->cols - w + 1;
    const int method = c->method;
    StatPlanes st{};
    st.pitch = (int)round_up((size_t)ow, 4);
    *out = st;
    const int rk = resolved_kernel(c, sc);
    const bool want_t_always = rk == MTM_KERNEL_MFMA || rk == MTM_KERNEL_MFMA16 || rk == MTM_KERNEL_MFMA_F32;
    const bool masked_mfma = sc.masked && rk == MTM_KERNEL_MFMA;
    if ((sc.masked && !masked_mfma) || (method == MTM_TM_CCORR && !want_t_always)) return MTM_OK;   // none needed
    const int num_type = masked_mfma ? 0
                       : (method == MTM_TM_CCORR_NORMED) ? 0
                       : (method == MTM_TM_CCOEFF || method == MTM_TM_CCOEFF_NORMED) ? 1 : 2;
    const bool normed = !masked_mfma && (method == MTM_TM_SQDIFF_NORMED || method == MTM_TM_CCORR_NORMED ||
                                         method == MTM_TM_CCOEFF_NORMED);
    const size_t plane = (size_t)st.pitch * oh;
    MTMC(c->stats.ensure(sizeof(double) * plane * (kMaxChans + 2)));
    double* base = c->stats.as<double>();
    double* tp[kMaxChans];
    for (int k = 0; k < kMaxChans; ++k) tp[k] = base + plane * k;
    double* sum2 = base + plane * kMaxChans;
    double* sq = base + plane * (kMaxChans + 1);
    const int hs_pitch = st.pitch;
    const long long hs_plane = (long long)hs_pitch * c->rows;
    const bool u8 = c->dtype == MTM_U8;
    const size_t esz = u8 ? sizeof(uint32_t) : sizeof(double);
    MTMC(c->hs1.ensure(esz * hs_plane * c->chans));
    MTMC(c->hs2.ensure(esz * hs_plane * c->chans));
    const ImageDev img = image_dev(c);
    const dim3 g1((ow + 256 * kHsumSeg - 1) / (256 * kHsumSeg), c->rows, c->chans);
    const dim3 g2((ow + 255) / 256, (oh + kVsumBand - 1) / kVsumBand);
    const int want_t = (num_type == 1 || want_t_always) ? 1 : 0;
    const double inv_area = 1.0 / ((double)h * (double)w);
    // fused single-kernel statistics for the common case
    const bool fused_stats = u8 && c->chans == 1 && w <= 768 && (double)w * h * 65025.0 < 4294967296.0 &&
                             c->fuse_stats;
    if (fused_stats) {
        const int want_sum2 = (num_type == 2 || (normed && num_type != 1) || !want_t_always || masked_mfma) ? 1 : 0;
        const int owg = stats_u8_owg(w);
        const int nsb = (oh + kStatBand4 - 1) / kStatBand4;
        const int b1 = sb1 < 0 ? nsb : std::min(sb1, nsb);
        double* rsq = nullptr;
        if (sc.rm_R > 0 && normed) {
            MTMC(c->stats_rsq.ensure(sizeof(double) * plane));
            rsq = c->stats_rsq.as<double>();
        }
        double* blk = nullptr;
        if (sc.r2 > 0 && normed) {             // multi-row MFMA variants: statistic ranges per 16-pixel column block
            st.blk_pitch = (st.pitch + 15) / 16;
            MTMC(c->stats_blk.ensure(sizeof(double) * 4 * (size_t)st.blk_pitch * oh));
            blk = c->stats_blk.as<double>();
            st.blk = blk;
        }
        if (b1 > sb0) {
            const dim3 gs((ow + owg - 1) / owg, b1 - sb0);
            hipLaunchKernelGGL(stats_u8_kernel, gs, dim3(256), 0, c->stats_stream ? c->stats_stream : c->stream, img.u8,
                               img.u8_pitch, h, w, oh, ow, owg, inv_area, num_type, normed ? 1 : 0, want_t, want_sum2, tp[0],
                               sum2, sq, st.pitch, rsq, sb0, blk, st.blk_pitch);
        }
    } else if (u8 && c->chans == 3 && w <= 768 && 3.0 * w * h * 65025.0 < 4294967296.0 && c->fuse_stats) {
        // RGB: the fused kernel with one scan per channel + one for the squares (sum2 always written:
        // vsum_stats_kernel does)
        const int owg = stats_u8_owg(w);
        const dim3 gs((ow + owg - 1) / owg, (oh + kStatBand4 - 1) / kStatBand4);
        hipLaunchKernelGGL(stats_u8_mc_kernel<3>, gs, dim3(256), 0, c->stream, img.u8, img.u8_pitch, img.u8_plane, h, w, oh,
                           ow, owg, inv_area, num_type, normed ? 1 : 0, want_t, 1, tp[0], (long long)plane, sum2, sq,
                           st.pitch);
    } else if (u8) {
        if (c->cols <= 8191)
            hipLaunchKernelGGL(hsum_u8_kernel, dim3(c->rows, c->chans), dim3(256), sizeof(uint32_t) * 2 * (c->cols + 1),
                               c->stream, img.u8, img.u8_pitch, img.u8_plane, c->cols, w, ow, c->hs1.as<uint32_t>(),
                               c->hs2.as<uint32_t>(), hs_pitch, hs_plane);
        else {
            MTMC(ensure_f32_plane(c));
            hipLaunchKernelGGL(hsum_kernel<uint32_t>, g1, dim3(256), 0, c->stream, img.f32, img.f32_pitch,
                               img.f32_plane, c->rows, w, ow, c->hs1.as<uint32_t>(), c->hs2.as<uint32_t>(),
                               hs_pitch, hs_plane);
        }
        hipLaunchKernelGGL((vsum_stats_kernel<uint32_t, unsigned long long>), g2, dim3(256), 0, c->stream,
                           c->hs1.as<uint32_t>(), c->hs2.as<uint32_t>(), hs_pitch, hs_plane, c->chans, h, oh,
                           ow, inv_area, num_type, normed ? 1 : 0, want_t, tp[0], tp[1], tp[2], tp[3], sum2, sq,
                           st.pitch);
    } else {
        MTMC(ensure_f32_plane(c));
        hipLaunchKernelGGL(hsum_kernel<double>, g1, dim3(256), 0, c->stream, img.f32, img.f32_pitch,
                           img.f32_plane, c->rows, w, ow, c->hs1.as<double>(), c->hs2.as<double>(), hs_pitch,
                           hs_plane);
        hipLaunchKernelGGL((vsum_stats_kernel<double, double>), g2, dim3(256), 0, c->stream,
                           c->hs1.as<double>(), c->hs2.as<double>(), hs_pitch, hs_plane, c->chans, h, oh, ow,
                           inv_area, num_type, normed ? 1 : 0, want_t, tp[0], tp[1], tp[2], tp[3], sum2, sq, st.pitch);
    }
    HIPC(hipGetLastError());
    for (int k = 0; k < kMaxChans; ++k) st.t[k] = tp[k];
    st.sum2 = sum2;
    st.sq = sq;
    if (masked_mfma && sc.mask_rm_off >= 0 && fused_stats) {
        // sum I^2 * M over every window on the matrix cores (see square_planes_kernel): two row-multiplexed
        // raw correlations of the byte planes of I^2 with the mask, combined into the sum2 plane
        const size_t plane_bytes = (size_t)img.u8_plane;
        if (!c->sq_valid) {
            MTMC(c->sq_planes.ensure(3 * plane_bytes));
            const size_t n16 = plane_bytes / 16;
            hipLaunchKernelGGL(square_planes_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, c->stream, img.u8, n16,
                               c->sq_planes.as<uint8_t>(), c->sq_planes.as<uint8_t>() + plane_bytes,
                               c->sq_planes.as<uint8_t>() + 2 * plane_bytes);
            c->sq_valid = true;
        }
        MTMC(c->stats_hi.ensure(sizeof(double) * plane));
        {
            const int owg = stats_u8_owg(w);
            const dim3 gs((ow + owg - 1) / owg, (oh + kStatBand4 - 1) / kStatBand4);
            hipLaunchKernelGGL(stats_u8_kernel, gs, dim3(256), 0, c->stream, c->sq_planes.as<uint8_t>(), img.u8_pitch, h, w, oh,
                               ow, owg, inv_area, 0, 0, 1, 0, c->stats_hi.as<double>(), (double*)nullptr, (double*)nullptr,
                               st.pitch);
        }
        const int map_pitch = (int)round_up((size_t)ow, 4);
        const long long raw_map = (long long)oh * map_pitch;
        MTMC(c->raw16.ensure(sizeof(int) * (size_t)(2 * raw_map)));
        MfmaParams p{};
        p.pitch = img.u8_pitch;
        p.plane = img.u8_plane;
        p.chans = 1;
        p.h = h;
        p.w = w;
        p.oh = oh;
        p.ow = ow;
        p.nb = (w + 63) / 64;
        p.n_list = 1;
        p.rm_R = 16;
        p.rm_nt = 1;
        p.rm_log2nt = 0;
        p.rm_steps = h + 2 * 16 - 1;
        p.nseg = (ow + kMfSeg - 1) / kMfSeg;
        p.nyb = (oh + 8 * 16 - 1) / (8 * 16);
        p.ntg = 1;
        p.n_work = p.nseg * p.nyb;
        p.method = method;
        p.lds_pitch = (16 + 4 * p.nb + 1) * 16;
        p.cpr = p.lds_pitch / 16;
        p.cpr_rstep = 256 / p.cpr;
        p.cpr_dstep = 256 % p.cpr;
        p.cpr_magic = 65536 / p.cpr + 1;
        p.group_bytes = -(long long)16 * p.nb * 1024;
        p.only_li = -1;
        p.raw_map = raw_map;
        p.raw_pitch = map_pitch;
        const int tile_rows = std::min(p.rm_steps, kMfChunkH) + (kMfRows - 1) * 2 * 16;
        const size_t lds_main = (std::max<size_t>((size_t)tile_rows * p.lds_pitch, (size_t)kMfRows * kMfEpiBytesPerWave) + 15) &
                                ~(size_t)15;
        p.tc_off = (int)lds_main;
        p.st_off = (int)((lds_main + sizeof(MfTemplConst) * 32 + kMfItemBytes + 15) & ~(size_t)15);
        const size_t lds = (size_t)p.st_off;
        constexpr int kSchedWords = 1 + 4096;
        MTMC(c->sched.ensure(sizeof(unsigned int) * kSchedWords));
        const uint8_t* ap = c->apacks.as<uint8_t>() + sc.mask_rm_off + (long long)16 * p.nb * 1024;
        const int grid = ((p.n_work + 7) / 8) * 8;
        for (int x = 0; x < 2; ++x) {
            p.img = c->sq_planes.as<uint8_t>() + (size_t)(1 + x) * plane_bytes;
            p.raw_out = c->raw16.as<int>() + (size_t)x * raw_map;
            hipLaunchKernelGGL(mfma_raw_fn(true, false), dim3(grid), dim3(256), lds, c->stream, p,
                               c->td.as<TemplDev>(), c->tlist.as<int>(), ap, st, c->maps.as<float>(),
                               c->sched.as<unsigned int>());
        }
        const double km = 128.0 * sc.mask_ones - 16384.0 * (double)h * (double)w;
        hipLaunchKernelGGL(masksq_combine_kernel, dim3((ow + 255) / 256, oh), dim3(256), 0, c->stream, c->raw16.as<int>(),
                           c->raw16.as<int>() + raw_map, map_pitch, c->stats_hi.as<double>(), sum2, st.pitch, km, oh, ow);
        HIPC(hipGetLastError());
    } else if (masked_mfma) {
        // sum I^2 * M over every window: dot4 kernel with the mask bytes as the "template", into the
        // sum2 plane (overwrites the unmasked window sum of squares, which the masked path never uses)
        if (sc.mask_pack_off < 0) {
            set_error("internal: masked class without a dot4 mask pack");
            return MTM_E_STATE;
        }
        const DotVariant v = {4, 4, 1, false, ncc_dot4_kernel<4, 4, 1, false, true>};
        DotParams p{};
        p.img = img.u8;
        p.pitch = img.u8_pitch;
        p.plane = img.u8_plane;
        p.chans = 1;
        p.h = h;
        p.w = w;
        p.oh = oh;
        p.ow = ow;
        const int w4 = (w + 3) & ~3;
        p.ncy = (h + kDotChunk - 1) / kDotChunk;
        p.ncx = (w4 + kDotChunk - 1) / kDotChunk;
        p.n_list = 1;
        p.ntx = (ow + 32 * v.px - 1) / (32 * v.px);
        p.nty = (oh + 8 * v.py - 1) / (8 * v.py);
        p.nchunks = 1;
        p.n_work = p.ntx * p.nty;
        p.method = method;
        p.sumsq_out = sum2;
        // the kernel addresses its single template through td[tlist[0]].pack_off: point a scratch
        // TemplDev at the class's mask pack (only pack_off is read on the MASKSQ path)
        TemplDev mk = c->td_host[sc.members[0]];
        mk.pack_off = sc.mask_pack_off;
        MTMC(c->mask_td.ensure(sizeof(TemplDev) + sizeof(int)));
        HIPC(hipMemcpyAsync(c->mask_td.p, &mk, sizeof(TemplDev), hipMemcpyHostToDevice, c->stream));
        HIPC(hipMemsetAsync(c->mask_td.as<uint8_t>() + sizeof(TemplDev), 0, sizeof(int), c->stream));
        const int grid = ((p.n_work + 7) / 8) * 8;
        hipLaunchKernelGGL(v.fn, dim3(grid), dim3(256), 0, c->stream, p, c->mask_td.as<TemplDev>(),
                           reinterpret_cast<const int*>(c->mask_td.as<uint8_t>() + sizeof(TemplDev)),
                           c->packs.as<uint8_t>(), st, c->maps.as<float>());
        HIPC(hipGetLastError());
    }
    *out = st;
    return MTM_OK;
}

// Score maps of `n_list` templates of class `sc` (device list at tlist + list_off).
// `yb0`, `yb1`: range of output row blocks (MFMA kernel only; banded image upload), yb1 < 0 = all.
int launch_ncc(mtm_ctx* c, const SizeClass& sc, int list_off, int n_list, const StatPlanes& st, int only_li = -1,
               int yb0 = 0, int yb1 = -1) {
    const int h = sc.h, w = sc.w;
    const int oh = c->rows - h + 1, ow = c->cols - w + 1;
    const ImageDev img = image_dev(c);
    const int* tl = c->tlist.as<int>() + list_off;
    const TemplDev* td = c->td.as<TemplDev>();
    float* maps = c->maps.as<float>();
    const int kernel = resolved_kernel(c, sc);      // the same decision place_templates packed for

    // timing events around the dominant kernel
    if ((int)c->ncc_ev.size() <= c->timing.ncc_launches) {
        hipEvent_t a, b;
        HIPC(hipEventCreate(&a));
        HIPC(hipEventCreate(&b));
        c->ncc_ev.emplace_back(a, b);
    }
    auto& evp = c->ncc_ev[c->timing.ncc_launches];
    hipStream_t ncc_s = (c->ncc_stream && kernel == MTM_KERNEL_MFMA) ? c->ncc_stream : c->stream;
    HIPC(hipEventRecord(evp.first, ncc_s));

    if (kernel == MTM_KERNEL_NAIVE) {
        const dim3 blk(64, 4), grd((ow + 63) / 64, (oh + 3) / 4, n_list);
        MTMC(ensure_f32_plane(c));
        hipLaunchKernelGGL(ncc_naive_kernel, grd, blk, 0, c->stream, img, td, tl, c->weights.as<double>(), st,
                           c->method, sc.masked ? 1 : 0, maps);
        c->timing.kernel_used = MTM_KERNEL_NAIVE;
    } else if (kernel == MTM_KERNEL_MFMA && !sc.slabs.empty()) {
        // large templates: one RAW launch per slab (a template of its own against the image shifted by the slab's
        // offset), then slab_combine_kernel adds the slabs up, restores the bias terms and normalises
        const int n_all = (int)sc.members.size();
        const int map_pitch = (int)round_up((size_t)ow, 4);
        const long long raw_map = (long long)oh * map_pitch;
        const int S = (int)sc.slabs.size();
        MTMC(c->slab_raw.ensure(sizeof(int) * (size_t)S * n_all * (size_t)raw_map));
        constexpr int kSchedWords = 1 + 4096;
        MTMC(c->sched.ensure(sizeof(unsigned int) * kSchedWords));
        const bool rmr = sc.slab_R > 0;
        for (int k = 0; k < S; ++k) {
            const SizeClass::Slab& sl = sc.slabs[(size_t)k];
            const int hs = sl.r1 - sl.r0, ws = sl.c1 - sl.c0;
            MfmaParams p{};
            p.img = c->slot[c->cur].u8b.as<uint8_t>() + (size_t)sl.ch * img.u8_plane + (size_t)sl.r0 * img.u8_pitch + sl.c0;
            p.pitch = img.u8_pitch;
            p.plane = img.u8_plane;
            p.chans = 1;
            p.h = hs;
            p.w = ws;
            p.oh = oh;
            p.ow = ow;
            p.nb = (ws + 63) / 64;
            p.n_list = n_all;
            p.nseg = (ow + kMfSeg - 1) / kMfSeg;
            p.method = c->method;
            p.lds_pitch = (16 + 4 * p.nb + 1) * 16;
            p.cpr = p.lds_pitch / 16;
            p.cpr_rstep = 256 / p.cpr;
            p.cpr_dstep = 256 % p.cpr;
        p.cpr_magic = 65536 / p.cpr + 1;
            p.only_li = -1;
            p.raw_map = raw_map;
            p.raw_pitch = map_pitch;
            p.raw_out = c->slab_raw.as<int>() + (size_t)k * n_all * (size_t)raw_map;
            int tile_rows;
            if (rmr) {
                const int R = sc.slab_R;
                p.rm_R = R;
                p.rm_nt = sc.slab_nt;
                while ((1 << p.rm_log2nt) < sc.slab_nt) ++p.rm_log2nt;
                p.rm_steps = hs + 2 * R - 1;
                p.rm_cstride = rm_pack_bytes(hs, ws, R);
                p.nyb = (oh + 8 * R - 1) / (8 * R);
                p.ntg = 1;
                p.group_bytes = -(long long)R * p.nb * 1024;
                tile_rows = std::min(p.rm_steps, kMfChunkH) + (kMfRows - 1) * 2 * R;
            } else {
                p.nyb = (oh + kMfRows - 1) / kMfRows;
                p.ntg = (n_all + 31) / 32;
                p.group_bytes = mfma_group_bytes(hs, ws, 1);
                tile_rows = std::min(hs, kMfChunkH) + kMfRows - 1;
            }
            p.n_work = p.nseg * p.nyb * p.ntg;
            const size_t lds_main = (std::max<size_t>((size_t)tile_rows * p.lds_pitch, (size_t)kMfRows * kMfEpiBytesPerWave) + 15) &
                                    ~(size_t)15;
            p.tc_off = (int)lds_main;
            p.st_off = (int)((lds_main + sizeof(MfTemplConst) * 32 + kMfItemBytes + 15) & ~(size_t)15);
            const size_t lds = (size_t)p.st_off + (rmr ? 0 : (size_t)kMfRows * kMfStatBytesPerWave);
            const int grid = ((p.n_work + 7) / 8) * 8;
            const uint8_t* ap = c->apacks.as<uint8_t>() + sl.apack_off + (rmr ? (long long)sc.slab_R * p.nb * 1024 : 0);
            hipLaunchKernelGGL(mfma_raw_fn(rmr, false), dim3(grid), dim3(256), lds, c->stream, p, td,
                               c->tlist.as<int>() + sc.tlist_off, ap, st, maps, c->sched.as<unsigned int>());
        }
        SlabParams q{};
        q.raw = c->slab_raw.as<int>();
        q.raw_slab = (long long)n_all * raw_map;
        q.raw_map = raw_map;
        q.n_slabs = S;
        q.oh = oh;
        q.ow = ow;
        q.pitch = map_pitch;
        q.n_list = n_all;
        q.method = c->method;
        q.w = w;
        q.h = h;
        q.chans = c->chans;
        q.cand_on = (c->cand_on && only_li < 0) ? 1 : 0;
        q.cand_min = c->cand_min ? 1 : 0;
        q.cand_thr = c->cand_thr;
        q.cand_cap = (unsigned long long)std::min<int64_t>(c->hit_cap, 4096LL * 256);
        q.cand_counter = c->cands.as<unsigned long long>();
        q.cand_hits = reinterpret_cast<mtm_hit*>(c->cands.as<uint8_t>() + 16);
        q.hits_only = (q.cand_on && c->hits_only_now) ? 1 : 0;
        hipLaunchKernelGGL(slab_combine_kernel, dim3((ow + 255) / 256, oh, n_all), dim3(256), 0, c->stream, q, td,
                           c->tlist.as<int>() + sc.tlist_off, st, maps, only_li);
        c->timing.kernel_used = MTM_KERNEL_MFMA;
    } else if (kernel == MTM_KERNEL_MFMA) {
        // the MFMA kernel works on whole 16-template groups of the class list; a single-template
        // request (mtm_score_map) computes its group and stores only that template
        const int n_all = (int)sc.members.size();
        const bool rm = sc.rm_R > 0;
        const bool r2 = sc.r2 > 0;
        const int mb = r2 ? sc.r2 : (n_all > 16 || rm) ? 2 : 1;
        const int tgsz = r2 ? 16 : 16 * mb;          // templates per work item
        MfmaParams p{};
        p.img = c->slot[c->cur].u8b.as<uint8_t>();        // int8 view (bytes ^ 0x80), same geometry as img.u8
        p.pitch = img.u8_pitch;
        p.plane = img.u8_plane;
        p.chans = c->chans;
        p.h = h;
        p.w = w;
        p.oh = oh;
        p.ow = ow;
        p.nb = (w + 63) / 64;
        p.n_list = n_all;
        p.nseg = (ow + kMfSeg - 1) / kMfSeg;
        p.nyb = (oh + kMfRows - 1) / kMfRows;
        p.ntg = (n_all + tgsz - 1) / tgsz;
        if (r2) p.nyb = (oh + mb * kMfRows - 1) / (mb * kMfRows);
        p.method = c->method;
        p.lds_pitch = (16 + 4 * p.nb + 1) * 16;
        p.cpr = p.lds_pitch / 16;
        p.cpr_rstep = 256 / p.cpr;
        p.cpr_dstep = 256 % p.cpr;
        p.cpr_magic = 65536 / p.cpr + 1;
        p.group_bytes = sc.group_bytes;
        p.only_li = only_li;
        p.dbg = c->mfma_dbg;
        p.cand_on = (c->cand_on && only_li < 0) ? 1 : 0;
        p.hits_only = (p.cand_on && c->hits_only_now) ? 1 : 0;
        p.cand_thr_lo = (double)c->cand_thr - 1e-6 * std::max(1.0, std::fabs((double)c->cand_thr));
        p.screen_hi = std::min(p.cand_thr_lo, 0.999999) - 1e-6;
        p.sq_floor = 0.99 / std::sqrt((double)w * (double)h);
        p.screen_l1 = c->screen_l1;
        p.cand_min = c->cand_min ? 1 : 0;
        p.cand_thr = c->cand_thr;
        p.cand_cap = (unsigned long long)c->hit_cap;
        p.cand_counter = c->cands.as<unsigned long long>();
        p.cand_hits = reinterpret_cast<mtm_hit*>(c->cands.as<uint8_t>() + 16);
        // the 8 spare bytes of the candidate header carry the shader clock the kernel measured (fetched with it)
        p.clk_out = (p.cand_on && c->cands.p) ? reinterpret_cast<float*>(c->cands.as<uint8_t>() + 8) : nullptr;
        int tg0 = 0;
        if (only_li >= 0 && !rm) {   // one template: just its group
            tg0 = only_li / tgsz;
            p.ntg = 1;
        }
        int tile_rows = r2 ? std::min(h + mb - 1, kMfChunkR2) + (kMfRows - 1) * mb : std::min(h, kMfChunkH) + kMfRows - 1;
        if (rm) {
            p.rm_R = sc.rm_R;
            p.rm_nt = sc.rm_nt;
            p.rm_log2nt = 0;
            while ((1 << p.rm_log2nt) < sc.rm_nt) ++p.rm_log2nt;
            p.rm_steps = h + 2 * sc.rm_R - 1;
            p.rm_cstride = class_rm_pack_bytes(sc);
            p.rm_rsq = c->stats_rsq.as<double>();
            p.nyb = (oh + 8 * sc.rm_R - 1) / (8 * sc.rm_R);
            p.ntg = 1;
            tile_rows = std::min(p.rm_steps, kMfChunkH) + (kMfRows - 1) * 2 * sc.rm_R;
        }
        if (yb1 >= 0) {                     // banded launch: row blocks yb0 .. yb1 - 1 (the caller keeps the range non-empty)
            p.yb0 = yb0;
            p.nyb = std::min(yb1, p.nyb) - yb0;
        }
        p.n_work = p.nseg * p.nyb * p.ntg;
        const size_t lds_main = (std::max<size_t>((size_t)tile_rows * p.lds_pitch,
                                                  (size_t)kMfRows * kMfEpiBytesPerWave) + 15) & ~(size_t)15;
        p.tc_off = (int)lds_main;
        p.st_off = (int)((lds_main + sizeof(MfTemplConst) * 32 + kMfItemBytes + 15) & ~(size_t)15);
        // statistics prefetch region: (channels + 2) planes per wave (RM loads its statistics directly)
        size_t lds = (size_t)p.st_off + (rm ? 0 : r2 ? (size_t)kMfRows * ((mb + 1) / 2) * 1024
                                                     : (size_t)kMfRows * mf_stat_bytes_per_wave(c->chans == 3 ? 3 : 1));
        const bool ext = c->ext_now && only_li < 0;      // find_matches_impl checked the class
        if (ext) {
            p.ext_off = (int)lds;                         // 4 waves x 32 keys
            lds += (size_t)kMfRows * 32 * sizeof(unsigned long long);
            p.ext_best = c->counters.as<unsigned long long>();
            p.cand_on = 1;
            p.hits_only = 1;
        }
        const int grid = ((p.n_work + 7) / 8) * 8;
        const int* tl_class = c->tlist.as<int>() + sc.tlist_off;
        p.kp_nseg = sc.kp_nseg;
        p.kp_blocks = sc.kp_nseg ? kp_blocks(h, sc.kp_nseg) : 0;
        const uint8_t* ap = c->apacks.as<uint8_t>() + sc.apack_off +
                            (rm ? (sc.kp_nseg ? 0LL : (long long)sc.rm_R * p.nb * 1024)
                                : (long long)tg0 * (r2 ? 1 : mb) * sc.group_bytes);
        // with a group offset the kernel's list positions must stay class-relative: shift the list
        // pointer and the counts instead (positions inside the kernel are relative to tg0)
        p.n_list = n_all - tg0 * tgsz;
        if (only_li >= 0) p.only_li = only_li - tg0 * tgsz;
        const int* tl_k = tl_class + tg0 * tgsz;
        // the instantiation of ncc_mfma_kernel for this class (the kernels live in the mtm_mfma_*.hip units)
        MfmaSel sel;
        sel.mb = mb;
        sel.exact_div = c->exact_div != 0;
        sel.masked = sc.masked;
        sel.rm = rm;
        sel.ch = (c->chans == 3 && !sc.masked) ? 3 : 1;
        sel.method = (c->chans == 1 || sel.ch == 3) ? c->method : -1;     // other channel counts: the generic epilogue
        sel.ext = ext;
        sel.r2 = r2;
        sel.kp = sc.kp_nseg > 0;
        const MfmaFn fn = mfma_kernel(sel);
        if (!fn) {
            set_error("internal: no ncc_mfma_kernel instantiation for this class");
            return MTM_E_STATE;
        }
        // persistent launch: as many work-groups as stay co-resident; items via an atomic counter
        constexpr int kSchedWords = 1 + 4096;
        MTMC(c->sched.ensure(sizeof(unsigned int) * kSchedWords));
        p.persistent = c->mfma_persistent;
        int grid_launch = grid;
        if (p.persistent) {
            // residency query, cached per (kernel, LDS size): both calls are slow on the host
            int per_cu = 0;
            const auto key = std::make_pair(reinterpret_cast<const void*>(fn), lds);
            auto it = c->occupancy_cache.find(key);
            if (it == c->occupancy_cache.end()) {
                HIPC(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds));
                it = c->occupancy_cache.emplace(key, per_cu).first;
            }
            per_cu = it->second;
            if (c->n_cus == 0) {
                hipDeviceProp_t prop;
                HIPC(hipGetDeviceProperties(&prop, c->device));
                c->n_cus = prop.multiProcessorCount;
            }
            per_cu = std::max(1, std::min(per_cu, c->mfma_per_cu));
            p.stagger_mode = c->mfma_stagger_mode;
            grid_launch = std::min(p.n_work, per_cu * c->n_cus);
            // one main loop is chans*h*nb*16*MB MFMAs of 16 cycles; s_sleep(127) is ~8128 cycles
            const double main_cycles = (double)c->chans * h * p.nb * 16.0 * mb * 16.0;
            p.stagger_sleeps = c->mfma_stagger >= 0 ? c->mfma_stagger : (int)(0.75 * main_cycles / 8128.0 + 0.5);
            HIPC(hipMemsetAsync(c->sched.p, 0, sizeof(unsigned int) * kSchedWords, c->stream));
        }
        if (!p.persistent && c->mfma_stagger_np > 0) {
            if (c->n_cus == 0) {
                hipDeviceProp_t prop;
                HIPC(hipGetDeviceProperties(&prop, c->device));
                c->n_cus = prop.multiProcessorCount;
            }
            p.stagger_first = c->mfma_per_cu * c->n_cus;
            p.stagger_mode = c->mfma_stagger_mode;
            p.stagger_sleeps = c->mfma_stagger_np;
            HIPC(hipMemsetAsync(c->sched.p, 0, sizeof(unsigned int) * kSchedWords, c->stream));
        }
        hipLaunchKernelGGL(fn, dim3(grid_launch), dim3(256), lds, ncc_s, p, td, tl_k, ap, st, maps,
                           c->sched.as<unsigned int>());
        c->timing.kernel_used = MTM_KERNEL_MFMA;
    } else if (kernel == MTM_KERNEL_MFMA16) {
        // uint16: two launches over the image's byte planes x [T_hi | T_lo] of 16 templates per work item.  The first
        // (high bytes) stores its raw accumulators, the second (low bytes) reads them back in its epilogue and
        // finishes the exact 16-bit correlation + normalisation there (kMfU16).
        const int n_all = (int)sc.members.size(), n_pad = sc.n_pad;
        const int map_pitch = (int)round_up((size_t)ow, 4);
        const long long raw_map = (long long)oh * map_pitch;
        MTMC(c->raw16.ensure(sizeof(int) * (size_t)(2LL * n_pad * raw_map)));
        MfmaParams p{};
        p.pitch = img.u8_pitch;
        p.plane = img.u8_plane;
        p.chans = 1;
        p.h = h;
        p.w = w;
        p.oh = oh;
        p.ow = ow;
        p.nb = (w + 63) / 64;
        p.n_list = 2 * n_pad;
        p.nseg = (ow + kMfSeg - 1) / kMfSeg;
        p.nyb = (oh + kMfRows - 1) / kMfRows;
        p.ntg = n_pad / 16;
        p.method = c->method;
        p.lds_pitch = (16 + 4 * p.nb + 1) * 16;
        p.cpr = p.lds_pitch / 16;
        p.cpr_rstep = 256 / p.cpr;
        p.cpr_dstep = 256 % p.cpr;
        p.cpr_magic = 65536 / p.cpr + 1;
        p.group_bytes = sc.group_bytes;
        p.only_li = -1;
        p.raw_map = raw_map;
        p.raw_pitch = map_pitch;
        p.raw_out = c->raw16.as<int>();
        int tg0 = 0;
        if (only_li >= 0) {                 // one template: just its group of 16
            tg0 = only_li / 16;
            p.ntg = 1;
            p.raw_out += (size_t)32 * tg0 * raw_map;
        }
        p.n_work = p.nseg * p.nyb * p.ntg;
        const size_t lds_main = (std::max<size_t>((size_t)(std::min(h, kMfChunkH) + kMfRows - 1) * p.lds_pitch,
                                                  (size_t)kMfRows * kMfEpiBytesPerWave) + 15) & ~(size_t)15;
        p.tc_off = (int)lds_main;
        p.st_off = (int)((lds_main + sizeof(MfTemplConst) * 32 + kMfItemBytes + 15) & ~(size_t)15);
        const size_t lds = (size_t)p.st_off + (size_t)kMfRows * kMfStatBytesPerWave;
        const int grid = ((p.n_work + 7) / 8) * 8;
        constexpr int kSchedWords = 1 + 4096;
        MTMC(c->sched.ensure(sizeof(unsigned int) * kSchedWords));
        const uint8_t* planes = c->slot[c->cur].u8b.as<uint8_t>();
        const uint8_t* ap = c->apacks.as<uint8_t>() + sc.apack_off + (long long)tg0 * 2 * sc.group_bytes;
        const int* tl_k = c->tlist.as<int>() + sc.tlist_off + tg0 * 16;
        p.img = planes;                                         // high bytes: raw accumulators
        p.kp_nseg = sc.kp_nseg;
        p.kp_blocks = sc.kp_nseg ? kp_blocks(h, sc.kp_nseg) : 0;
        hipLaunchKernelGGL(mfma_raw_fn(false, sc.kp_nseg > 0), dim3(grid), dim3(256), lds, c->stream, p, td, tl_k, ap, st, maps,
                           c->sched.as<unsigned int>());
        p.img = planes + (size_t)img.u8_plane;                  // low bytes: finish
        p.n_list = n_all - tg0 * 16;                            // list positions inside the kernel are relative to tg0
        p.only_li = only_li >= 0 ? only_li - tg0 * 16 : -1;
        const double* ts = c->tsum.as<double>() + sc.tsum_off;
        p.u16_tsum = ts + tg0 * 16;
        p.u16_npad = n_pad;
        p.u16_area = (double)h * (double)w;
        p.cand_on = (c->cand_on && only_li < 0) ? 1 : 0;
        p.hits_only = (p.cand_on && c->hits_only_now) ? 1 : 0;
        p.cand_min = c->cand_min ? 1 : 0;
        p.cand_thr = c->cand_thr;
        p.cand_cap = (unsigned long long)std::min<int64_t>(c->hit_cap, 4096LL * 256);
        p.cand_counter = c->cands.as<unsigned long long>();
        p.cand_hits = reinterpret_cast<mtm_hit*>(c->cands.as<uint8_t>() + 16);
        size_t lds2 = lds;
        const bool ext = c->ext_now && only_li < 0;      // fused global extremum (find_matches_impl checked the classes)
        if (ext) {
            p.ext_off = (int)lds2;                        // 4 waves x 32 keys
            lds2 += (size_t)kMfRows * 32 * sizeof(unsigned long long);
            p.ext_best = c->counters.as<unsigned long long>();
            p.cand_on = 1;
            p.hits_only = 1;
        }
        MfmaSel sel16;
        sel16.method = kMfU16;
        sel16.kp = sc.kp_nseg > 0;
        sel16.ext = ext;
        sel16.exact_div = c->exact_div != 0;
        hipLaunchKernelGGL(mfma_kernel(sel16), dim3(grid), dim3(256), lds2, c->stream, p, td, tl_k, ap, st, maps,
                           c->sched.as<unsigned int>());
        c->timing.kernel_used = MTM_KERNEL_MFMA16;
    } else if (kernel == MTM_KERNEL_MFMA_F32 && !c->f32_exact_now) {
        const int n_all = (int)sc.members.size();
        const int mb = n_all > 16 ? 2 : 1;
        Bf16Params p{};
        p.img = img.f32;
        p.pitch = img.f32_pitch;
        p.plane = img.f32_plane;
        p.chans = c->chans;
        p.rows = c->rows;
        p.cols = c->cols;
        p.h = h;
        p.w = w;
        p.oh = oh;
        p.ow = ow;
        p.nkb = bf16_nkb(w);
        p.chunk_h = p.nkb <= 2 ? 64 : 32;
        p.lds_cols = kBfSeg + 32 * p.nkb;
        p.n_list = n_all;
        p.nseg = (ow + kBfSeg - 1) / kBfSeg;
        p.nyb = (oh + kBfRows - 1) / kBfRows;
        p.ntg = (n_all + 16 * mb - 1) / (16 * mb);
        p.method = c->method;
        p.group_bytes = sc.group_bytes;
        p.piece_bytes = sc.group_bytes * mfma_groups_alloc(n_all);
        p.only_li = only_li;
        int tg0 = 0;
        if (only_li >= 0) {
            tg0 = only_li / (16 * mb);
            p.ntg = 1;
            p.n_list = n_all - tg0 * 16 * mb;
            p.only_li = only_li - tg0 * 16 * mb;
        }
        p.n_work = p.nseg * p.nyb * p.ntg;
        p.cand_on = (c->cand_on && only_li < 0) ? 1 : 0;
        p.cand_min = c->cand_min ? 1 : 0;
        p.cand_thr = c->cand_thr;
        p.cand_cap = (unsigned long long)std::min<int64_t>(c->hit_cap, 4096LL * 256);
        p.cand_counter = c->cands.as<unsigned long long>();
        p.cand_hits = reinterpret_cast<mtm_hit*>(c->cands.as<uint8_t>() + 16);
        p.hits_only = (p.cand_on && c->hits_only_now) ? 1 : 0;
        if (c->ext_now && only_li < 0) {                  // fused global extremum (find_matches_impl checked the classes)
            p.ext_on = 1;
            p.ext_best = c->counters.as<unsigned long long>();
            p.cand_on = 1;
            p.hits_only = 1;
            p.ext_margin = c->refine_now ? kRefineThrMargin : 0.0f;
        }
        const size_t lds = bf16_lds_bytes(p.chunk_h, p.lds_cols);
        const int grid = ((p.n_work + 7) / 8) * 8;
        const uint8_t* ap = c->apacks.as<uint8_t>() + sc.apack_off + (long long)tg0 * mb * sc.group_bytes;
        const int* tl_k = c->tlist.as<int>() + sc.tlist_off + tg0 * 16 * mb;
        hipLaunchKernelGGL(bf16_kernel(mb), dim3(grid), dim3(256), lds, c->stream, p, td, tl_k, ap, st, maps);
        c->timing.kernel_used = MTM_KERNEL_MFMA_F32;
    } else if (kernel == MTM_KERNEL_DOT4) {
        const bool wide = (double)c->chans * w * h * 65025.0 >= 4294967296.0;
        const DotVariant& v = kDotVariants[wide ? kDotWideVariant : c->dot_variant];
        DotParams p{};
        p.img = img.u8;
        p.pitch = img.u8_pitch;
        p.plane = img.u8_plane;
        p.chans = c->chans;
        p.h = h;
        p.w = w;
        p.oh = oh;
        p.ow = ow;
        const int w4 = (w + 3) & ~3;
        p.ncy = (h + kDotChunk - 1) / kDotChunk;
        p.ncx = (w4 + kDotChunk - 1) / kDotChunk;
        p.n_list = n_list;
        p.ntx = (ow + 32 * v.px - 1) / (32 * v.px);
        p.nty = (oh + 8 * v.py - 1) / (8 * v.py);
        p.nchunks = (n_list + v.nt - 1) / v.nt;
        p.n_work = p.ntx * p.nty * p.nchunks;
        p.method = c->method;
        const int grid = ((p.n_work + 7) / 8) * 8;
        hipLaunchKernelGGL(v.fn, dim3(grid), dim3(256), 0, c->stream, p, td, tl, c->packs.as<uint8_t>(), st, maps);
        c->timing.kernel_used = MTM_KERNEL_DOT4;
    } else {
        const int ntx = (ow + kF64BX - 1) / kF64BX, nty = (oh + kF64BY - 1) / kF64BY;
        const dim3 grd(ntx * nty, n_list);
        MTMC(ensure_f32_plane(c));
        if (sc.masked)
            hipLaunchKernelGGL(ncc_f64_kernel<true>, grd, dim3(256), 0, c->stream, img, td, tl,
                               c->weights.as<double>(), st, c->method, maps, ntx);
        else
            hipLaunchKernelGGL(ncc_f64_kernel<false>, grd, dim3(256), 0, c->stream, img, td, tl,
                               c->weights.as<double>(), st, c->method, maps, ntx);
        if (c->timing.kernel_used == 0) c->timing.kernel_used = MTM_KERNEL_AUTO;
    }
    HIPC(hipGetLastError());
    HIPC(hipEventRecord(evp.second, ncc_s));
    c->timing.ncc_launches++;
    return MTM_OK;
}

int resolved_kernel(const mtm_ctx* c, const SizeClass& sc) {
    const bool dot_ok = c->dtype == MTM_U8 && sc.all_u8 && !sc.masked;
    int kernel = c->opt_kernel;
    if (c->dtype == MTM_F32) {
        if ((kernel == MTM_KERNEL_AUTO || kernel == MTM_KERNEL_MFMA) && sc.bf16_ok) return MTM_KERNEL_MFMA_F32;
        return kernel == MTM_KERNEL_NAIVE ? MTM_KERNEL_NAIVE : MTM_KERNEL_AUTO;
    }
    if (c->dtype == MTM_U16) {
        if ((kernel == MTM_KERNEL_AUTO || kernel == MTM_KERNEL_MFMA) && sc.mfma16_ok) return MTM_KERNEL_MFMA16;
        return kernel == MTM_KERNEL_NAIVE ? MTM_KERNEL_NAIVE : MTM_KERNEL_AUTO;
    }
    if (kernel == MTM_KERNEL_AUTO) kernel = c->auto_kernel;
    if (kernel == MTM_KERNEL_MFMA && (!sc.mfma_ok || (sc.masked && c->method > 3))) kernel = MTM_KERNEL_DOT4;
    if (kernel == MTM_KERNEL_DOT4 && !dot_ok) kernel = MTM_KERNEL_AUTO;
    return kernel;
}

int ensure_maps(mtm_ctx* c) { return c->maps.ensure(sizeof(float) * std::max<size_t>(4, c->maps_floats)); }

// float32 refinement (mtm_refine.hip.h): the records of the candidate buffer that belong to class `sc` get the scores
// of the exact float64 kernel - `ring`: their whole 3x3 neighbourhoods, written into the maps.  Runs right behind the
// class's score launch: the statistics planes are shared by all classes and only live until the next one starts.
int launch_refine(mtm_ctx* c, const SizeClass& sc, const StatPlanes& st, bool ring, bool patch_maps) {
    MTMC(ensure_f32_plane(c));
    RefineParams p{};
    p.img = image_dev(c);
    p.td = c->td.as<TemplDev>();
    p.weights = c->weights.as<double>();
    p.st = st;
    p.method = c->method;
    p.cls = (int)(&sc - c->classes.data());
    p.ring = ring ? 1 : 0;
    p.list = reinterpret_cast<mtm_hit*>(c->cands.as<uint8_t>() + 16);
    p.count = c->cands.as<unsigned long long>();
    p.cap = (unsigned long long)std::min<int64_t>(c->hit_cap, 4096LL * 256);
    p.maps = patch_maps ? c->maps.as<float>() : nullptr;
    // one wave per record, records strided over a grid that fills the chip a few times (the list length is only known
    // on the device; most calls list a few hundred records)
    const unsigned grid = (unsigned)std::min<unsigned long long>(p.cap, 8192ull);
    hipLaunchKernelGGL(refine_rescore_kernel, dim3(grid), dim3(64), 0, c->stream, p);
    HIPC(hipGetLastError());
    return MTM_OK;
}

// the map scan of the refined route: potential peaks of class `sc` (approximate maps in memory) -> candidate buffer
int launch_refine_scan(mtm_ctx* c, const SizeClass& sc) {
    const int oh = c->rows - sc.h + 1, ow = c->cols - sc.w + 1;
    const dim3 grd((ow + kPkCols - 1) / kPkCols, (oh + 4 * kPkRows - 1) / (4 * kPkRows), (unsigned)sc.members.size());
    hipLaunchKernelGGL(refine_scan_kernel, grd, dim3(256), 0, c->stream, c->maps.as<float>(), c->td.as<TemplDev>(),
                       c->tlist.as<int>() + sc.tlist_off, c->cand_min ? 1 : 0, c->cand_thr, kRefineNbrTol, c->opt_border,
                       reinterpret_cast<mtm_hit*>(c->cands.as<uint8_t>() + 16),
                       (unsigned long long)std::min<int64_t>(c->hit_cap, 4096LL * 256), c->cands.as<unsigned long long>());
    HIPC(hipGetLastError());
    return MTM_OK;
}

int run_score_all(mtm_ctx* c) {
    if (!c->hits_only_now) MTMC(ensure_maps(c));
    for (const SizeClass& sc : c->classes) {
        StatPlanes st;
        MTMC(launch_stats(c, sc, &st));
        MTMC(launch_ncc(c, sc, sc.tlist_off, (int)sc.members.size(), st));
        if (c->refine_now && !c->f32_exact_now && resolved_kernel(c, sc) == MTM_KERNEL_MFMA_F32) {
            if (c->refine_scan_now) {
                MTMC(launch_refine_scan(c, sc));
                MTMC(launch_refine(c, sc, st, true, true));
            } else {
                MTMC(launch_refine(c, sc, st, false, !c->hits_only_now));
            }
        }
    }
    if (c->refine_now && !c->f32_exact_now && c->ext_now) {
        // global extremum: the keys the score kernel kept are approximate - rebuild them from the re-scored list
        const size_t n = c->templs.size();
        HIPC(hipMemsetAsync(c->counters.p, 0, sizeof(unsigned long long) * 2 * n, c->stream));
        const unsigned long long cap = (unsigned long long)std::min<int64_t>(c->hit_cap, 4096LL * 256);
        hipLaunchKernelGGL(refine_extremum_kernel, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, c->stream,
                           reinterpret_cast<const mtm_hit*>(c->cands.as<uint8_t>() + 16), c->cands.as<unsigned long long>(), cap,
                           c->td.as<TemplDev>(), c->cand_min ? 1 : 0, c->counters.as<unsigned long long>());
        HIPC(hipGetLastError());
    }
    return MTM_OK;
}

// Time during which at least one score-kernel launch of the call was running: the launches of a banded call
// overlap (two compute streams), so their intervals are laid on the timeline of the first one and united.
int collect_ncc_time(mtm_ctx* c) {
    const int n = c->timing.ncc_launches;
    std::vector<std::pair<float, float>> iv;
    for (int i = 0; i < n; ++i) {
        float a = 0.f, d = 0.f;
        if (i > 0) HIPC(hipEventElapsedTime(&a, c->ncc_ev[0].first, c->ncc_ev[i].first));
        HIPC(hipEventElapsedTime(&d, c->ncc_ev[i].first, c->ncc_ev[i].second));
        iv.emplace_back(a, a + d);
    }
    std::sort(iv.begin(), iv.end());
    float total = 0.f, lo = 0.f, hi = -1.f, sum = 0.f;
    for (const auto& x : iv) sum += x.second - x.first;
    c->timing.ncc_sum_ms = sum;
    for (const auto& x : iv) {
        if (hi < lo || x.first > hi) {
            if (hi >= lo) total += hi - lo;
            lo = x.first;
            hi = x.second;
        } else {
            hi = std::max(hi, x.second);
        }
    }
    if (hi >= lo) total += hi - lo;
    c->timing.ncc_kernel_ms = total;
    return MTM_OK;
}

inline float decode_order(uint32_t o) {
    const uint32_t b = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    float v;
    std::memcpy(&v, &b, 4);
    return v;
}

}  // namespace

extern "C" {

int mtm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void* mtm_host_alloc(size_t bytes) {
    void* p = nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        set_error(std::string("mtm_host_alloc: ") + hipGetErrorString(e));
        return nullptr;
    }
    return p;
}

void mtm_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int mtm_ctx_create(mtm_ctx** out, int device_id) {
    if (!out) {
        set_error("mtm_ctx_create: null output");
        return MTM_E_INVALID;
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device visible (libmtm_hip has no CPU fallback)");
        return MTM_E_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= n) {
        set_error("device id out of range");
        return MTM_E_INVALID;
    }
    HIPC(hipSetDevice(device_id));
    mtm_ctx* c = new mtm_ctx();
    c->device = device_id;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreate(&c->ev[i]);
    if (e != hipSuccess) {
        set_error(std::string("context creation: ") + hipGetErrorString(e));
        delete c;
        return MTM_E_HIP;
    }
    if (const char* v = std::getenv("MTM_AUTO_KERNEL")) {
        if (!std::strcmp(v, "mfma")) c->auto_kernel = MTM_KERNEL_MFMA;
        if (!std::strcmp(v, "dot4")) c->auto_kernel = MTM_KERNEL_DOT4;
    }
    if (const char* v = std::getenv("MTM_UPLOAD_BANDS")) {      // e.g. "0.2,0.6,1": cumulative row fractions; "1": one piece
        std::vector<double> f;
        for (const char* q = v; *q;) {
            char* end = nullptr;
            const double x = std::strtod(q, &end);
            if (end == q) break;
            if (x > 0.0 && x <= 1.0 && (f.empty() || x > f.back())) f.push_back(x);
            q = *end == ',' ? end + 1 : end;
        }
        if (!f.empty()) {
            f.back() = 1.0;
            c->upload_bands = f;
        }
    }
    if (const char* v = std::getenv("MTM_COPY_PRIO")) c->copy_prio = std::atoi(v);
    if (const char* v = std::getenv("MTM_DUAL_STREAM")) c->dual_stream = std::atoi(v);
    if (const char* v = std::getenv("MTM_MFMA_R2")) c->mfma_r2 = std::atoi(v);
    if (const char* v = std::getenv("MTM_F32_MFMA")) c->f32_mfma = std::atoi(v);
    if (const char* v = std::getenv("MTM_SKIP_F32")) c->skip_f32 = std::atoi(v);
    if (const char* v = std::getenv("MTM_KPACK")) c->kpack = std::atoi(v);
    if (const char* v = std::getenv("MTM_SCREEN_L1")) c->screen_l1 = std::atoi(v);
    if (const char* v = std::getenv("MTM_COMM_TIMEOUT_S")) c->comm_timeout_s = std::atof(v);
    if (const char* v = std::getenv("MTM_SLAB_MFMA")) c->slab_mfma = std::atoi(v);
    if (const char* v = std::getenv("MTM_TEMPL_ON_DEVICE")) c->templ_on_device = std::atoi(v);
    if (const char* v = std::getenv("MTM_MFMA_DBG")) c->mfma_dbg = std::atoi(v);
    if (const char* v = std::getenv("MTM_FUSE_PEAKS")) c->fuse_peaks = std::atoi(v);
    if (const char* v = std::getenv("MTM_HITS_ONLY")) c->hits_only = std::atoi(v);
    if (const char* v = std::getenv("MTM_ROW_MUX")) c->row_mux = std::atoi(v);
    if (const char* v = std::getenv("MTM_FUSE_STATS")) c->fuse_stats = std::atoi(v);
    if (const char* v = std::getenv("MTM_EXACT_DIV")) c->exact_div = std::atoi(v);
    if (const char* v = std::getenv("MTM_MFMA_PERSISTENT")) c->mfma_persistent = std::atoi(v);
    if (const char* v = std::getenv("MTM_MFMA_STAGGER")) c->mfma_stagger = std::atoi(v);
    if (const char* v = std::getenv("MTM_MFMA_STAGGER_MODE")) c->mfma_stagger_mode = std::atoi(v);
    if (const char* v = std::getenv("MTM_MFMA_PER_CU")) c->mfma_per_cu = std::atoi(v);
    if (const char* v = std::getenv("MTM_MFMA_STAGGER_NP")) c->mfma_stagger_np = std::atoi(v);
    if (const char* v = std::getenv("MTM_DOT4_VARIANT")) {
        const int k = std::atoi(v);
        if (k >= 0 && k < kNumDotVariants && !kDotVariants[k].wide) c->dot_variant = k;
    }
    *out = c;
    return MTM_OK;
}

void mtm_ctx_destroy(mtm_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    mtm_comm_destroy(c);
    (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    for (auto& sl : c->slot)
        for (DevBuf* b : {&sl.raw, &sl.u8, &sl.u8b, &sl.f32}) b->release();
    for (DevBuf* b : {&c->tsrc, &c->usrc_dev, &c->tsums_dev, &c->tgather, &c->slab_raw}) b->release();
    for (DevBuf* b : {&c->td, &c->tlist, &c->weights, &c->packs, &c->apacks, &c->maps, &c->hs1, &c->hs2, &c->stats, &c->hits,
                      &c->counters, &c->sched, &c->cands, &c->mask_td, &c->chash, &c->raw16, &c->stats_hi, &c->tsum, &c->stats_rsq, &c->stats_blk, &c->sq_planes, &c->comm_send,
                      &c->comm_recv})
        b->release();
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->comm_pin) (void)hipHostFree(c->comm_pin);
    if (c->next_ready) (void)hipEventDestroy(c->next_ready);
    for (hipEvent_t e : c->band_ev) (void)hipEventDestroy(e);
    if (c->stream2_done) (void)hipEventDestroy(c->stream2_done);
    if (c->stream2) {
        (void)hipStreamSynchronize(c->stream2);
        (void)hipStreamDestroy(c->stream2);
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (auto& p : c->ncc_ev) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    for (int i = 0; i < 4; ++i)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int mtm_set_option(mtm_ctx* c, int option, int64_t value) {
    if (!c) return MTM_E_INVALID;
    MTM_NOT_IN_FLIGHT(c, "mtm_set_option");
    switch (option) {
        case MTM_OPT_KERNEL:
            if (value < MTM_KERNEL_AUTO || value > MTM_KERNEL_MFMA) break;
            if (c->opt_kernel != (int)value) c->placed = false;     // the packs follow the kernel
            c->opt_kernel = (int)value;
            return MTM_OK;
        case MTM_OPT_PEAK_BORDER:
            if (value != MTM_BORDER_CONSTANT && value != MTM_BORDER_NEAREST) break;
            c->opt_border = (int)value;
            return MTM_OK;
        case MTM_OPT_HIT_CAPACITY:
            if (value < 1) break;
            c->hit_cap = value;
            return MTM_OK;
        case MTM_OPT_EXACT_DIV:
            c->exact_div = value ? 1 : 0;
            return MTM_OK;
        case MTM_OPT_HITS_ONLY:
            c->hits_only = value ? 1 : 0;
            c->fuse_backoff = 0;
            c->backoff_len = 16;
            return MTM_OK;
        case MTM_OPT_F32_MFMA:
            if (value < 0 || value > 2) break;
            if ((c->f32_mfma != 0) != (value != 0)) c->placed = false;     // the packs follow the kernel
            c->f32_mfma = (int)value;
            return MTM_OK;
        case MTM_OPT_DOT4_VARIANT:
            if (value < 0 || value >= kNumDotVariants || kDotVariants[value].wide) break;
            c->dot_variant = (int)value;
            return MTM_OK;
        default: break;
    }
    set_error("mtm_set_option: bad option or value");
    return MTM_E_INVALID;
}

namespace {

// Geometry of the planar device copies of an image (after an optional integer downscale).
struct SlotGeom {
    int rows, cols, rows_alloc, pitch;
    size_t u8_bytes;
};

// Allocates the raw + planar buffers of `sl` for an image and (re)writes their padding when the geometry is new.
int prepare_slot(mtm_ctx* c, mtm_ctx::ImageSlot& sl, int src_rows, int src_cols, int chans, int dtype, hipStream_t stream,
                 int factor, SlotGeom* g) {
    (void)c;
    const size_t esz = elem_size(dtype);
    const size_t tight = (size_t)src_cols * chans * esz;
    MTMC(sl.raw.ensure(tight * src_rows));
    const int rows = src_rows / factor, cols = src_cols / factor;      // the planes hold the downscaled image
    const int rows_alloc = rows + kPadRows;
    const int pitch = (int)round_up((size_t)cols + kPadCols, 64);
    const size_t f32_bytes = sizeof(float) * pitch * rows_alloc * chans;
    const size_t u8_bytes = (size_t)pitch * rows_alloc * chans;
    const long long geom = (((long long)rows * 65536 + cols) * 8 + chans) * 4 + dtype;
    const size_t caps[3] = {sl.f32.cap, sl.u8.cap, sl.u8b.cap};
    MTMC(sl.f32.ensure(f32_bytes));
    // uint16, one channel: u8 = high-byte plane, u8b = [high ^ 0x80 plane][low ^ 0x80 plane]
    const bool u16_planes = dtype == MTM_U16 && chans == 1;
    const size_t u8b_bytes = u16_planes ? 2 * u8_bytes : u8_bytes;
    if (dtype == MTM_U8 || u16_planes) {
        MTMC(sl.u8.ensure(u8_bytes));
        MTMC(sl.u8b.ensure(u8b_bytes));
    }
    // the padding (zeros; 0x80 in the int8 view) only needs writing when the planes are new
    if (sl.geom != geom || caps[0] != sl.f32.cap || caps[1] != sl.u8.cap || caps[2] != sl.u8b.cap) {
        HIPC(hipMemsetAsync(sl.f32.p, 0, f32_bytes, stream));
        if (dtype == MTM_U8 || u16_planes) {
            HIPC(hipMemsetAsync(sl.u8.p, 0, u8_bytes, stream));
            HIPC(hipMemsetAsync(sl.u8b.p, 0x80, u8b_bytes, stream));
        }
        sl.geom = geom;
    }
    g->rows = rows;
    g->cols = cols;
    g->rows_alloc = rows_alloc;
    g->pitch = pitch;
    g->u8_bytes = u8_bytes;
    return MTM_OK;
}

// Rows [r0, r1) of a single-channel uint8 image: copy into the raw buffer and convert into the planes of `sl`
// (prepared by prepare_slot) on `stream`.  A pageable source makes the copy call block the host until the rows
// are staged; work queued on OTHER streams before the call runs under it.
int upload_rows_u8c1(mtm_ctx::ImageSlot& sl, const SlotGeom& g, const void* src, int64_t src_stride, int r0, int r1,
                     hipStream_t stream, bool skip_f32) {
    const int cols = g.cols, nrows = r1 - r0;
    if (nrows <= 0) return MTM_OK;
    uint8_t* raw = sl.raw.as<uint8_t>() + (size_t)r0 * cols;
    HIPC(hipMemcpy2DAsync(raw, (size_t)cols, (const uint8_t*)src + (size_t)r0 * src_stride, (size_t)src_stride, (size_t)cols,
                          nrows, hipMemcpyHostToDevice, stream));
    uint8_t* u8 = sl.u8.as<uint8_t>() + (size_t)r0 * g.pitch;
    uint8_t* u8b = sl.u8b.as<uint8_t>() + (size_t)r0 * g.pitch;
    // no float32 plane here: nothing in a banded call reads it (33 of the 50 MB this conversion would write at 4K);
    // ensure_f32_plane() makes it from the uint8 plane if a later call on this image needs it
    float* f32 = skip_f32 ? nullptr : sl.f32.as<float>() + (size_t)r0 * g.pitch;
    sl.f32_valid = !skip_f32;
    int x_begin = 0;
    if (cols >= 16) {        // 16 pixels per thread; the generic kernel takes the tail columns
        const int cols16 = cols / 16;
        hipLaunchKernelGGL(planarize_u8_c1_kernel, dim3((cols16 + 255) / 256, nrows), dim3(256), 0, stream, raw, nrows, cols,
                           cols16, u8, u8b, g.pitch, f32, g.pitch);
        x_begin = cols16 * 16;
    }
    if (x_begin < cols)
        hipLaunchKernelGGL(planarize_u8_kernel, dim3((cols - x_begin + 255) / 256, nrows), dim3(256), 0, stream, raw, nrows,
                           cols, 1, u8, u8b, g.pitch, (long long)g.pitch * g.rows_alloc, f32, g.pitch,
                           (long long)g.pitch * g.rows_alloc, x_begin);
    HIPC(hipGetLastError());
    return MTM_OK;
}

// Upload one image into `sl` and build its planar padded planes on `stream`.  `src` has tightly
// packed rows when `src_stride` == cols * chans * elem size or any larger stride.
int upload_image(mtm_ctx* c, mtm_ctx::ImageSlot& sl, const void* src, int64_t src_stride, int src_rows, int src_cols,
                 int chans, int dtype, hipStream_t stream, int factor = 1) {
    SlotGeom g{};
    MTMC(prepare_slot(c, sl, src_rows, src_cols, chans, dtype, stream, factor, &g));
    sl.f32_valid = true;
    const size_t tight = (size_t)src_cols * chans * elem_size(dtype);
    HIPC(hipMemcpy2DAsync(sl.raw.p, tight, src, (size_t)src_stride, tight, src_rows, hipMemcpyHostToDevice, stream));
    const int rows = g.rows, cols = g.cols, rows_alloc = g.rows_alloc, pitch = g.pitch;
    const size_t u8_bytes = g.u8_bytes;
    const bool u16_planes = dtype == MTM_U16 && chans == 1;
    const dim3 grd((cols + 255) / 256, rows);
    if (dtype == MTM_U16)
        hipLaunchKernelGGL(planarize_u16_kernel, grd, dim3(256), 0, stream, sl.raw.as<uint16_t>(), src_cols, chans, factor,
                           rows, cols, u16_planes ? sl.u8.as<uint8_t>() : (uint8_t*)nullptr, sl.u8b.as<uint8_t>(),
                           u16_planes ? sl.u8b.as<uint8_t>() + u8_bytes : (uint8_t*)nullptr, pitch, sl.f32.as<float>(),
                           pitch, (long long)pitch * rows_alloc);
    else if (factor > 1 && dtype == MTM_U8)
        hipLaunchKernelGGL(planarize_u8_down_kernel, grd, dim3(256), 0, stream, sl.raw.as<uint8_t>(), src_cols, chans,
                           factor, rows, cols, sl.u8.as<uint8_t>(), sl.u8b.as<uint8_t>(), pitch,
                           (long long)pitch * rows_alloc, sl.f32.as<float>(), pitch, (long long)pitch * rows_alloc);
    else if (factor > 1)
        hipLaunchKernelGGL(planarize_f32_down_kernel, grd, dim3(256), 0, stream, sl.raw.as<float>(), src_cols, chans,
                           factor, rows, cols, sl.f32.as<float>(), pitch, (long long)pitch * rows_alloc);
    else if (dtype == MTM_U8) {
        int x_begin = 0;
        if (chans == 1 && cols >= 16) {        // 16 pixels per thread; the generic kernel takes the tail columns
            const int cols16 = cols / 16;
            hipLaunchKernelGGL(planarize_u8_c1_kernel, dim3((cols16 + 255) / 256, rows), dim3(256), 0, stream,
                               sl.raw.as<uint8_t>(), rows, cols, cols16, sl.u8.as<uint8_t>(), sl.u8b.as<uint8_t>(), pitch,
                               sl.f32.as<float>(), pitch);
            x_begin = cols16 * 16;
        }
        if (x_begin < cols)
            hipLaunchKernelGGL(planarize_u8_kernel, dim3((cols - x_begin + 255) / 256, rows), dim3(256), 0, stream,
                               sl.raw.as<uint8_t>(), rows, cols, chans, sl.u8.as<uint8_t>(), sl.u8b.as<uint8_t>(), pitch,
                               (long long)pitch * rows_alloc, sl.f32.as<float>(), pitch, (long long)pitch * rows_alloc,
                               x_begin);
    }
    else
        hipLaunchKernelGGL(planarize_f32_kernel, grd, dim3(256), 0, stream, sl.raw.as<float>(), rows, cols, chans,
                           sl.f32.as<float>(), pitch, (long long)pitch * rows_alloc);
    HIPC(hipGetLastError());
    return MTM_OK;
}

void adopt_image(mtm_ctx* c, int rows, int cols, int chans, int dtype) {
    c->sq_valid = false;
    if (rows != c->rows || cols != c->cols || chans != c->chans || dtype != c->dtype) c->placed = false;
    c->rows = rows;
    c->cols = cols;
    c->chans = chans;
    c->dtype = dtype;
    c->rows_alloc = rows + kPadRows;
    c->u8_pitch = (int)round_up((size_t)cols + kPadCols, 64);
    c->f32_pitch = c->u8_pitch;
    c->have_image = true;
}

int check_image_args(const void* px, int rows, int cols, int chans, int dtype, int64_t row_stride_bytes,
                     const char* who) {
    if (!px || rows <= 0 || cols <= 0 || chans < 1 || chans > kMaxChans ||
        (dtype != MTM_U8 && dtype != MTM_F32 && dtype != MTM_U16)) {
        set_error(std::string(who) + ": bad arguments (1..4 channels, uint8, uint16 or float32)");
        return MTM_E_INVALID;
    }
    if (row_stride_bytes < (int64_t)((size_t)cols * chans * elem_size(dtype))) {
        set_error(std::string(who) + ": row stride smaller than a row");
        return MTM_E_INVALID;
    }
    return MTM_OK;
}

}  // namespace

int mtm_set_image_downscaled(mtm_ctx* c, const void* px, int rows, int cols, int chans, int dtype,
                             int64_t row_stride_bytes, int factor) {
    if (!c) {
        set_error("mtm_set_image: null context");
        return MTM_E_INVALID;
    }
    MTM_NOT_IN_FLIGHT(c, "mtm_set_image");
    MTMC(check_image_args(px, rows, cols, chans, dtype, row_stride_bytes, "mtm_set_image"));
    if (factor < 1 || factor > 64 || rows / factor < 1 || cols / factor < 1) {
        set_error("mtm_set_image_downscaled: factor must be in 1..64 and leave at least one pixel");
        return MTM_E_INVALID;
    }
    HIPC(hipSetDevice(c->device));
    MTMC(upload_image(c, c->slot[c->cur], px, row_stride_bytes, rows, cols, chans, dtype, c->stream, factor));
    HIPC(hipStreamSynchronize(c->stream));
    adopt_image(c, rows / factor, cols / factor, chans, dtype);
    return MTM_OK;
}

int mtm_set_image(mtm_ctx* c, const void* px, int rows, int cols, int chans, int dtype,
                  int64_t row_stride_bytes) {
    return mtm_set_image_downscaled(c, px, rows, cols, chans, dtype, row_stride_bytes, 1);
}

namespace {

// unit (y, x) -> source (row, col) of a view, composed from the augmentation steps (numpy semantics)
struct View {
    int ay = 1, by = 0, cy = 0, ax = 0, bx = 1, cx = 0, h = 0, w = 0;
    void fliplr() {                  // new(y, x) = cur(y, w - 1 - x)
        cy += by * (w - 1);
        cx += bx * (w - 1);
        by = -by;
        bx = -bx;
    }
    void flipud() {                  // new(y, x) = cur(h - 1 - y, x)
        cy += ay * (h - 1);
        cx += ax * (h - 1);
        ay = -ay;
        ax = -ax;
    }
    void rot90() {                   // np.rot90: new(i, j) = cur(j, w - 1 - i); new dims (w, h)
        const int nay = -by, nby = ay, ncy = cy + by * (w - 1);
        const int nax = -bx, nbx = ax, ncx = cx + bx * (w - 1);
        ay = nay; by = nby; cy = ncy;
        ax = nax; bx = nbx; cx = ncx;
        std::swap(h, w);
    }
};

inline unsigned long long fnv(unsigned long long h, const void* p, size_t n) {
    const uint8_t* b = (const uint8_t*)p;
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

// uint8 template sets, plain or augmented: the bases go to the device source arena (planar), resized copies are
// made there, every unit becomes a view (UnitSrc), statistics come from exact sums per source - on the host for
// the bases (one pass over the bytes that are being copied anyway), by a device reduction for resized copies.
// No per-unit pixel work on the host, no per-unit upload.
int set_templates_device(mtm_ctx* c, const mtm_templ* bases, int n_bases, const mtm_variant* variants, int n_var,
                         int method, std::vector<HostTempl>& hts) {
    HIPC(hipSetDevice(c->device));
    static const mtm_variant kIdentity = {0, 0, 0, 0, 0, 0};
    if (n_var <= 0) {
        variants = &kIdentity;
        n_var = 1;
    }
    struct Source {
        long long off = 0, moff = -1;
        int sh = 0, sw = 0, chans = 0;
        bool on_host = false;                       // sums computed on the host
        double s[kMaxChans] = {0, 0, 0, 0}, sq[kMaxChans] = {0, 0, 0, 0}, sm[kMaxChans] = {0, 0, 0, 0},
               sqm[kMaxChans] = {0, 0, 0, 0}, mones = 0.0;
        unsigned long long mask_hash = 0;
    };
    std::vector<Source> srcs;
    std::vector<uint8_t> stage;                     // host image of the arena prefix (the bases)
    auto alloc = [&](size_t& cursor, size_t bytes) {
        const size_t off = cursor;
        cursor = round_up(cursor + bytes, 16);
        return (long long)off;
    };
    size_t cursor = 0;
    // ---- bases: planarise into the staging image, exact sums on the way
    for (int b = 0; b < n_bases; ++b) {
        const mtm_templ& t = bases[b];
        Source sc;
        sc.sh = t.rows;
        sc.sw = t.cols;
        sc.chans = t.chans;
        sc.on_host = true;
        const size_t plane = (size_t)t.rows * t.cols, n = plane * t.chans;
        sc.off = alloc(cursor, n);
        if (t.mask) sc.moff = alloc(cursor, n);
        stage.resize(cursor);
        uint8_t* dpx = stage.data() + sc.off;
        uint8_t* dmk = t.mask ? stage.data() + sc.moff : nullptr;
        unsigned long long mh = 1469598103934665603ull;
        for (int ch = 0; ch < t.chans; ++ch) {
            unsigned long long s = 0, sq = 0, sm = 0, sqm = 0, ones = 0;
            for (int y = 0; y < t.rows; ++y) {
                const uint8_t* rp = (const uint8_t*)t.px + (size_t)y * t.row_stride + ch;
                const uint8_t* mp = t.mask ? (const uint8_t*)t.mask + (size_t)y * t.mask_row_stride + ch : nullptr;
                uint8_t* o = dpx + ch * plane + (size_t)y * t.cols;
                uint8_t* om = dmk ? dmk + ch * plane + (size_t)y * t.cols : nullptr;
                for (int x = 0; x < t.cols; ++x) {
                    const unsigned v = rp[(size_t)x * t.chans];
                    o[x] = (uint8_t)v;
                    s += v;
                    sq += v * v;
                    if (mp) {
                        // the arena keeps the mask bytes as given (a resized mask is the resize of THOSE bytes);
                        // every reader binarises: CV_8U masks are binary masks (matchTemplateMask)
                        om[x] = mp[(size_t)x * t.chans];
                        const unsigned m = om[x] > 0 ? 1u : 0u;
                        sm += v * m;
                        sqm += v * v * m;
                        ones += m;
                    }
                }
                if (om) mh = fnv(mh, om, (size_t)t.cols);
            }
            sc.s[ch] = (double)s;
            sc.sq[ch] = (double)sq;
            sc.sm[ch] = (double)sm;
            sc.sqm[ch] = (double)sqm;
            if (ch == 0) sc.mones = (double)ones;
        }
        sc.mask_hash = t.mask ? mh : 0ull;
        srcs.push_back(sc);
    }
    // ---- resized copies: one source per (base, resize rule), shared by the variants that use it
    struct Resize { int rows, cols, down; };
    std::vector<Resize> rules;
    std::vector<int> rule_of_var((size_t)n_var, -1);
    for (int v = 0; v < n_var; ++v) {
        const mtm_variant& q = variants[v];
        if (q.rot90 < 0 || q.rot90 > 3 || q.rows < 0 || q.cols < 0 || q.down < 0 || ((q.rows > 0) != (q.cols > 0)) ||
            (q.rows > 0 && q.down > 1)) {
            set_error("mtm_set_templates_augmented: bad variant " + std::to_string(v));
            return MTM_E_INVALID;
        }
        if (q.rows == 0 && q.down <= 1) continue;
        for (size_t r = 0; r < rules.size(); ++r)
            if (rules[r].rows == q.rows && rules[r].cols == q.cols && rules[r].down == (q.down > 1 ? q.down : 0)) rule_of_var[(size_t)v] = (int)r;
        if (rule_of_var[(size_t)v] < 0) {
            rule_of_var[(size_t)v] = (int)rules.size();
            rules.push_back(Resize{q.rows, q.cols, q.down > 1 ? q.down : 0});
        }
    }
    const size_t n_host_src = srcs.size();
    for (int b = 0; b < n_bases; ++b)
        for (size_t r = 0; r < rules.size(); ++r) {
            const Source& base = srcs[(size_t)b];
            Source sc;
            sc.chans = base.chans;
            sc.sh = rules[r].down ? base.sh / rules[r].down : rules[r].rows;
            sc.sw = rules[r].down ? base.sw / rules[r].down : rules[r].cols;
            if (sc.sh < 1 || sc.sw < 1) {
                set_error("mtm_set_templates_augmented: a resize leaves no pixel of base " + std::to_string(b));
                return MTM_E_INVALID;
            }
            const size_t n = (size_t)sc.sh * sc.sw * sc.chans;
            sc.off = alloc(cursor, n);
            if (base.moff >= 0) sc.moff = alloc(cursor, n);
            const int key[4] = {sc.sh, sc.sw, rules[r].down, 0};
            sc.mask_hash = base.moff >= 0 ? fnv(base.mask_hash, key, sizeof(key)) : 0ull;
            srcs.push_back(sc);
        }
    MTMC(c->tsrc.ensure(std::max<size_t>(16, cursor)));
    if (!stage.empty()) HIPC(hipMemcpyAsync(c->tsrc.p, stage.data(), stage.size(), hipMemcpyHostToDevice, c->stream));
    uint8_t* arena = c->tsrc.as<uint8_t>();
    for (int b = 0; b < n_bases; ++b)
        for (size_t r = 0; r < rules.size(); ++r) {
            const Source& base = srcs[(size_t)b];
            const Source& d = srcs[n_host_src + (size_t)b * rules.size() + r];
            const dim3 grd((d.sw + 63) / 64, d.sh, d.chans);
            for (int pass = 0; pass < (base.moff >= 0 ? 2 : 1); ++pass) {
                const uint8_t* sp = arena + (pass ? base.moff : base.off);
                uint8_t* dp = arena + (pass ? d.moff : d.off);
                if (rules[r].down)
                    hipLaunchKernelGGL(downscale_int_kernel, grd, dim3(64), 0, c->stream, sp, base.sh, base.sw, dp, rules[r].down,
                                       d.chans);
                else
                    hipLaunchKernelGGL(resize_area_kernel, grd, dim3(64), 0, c->stream, sp, base.sh, base.sw, dp, d.sh, d.sw,
                                       d.chans);
            }
        }
    // exact sums of the device-made sources: one reduction, one small copy back
    if (srcs.size() > n_host_src) {
        const size_t nd = srcs.size() - n_host_src;
        std::vector<SourceDesc> desc(nd);
        for (size_t k = 0; k < nd; ++k) {
            const Source& d = srcs[n_host_src + k];
            desc[k] = SourceDesc{d.off, d.moff, d.sh, d.sw, d.chans, 0};
        }
        MTMC(c->tsums_dev.ensure(sizeof(SourceDesc) * nd + sizeof(unsigned long long) * kSumsPerSource * nd));
        SourceDesc* ddesc = c->tsums_dev.as<SourceDesc>();
        unsigned long long* dsums = reinterpret_cast<unsigned long long*>(c->tsums_dev.as<uint8_t>() + sizeof(SourceDesc) * nd);
        HIPC(hipMemcpyAsync(ddesc, desc.data(), sizeof(SourceDesc) * nd, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(source_sums_kernel, dim3((unsigned)nd), dim3(256), 0, c->stream, arena, ddesc, dsums);
        HIPC(hipGetLastError());
        std::vector<unsigned long long> sums((size_t)kSumsPerSource * nd);
        HIPC(hipMemcpyAsync(sums.data(), dsums, sizeof(unsigned long long) * sums.size(), hipMemcpyDeviceToHost, c->stream));
        HIPC(hipStreamSynchronize(c->stream));
        for (size_t k = 0; k < nd; ++k) {
            Source& d = srcs[n_host_src + k];
            const unsigned long long* q = sums.data() + k * kSumsPerSource;
            for (int ch = 0; ch < d.chans; ++ch) {
                d.s[ch] = (double)q[4 * ch + 0];
                d.sq[ch] = (double)q[4 * ch + 1];
                d.sm[ch] = (double)q[4 * ch + 2];
                d.sqm[ch] = (double)q[4 * ch + 3];
            }
            d.mones = (double)q[4 * kMaxChans];
        }
    }
    // ---- units: base-major, variants in the order given
    const int n_units = n_bases * n_var;
    hts.assign((size_t)n_units, HostTempl{});
    std::vector<UnitSrc> units((size_t)n_units);
    for (int b = 0; b < n_bases; ++b)
        for (int v = 0; v < n_var; ++v) {
            const mtm_variant& q = variants[v];
            const Source& sc = rule_of_var[(size_t)v] < 0 ? srcs[(size_t)b]
                                                          : srcs[n_host_src + (size_t)b * rules.size() + (size_t)rule_of_var[(size_t)v]];
            View vw;
            vw.h = sc.sh;
            vw.w = sc.sw;
            if (q.flip_lr) vw.fliplr();
            if (q.flip_ud) vw.flipud();
            for (int k = 0; k < q.rot90; ++k) vw.rot90();
            const int ui = b * n_var + v;
            UnitSrc& u = units[(size_t)ui];
            u = UnitSrc{sc.off, sc.moff, sc.sh, sc.sw, vw.ay, vw.by, vw.cy, vw.ax, vw.bx, vw.cx, vw.h, vw.w, sc.chans, 0};
            HostTempl& t = hts[(size_t)ui];
            t.rows = vw.h;
            t.cols = vw.w;
            t.chans = sc.chans;
            t.dtype = MTM_U8;
            t.masked = sc.moff >= 0;
            t.on_device = true;
            t.src = u;
            double tm2 = 0.0;
            for (int ch = 0; ch < sc.chans; ++ch) {
                t.sum_t += t.masked ? sc.sm[ch] : sc.s[ch];
                tm2 += sc.sqm[ch];
            }
            t.mask_ones = sc.mones;
            t.st = templ_stats_from_sums(sc.s, sc.sq, tm2, t.masked, t.rows, t.cols, t.chans, method);
            if (t.masked) {
                const int key[8] = {vw.ay, vw.by, vw.cy, vw.ax, vw.bx, vw.cx, vw.h, vw.w};
                t.mask_key = fnv(sc.mask_hash, key, sizeof(key));
                if (t.mask_key == 0) t.mask_key = 1;
            }
        }
    MTMC(c->usrc_dev.ensure(sizeof(UnitSrc) * std::max<size_t>(1, units.size())));
    if (!units.empty())
        HIPC(hipMemcpyAsync(c->usrc_dev.p, units.data(), sizeof(UnitSrc) * units.size(), hipMemcpyHostToDevice, c->stream));
    HIPC(hipStreamSynchronize(c->stream));      // `stage` goes out of scope
    c->usrc_units = units.size();
    c->usrc_host.swap(units);
    return MTM_OK;
}

int set_templates_impl(mtm_ctx* c, const mtm_templ* templs, int n_templ, const mtm_variant* variants, int n_var, int method,
                       const char* who);

}  // namespace

int mtm_set_templates(mtm_ctx* c, const mtm_templ* templs, int n_templ, int method) {
    return set_templates_impl(c, templs, n_templ, nullptr, 0, method, "mtm_set_templates");
}

int mtm_set_templates_augmented(mtm_ctx* c, const mtm_templ* bases, int n_bases, const mtm_variant* variants, int n_variants,
                                int method) {
    if (n_variants < 1 || !variants) {
        set_error("mtm_set_templates_augmented: at least one variant is needed");
        return MTM_E_INVALID;
    }
    return set_templates_impl(c, bases, n_bases, variants, n_variants, method, "mtm_set_templates_augmented");
}

namespace {

int set_templates_impl(mtm_ctx* c, const mtm_templ* templs, int n_templ, const mtm_variant* variants, int n_var, int method,
                       const char* who) {
    if (!c || n_templ < 0 || (n_templ > 0 && !templs) || method < 0 || method > 5) {
        set_error(std::string(who) + ": bad arguments");
        return MTM_E_INVALID;
    }
    MTM_NOT_IN_FLIGHT(c, who);
    bool all_u8 = true;
    for (int i = 0; i < n_templ; ++i) {
        const mtm_templ& s = templs[i];
        if (!s.px || s.rows <= 0 || s.cols <= 0 || s.chans < 1 || s.chans > kMaxChans ||
            (s.dtype != MTM_U8 && s.dtype != MTM_F32 && s.dtype != MTM_U16)) {
            set_error(std::string(who) + ": bad template " + std::to_string(i));
            return MTM_E_INVALID;
        }
        all_u8 = all_u8 && s.dtype == MTM_U8;
    }
    if (n_var > 0 && !all_u8) {
        set_error("mtm_set_templates_augmented takes uint8 bases (augment other pixel types on the host)");
        return MTM_E_INVALID;
    }
    // The same templates again (a loop of matchTemplates calls over different images): keep everything that
    // was derived from them - statistics, size classes, device packs.  The test is on the pixel bytes: the
    // caller's rows are compared in place with the copy kept from the call that built the current state.
    {
        auto walk = [&](auto&& emit) {
            emit(&n_templ, sizeof(n_templ));
            emit(&method, sizeof(method));
            emit(&n_var, sizeof(n_var));
            if (n_var > 0) emit(variants, sizeof(mtm_variant) * (size_t)n_var);
            for (int i = 0; i < n_templ; ++i) {
                const mtm_templ& s = templs[i];
                const int hdr[5] = {s.rows, s.cols, s.chans, s.dtype, s.mask ? 1 : 0};
                emit(hdr, sizeof(hdr));
                const size_t row = (size_t)s.cols * s.chans * elem_size(s.dtype);
                if (!s.mask && s.row_stride == (int64_t)row) {      // contiguous template: one piece
                    emit(s.px, row * s.rows);
                    continue;
                }
                for (int y = 0; y < s.rows; ++y) {
                    emit((const uint8_t*)s.px + (size_t)y * s.row_stride, row);
                    if (s.mask) emit((const uint8_t*)s.mask + (size_t)y * s.mask_row_stride, row);
                }
            }
        };
        if (c->have_templ) {
            size_t off = 0;
            bool same = true;
            const std::vector<uint8_t>& old = c->templ_blob;
            walk([&](const void* p, size_t n) {
                if (!same) return;
                if (off + n > old.size() || std::memcmp(old.data() + off, p, n) != 0) same = false;
                off += n;
            });
            if (same && off == old.size()) return MTM_OK;
        }
        std::vector<uint8_t> blob;
        size_t total = 0;
        walk([&](const void*, size_t n) { total += n; });
        blob.reserve(total);
        walk([&](const void* p, size_t n) { blob.insert(blob.end(), (const uint8_t*)p, (const uint8_t*)p + n); });
        c->templ_blob.swap(blob);
        c->have_templ = false;          // until the new set is complete
    }
    std::vector<HostTempl> hts;
    if (all_u8 && (c->templ_on_device || n_var > 0)) {       // augmented sets only exist as device views
        const int rc = set_templates_device(c, templs, n_templ, variants, n_var, method, hts);
        if (rc != MTM_OK) {
            c->templ_blob.clear();
            return rc;
        }
        n_templ = (int)hts.size();
    } else {
    hts.assign((size_t)n_templ, HostTempl{});
    for (int i = 0; i < n_templ; ++i) {
        const mtm_templ& s = templs[i];
        HostTempl& t = hts[i];
        t.rows = s.rows;
        t.cols = s.cols;
        t.chans = s.chans;
        t.dtype = s.dtype;
        t.masked = s.mask != nullptr;
        const size_t plane = (size_t)s.rows * s.cols;
        t.px.resize(plane * s.chans);
        if (t.masked) t.mask.resize(plane * s.chans);
        for (int y = 0; y < s.rows; ++y) {
            const uint8_t* rp = (const uint8_t*)s.px + (size_t)y * s.row_stride;
            const uint8_t* mp = t.masked ? (const uint8_t*)s.mask + (size_t)y * s.mask_row_stride : nullptr;
            for (int x = 0; x < s.cols; ++x)
                for (int k = 0; k < s.chans; ++k) {
                    const size_t src = (size_t)x * s.chans + k;
                    const size_t dst = (size_t)k * plane + (size_t)y * s.cols + x;
                    if (s.dtype == MTM_U8) {
                        t.px[dst] = (double)rp[src];
                        // CV_8U masks are binary masks (matchTemplateMask)
                        if (mp) t.mask[dst] = mp[src] > 0 ? 1.0 : 0.0;
                    } else if (s.dtype == MTM_U16) {
                        // the reference casts uint16 to float32 (exact) before cv2 (MTM/__init__.py:71-74):
                        // a mask is then a float32 weight image, not a binary mask
                        t.px[dst] = (double)((const uint16_t*)rp)[src];
                        if (mp) t.mask[dst] = (double)((const uint16_t*)mp)[src];
                    } else {
                        t.px[dst] = (double)((const float*)rp)[src];
                        if (mp) t.mask[dst] = (double)((const float*)mp)[src];
                    }
                }
        }
        t.st = compute_templ_stats(t.px.data(), t.masked ? t.mask.data() : nullptr, t.rows, t.cols, t.chans,
                                   method, s.dtype == MTM_U8 || s.dtype == MTM_U16);
    }
    }   // host path
    // size classes, in order of first appearance
    std::vector<SizeClass> classes;
    // masked templates only share a class (and its masked window statistics) when their masks are equal
    auto mask_hash = [](const HostTempl& t) {
        unsigned long long hsh = 1469598103934665603ull;
        for (double v : t.mask) {
            unsigned long long bits;
            std::memcpy(&bits, &v, 8);
            hsh = (hsh ^ bits) * 1099511628211ull;
        }
        if (t.on_device) return t.masked ? t.mask_key : 0ull;
        return t.masked ? hsh : 0ull;
    };
    std::map<std::tuple<int, int, bool, unsigned long long>, int> index;
    for (int i = 0; i < n_templ; ++i) {
        const auto key = std::make_tuple(hts[i].rows, hts[i].cols, hts[i].masked, mask_hash(hts[i]));
        auto it = index.find(key);
        if (it == index.end()) {
            SizeClass sc;
            sc.h = hts[i].rows;
            sc.w = hts[i].cols;
            sc.masked = hts[i].masked;
            sc.mask_hash = std::get<3>(key);
            it = index.emplace(key, (int)classes.size()).first;
            classes.push_back(sc);
        }
        SizeClass& sc = classes[it->second];
        sc.members.push_back(i);
        sc.all_u8 = sc.all_u8 && hts[i].dtype == MTM_U8;
        sc.all_u16 = sc.all_u16 && hts[i].dtype == MTM_U16;
        sc.all_f32 = sc.all_f32 && hts[i].dtype == MTM_F32;
        hts[i].cls = it->second;
    }
    c->templs.swap(hts);
    c->classes.swap(classes);
    c->method = method;
    c->have_templ = true;
    c->placed = false;
    return MTM_OK;
}

}  // namespace

int mtm_score_map(mtm_ctx* c, int templ_idx, float* out, int64_t out_row_stride_bytes) {
    if (!c || !out) {
        set_error("mtm_score_map: bad arguments");
        return MTM_E_INVALID;
    }
    MTM_NOT_IN_FLIGHT(c, "mtm_score_map");
    HIPC(hipSetDevice(c->device));
    MTMC(place_templates(c));
    if (templ_idx < 0 || templ_idx >= (int)c->templs.size()) {
        set_error("mtm_score_map: template index out of range");
        return MTM_E_INVALID;
    }
    const TemplDev& d = c->td_host[templ_idx];
    if (out_row_stride_bytes < (int64_t)(sizeof(float) * d.ow)) {
        set_error("mtm_score_map: output row stride too small");
        return MTM_E_INVALID;
    }
    const SizeClass& sc = c->classes[c->templs[templ_idx].cls];
    // position of the template inside its class list
    int pos = 0;
    while (sc.members[pos] != templ_idx) ++pos;
    c->timing = mtm_timing{};
    StatPlanes st;
    MTMC(ensure_maps(c));
    MTMC(launch_stats(c, sc, &st));
    MTMC(launch_ncc(c, sc, sc.tlist_off + pos, 1, st, pos));
    HIPC(hipMemcpy2DAsync(out, (size_t)out_row_stride_bytes, c->maps.as<float>() + d.map_off,
                          sizeof(float) * d.map_pitch, sizeof(float) * d.ow, d.oh, hipMemcpyDeviceToHost,
                          c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    MTMC(collect_ncc_time(c));
    return MTM_OK;
}

namespace {

struct NextImage {
    const void* px;
    int rows, cols, chans, dtype;
    int64_t stride;
    bool staged;
};

// the image of a fused "upload + search" call (mtm_find_matches_image)
struct ImageArgs {
    const void* px;
    int rows, cols, chans, dtype;
    int64_t stride;
};

// The producer side of the upload pipelines (copies, layout conversion, window statistics of a band): its short
// kernels must not queue behind the score kernel's work-groups for a free CU, hence the highest stream priority.
int ensure_copy_stream(mtm_ctx* c) {
    if (c->copy_stream) return MTM_OK;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    HIPC(hipStreamCreateWithPriority(&c->copy_stream, hipStreamNonBlocking, c->copy_prio ? hi : 0));
    return MTM_OK;
}

// Enqueue the upload + plane conversion of the next image of a stream on the copy stream, into the
// image slot the kernels are not reading.  Called by find_matches_impl after the kernels of the
// current image are enqueued and before it waits for them: the PCIe transfer (and the host-side
// staging the runtime does for pageable memory) runs under the kernels.
int stage_next_image(mtm_ctx* c, NextImage* nx) {
    if (!nx || nx->staged) return MTM_OK;
    MTMC(ensure_copy_stream(c));
    if (!c->next_ready) HIPC(hipEventCreateWithFlags(&c->next_ready, hipEventDisableTiming));
    // The runtime batches stream commands and only submits them when somebody asks about the stream:
    // push the kernels of the current image out first, then (below) the copy, so that they overlap.
    (void)hipStreamQuery(c->stream);
    // straight from the caller's (pageable) rows: the runtime stages them through its own pinned
    // buffers, which measured 5x faster than a host copy into hipHostMalloc memory on this platform
    MTMC(upload_image(c, c->slot[1 - c->cur], nx->px, nx->stride, nx->rows, nx->cols, nx->chans, nx->dtype,
                      c->copy_stream));
    HIPC(hipEventRecord(c->next_ready, c->copy_stream));
    (void)hipStreamQuery(c->copy_stream);
    nx->staged = true;
    return MTM_OK;
}

int find_matches_impl(mtm_ctx* c, int mode, double score_threshold, mtm_hit* out, int64_t capacity,
                      int64_t* n_out, NextImage* next, const ImageArgs* up = nullptr);

}  // namespace

int mtm_find_matches(mtm_ctx* c, int mode, double score_threshold, mtm_hit* out, int64_t capacity,
                     int64_t* n_out) {
    return find_matches_impl(c, mode, score_threshold, out, capacity, n_out, nullptr);
}

int mtm_find_matches_image(mtm_ctx* c, const void* px, int rows, int cols, int chans, int dtype, int64_t row_stride_bytes,
                           int mode, double score_threshold, mtm_hit* out, int64_t capacity, int64_t* n_out) {
    if (!c) {
        set_error("mtm_find_matches_image: null context");
        return MTM_E_INVALID;
    }
    MTMC(check_image_args(px, rows, cols, chans, dtype, row_stride_bytes, "mtm_find_matches_image"));
    const ImageArgs up{px, rows, cols, chans, dtype, row_stride_bytes};
    return find_matches_impl(c, mode, score_threshold, out, capacity, n_out, nullptr, &up);
}

int mtm_find_matches_next(mtm_ctx* c, int mode, double score_threshold, mtm_hit* out, int64_t capacity,
                          int64_t* n_out, const void* next_px, int rows, int cols, int chans, int dtype,
                          int64_t row_stride_bytes) {
    if (!c) {
        set_error("mtm_find_matches_next: null context");
        return MTM_E_INVALID;
    }
    MTMC(check_image_args(next_px, rows, cols, chans, dtype, row_stride_bytes, "mtm_find_matches_next"));
    NextImage nx{next_px, rows, cols, chans, dtype, row_stride_bytes, false};
    const int rc = find_matches_impl(c, mode, score_threshold, out, capacity, n_out, &nx);
    if (rc != MTM_OK && rc != MTM_E_OVERFLOW) {
        if (nx.staged) (void)hipStreamSynchronize(c->copy_stream);   // drop the staged image
        return rc;
    }
    // the results of the current image are final: make the staged image current
    if (!nx.staged) MTMC(stage_next_image(c, &nx));
    HIPC(hipEventSynchronize(c->next_ready));
    c->cur = 1 - c->cur;
    adopt_image(c, rows, cols, chans, dtype);
    return rc;
}

namespace {

constexpr size_t kHitPrefetch = 1024;       // candidate / hit records fetched together with the counters

// Asynchronous half of mtm_find_matches: statistics, score kernels, the upload of the next image and (usual
// case) the copy of the candidate list into pinned memory are queued; nothing waits for the GPU.
// Can the image of a fused call arrive in row bands (copy / layout / statistics of band k+1 under the score
// kernel of band k)?  One unmasked single-channel uint8 size class on the MFMA kernel with the fused
// statistics kernel; anything else uploads the image in one piece (still without a round trip to the host).
bool banded_ok(const mtm_ctx* c, const ImageArgs& a) {
    if (c->upload_bands.size() < 2 || a.dtype != MTM_U8 || a.chans != 1 || c->classes.size() != 1) return false;
    const SizeClass& sc = c->classes[0];
    if (sc.masked || !c->fuse_stats || !sc.slabs.empty() || resolved_kernel(c, sc) != MTM_KERNEL_MFMA) return false;
    if (!(sc.w <= 768 && (double)sc.w * sc.h * 65025.0 < 4294967296.0)) return false;
    return (size_t)a.rows * a.cols >= ((size_t)1 << 20) && a.rows - sc.h + 1 >= 256;
}

// The score pass of a fused call with a banded upload.  copy_stream: per band the rows' copy, their layout
// conversion and the window statistics of the output rows that became computable; c->stream: the score kernel
// over the row blocks whose statistics exist, behind the band's event.  With a pageable source every copy call
// blocks the host while its rows are staged - the kernels queued before it run meanwhile.
int run_score_banded(mtm_ctx* c, const ImageArgs& a) {
    const SizeClass& sc = c->classes[0];
    if (!c->hits_only_now) MTMC(ensure_maps(c));
    MTMC(ensure_copy_stream(c));
    mtm_ctx::ImageSlot& sl = c->slot[c->cur];
    SlotGeom g{};
    MTMC(prepare_slot(c, sl, a.rows, a.cols, 1, MTM_U8, c->copy_stream, 1, &g));
    const int nb = (int)c->upload_bands.size();
    while ((int)c->band_ev.size() < nb) {
        hipEvent_t e;
        HIPC(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->band_ev.push_back(e);
    }
    const int h = sc.h, oh = a.rows - h + 1;
    const int RB = sc.rm_R > 0 ? 8 * sc.rm_R : (sc.r2 ? sc.r2 * kMfRows : kMfRows);   // output rows per score-kernel row block
    const int nyb = (oh + RB - 1) / RB, nsb = (oh + kStatBand4 - 1) / kStatBand4;
    int r_done = 0, sb_done = 0, yb_done = 0, n_launch = 0;
    bool used2 = false;
    if (!c->stream2) HIPC(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    if (!c->stream2_done) HIPC(hipEventCreateWithFlags(&c->stream2_done, hipEventDisableTiming));
    // stream2 starts behind whatever the call queued on c->stream so far (counter reset, statistics buffers ...)
    HIPC(hipEventRecord(c->stream2_done, c->stream));
    HIPC(hipStreamWaitEvent(c->stream2, c->stream2_done, 0));
    for (int k = 0; k < nb; ++k) {
        const bool last = k == nb - 1;
        int r1 = last ? a.rows : std::min(a.rows, (int)(c->upload_bands[(size_t)k] * a.rows) & ~7);
        if (r1 <= r_done) continue;
        MTMC(upload_rows_u8c1(sl, g, a.px, a.stride, r_done, r1, c->copy_stream, c->skip_f32 != 0));
        r_done = r1;
        const int avail = r1 - h + 1;                            // output rows whose windows are complete
        const int sb1 = last ? nsb : std::max(sb_done, avail > 0 ? avail / kStatBand4 : 0);
        StatPlanes st;
        c->stats_stream = c->copy_stream;
        const int rc = launch_stats(c, sc, &st, sb_done, sb1);
        c->stats_stream = nullptr;
        MTMC(rc);
        sb_done = sb1;
        HIPC(hipEventRecord(c->band_ev[(size_t)k], c->copy_stream));
        (void)hipStreamQuery(c->copy_stream);                    // submit now (the runtime batches commands)
        const int yb1 = last ? nyb : (sb1 * kStatBand4) / RB;
        if (yb1 > yb_done) {
            hipStream_t s = (c->dual_stream && (n_launch & 1)) ? c->stream2 : c->stream;
            HIPC(hipStreamWaitEvent(s, c->band_ev[(size_t)k], 0));
            c->ncc_stream = s;
            const int rc2 = launch_ncc(c, sc, sc.tlist_off, (int)sc.members.size(), st, -1, yb_done, yb1);
            c->ncc_stream = nullptr;
            MTMC(rc2);
            (void)hipStreamQuery(s);
            used2 = used2 || s == c->stream2;
            ++n_launch;
            yb_done = yb1;
        }
    }
    if (used2) {                                    // everything after the score pass is queued on c->stream
        HIPC(hipEventRecord(c->stream2_done, c->stream2));
        HIPC(hipStreamWaitEvent(c->stream, c->stream2_done, 0));
    }
    return MTM_OK;
}

int fm_begin(mtm_ctx* c, int mode, double score_threshold, NextImage* next, FmState& S, const ImageArgs* up = nullptr) {
    HIPC(hipSetDevice(c->device));
    bool banded = false;
    if (up) {
        // the geometry first (placement depends on it); the pixels follow in stream order
        adopt_image(c, up->rows, up->cols, up->chans, up->dtype);
        c->have_image = false;                       // until the upload is queued: an error below leaves no stale image
        if (!c->have_templ) {
            set_error("set the templates first");
            return MTM_E_STATE;
        }
        c->have_image = true;
    }
    MTMC(place_templates(c));
    if (up) {
        banded = banded_ok(c, *up);
        if (!banded) {
            const int rc = upload_image(c, c->slot[c->cur], up->px, up->stride, up->rows, up->cols, up->chans, up->dtype,
                                        c->stream);
            if (rc != MTM_OK) {
                c->have_image = false;
                return rc;
            }
        }
    }
    const int n = (int)c->templs.size();
    const bool mode_min = c->method == MTM_TM_SQDIFF || c->method == MTM_TM_SQDIFF_NORMED;
    // numpy compares the float32 map with the python-float threshold in float32
    const float thr = (float)score_threshold;
    c->timing = mtm_timing{};
    c->maps_valid = false;

    // fused peak candidates: only when every class runs the MFMA kernel - and not while the maps of this context are
    // known to be dense (the last attempts overflowed the candidate list: smooth images at a low threshold), where the
    // full peak pass over the maps is the cheaper route
    bool fused = mode == MTM_PEAKS_LOCAL && c->fuse_peaks && n > 0;
    if (fused && c->fuse_backoff > 0) {
        --c->fuse_backoff;
        fused = false;
    }
    for (const SizeClass& sc : c->classes) {
        const int rk = resolved_kernel(c, sc);
        fused = fused && (rk == MTM_KERNEL_MFMA || rk == MTM_KERNEL_MFMA16 || rk == MTM_KERNEL_MFMA_F32);
    }
    c->cand_on = false;
    c->hits_only_now = false;
    c->ext_now = false;
    // float32 images on the bf16 matrix cores: the kernel's scores are a screen, the decisions are taken on exact
    // float64 scores (mtm_refine.hip.h).  Calls that mix bf16 classes with float64-kernel ones (float masks) run
    // everything on the float64 kernel.
    c->refine_now = c->refine_scan_now = c->f32_exact_now = false;
    {
        bool any_bf16 = false, all_bf16 = n > 0;
        for (const SizeClass& sc : c->classes) {
            const bool b = resolved_kernel(c, sc) == MTM_KERNEL_MFMA_F32;
            any_bf16 = any_bf16 || b;
            all_bf16 = all_bf16 && b;
        }
        if (any_bf16 && c->f32_mfma == 1) {
            if (all_bf16) c->refine_now = true;
            else c->f32_exact_now = true;
        }
    }
    if (c->f32_exact_now) fused = false;            // the float64 kernel writes maps and lists no candidates
    // fused global extremum (cv2.minMaxLoc inside the score kernel): every class on the 1- or 3-channel MFMA kernel
    // (plain, two-row or row-multiplexed; binary masks with the reciprocal normalisation), the uint16 byte-plane passes
    // or the float32 kernel; same switch as the hits-only mode (MTM_OPT_HITS_ONLY)
    if (mode == MTM_PEAKS_GLOBAL && c->hits_only && c->fuse_peaks && n > 0 && (c->chans == 1 || c->chans == 3) &&
        !c->f32_exact_now) {
        bool ok = true;
        for (const SizeClass& sc : c->classes)
            ok = ok && ((resolved_kernel(c, sc) == MTM_KERNEL_MFMA && sc.slabs.empty() &&
                         (!sc.masked || (!c->exact_div && c->chans == 1 && c->method <= MTM_TM_CCORR_NORMED))) ||
                        resolved_kernel(c, sc) == MTM_KERNEL_MFMA16 || resolved_kernel(c, sc) == MTM_KERNEL_MFMA_F32);
        if (ok) {
            MTMC(c->counters.ensure(sizeof(unsigned long long) * 2 * (size_t)n));
            HIPC(hipMemsetAsync(c->counters.p, 0, sizeof(unsigned long long) * 2 * (size_t)n, c->stream));
            c->ext_now = true;
            c->cand_on = true;
            c->hits_only_now = true;
            c->cand_min = mode_min;
            c->cand_thr = 0.0f;
        }
    }
    const int64_t cand_cap = std::min<int64_t>(c->hit_cap, 4096LL * 256);
    if (mode == MTM_PEAKS_GLOBAL && c->refine_now && !c->ext_now) {
        // no fused extremum in this configuration (maps requested, MTM_FUSE_PEAKS=0): the float64 kernel + extremum_kernel
        c->refine_now = false;
        c->f32_exact_now = true;
    }
    // the refined routes list their records in the candidate buffer: the outputs within the margin of the running best
    // (global extremum), the potential peaks of the map scan (local extrema without kernel candidates)
    const bool pp_mode = mode == MTM_PEAKS_LOCAL && c->refine_now && !fused && n > 0;
    if ((c->refine_now && c->ext_now) || pp_mode) {
        const size_t cands_cap = c->cands.cap;
        MTMC(c->cands.ensure(16 + sizeof(mtm_hit) * (size_t)c->hit_cap));
        if (c->cands.cap != cands_cap) c->cands_zeroed = nullptr;
        if (c->cands.p != c->cands_zeroed) HIPC(hipMemsetAsync(c->cands.p, 0, 16, c->stream));
        c->cands_zeroed = nullptr;
    }
    if (pp_mode) {
        c->refine_scan_now = true;
        c->cand_min = mode_min;
        const float tq = mode_min ? -thr : thr;
        c->cand_thr = tq - kRefineThrMargin * std::max(1.0f, std::fabs(tq));
    }
    if (fused) {
        const size_t cands_cap = c->cands.cap;
        MTMC(c->cands.ensure(16 + sizeof(mtm_hit) * (size_t)c->hit_cap));
        if (c->cands.cap != cands_cap) c->cands_zeroed = nullptr;      // reallocated (possibly at the same address)
        // the counter is normally cleared right after the previous call fetched it (off the critical path)
        if (c->cands.p != c->cands_zeroed) HIPC(hipMemsetAsync(c->cands.p, 0, 16, c->stream));
        c->cands_zeroed = nullptr;
        c->cand_on = true;
        c->cand_min = mode_min;
        c->cand_thr = mode_min ? -thr : thr;
        // (float32 refinement: everything within the margin of the threshold is listed and re-scored)
        if (c->refine_now) c->cand_thr -= kRefineThrMargin * std::max(1.0f, std::fabs(c->cand_thr));
        // hits-only: single-channel MFMA classes, every map 2-D, no recent candidate overflow
        bool honly = c->hits_only && (c->chans == 1 || c->chans == 3) && (int)c->list2d.size() == n;
        c->hits_only_now = honly;
    }
    // hash table of the candidate positions (hits-only verification on the device: only when the
    // candidates are too many to be checked on the host, see below)
    unsigned hash_mask = 0;
    if (c->hits_only_now && !c->ext_now) {
        size_t hsz = 1024;
        while (hsz < 2 * (size_t)cand_cap) hsz <<= 1;
        hash_mask = (unsigned)(hsz - 1);
        MTMC(c->chash.ensure(hsz * (sizeof(unsigned long long) + sizeof(int))));
    }

    HIPC(hipEventRecord(c->ev[0], c->stream));
    if (banded) {
        const int rc = run_score_banded(c, *up);
        if (rc != MTM_OK) {
            c->have_image = false;                   // possibly half an image on the device
            (void)hipStreamSynchronize(c->copy_stream);
            return rc;
        }
    } else {
        MTMC(run_score_all(c));
    }
    HIPC(hipEventRecord(c->ev[1], c->stream));
    c->cand_on = false;
    // stream mode: the kernels of this image are on their way - start the upload of the next one now.
    // (Not later: the device-to-host copy of the hit records below lands in pageable memory, which
    // the runtime executes synchronously, i.e. after the kernels.)
    MTMC(stage_next_image(c, next));

    S.mode = mode;
    S.thr = thr;
    S.mode_min = mode_min;
    S.fused = fused;
    S.n = n;
    S.cand_cap = cand_cap;
    S.hash_mask = hash_mask;
    S.prefetched = false;
    S.pp_mode = pp_mode;
    if (mode == MTM_PEAKS_LOCAL && fused && !c->list2d.empty()) {
        // Few candidates (the usual case): they come back in one copy and the 3x3 test runs on the host
        // (fm_end).  Pinned landing buffer: the copy is a plain DMA instead of a staged one.
        const size_t nfetch = std::min<size_t>(kHitPrefetch, (size_t)cand_cap);
        const size_t fetch_bytes = 16 + sizeof(mtm_hit) * nfetch;
        if (c->pinned_cap < fetch_bytes) {
            if (c->pinned) (void)hipHostFree(c->pinned);
            c->pinned = nullptr;
            c->pinned_cap = 0;
            HIPC(hipHostMalloc(&c->pinned, fetch_bytes, hipHostMallocDefault));
            c->pinned_cap = fetch_bytes;
        }
        HIPC(hipMemcpyAsync(c->pinned, c->cands.p, fetch_bytes, hipMemcpyDeviceToHost, c->stream));
        HIPC(hipEventRecord(c->ev[2], c->stream));
        S.prefetched = true;
    }
    return MTM_OK;
}

// Synchronising half: waits for the stream, verifies / extracts the peaks, delivers the hits.
int fm_end(mtm_ctx* c, const FmState& S, mtm_hit* out, int64_t capacity, int64_t* n_out) {
    HIPC(hipSetDevice(c->device));
    const int mode = S.mode, n = S.n;
    const float thr = S.thr;
    const bool mode_min = S.mode_min, fused = S.fused;
    const int64_t cand_cap = S.cand_cap;
    const unsigned hash_mask = S.hash_mask;
    std::vector<mtm_hit> hits;

    if (mode == MTM_PEAKS_GLOBAL) {
        std::vector<unsigned long long> best(2 * (size_t)std::max(1, n));
        for (int attempt = 0; attempt < 2; ++attempt) {
            if (!c->ext_now) {
                MTMC(c->counters.ensure(sizeof(unsigned long long) * 2 * std::max(1, n)));
                HIPC(hipMemsetAsync(c->counters.p, 0, sizeof(unsigned long long) * 2 * std::max(1, n), c->stream));
            }
            if (n > 0 && !c->ext_now) {
                const int nb = 256;
                hipLaunchKernelGGL(extremum_kernel, dim3(nb, n), dim3(256), 0, c->stream, c->maps.as<float>(),
                                   c->td.as<TemplDev>(), nb, c->counters.as<unsigned long long>());
                HIPC(hipGetLastError());
            }
            HIPC(hipEventRecord(c->ev[2], c->stream));
            HIPC(hipMemcpyAsync(best.data(), c->counters.p, sizeof(unsigned long long) * 2 * std::max(1, n),
                                hipMemcpyDeviceToHost, c->stream));
            unsigned long long nlisted = 0;
            const bool refined = c->refine_now && c->ext_now;
            if (refined)
                HIPC(hipMemcpyAsync(&nlisted, c->cands.p, sizeof(nlisted), hipMemcpyDeviceToHost, c->stream));
            HIPC(hipStreamSynchronize(c->stream));
            if (!refined || (int64_t)nlisted <= cand_cap) break;
            // float32 refinement: more outputs within the margin of their template's best than the list holds (near-flat
            // maps) - the float64 kernel decides, on maps in memory
            c->refine_now = false;
            c->f32_exact_now = true;
            c->ext_now = false;
            c->hits_only_now = false;
            c->cand_on = false;
            c->timing.ncc_launches = 0;
            MTMC(run_score_all(c));
            HIPC(hipEventRecord(c->ev[1], c->stream));
        }
        for (int t = 0; t < n; ++t) {
            const unsigned long long key = best[2 * t + (mode_min ? 1 : 0)];
            const TemplDev& d = c->td_host[t];
            uint32_t o = (uint32_t)(key >> 32);
            if (mode_min) o = ~o;
            const uint32_t idx = key ? (0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFu)) : 0u;
            mtm_hit hrec;
            hrec.templ_idx = t;
            hrec.x = (int)(idx % (uint32_t)d.ow);
            hrec.y = (int)(idx / (uint32_t)d.ow);
            hrec.w = d.cols;
            hrec.h = d.rows;
            hrec.score = key ? decode_order(o) : NAN;
            hits.push_back(hrec);
        }
    } else {
        // ---- 2-D maps.  One device buffer holds [64-bit counter | per-template ints | hit records];
        // the header and the first kHitPrefetch records come back in ONE copy.
        // Fused path: the score-map kernel already appended every pixel above the threshold to the
        // candidate list; verify_peaks_kernel keeps the 3x3 local maxima.  If the candidate list
        // overflowed (dense maps), or on any non-MFMA class, the full peaks_kernel pass runs instead.
        const int n2d = (int)c->list2d.size();
        const size_t hdr_bytes = round_up(2 * sizeof(unsigned long long) + sizeof(int) * (size_t)std::max(1, n), 16);
        unsigned long long count = 0;
        std::vector<int> tflags((size_t)std::max(1, n), 0);
        std::vector<uint8_t> host_buf;
        bool use_fused = fused;
        // Few candidates (the usual case): they come back in one copy and the 3x3 test runs on the host.
        // Every pixel above the threshold is in the list (in both modes), so a neighbour that is not
        // is <= threshold < candidate: the list alone decides.  Saves two kernels, three fills and a copy.
        bool verified_on_host = false;
        bool pp_mode = S.pp_mode;
        const float thr_q = mode_min ? -thr : thr;      // a hit's quality (score, or -score for minima) exceeds this
        if (use_fused && n2d > 0 && !pp_mode) {
            // the candidate list is already on its way into the pinned landing buffer (fm_begin)
            const size_t nfetch = std::min<size_t>(kHitPrefetch, (size_t)cand_cap);
            if (!S.prefetched) {
                set_error("mtm_find_matches: internal state (candidate fetch not queued)");
                return MTM_E_INVALID;
            }
            HIPC(hipStreamSynchronize(c->stream));
            const uint8_t* land = static_cast<const uint8_t*>(c->pinned);
            unsigned long long ncand = 0;
            std::memcpy(&ncand, land, sizeof(ncand));
            std::memcpy(&c->timing.sclk_mhz, land + 8, sizeof(float));
            if (ncand <= nfetch) {
                // everything needed is on the host: clear the counter for the next call while this one finishes
                if (hipMemsetAsync(c->cands.p, 0, 16, c->stream) == hipSuccess) c->cands_zeroed = c->cands.p;
                const mtm_hit* cd = reinterpret_cast<const mtm_hit*>(land + 16);
                // open-addressing table over the candidates (key -> index), kept in the context between calls
                size_t tsize = 64;
                while (tsize < 2 * (size_t)ncand + 8) tsize <<= 1;
                std::vector<unsigned long long>& hk = c->vh_keys;
                std::vector<int>& hv = c->vh_vals;
                hk.assign(tsize, 0ull);
                hv.resize(tsize);
                const size_t tmask = tsize - 1;
                auto key = [](int t, int y, int x) {
                    return ((unsigned long long)(t + 1) << 42) | ((unsigned long long)y << 21) | (unsigned long long)x;
                };
                auto slot_of = [&](unsigned long long k) {
                    size_t sidx = (size_t)((k * 0x9E3779B97F4A7C15ull) >> 20) & tmask;
                    while (hk[sidx] != 0ull && hk[sidx] != k) sidx = (sidx + 1) & tmask;
                    return sidx;
                };
                for (int i = 0; i < (int)ncand; ++i) {
                    const unsigned long long k = key(cd[i].templ_idx, cd[i].y, cd[i].x);
                    const size_t sidx = slot_of(k);
                    if (hk[sidx] == 0ull) {         // (a pixel is listed once; keep the first if it ever were not)
                        hk[sidx] = k;
                        hv[sidx] = i;
                    }
                }
                const float padv = (c->opt_border == MTM_BORDER_CONSTANT) ? 0.0f : -INFINITY;
                for (int i = 0; i < (int)ncand; ++i) {
                    const mtm_hit& h = cd[i];
                    const TemplDev& d = c->td_host[h.templ_idx];
                    const float v = mode_min ? -h.score : h.score;
                    float mx = v;
                    for (int dy = -1; dy <= 1; ++dy)
                        for (int dx = -1; dx <= 1; ++dx) {
                            if (!dy && !dx) continue;
                            const int yy = h.y + dy, xx = h.x + dx;
                            if (yy < 0 || yy >= d.oh || xx < 0 || xx >= d.ow) {
                                mx = fmaxf(mx, padv);
                                continue;
                            }
                            const size_t sidx = slot_of(key(h.templ_idx, yy, xx));
                            if (hk[sidx] != 0ull) mx = fmaxf(mx, mode_min ? -cd[hv[sidx]].score : cd[hv[sidx]].score);
                        }
                    // (v > thr_q: every record the integer kernels list passes; the float32 screen lists with a margin)
                    if (v == mx && v > thr_q) {
                        hits.push_back(h);
                        ++tflags[(size_t)h.templ_idx];
                    }
                }
                count = hits.size();
                verified_on_host = true;
            }
        }
        if (!verified_on_host && c->hits_only_now)
            HIPC(hipMemsetAsync(c->chash.p, 0, ((size_t)hash_mask + 1) * sizeof(unsigned long long), c->stream));
        if (pp_mode) use_fused = true;          // the potential peaks are in the candidate buffer, their neighbourhoods in the maps
        for (int attempt = 0; attempt < 5 && n2d > 0 && !verified_on_host; ++attempt) {
            MTMC(c->hits.ensure(hdr_bytes + sizeof(mtm_hit) * (size_t)c->hit_cap));
            uint8_t* dbase = c->hits.as<uint8_t>();
            HIPC(hipMemsetAsync(dbase, 0, hdr_bytes, c->stream));
            unsigned long long* counter = reinterpret_cast<unsigned long long*>(dbase);
            int* flags = reinterpret_cast<int*>(counter + 2);
            mtm_hit* dhits = reinterpret_cast<mtm_hit*>(dbase + hdr_bytes);
            if (use_fused) {
                // counter[1] <- candidate count (for the overflow check on the host)
                HIPC(hipMemcpyAsync(counter + 1, c->cands.p, sizeof(unsigned long long), hipMemcpyDeviceToDevice,
                                    c->stream));
                const unsigned blocks = std::min((unsigned)((c->hit_cap + 255) / 256), 4096u);
                const mtm_hit* dcands = reinterpret_cast<const mtm_hit*>(c->cands.as<uint8_t>() + 16);
                if (c->hits_only_now) {
                    unsigned long long* keys = c->chash.as<unsigned long long>();
                    int* vals = reinterpret_cast<int*>(keys + (size_t)hash_mask + 1);
                    hipLaunchKernelGGL(cand_hash_insert_kernel, dim3(blocks), dim3(256), 0, c->stream, dcands,
                                       c->cands.as<unsigned long long>(), (unsigned long long)cand_cap, keys, vals,
                                       hash_mask);
                    hipLaunchKernelGGL(verify_hash_kernel, dim3(blocks), dim3(256), 0, c->stream, c->td.as<TemplDev>(),
                                       mode_min ? 1 : 0, c->opt_border, dcands, c->cands.as<unsigned long long>(),
                                       (unsigned long long)cand_cap, keys, vals, hash_mask, dhits,
                                       (unsigned long long)c->hit_cap, counter, flags, thr_q);
                } else {
                    hipLaunchKernelGGL(verify_peaks_kernel, dim3(blocks), dim3(256), 0, c->stream,
                                       c->maps.as<float>(), c->td.as<TemplDev>(), mode_min ? 1 : 0, c->opt_border,
                                       dcands, c->cands.as<unsigned long long>(), (unsigned long long)cand_cap, dhits,
                                       (unsigned long long)c->hit_cap, counter, flags, thr_q);
                }
            } else {
                int max_oh = 0, max_ow = 0;
                for (int t : c->list2d) {
                    max_oh = std::max(max_oh, c->td_host[t].oh);
                    max_ow = std::max(max_ow, c->td_host[t].ow);
                }
                const dim3 grd((max_ow + kPkCols - 1) / kPkCols, (max_oh + 4 * kPkRows - 1) / (4 * kPkRows), n2d);
                hipLaunchKernelGGL(peaks_kernel, grd, dim3(256), 0, c->stream, c->maps.as<float>(),
                                   c->td.as<TemplDev>(), c->tlist.as<int>() + c->list2d_off, mode_min ? 1 : 0, thr,
                                   c->opt_border, dhits, (unsigned long long)c->hit_cap, counter, flags);
            }
            HIPC(hipGetLastError());
            HIPC(hipEventRecord(c->ev[2], c->stream));
            const size_t first = std::min<size_t>(kHitPrefetch, (size_t)c->hit_cap);
            host_buf.resize(hdr_bytes + sizeof(mtm_hit) * first);
            HIPC(hipMemcpyAsync(host_buf.data(), dbase, host_buf.size(), hipMemcpyDeviceToHost, c->stream));
            HIPC(hipStreamSynchronize(c->stream));
            unsigned long long ncand = 0;
            std::memcpy(&count, host_buf.data(), sizeof(count));
            std::memcpy(&ncand, host_buf.data() + sizeof(count), sizeof(ncand));
            std::memcpy(tflags.data(), host_buf.data() + 2 * sizeof(count), sizeof(int) * n);
            if (use_fused && (int64_t)ncand > cand_cap && c->refine_now) {
                // float32 refinement, list overflowed.  Kernel candidates (everything above the threshold): take the
                // potential peaks of a map scan instead - far fewer.  Those too (plateau-rich maps): the float64 kernel.
                c->cand_on = false;
                c->hits_only_now = false;
                c->timing.ncc_launches = 0;
                if (!pp_mode) {
                    c->fuse_backoff = c->backoff_len;
                    c->backoff_len = std::min(2 * c->backoff_len, 1024);
                    pp_mode = true;
                    c->refine_scan_now = true;
                    HIPC(hipMemsetAsync(c->cands.p, 0, 16, c->stream));
                } else {
                    pp_mode = false;
                    use_fused = false;
                    c->refine_now = c->refine_scan_now = false;
                    c->f32_exact_now = true;
                }
                MTMC(run_score_all(c));
                HIPC(hipEventRecord(c->ev[1], c->stream));
                continue;
            }
            if (use_fused && (int64_t)ncand > cand_cap) {
                use_fused = false;                  // dense maps: candidate list overflowed
                // the next calls on this context go straight to map mode + full peak pass; the period doubles while
                // the retries keep overflowing
                c->fuse_backoff = c->backoff_len;
                c->backoff_len = std::min(2 * c->backoff_len, 1024);
                if (c->hits_only_now) {
                    // no maps in memory: compute them (this call pays twice - the overflowed launch left early)
                    c->hits_only_now = false;
                    c->cand_on = false;
                    c->timing.ncc_launches = 0;
                    MTMC(run_score_all(c));
                    HIPC(hipEventRecord(c->ev[1], c->stream));
                }
                continue;
            }
            if (use_fused && !pp_mode) c->backoff_len = 16;     // the candidates fitted
            if ((int64_t)count <= c->hit_cap) {
                hits.resize((size_t)count);
                const size_t got = std::min<size_t>((size_t)count, first);
                if (got) std::memcpy(hits.data(), host_buf.data() + hdr_bytes, sizeof(mtm_hit) * got);
                if (count > got) {
                    HIPC(hipMemcpyAsync(hits.data() + got, dhits + got, sizeof(mtm_hit) * ((size_t)count - got),
                                        hipMemcpyDeviceToHost, c->stream));
                    HIPC(hipStreamSynchronize(c->stream));
                }
                break;
            }
            c->hit_cap = (int64_t)count + 1024;     // grow and rerun the compaction pass
            use_fused = false;
        }
        if (n2d == 0) {
            HIPC(hipEventRecord(c->ev[2], c->stream));
            HIPC(hipStreamSynchronize(c->stream));
        }
        if (!hits.empty()) {
            // skimage: a map in which every pixel equals its local maximum has no peaks at all.
            // peaks_kernel: tflags[t] = "some pixel differs from its local max";
            // fused path:   tflags[t] = number of peaks of t (all pixels <=> trivial).
            hits.erase(std::remove_if(hits.begin(), hits.end(),
                                      [&](const mtm_hit& h) {
                                          const TemplDev& d = c->td_host[h.templ_idx];
                                          if (d.oh <= 1 || d.ow <= 1) return true;   // 1-D / 1x1 maps: host path below
                                          if (use_fused) return (long long)tflags[h.templ_idx] == (long long)d.oh * d.ow;
                                          return tflags[h.templ_idx] == 0;
                                      }),
                       hits.end());
        }
        // ---- 1x1 and 1-D maps (MTM/__init__.py:25-41) on the host
        for (int t = 0; t < n; ++t) {
            const TemplDev& d = c->td_host[t];
            if (d.oh > 1 && d.ow > 1) continue;
            const int len = std::max(d.oh, d.ow);
            std::vector<float> line((size_t)len);
            HIPC(hipMemcpy2DAsync(line.data(), sizeof(float) * (d.oh == 1 ? len : 1),
                                  c->maps.as<float>() + d.map_off, sizeof(float) * d.map_pitch,
                                  sizeof(float) * d.ow, d.oh, hipMemcpyDeviceToHost, c->stream));
            HIPC(hipStreamSynchronize(c->stream));
            std::vector<int> pk;
            if (len == 1) {
                const float v = mode_min ? -line[0] : line[0];
                if (v >= (mode_min ? -thr : thr)) pk.push_back(0);
            } else {
                pk = find_peaks_1d(line.data(), len, 1, mode_min ? -thr : thr, mode_min);
            }
            for (int i : pk) {
                mtm_hit hrec;
                hrec.templ_idx = t;
                hrec.x = d.oh == 1 ? i : 0;
                hrec.y = d.oh == 1 ? 0 : i;
                hrec.w = d.cols;
                hrec.h = d.rows;
                hrec.score = line[(size_t)i];
                hits.push_back(hrec);
            }
        }
        // deterministic order: template, then descending quality, then row-major position
        sort_hits(hits, mode_min);
    }
    HIPC(hipEventSynchronize(c->ev[2]));       // already complete: every path above synchronised the stream
    HIPC(hipEventElapsedTime(&c->timing.score_ms, c->ev[0], c->ev[1]));
    HIPC(hipEventElapsedTime(&c->timing.peaks_ms, c->ev[1], c->ev[2]));
    HIPC(hipEventElapsedTime(&c->timing.total_ms, c->ev[0], c->ev[2]));
    MTMC(collect_ncc_time(c));
    c->timing.n_hits = (int64_t)hits.size();
    c->timing.hits_only = c->hits_only_now ? 1 : 0;
    c->timing.f32_route = c->f32_exact_now ? 3 : !c->refine_now ? 0 : (c->refine_scan_now ? 2 : 1);
    c->maps_valid = !c->hits_only_now && !c->ext_now;
    c->refine_now = c->refine_scan_now = c->f32_exact_now = false;      // states of this call only
    *n_out = (int64_t)hits.size();
    c->last_hits.swap(hits);
    if ((int64_t)c->last_hits.size() > capacity) {
        set_error("mtm_find_matches: output capacity too small (fetch the result with mtm_last_hits)");
        return MTM_E_OVERFLOW;
    }
    if (!c->last_hits.empty()) std::memcpy(out, c->last_hits.data(), sizeof(mtm_hit) * c->last_hits.size());
    return MTM_OK;
}

int find_matches_impl(mtm_ctx* c, int mode, double score_threshold, mtm_hit* out, int64_t capacity,
                      int64_t* n_out, NextImage* next, const ImageArgs* up) {
    if (!c || !n_out || capacity < 0 || (capacity > 0 && !out) ||
        (mode != MTM_PEAKS_LOCAL && mode != MTM_PEAKS_GLOBAL)) {
        set_error("mtm_find_matches: bad arguments");
        return MTM_E_INVALID;
    }
    if (c->fm_in_flight) {
        set_error("mtm_find_matches: a mtm_find_matches_async call is in flight (collect it with mtm_find_matches_wait)");
        return MTM_E_INVALID;
    }
    FmState S;
    MTMC(fm_begin(c, mode, score_threshold, next, S, up));
    return fm_end(c, S, out, capacity, n_out);
}

}  // namespace

int mtm_last_hits(mtm_ctx* c, mtm_hit* out, int64_t capacity, int64_t* n_out) {
    if (!c || !n_out || capacity < 0 || (capacity > 0 && !out)) {
        set_error("mtm_last_hits: bad arguments");
        return MTM_E_INVALID;
    }
    *n_out = (int64_t)c->last_hits.size();
    if ((int64_t)c->last_hits.size() > capacity) {
        set_error("mtm_last_hits: output capacity too small");
        return MTM_E_OVERFLOW;
    }
    if (!c->last_hits.empty()) std::memcpy(out, c->last_hits.data(), sizeof(mtm_hit) * c->last_hits.size());
    return MTM_OK;
}

int mtm_last_score_map(mtm_ctx* c, int templ_idx, float* out, int64_t out_row_stride_bytes) {
    if (!c || !out) {
        set_error("mtm_last_score_map: bad arguments");
        return MTM_E_INVALID;
    }
    MTM_NOT_IN_FLIGHT(c, "mtm_last_score_map");
    if (!c->placed || !c->maps_valid) {
        set_error("mtm_last_score_map: the last mtm_find_matches did not materialise the score maps "
                  "(MTM_OPT_HITS_ONLY = 0 makes it), or the inputs changed since");
        return MTM_E_STATE;
    }
    if (templ_idx < 0 || templ_idx >= (int)c->templs.size()) {
        set_error("mtm_last_score_map: template index out of range");
        return MTM_E_INVALID;
    }
    const TemplDev& d = c->td_host[templ_idx];
    if (out_row_stride_bytes < (int64_t)(sizeof(float) * d.ow)) {
        set_error("mtm_last_score_map: output row stride too small");
        return MTM_E_INVALID;
    }
    HIPC(hipSetDevice(c->device));
    HIPC(hipMemcpy2DAsync(out, (size_t)out_row_stride_bytes, c->maps.as<float>() + d.map_off, sizeof(float) * d.map_pitch,
                          sizeof(float) * d.ow, d.oh, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    return MTM_OK;
}

int mtm_find_matches_async(mtm_ctx* c, int mode, double score_threshold) {
    if (!c || (mode != MTM_PEAKS_LOCAL && mode != MTM_PEAKS_GLOBAL)) {
        set_error("mtm_find_matches_async: bad arguments");
        return MTM_E_INVALID;
    }
    if (c->fm_in_flight) {
        set_error("mtm_find_matches_async: a call is already in flight (collect it with mtm_find_matches_wait)");
        return MTM_E_INVALID;
    }
    MTMC(fm_begin(c, mode, score_threshold, nullptr, c->fm));
    c->fm_in_flight = true;
    return MTM_OK;
}

int mtm_find_matches_wait(mtm_ctx* c, mtm_hit* out, int64_t capacity, int64_t* n_out) {
    if (!c || !n_out || capacity < 0 || (capacity > 0 && !out)) {
        set_error("mtm_find_matches_wait: bad arguments");
        return MTM_E_INVALID;
    }
    if (!c->fm_in_flight) {
        set_error("mtm_find_matches_wait: no call in flight");
        return MTM_E_INVALID;
    }
    c->fm_in_flight = false;
    return fm_end(c, c->fm, out, capacity, n_out);
}

int mtm_get_timing(mtm_ctx* c, mtm_timing* out) {
    if (!c || !out) return MTM_E_INVALID;
    *out = c->timing;
    return MTM_OK;
}

// ---------------------------------------------------------------------------------------------
// RCCL hit exchange.  librccl is loaded lazily so that the library itself has no link-time
// dependency on it (single-GPU users never touch it).
// ---------------------------------------------------------------------------------------------
namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
    if (g_rccl.lib) return MTM_OK;
    void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) {
        set_error(std::string("cannot load librccl.so: ") + dlerror());
        return MTM_E_COMM;
    }
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(dlsym(lib, "ncclAllGather"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    g_rccl.CommAbort = reinterpret_cast<decltype(g_rccl.CommAbort)>(dlsym(lib, "ncclCommAbort"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.CommDestroy) {
        set_error("librccl.so lacks an expected symbol");
        dlclose(lib);
        return MTM_E_COMM;
    }
    g_rccl.lib = lib;
    return MTM_OK;
}

#define NCCLC(expr)                                                                         \
    do {                                                                                    \
        ncclResult_t r_ = (expr);                                                           \
        if (r_ != ncclSuccess) {                                                            \
            set_error(std::string(#expr) + ": " +                                           \
                      (g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error"));  \
            return MTM_E_COMM;                                                              \
        }                                                                                   \
    } while (0)
}  // namespace

int mtm_comm_unique_id(void* id_out) {
    if (!id_out) return MTM_E_INVALID;
    static_assert(sizeof(ncclUniqueId) == MTM_COMM_ID_BYTES, "unique id size");
    MTMC(load_rccl());
    ncclUniqueId id;
    NCCLC(g_rccl.GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return MTM_OK;
}

int mtm_comm_init(mtm_ctx* c, const void* id, int n_ranks, int rank) {
    if (!c || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) {
        set_error("mtm_comm_init: bad arguments");
        return MTM_E_INVALID;
    }
    MTMC(load_rccl());
    HIPC(hipSetDevice(c->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    NCCLC(g_rccl.CommInitRank(&c->comm, n_ranks, uid, rank));
    c->comm_slot_hits = 512;
    c->n_ranks = n_ranks;
    c->rank = rank;
    return MTM_OK;
}

int mtm_comm_allgather_hits(mtm_ctx* c, const mtm_hit* local, int64_t n_local, mtm_hit* out,
                            int64_t capacity, int64_t* counts_out, int64_t* n_out) {
    if (!c || !c->comm || n_local < 0 || (n_local > 0 && !local) || !counts_out || !n_out) {
        set_error("mtm_comm_allgather_hits: bad arguments or communicator not initialised");
        return MTM_E_INVALID;
    }
    MTM_NOT_IN_FLIGHT(c, "mtm_comm_allgather_hits");
    HIPC(hipSetDevice(c->device));
    const int R = c->n_ranks;
    // One all-gather of fixed-size slots: [count (16-byte header) | slot_hits records].  Every rank
    // sees every count; only if some rank produced more than kSlotHits hits is a second all-gather
    // issued with slots of the (globally known) maximum count.  The usual case is ONE collective of
    // ~12 KB per rank: latency-bound on xGMI, ring bandwidth irrelevant.
    // The slot size adapts to the data: it starts at 512 records and follows twice the largest count
    // of the previous exchange (a value every rank knows, so the ranks always agree on it).
    std::vector<long long> counts((size_t)R, 0);
    const uint8_t* all = nullptr;      // the gathered slots of the last round (pinned staging)
    long long slot_hits = c->comm_slot_hits;
    for (int round = 0; round < 2; ++round) {
        const size_t slot = 16 + sizeof(mtm_hit) * (size_t)slot_hits;
        MTMC(c->comm_send.ensure(slot));
        MTMC(c->comm_recv.ensure(slot * R));
        // pinned staging [my slot | R gathered slots]: both copies are plain DMAs queued behind each other
        // on the stream, one synchronisation per exchange
        const size_t pin_bytes = slot * (size_t)(R + 1);
        if (c->comm_pin_cap < pin_bytes) {
            if (c->comm_pin) (void)hipHostFree(c->comm_pin);
            c->comm_pin = nullptr;
            c->comm_pin_cap = 0;
            HIPC(hipHostMalloc(&c->comm_pin, pin_bytes, hipHostMallocDefault));
            c->comm_pin_cap = pin_bytes;
        }
        uint8_t* mine = static_cast<uint8_t*>(c->comm_pin);
        uint8_t* gathered = mine + slot;
        const size_t mine_bytes = 16 + sizeof(mtm_hit) * (size_t)std::min<long long>(n_local, slot_hits);
        std::memset(mine, 0, 16);
        const long long cnt = n_local;
        std::memcpy(mine, &cnt, sizeof(cnt));
        if (n_local > 0) std::memcpy(mine + 16, local, mine_bytes - 16);
        HIPC(hipMemcpyAsync(c->comm_send.p, mine, mine_bytes, hipMemcpyHostToDevice, c->stream));
        NCCLC(g_rccl.AllGather(c->comm_send.p, c->comm_recv.p, slot, ncclInt8, c->comm, c->stream));
        HIPC(hipMemcpyAsync(gathered, c->comm_recv.p, slot * R, hipMemcpyDeviceToHost, c->stream));
        // A rank that never arrives (crashed, or took another branch) must not hang the others for ever: the
        // exchange has a deadline (MTM_COMM_TIMEOUT_S, default 300 s; 0 = wait without limit), after which the
        // communicator is aborted - the queued collective is cancelled, the context stays usable without it.
        if (c->comm_timeout_s > 0.0) {
            const auto t0 = std::chrono::steady_clock::now();
            hipError_t qs;
            int spins = 0;
            while ((qs = hipStreamQuery(c->stream)) == hipErrorNotReady) {
                if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->comm_timeout_s) {
                    if (g_rccl.CommAbort) (void)g_rccl.CommAbort(c->comm);
                    else if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
                    c->comm = nullptr;
                    c->n_ranks = 1;
                    c->rank = 0;
                    (void)hipStreamSynchronize(c->stream);
                    set_error("mtm_comm_allgather_hits: no answer from the other ranks within the deadline "
                              "(MTM_COMM_TIMEOUT_S); communicator aborted");
                    return MTM_E_COMM;
                }
            }
            HIPC(qs);
        } else {
            HIPC(hipStreamSynchronize(c->stream));
        }
        all = gathered;
        long long mx = 0;
        for (int r = 0; r < R; ++r) {
            std::memcpy(&counts[r], all + slot * r, sizeof(long long));
            mx = std::max(mx, counts[r]);
        }
        long long want = 512;
        while (want < 2 * mx) want <<= 1;
        c->comm_slot_hits = want;       // next exchange (identical on every rank)
        if (mx <= slot_hits) break;
        slot_hits = mx;                 // every rank computes the same maximum: the collective stays matched
    }
    c->comm_last_counts = counts;
    c->comm_last_slot = 16 + sizeof(mtm_hit) * (size_t)slot_hits;
    (void)all;
    // The collective is over.  A too-small output buffer is a purely local matter: the gathered slots stay in
    // the pinned staging area and mtm_comm_last_gather() delivers them - the exchange is NEVER repeated (the
    // other ranks, whose buffers were large enough, have already moved on).
    return mtm_comm_last_gather(c, out, capacity, counts_out, n_out);
}

int mtm_comm_last_gather(mtm_ctx* c, mtm_hit* out, int64_t capacity, int64_t* counts_out, int64_t* n_out) {
    if (!c || !counts_out || !n_out || capacity < 0 || (capacity > 0 && !out)) {
        set_error("mtm_comm_last_gather: bad arguments");
        return MTM_E_INVALID;
    }
    if (c->comm_last_counts.empty() || !c->comm_pin) {
        set_error("mtm_comm_last_gather: no exchange has run on this context");
        return MTM_E_STATE;
    }
    const int R = (int)c->comm_last_counts.size();
    long long total = 0;
    for (int r = 0; r < R; ++r) {
        counts_out[r] = c->comm_last_counts[(size_t)r];
        total += c->comm_last_counts[(size_t)r];
    }
    *n_out = total;
    if (total > capacity) {
        set_error("mtm_comm_allgather_hits: output capacity too small (fetch the result with mtm_comm_last_gather)");
        return MTM_E_OVERFLOW;
    }
    const size_t slot = c->comm_last_slot;
    const uint8_t* all = static_cast<const uint8_t*>(c->comm_pin) + slot;      // [my slot | R gathered slots]
    int64_t o = 0;
    for (int r = 0; r < R; ++r) {
        const long long cnt = c->comm_last_counts[(size_t)r];
        if (cnt) std::memcpy(out + o, all + slot * r + 16, sizeof(mtm_hit) * (size_t)cnt);
        o += cnt;
    }
    return MTM_OK;
}

int mtm_comm_destroy(mtm_ctx* c) {
    if (!c) return MTM_E_INVALID;
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    c->comm = nullptr;
    c->n_ranks = 1;
    c->rank = 0;
    return MTM_OK;
}

}  // extern "C"
