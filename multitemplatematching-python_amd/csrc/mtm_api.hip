// libmtm_hip.so - the search entry points: mtm_find_matches and its variants (score pass, peak extraction or
// verification of the kernels' candidates, float32 refinement routes, hit lists), score maps, timing.
#include "mtm_ctx.h"

using namespace mtm;
using namespace mtmi;
#include "mtm_k_peaks.hip.h"
#include "mtm_k_nms.hip.h"

namespace {

inline float decode_order(uint32_t o) {
    const uint32_t b = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    float v;
    std::memcpy(&v, &b, 4);
    return v;
}

struct NextImage {
    const void* px;
    int rows, cols, chans, dtype;
    int64_t stride;
    bool staged;
};

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

constexpr size_t kHitPrefetch = 1024;       // candidate / hit records fetched together with the counters

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
    host_trace(c, 1);
    MTMC(place_templates(c));
    host_trace(c, 2);
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
    c->seg_skip_used = false;
    // masked float32 classes: the bf16 screen needs the threshold (local extrema only; mtm_score_map and N_object == 1 keep
    // the float64 kernel).  What the peak pass compares with: the float32 threshold, on float32 scores.
    c->mbf_used = false;
    c->mbf_thr_on = (mode == MTM_PEAKS_LOCAL || mode == MTM_PEAKS_GLOBAL) && n > 0 && f32_refined(c);
    c->mbf_global = mode == MTM_PEAKS_GLOBAL;
    c->mbf_thr = thr;

    // fused peak candidates: only when every class runs the MFMA kernel - and not while the maps of this context are
    // known to be dense (the last attempts overflowed the candidate list: smooth images at a low threshold), where the
    // full peak pass over the maps is the cheaper route
    bool fused = mode == MTM_PEAKS_LOCAL && n > 0;
    if (fused && c->fuse_backoff > 0) {
        --c->fuse_backoff;
        fused = false;
    }
    // Segment flags (the default route while the back-off lasts): the maps go to memory as in map mode, and the score kernel
    // sets a flag per row segment (a wave's 256 outputs of one template row) in which something passes the threshold;
    // peaks_sparse_kernel visits those instead of scanning 1 GB of maps (4K x 32 templates).  A first version wrote only
    // the flagged segments, from the hits-only screens: no faster - on such images the screens pass nearly everywhere.
    // uint8 classes on the lean 1- / 3-channel MFMA epilogue, every map 2-D.
    c->sparse_now = false;
    if (mode == MTM_PEAKS_LOCAL && n > 0 && !fused && c->sparse_maps && c->hits_only &&
        c->dtype == MTM_U8 && (c->chans == 1 || c->chans == 3) && (int)c->list2d.size() == n) {
        bool ok = true;
        int max_oh = 0, max_nseg = 0;
        for (const SizeClass& sc : c->classes) {
            ok = ok && resolved_kernel(c, sc) == MTM_KERNEL_MFMA && sc.slabs.empty();
            max_oh = std::max(max_oh, c->rows - sc.h + 1);
            max_nseg = std::max(max_nseg, (c->cols - sc.w + 1 + kMfSeg - 1) / kMfSeg);
        }
        // (bounded: the flags are cleared on every call and compact_hits_kernel sums the per-(template, strip) counters in
        // every block - with thousands of templates the flagged route would cost more than the full scan it replaces)
        if (ok && (long long)n * max_oh * max_nseg <= (64ll << 20) && (long long)n * max_nseg <= 4096) {
            c->flag_rstride = max_nseg;
            c->flag_tstride = max_oh * max_nseg;
            MTMC(c->seg_flags.ensure((size_t)n * c->flag_tstride));
            HIPC(hipMemsetAsync(c->seg_flags.p, 0, (size_t)n * c->flag_tstride, c->stream));
            c->sparse_now = true;
        }
    }
    for (const SizeClass& sc : c->classes) {
        const int rk = resolved_kernel(c, sc);
        fused = fused && (rk == MTM_KERNEL_MFMA || rk == MTM_KERNEL_MFMA16 || rk == MTM_KERNEL_MFMA_F32);
    }
    c->cand_on = false;
    c->hits_only_now = false;
    c->ext_now = false;
    if (c->sparse_now) {                // the threshold the score kernel flags against (launch_ncc)
        c->cand_min = mode_min;
        c->cand_thr = mode_min ? -thr : thr;
    }
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
        // raw sums (TM_SQDIFF / TM_CCORR / TM_CCOEFF): only the refined global extremum runs on the matrix cores - maps and
        // thresholds on unnormalised sums have no error the bf16 pieces could promise (an exact copy is TM_SQDIFF 0)
        const bool raw_m = c->method == MTM_TM_SQDIFF || c->method == MTM_TM_CCORR || c->method == MTM_TM_CCOEFF;
        // ... except, round 5, local extrema against a threshold while the kernel's candidate list is available: every output
        // whose upper bound (score + E, E in the sum's own units) passes the threshold is listed and re-scored exactly -
        // route 1 only; maps, the map scan and every overflow keep the float64 kernel
        c->raw_rig_now = any_bf16 && all_bf16 && raw_m && mode == MTM_PEAKS_LOCAL && f32_refined(c) && fused;
        if (any_bf16 && raw_m && (mode != MTM_PEAKS_GLOBAL || !f32_refined(c)) && !c->raw_rig_now) {
            c->f32_exact_now = true;
        } else if (any_bf16 && f32_refined(c)) {
            if (all_bf16) c->refine_now = true;
            else c->f32_exact_now = true;
        }
    }
    if (c->f32_exact_now) fused = false;            // the float64 kernel writes maps and lists no candidates
    // fused global extremum (cv2.minMaxLoc inside the score kernel): every class on the 1- or 3-channel MFMA kernel
    // (plain, two-row, row-multiplexed or in slabs - there in slab_combine_kernel; binary masks with the reciprocal
    // normalisation), the uint16 byte-plane kernel
    // or the float32 kernel; same switch as the hits-only mode (MTM_OPT_HITS_ONLY)
    if (mode == MTM_PEAKS_GLOBAL && c->hits_only && n > 0 && (c->chans == 1 || c->chans == 3) &&
        !c->f32_exact_now) {
        bool ok = true;
        for (const SizeClass& sc : c->classes)
            ok = ok && ((resolved_kernel(c, sc) == MTM_KERNEL_MFMA &&
                         (!sc.masked || (c->exact_div < 2 && c->chans == 1 && c->method <= MTM_TM_CCORR_NORMED))) ||
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
    host_trace(c, 18);
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
    // the refined routes' own thresholds: rig_thr the exact one, rig_cap the widest error bound the map scan's tolerances
    // cover (4 x the largest class constant: windows whose mean lies within ~4 standard deviations of their tile's)
    if (c->refine_now) {
        const float tq = mode_min ? -thr : thr;
        c->rig_thr = tq;
        float eps = 0.0f;
        for (const SizeClass& sc : c->classes)
            if (resolved_kernel(c, sc) == MTM_KERNEL_MFMA_F32) eps = std::max(eps, bf16_rig_eps(c->chans, sc.h, bf16_nkb(sc.w)));
        c->rig_cap = std::max(kRefineThrMargin, 4.0f * eps);
        c->scan_thr = tq - c->rig_cap * std::max(1.0f, std::fabs(tq));
    }
    if (pp_mode) {
        c->refine_scan_now = true;
        c->cand_min = mode_min;
        c->cand_thr = c->scan_thr;
    }
    if (fused) {
        const size_t cands_cap = c->cands.cap;
        MTMC(c->cands.ensure(16 + sizeof(mtm_hit) * (size_t)c->hit_cap));
        if (c->cands.cap != cands_cap) c->cands_zeroed = nullptr;      // reallocated (possibly at the same address)
        // the counter is normally cleared right after the previous call fetched it (off the critical path); round 5: a banded
        // uint8 call lets its first statistics launch do it (zero_pending; run_score_banded) - no fill command at all
        c->zero_pending = false;
        if (banded && c->dtype == MTM_U8) c->zero_pending = true;
        else if (c->cands.p != c->cands_zeroed) HIPC(hipMemsetAsync(c->cands.p, 0, 16, c->stream));
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

    // The landing buffer of the candidate list (pinned).  Round 5: when every class of the call runs ncc_mfma_kernel's own
    // epilogue, the waves that fill the first slots of the list write them there as well (MfmaParams::cand_pin) and the
    // host finds them when the last score launch has ended - no fetch kernel (or copy command) with its kernel boundary
    // behind the score pass.  The window starts out as "no record" (template index -1) in every slot.
    c->cand_pin_now = false;
    const bool want_prefetch = mode == MTM_PEAKS_LOCAL && fused && !c->list2d.empty();
    const size_t nfetch_w = std::min<size_t>(kHitPrefetch, (size_t)cand_cap);
    if (want_prefetch) {
        const size_t fetch_bytes = 16 + sizeof(mtm_hit) * nfetch_w;
        if (c->pinned_cap < fetch_bytes) {
            if (c->pinned) (void)hipHostFree(c->pinned);
            c->pinned = nullptr;
            c->pinned_cap = 0;
            HIPC(hipHostMalloc(&c->pinned, fetch_bytes, hipHostMallocDefault));
            c->pinned_cap = fetch_bytes;
        }
        bool pin = c->cand_pinned != 0 && !c->refine_now;
        for (const SizeClass& sc : c->classes) {
            const int rk = resolved_kernel(c, sc);
            pin = pin && ((rk == MTM_KERNEL_MFMA && sc.slabs.empty()) || rk == MTM_KERNEL_MFMA16);
        }
        if (pin) {
            uint8_t* land = static_cast<uint8_t*>(c->pinned);
            std::memset(land, 0, 16);
            mtm_hit* w = reinterpret_cast<mtm_hit*>(land + 16);
            for (size_t i = 0; i < nfetch_w; ++i) w[i].templ_idx = -1;
            c->cand_pin_now = true;
            c->cand_pin_n = nfetch_w;
        }
    }
    // float32: the hits-only refined routes (kernel candidates re-scored; the fused extremum by bounds) and the masked
    // classes' screen start with ONE piece product (mtm_ctx::bf16_np_now) unless a recent call overflowed its list that way
    c->bf16_np_now = 3;
    if (c->dtype == MTM_F32 && c->f32_mfma == 1) {
        if (c->np1_backoff > 0) --c->np1_backoff;
        else c->bf16_np_now = 1;
    }
    host_trace(c, 3);
    // start of the GPU time of the call (timing.total_ms).  Banded: recorded by run_score_banded once the first band's
    // copy is on its way - nothing is queued ahead of that copy that does not have to be (every API call is 5-10 us)
    if (!banded) HIPC(hipEventRecord(c->ev[0], c->stream));
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
    host_trace(c, 9);
    c->cand_on = false;
    const bool pin_direct = c->cand_pin_now;
    c->cand_pin_now = false;
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
    S.pin_direct = pin_direct;
    S.banded_u8 = banded && c->dtype == MTM_U8;
    if (want_prefetch) {
        // Few candidates (the usual case): they are in the pinned landing buffer when the stream is done and the 3x3 test
        // runs on the host (fm_end) - written there by the score kernel itself (pin_direct), else by a one-group kernel
        const size_t nfetch = nfetch_w;
        if (!pin_direct) {
            hipLaunchKernelGGL(fetch_cands_kernel, dim3(1), dim3(256), 0, c->stream, c->cands.as<uint4>(),
                               static_cast<uint4*>(c->pinned), (unsigned long long)nfetch);
            HIPC(hipGetLastError());
        }
        HIPC(hipEventRecord(c->ev[2], c->stream));
        S.prefetched = true;
    }
    return MTM_OK;
}

// The device's share of the non-maxima suppression (mtm_k_nms.hip.h), queued right behind the peak pass: the length of the
// peak list at `dhits` is still on the device (`dcount`), the launches read it there and do nothing unless it lies in
// [nms_device_min, n_max].  What a neighbourhood's best hit suppresses stays on the device; fetch_device_nms() brings the
// rest: first the hits that are certainly kept (nothing earlier overlaps them), then the undecided ones.
struct DeviceNms {
    bool queued = false;
    unsigned n_max = 0;
    const unsigned long long* cnt_pin = nullptr;   // landing buffer of the two counters (champions, undecided), page-locked
    const mtm_hit* out = nullptr;
};

int queue_device_nms(mtm_ctx* c, const mtm_hit* dhits, const unsigned long long* dcount, bool ascending, DeviceNms* q) {
    // (a cell larger than the largest box side is still correct - the 3x3 cell neighbourhood covers every partner - and
    // small templates on a large image would otherwise make millions of cells to clear and scan on every attempt)
    int cell = 32;
    for (const TemplDev& d : c->td_host) cell = std::max(cell, std::max(d.rows, d.cols));
    const size_t n_max = (size_t)std::min<int64_t>(c->hit_cap, 1ll << 18);
    NmsParams p{};
    p.hits = dhits;
    p.n_ptr = dcount;
    p.n_min = (unsigned)std::min<long long>(c->nms_device_min, 1ll << 30);
    p.n_max = (unsigned)n_max;
    p.ascending = ascending ? 1 : 0;
    // MTM/NMS.py:73-78: the scores are float32 (1 - score for the difference methods), the threshold a python float
    // transformed in double and narrowed by the cv2 binding
    p.thr_score = (float)(ascending ? (1.0 - c->nms_req.score_threshold) : c->nms_req.score_threshold);
    p.thr_overlap = (float)c->nms_req.max_overlap;
    p.cell = cell;
    p.gw = c->cols / cell + 3;
    p.gh = c->rows / cell + 3;
    const size_t n_cells = (size_t)p.gw * p.gh;
    const size_t off_rank = round_up(sizeof(unsigned) * (n_cells + 1), 256), off_status = off_rank + round_up(sizeof(unsigned) * n_max, 256);
    const size_t off_sorted = off_status + round_up(sizeof(int) * n_max, 256);
    const size_t off_hdr = off_sorted + round_up(sizeof(mtm_hit) * n_max, 256), off_out = off_hdr + 256;
    MTMC(c->nms_buf.ensure(off_out + sizeof(mtm_hit) * n_max));
    uint8_t* b = c->nms_buf.as<uint8_t>();
    p.cell_cnt = reinterpret_cast<unsigned*>(b);
    p.rank = reinterpret_cast<unsigned*>(b + off_rank);
    p.status = reinterpret_cast<int*>(b + off_status);
    p.sorted = reinterpret_cast<mtm_hit*>(b + off_sorted);
    p.out_count = reinterpret_cast<unsigned long long*>(b + off_hdr);
    p.out = reinterpret_cast<mtm_hit*>(b + off_out);
    HIPC(hipMemsetAsync(p.cell_cnt, 0, sizeof(unsigned) * (n_cells + 1), c->stream));
    HIPC(hipMemsetAsync(b + off_hdr, 0, 16, c->stream));
    const unsigned blocks = (unsigned)((n_max + 255) / 256);
    hipLaunchKernelGGL(nms_count_kernel, dim3(blocks), dim3(256), 0, c->stream, p);
    hipLaunchKernelGGL(nms_offsets_kernel, dim3(1), dim3(1024), 0, c->stream, p);
    hipLaunchKernelGGL(nms_scatter_kernel, dim3(blocks), dim3(256), 0, c->stream, p);
    hipLaunchKernelGGL(nms_champion_kernel, dim3(blocks), dim3(256), 0, c->stream, p);
    hipLaunchKernelGGL(nms_prune_kernel, dim3(blocks), dim3(256), 0, c->stream, p);
    HIPC(hipGetLastError());
    // (into page-locked memory: a device-to-host copy into a pageable stack slot may hold the host until the whole peak pass
    // is done - the round trip this queueing exists to avoid)
    if (!c->pin_small) HIPC(hipHostMalloc(&c->pin_small, 64, hipHostMallocDefault));
    q->cnt_pin = static_cast<unsigned long long*>(c->pin_small);
    HIPC(hipMemcpyAsync(c->pin_small, b + off_hdr, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    q->queued = true;
    q->n_max = (unsigned)n_max;
    q->out = p.out;
    return MTM_OK;
}

// after the stream was synchronised (q.cnt has landed): the pruned list of `count` peaks -> `rest`, *n_sure = champions
int fetch_device_nms(mtm_ctx* c, const DeviceNms& qd, unsigned long long count, std::vector<mtm_hit>& rest, long long* n_sure) {
    struct {
        unsigned long long cnt[2];
        const mtm_hit* out;
    } q{{qd.cnt_pin ? qd.cnt_pin[0] : 0ull, qd.cnt_pin ? qd.cnt_pin[1] : 0ull}, qd.out};
    if (q.cnt[0] + q.cnt[1] > count) {
        set_error("mtm_find_matches_image_nms: internal state (pruned list longer than the peak list)");
        return MTM_E_STATE;
    }
    if (c->host_trace && c->trace_calls <= 12)
        std::fprintf(stderr, "[mtm host trace] device NMS: %llu peaks -> %llu champions + %llu for the host's pass\n", count,
                     q.cnt[0], q.cnt[1]);
    rest.resize((size_t)(q.cnt[0] + q.cnt[1]));
    if (q.cnt[0]) HIPC(hipMemcpyAsync(rest.data(), q.out, sizeof(mtm_hit) * (size_t)q.cnt[0], hipMemcpyDeviceToHost, c->stream));
    if (q.cnt[1])
        HIPC(hipMemcpyAsync(rest.data() + q.cnt[0], q.out + (count - q.cnt[1]), sizeof(mtm_hit) * (size_t)q.cnt[1],
                            hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    *n_sure = (long long)q.cnt[0];
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
        for (int attempt = 0; attempt < 3; ++attempt) {
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
            if (!refined || (int64_t)nlisted <= cand_cap) {
                if (refined && c->bf16_np_now == 1) c->np1_backoff_len = 16;
                break;
            }
            if (c->bf16_np_now == 1) {
                // the one-product screen's bounds let more outputs reach their template's best than the list holds: the
                // same route with three piece products (and the next calls start there)
                c->bf16_np_now = 3;
                c->np1_backoff = c->np1_backoff_len;
                c->np1_backoff_len = std::min(2 * c->np1_backoff_len, 1024);
                c->timing.ncc_launches = 0;
                c->timing.sq_launches = 0;
                HIPC(hipMemsetAsync(c->counters.p, 0, sizeof(unsigned long long) * 2 * std::max(1, n), c->stream));
                HIPC(hipMemsetAsync(c->cands.p, 0, 16, c->stream));
                MTMC(run_score_all(c));
                HIPC(hipEventRecord(c->ev[1], c->stream));
                continue;
            }
            // float32 refinement: more outputs within the margin of their template's best than the list holds (near-flat
            // maps) - the float64 kernel decides, on maps in memory
            c->refine_now = false;
            c->f32_exact_now = true;
            c->ext_now = false;
            c->hits_only_now = false;
            c->cand_on = false;
            c->timing.ncc_launches = 0;
            c->timing.sq_launches = 0;
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
        // [hit count | candidate count | spare word of the candidate header (float32 map mode: the "bound too wide" flag) | per-template ints]
        const size_t hdr_bytes = round_up(3 * sizeof(unsigned long long) + sizeof(int) * (size_t)std::max(1, n), 16);
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
            host_trace(c, 10);
            uint8_t* land = static_cast<uint8_t*>(c->pinned);
            unsigned long long ncand = 0;
            if (S.pin_direct) {
                // the window's slots fill from 0 upwards (every reserved slot below the capacity is written before the
                // launch ends): the count is the first slot that still says "no record"
                const mtm_hit* w = reinterpret_cast<const mtm_hit*>(land + 16);
                while (ncand < nfetch && w[ncand].templ_idx >= 0) ++ncand;
                if (ncand == nfetch) {      // window full: the list's real length is on the device (dense maps; rare)
                    HIPC(hipMemcpyAsync(&ncand, c->cands.p, sizeof(ncand), hipMemcpyDeviceToHost, c->stream));
                    HIPC(hipStreamSynchronize(c->stream));
                }
                std::memcpy(land, &ncand, sizeof(ncand));
            }
            std::memcpy(&ncand, land, sizeof(ncand));
            std::memcpy(&c->timing.sclk_mhz, land + 8, sizeof(float));
            if (ncand <= nfetch) {
                // everything needed is on the host: clear the counter for the next call while this one finishes
                // (unless this context's calls clear it in their own first kernel: banded uint8 calls)
                if (!S.banded_u8 && hipMemsetAsync(c->cands.p, 0, 16, c->stream) == hipSuccess)
                    c->cands_zeroed = c->cands.p;
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
                if (c->refine_now && c->bf16_np_now == 1) c->np1_backoff_len = 16;
            }
        }
        if (!verified_on_host && c->hits_only_now)
            HIPC(hipMemsetAsync(c->chash.p, 0, ((size_t)hash_mask + 1) * sizeof(unsigned long long), c->stream));
        if (pp_mode) use_fused = true;          // the potential peaks are in the candidate buffer, their neighbourhoods in the maps
        DeviceNms dnms;                 // (the device's share of a suppression request, queued behind the flagged-segment peak pass)
        for (int attempt = 0; attempt < 5 && n2d > 0 && !verified_on_host; ++attempt) {
            dnms.queued = false;
            MTMC(c->hits.ensure(hdr_bytes + sizeof(mtm_hit) * (size_t)c->hit_cap));
            uint8_t* dbase = c->hits.as<uint8_t>();
            HIPC(hipMemsetAsync(dbase, 0, hdr_bytes, c->stream));
            unsigned long long* counter = reinterpret_cast<unsigned long long*>(dbase);
            int* flags = reinterpret_cast<int*>(counter + 3);
            mtm_hit* dhits = reinterpret_cast<mtm_hit*>(dbase + hdr_bytes);
            if (use_fused) {
                // counter[1] <- candidate count (for the overflow check on the host)
                HIPC(hipMemcpyAsync(counter + 1, c->cands.p, 2 * sizeof(unsigned long long), hipMemcpyDeviceToDevice,
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
                if (c->sparse_now) {
                    const dim3 grd((max_ow + kPkCols - 1) / kPkCols, (max_oh + 4 * kPkSparseRows - 1) / (4 * kPkSparseRows), n2d);
                    // a list per (template, strip column) (at most 64 MB of them) + their counters, then one list for the host
                    const unsigned long long n_lists = (unsigned long long)n2d * grd.x;
                    const unsigned long long cap_t = std::max<unsigned long long>(
                        256ull, std::min<unsigned long long>((unsigned long long)c->hit_cap / 8, (64ull << 20) / sizeof(mtm_hit) / n_lists));
                    const size_t cnt_bytes = round_up(sizeof(unsigned long long) * (size_t)n_lists, 256);
                    MTMC(c->hits_t.ensure(cnt_bytes + sizeof(mtm_hit) * (size_t)cap_t * (size_t)n_lists));
                    unsigned long long* counts_t = c->hits_t.as<unsigned long long>();
                    mtm_hit* hits_t = reinterpret_cast<mtm_hit*>(c->hits_t.as<uint8_t>() + cnt_bytes);
                    HIPC(hipMemsetAsync(counts_t, 0, cnt_bytes, c->stream));
                    hipLaunchKernelGGL(peaks_sparse_kernel, grd, dim3(256), 0, c->stream, c->maps.as<float>(),
                                       c->td.as<TemplDev>(), c->tlist.as<int>() + c->list2d_off, mode_min ? 1 : 0, thr,
                                       c->opt_border, hits_t, cap_t, counts_t, flags, c->seg_flags.as<uint8_t>(),
                                       c->flag_tstride, c->flag_rstride, c->seg_skip_used ? 1 : 0);
                    hipLaunchKernelGGL(compact_hits_kernel, dim3((unsigned)n_lists), dim3(256), 0, c->stream, hits_t, cap_t, counts_t,
                                       (int)n_lists, dhits, (unsigned long long)c->hit_cap, counter);
                    // a suppression request: its device share follows at once (it reads the list's length on the device)
                    dnms = DeviceNms{};
                    if (c->nms_req.on && c->nms_req.max_overlap >= 0.0)
                        MTMC(queue_device_nms(c, dhits, counter, mode_min, &dnms));
                } else
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
            std::memcpy(tflags.data(), host_buf.data() + 3 * sizeof(count), sizeof(int) * n);
            unsigned int rig_wide = 0;
            if (use_fused) std::memcpy(&rig_wide, host_buf.data() + 2 * sizeof(count), sizeof(rig_wide));
            if (pp_mode && c->refine_now && rig_wide != 0) {
                // float32 map mode: some output that could pass the threshold has an error bound beyond what the scan's
                // tolerances cover (a low-contrast window beside a brightness step) - the float64 kernel decides
                c->cand_on = false;
                c->hits_only_now = false;
                c->timing.ncc_launches = 0;
                c->timing.sq_launches = 0;
                pp_mode = false;
                use_fused = false;
                c->refine_now = c->refine_scan_now = false;
                c->f32_exact_now = true;
                MTMC(run_score_all(c));
                HIPC(hipEventRecord(c->ev[1], c->stream));
                continue;
            }
            if (use_fused && !pp_mode && (int64_t)ncand > cand_cap && c->refine_now && c->bf16_np_now == 1) {
                // float32 refinement, the ONE-PRODUCT screen listed more than the list holds: the same route with three piece
                // products - bounds 2^8 times tighter - before anything slower is tried (and the next calls start there)
                c->bf16_np_now = 3;
                c->np1_backoff = c->np1_backoff_len;
                c->np1_backoff_len = std::min(2 * c->np1_backoff_len, 1024);
                c->timing.ncc_launches = 0;
                c->timing.sq_launches = 0;
                HIPC(hipMemsetAsync(c->cands.p, 0, 16, c->stream));
                if (c->hits_only_now)
                    HIPC(hipMemsetAsync(c->chash.p, 0, ((size_t)hash_mask + 1) * sizeof(unsigned long long), c->stream));
                c->cand_on = true;
                const int rc_np = run_score_all(c);
                c->cand_on = false;
                MTMC(rc_np);
                HIPC(hipEventRecord(c->ev[1], c->stream));
                continue;
            }
            if (use_fused && (int64_t)ncand > cand_cap && c->refine_now) {
                // float32 refinement, list overflowed.  Kernel candidates (everything above the threshold): take the
                // potential peaks of a map scan instead - far fewer.  Those too (plateau-rich maps): the float64 kernel.
                c->cand_on = false;
                c->hits_only_now = false;
                c->timing.ncc_launches = 0;
            c->timing.sq_launches = 0;
                if (!pp_mode && !c->raw_rig_now) {
                    c->fuse_backoff = c->backoff_len;
                    c->backoff_len = std::min(2 * c->backoff_len, 1024);
                    pp_mode = true;
                    c->refine_scan_now = true;
                    HIPC(hipMemsetAsync(c->cands.p, 0, 16, c->stream));
                } else {
                    if (c->raw_rig_now) {           // raw sums have no map-scan route: the next calls start on the float64 kernel
                        c->fuse_backoff = c->backoff_len;
                        c->backoff_len = std::min(2 * c->backoff_len, 1024);
                        c->raw_rig_now = false;
                    }
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
                // the next calls on this context go straight to map mode (dense route, or the full peak pass); the period
                // doubles while the retries keep overflowing.  (An overflow of the dense route's own list - row maxima
                // only - is no retry: the back-off it runs under keeps counting down, this call takes the full peak pass.)
                c->fuse_backoff = c->backoff_len;
                c->backoff_len = std::min(2 * c->backoff_len, 1024);
                if (c->hits_only_now) {
                    // no maps in memory: compute them (this call pays twice - the overflowed launch left early)
                    c->hits_only_now = false;
                    c->cand_on = false;
                    c->timing.ncc_launches = 0;
            c->timing.sq_launches = 0;
                    MTMC(run_score_all(c));
                    HIPC(hipEventRecord(c->ev[1], c->stream));
                }
                continue;
            }
            if (use_fused && !pp_mode) c->backoff_len = 16;     // the candidates fitted
            if (use_fused && !pp_mode && c->refine_now && c->bf16_np_now == 1) c->np1_backoff_len = 16;
            if ((int64_t)count <= c->hit_cap) {
                // thousands of peaks and a suppression request: decide on the device, fetch the kept ones
                if (dnms.queued && c->sparse_now && !use_fused && (long long)count >= c->nms_device_min && count <= dnms.n_max) {
                    bool trivial = false;       // (a map every pixel of which equals its local maximum loses its peaks below)
                    for (int t : c->list2d) {
                        const unsigned f = (unsigned)tflags[(size_t)t];
                        trivial = trivial || ((f & 0xFFu) == 0 && !((f & 0xFF00u) != 0 && (f & 0xFF0000u) != 0));
                    }
                    if (!trivial) {
                        MTMC(fetch_device_nms(c, dnms, count, hits, &c->nms_sure));
                        c->nms_raw_count = (long long)count;
                        break;
                    }
                }
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
            // (the per-segment lists of the flagged route are bounded in total: if growing did not help, the full scan with
            // its single list takes over - the maps are complete)
            if (c->sparse_now && attempt >= 1) {
                c->sparse_now = false;
                if (c->seg_skip_used) {         // (round 5: ... unless the score pass left the unflagged segments out - once more, in full)
                    c->timing.ncc_launches = 0;
                    c->timing.sq_launches = 0;
                    c->seg_skip_used = false;   // (the re-run writes every output: the maps are complete and may be published)
                    MTMC(run_score_all(c));
                    HIPC(hipEventRecord(c->ev[1], c->stream));
                }
            }
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
                                          // (bytes 1 and 2: peaks_sparse_kernel - segments above and below the threshold exist)
                                          const unsigned f = (unsigned)tflags[h.templ_idx];
                                          return (f & 0xFFu) == 0 && !((f & 0xFF00u) != 0 && (f & 0xFF0000u) != 0);
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
        host_trace(c, 11);
        // (the device may have pruned the list already - queue_device_nms / fetch_device_nms -: the count of peaks is the one before that)
        const int64_t n_raw = c->nms_raw_count >= 0 ? (int64_t)c->nms_raw_count : (int64_t)hits.size();
        if (c->nms_req.on && n_raw > 1) {               // MTM.NMS (a list of one hit is returned as it is: MTM/NMS.py:53-55)
            const float thr_s = (float)(mode_min ? (1.0 - c->nms_req.score_threshold) : c->nms_req.score_threshold);
            std::vector<int32_t> keep;
            nms_select(hits.data(), (int64_t)hits.size(), mode_min ? 1 : 0, thr_s, (float)c->nms_req.max_overlap, keep,
                       c->nms_raw_count >= 0 ? c->nms_sure : 0);
            std::vector<mtm_hit> kept(keep.size());
            for (size_t i = 0; i < keep.size(); ++i) kept[i] = hits[(size_t)keep[i]];
            hits.swap(kept);
        } else {
            sort_hits(hits, mode_min);
        }
        if (c->nms_req.on && c->nms_req.n_object >= 0 && (long long)hits.size() > c->nms_req.n_object)
            hits.resize((size_t)c->nms_req.n_object);          // MTM/NMS.py:81-82
        c->timing.n_hits = n_raw;
        host_trace(c, 12);
    }
    HIPC(hipEventSynchronize(c->ev[2]));       // already complete: every path above synchronised the stream
    HIPC(hipEventElapsedTime(&c->timing.score_ms, c->ev[0], c->ev[1]));
    HIPC(hipEventElapsedTime(&c->timing.peaks_ms, c->ev[1], c->ev[2]));
    HIPC(hipEventElapsedTime(&c->timing.total_ms, c->ev[0], c->ev[2]));
    MTMC(collect_ncc_time(c));
    if (mode != MTM_PEAKS_LOCAL || !c->nms_req.on) c->timing.n_hits = (int64_t)hits.size();
    c->nms_raw_count = -1;
    c->nms_sure = 0;
    c->timing.hits_only = c->sparse_now ? 2 : c->hits_only_now ? 1 : 0;
    c->timing.f32_route = c->mbf_used ? 4 : c->f32_exact_now ? 3 : !c->refine_now ? 0 : (c->refine_scan_now ? 2 : 1);
    c->maps_valid = !c->hits_only_now && !c->ext_now && !c->seg_skip_used && !c->mbf_used;
    c->mbf_thr_on = false;
    c->refine_now = c->refine_scan_now = c->f32_exact_now = false;      // states of this call only
    c->sparse_now = false;
    c->raw_rig_now = false;
    c->zero_pending = false;
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

extern "C" {

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
    // float32 classes: maps of the raw-sum methods come from the float64 kernel (see fm_begin)
    c->f32_exact_now = resolved_kernel(c, sc) == MTM_KERNEL_MFMA_F32 &&
                       (c->method == MTM_TM_SQDIFF || c->method == MTM_TM_CCORR || c->method == MTM_TM_CCOEFF);
    const int rc_st = launch_stats(c, sc, &st);
    const int rc_nc = rc_st == MTM_OK ? launch_ncc(c, sc, sc.tlist_off + pos, 1, st, pos) : rc_st;
    c->f32_exact_now = false;
    MTMC(rc_nc);
    HIPC(hipMemcpy2DAsync(out, (size_t)out_row_stride_bytes, c->maps.as<float>() + d.map_off,
                          sizeof(float) * d.map_pitch, sizeof(float) * d.ow, d.oh, hipMemcpyDeviceToHost,
                          c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    MTMC(collect_ncc_time(c));
    return MTM_OK;
}

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
    host_trace(c, 0);
    MTMC(check_image_args(px, rows, cols, chans, dtype, row_stride_bytes, "mtm_find_matches_image"));
    const ImageArgs up{px, rows, cols, chans, dtype, row_stride_bytes};
    const int rc = find_matches_impl(c, mode, score_threshold, out, capacity, n_out, nullptr, &up);
    host_trace(c, 15);
    return rc;
}

int mtm_find_matches_image_nms(mtm_ctx* c, const void* px, int rows, int cols, int chans, int dtype, int64_t row_stride_bytes,
                               double score_threshold, double max_overlap, int64_t n_object, mtm_hit* out, int64_t capacity,
                               int64_t* n_out) {
    if (!c) {
        set_error("mtm_find_matches_image_nms: null context");
        return MTM_E_INVALID;
    }
    host_trace(c, 0);
    MTMC(check_image_args(px, rows, cols, chans, dtype, row_stride_bytes, "mtm_find_matches_image_nms"));
    const ImageArgs up{px, rows, cols, chans, dtype, row_stride_bytes};
    c->nms_req.on = true;
    c->nms_req.score_threshold = score_threshold;
    c->nms_req.max_overlap = max_overlap;
    c->nms_req.n_object = n_object;
    const int rc = find_matches_impl(c, MTM_PEAKS_LOCAL, score_threshold, out, capacity, n_out, nullptr, &up);
    c->nms_req.on = false;
    c->nms_raw_count = -1;
    host_trace(c, 15);
    return rc;
}

// One step of the process-per-GPU form in ONE native call (round 5): this rank's shard is searched, its hits get their list
// positions in the caller's whole template list (global_idx), the ranks' lists are exchanged (mtm_comm_allgather_hits: the
// context's communicator; none / one rank: no exchange), merged in template order and suppressed - every rank returns
// the same kept hits.  What MTM.distributed.matchTemplates_sharded did in four steps from Python (search, remap,
// all-gather, mtm_nms) with the reference's fan-in (MTM/__init__.py:173-177) and MTM/NMS.py:53-84 behind it.
int mtm_find_matches_image_sharded_nms(mtm_ctx* c, const void* px, int rows, int cols, int chans, int dtype, int64_t row_stride_bytes,
                                       double score_threshold, double max_overlap, int64_t n_object, int method,
                                       const int32_t* global_idx, int n_local_templ, mtm_hit* out, int64_t capacity,
                                       int64_t* n_out) {
    if (!c || !n_out || capacity < 0 || (capacity > 0 && !out) || n_local_templ < 0 || (n_local_templ > 0 && !global_idx)) {
        set_error("mtm_find_matches_image_sharded_nms: bad arguments");
        return MTM_E_INVALID;
    }
    // This is a COLLECTIVE: whatever goes wrong on this rank before the exchange (arguments that differ per rank, a bad
    // image, a HIP error in the search) must not keep it away from the all-gather - the other ranks would wait in it without
    // a time-out of their own.  A failing rank joins with zero hits and its error code in the slot header; every rank then
    // returns an error (this one its own, the others MTM_E_COMM naming the rank) instead of a list with hits missing.
    std::vector<mtm_hit> local;
    int local_rc = MTM_OK;
    std::string local_msg;
    if (n_local_templ > 0) {                    // (a rank without units still takes part in the exchange)
        if ((int)c->templs.size() != n_local_templ || c->method != method) {
            set_error("mtm_find_matches_image_sharded_nms: global_idx / method do not match the context's template set");
            local_rc = MTM_E_INVALID;
        }
        host_trace(c, 0);
        if (local_rc == MTM_OK)
            local_rc = check_image_args(px, rows, cols, chans, dtype, row_stride_bytes, "mtm_find_matches_image_sharded_nms");
        if (local_rc == MTM_OK) {
            const ImageArgs up{px, rows, cols, chans, dtype, row_stride_bytes};
            int64_t n = 0;
            const int rc = find_matches_impl(c, MTM_PEAKS_LOCAL, score_threshold, nullptr, 0, &n, nullptr, &up);
            if (rc != MTM_OK && rc != MTM_E_OVERFLOW) local_rc = rc;
        }
        if (local_rc == MTM_OK) {
            local = c->last_hits;               // (find_matches_impl keeps the whole list there whatever the capacity)
            for (mtm_hit& h : local) h.templ_idx = global_idx[h.templ_idx];
        } else {
            local_msg = mtm_last_error();
        }
    }
    std::vector<mtm_hit> all;
    if (c->comm && c->n_ranks > 1) {
        std::vector<int64_t> counts((size_t)c->n_ranks, 0);
        std::vector<int32_t> flags;
        int64_t n_all = 0;
        all.resize(std::max<size_t>(4096, local.size() * (size_t)c->n_ranks));
        int rc = comm_allgather_hits_flagged(c, local.data(), (int64_t)local.size(), (int32_t)local_rc, all.data(), (int64_t)all.size(),
                                             counts.data(), &n_all, &flags);
        if (rc == MTM_E_OVERFLOW) {             // (a local matter: the gathered slots are still in the staging area)
            all.resize((size_t)n_all);
            rc = mtm_comm_last_gather(c, all.data(), n_all, counts.data(), &n_all);
        }
        if (local_rc != MTM_OK) {               // this rank's own failure: its message, its code (the exchange has been served)
            set_error(local_msg);
            return local_rc;
        }
        if (rc != MTM_OK) return rc;
        for (size_t r = 0; r < flags.size(); ++r)
            if (flags[r] != 0) {
                set_error("mtm_find_matches_image_sharded_nms: rank " + std::to_string(r) + " failed its local search (code " +
                          std::to_string(flags[r]) + "); no list is returned");
                return MTM_E_COMM;
            }
        all.resize((size_t)n_all);
    } else {
        if (local_rc != MTM_OK) {
            set_error(local_msg);
            return local_rc;
        }
        all.swap(local);
    }
    std::stable_sort(all.begin(), all.end(), [](const mtm_hit& a, const mtm_hit& b) { return a.templ_idx < b.templ_idx; });
    const bool ascending = method == MTM_TM_SQDIFF || method == MTM_TM_SQDIFF_NORMED;
    if (all.size() > 1) {                       // (MTM/NMS.py:53-55: a list of one hit is returned as it is)
        const float thr_s = (float)(ascending ? (1.0 - score_threshold) : score_threshold);
        std::vector<int32_t> keep;
        nms_select(all.data(), (int64_t)all.size(), ascending ? 1 : 0, thr_s, (float)max_overlap, keep);
        std::vector<mtm_hit> kept(keep.size());
        for (size_t i = 0; i < keep.size(); ++i) kept[i] = all[(size_t)keep[i]];
        all.swap(kept);
    }
    if (n_object >= 0 && (int64_t)all.size() > n_object) all.resize((size_t)n_object);
    *n_out = (int64_t)all.size();
    c->last_hits.swap(all);
    if ((int64_t)c->last_hits.size() > capacity) {
        set_error("mtm_find_matches_image_sharded_nms: output capacity too small (fetch the result with mtm_last_hits)");
        return MTM_E_OVERFLOW;
    }
    if (!c->last_hits.empty()) std::memcpy(out, c->last_hits.data(), sizeof(mtm_hit) * c->last_hits.size());
    return MTM_OK;
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
}  // extern "C"
