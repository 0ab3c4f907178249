// libmtm_hip.so - launches: window statistics and score maps of a size class on whichever kernel runs it, the float32
// refinement behind the bf16 kernel, the score pass of a whole call (plain and with a banded image upload).
#include "mtm_ctx.h"

using namespace mtm;
using namespace mtmi;
#include "mtm_k_stats.hip.h"
#include "mtm_k_score.hip.h"
#include "mtm_refine.hip.h"
#include "mtm_maskf32.hip.h"

namespace {

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

// raw-mode instantiations of ncc_mfma_kernel (biased int32 accumulators stored as they are)
MfmaFn mfma_raw_fn(bool row_mux, bool packed_k) {
    MfmaSel s;
    s.method = kMfRaw;
    s.rm = row_mux;
    s.kp = packed_k;
    return mfma_kernel(s);
}

}  // namespace

namespace mtmi {

bool dot_variant_ok(int64_t v) { return v >= 0 && v < kNumDotVariants && !kDotVariants[v].wide; }

// The byte planes of I^2 (masked classes: sum I^2 M on the matrix cores), once per image; a no-op for everything but
// single-channel uint8 images with a masked class on the MFMA kernel.
int ensure_square_planes(mtm_ctx* c) {
    if (c->sq_valid || c->dtype != MTM_U8 || c->chans != 1) return MTM_OK;
    bool any = false;
    for (const SizeClass& sc : c->classes)
        any = any || (sc.masked && sc.mask_rm_off >= 0 && resolved_kernel(c, sc) == MTM_KERNEL_MFMA);
    if (!any) return MTM_OK;
    const ImageDev img = image_dev(c);
    const size_t plane_bytes = (size_t)img.u8_plane;
    MTMC(c->sq_planes.ensure(2 * plane_bytes));
    const size_t n16 = plane_bytes / 16;
    hipLaunchKernelGGL(square_planes_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, c->stream, img.u8, n16,
                       c->sq_planes.as<uint8_t>(), c->sq_planes.as<uint8_t>() + plane_bytes);
    HIPC(hipGetLastError());
    c->sq_valid = true;
    return MTM_OK;
}

// Window statistics of one size class (two kernels), into c->stats.  Returns the plane table.
// `sb0`, `sb1`: range of kStatBand4-row output blocks to compute (banded image upload; fused single-channel
// kernel only), sb1 < 0 = all.
int launch_stats(mtm_ctx* c, const SizeClass& sc, StatPlanes* out, int sb0, int sb1) {
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
        // (masked classes on the matrix cores: the sum2 plane is written by the sum I^2 M pass below, not here)
        const int want_sum2 = (!masked_mfma && (num_type == 2 || (normed && num_type != 1) || !want_t_always)) ? 1 : 0;
        const int owg = stats_u8_owg(w);
        const int nsb = (oh + kStatBand4 - 1) / kStatBand4;
        const int b1 = sb1 < 0 ? nsb : std::min(sb1, nsb);
        double* rsq = nullptr;
        if (sc.rm_R > 0 && normed) {
            MTMC(c->stats_rsq.ensure(sizeof(double) * plane));
            rsq = c->stats_rsq.as<double>();
        }
        double* blk = nullptr;
        // statistic ranges per 16-pixel column block: the hits-only screens of the multi-row and row-multiplexed tilings
        if (c->chans == 1 && ((sc.r2 > 0 && normed) || (sc.rm_R > 0 && (normed || (masked_mfma && method == MTM_TM_CCORR_NORMED))))) {
            st.blk_pitch = (st.pitch + 15) / 16;
            MTMC(c->stats_blk.ensure(sizeof(double) * 4 * (size_t)st.blk_pitch * oh));
            blk = c->stats_blk.as<double>();
            st.blk = blk;
        }
        if (b1 > sb0) {
            const dim3 gs((ow + owg - 1) / owg, b1 - sb0);
            // banded upload with the layout conversion of the band's rows on this launch (run_score_banded): the kernel reads
            // the raw upload buffer (every row that has arrived so far is there) and writes the planes of the new rows
            StatLayout lay{};
            const uint8_t* src = img.u8;
            int src_pitch = img.u8_pitch;
            if (c->lay_r1 > c->lay_r0) {
                mtm_ctx::ImageSlot& sl = c->slot[c->cur];
                lay.u8 = sl.u8.as<uint8_t>();
                lay.u8b = sl.u8b.as<uint8_t>();
                lay.pitch = img.u8_pitch;
                lay.r0 = c->lay_r0;
                lay.r1 = c->lay_r1;
                src = sl.raw.as<uint8_t>();
                src_pitch = c->cols;
            }
            if (c->zero_pending && c->stats_stream != nullptr) {        // banded call: the candidate header is cleared here
                lay.zero16 = c->cands.as<unsigned long long>();
                c->zero_pending = false;
            }
            hipLaunchKernelGGL(stats_u8_kernel, gs, dim3(256), 0, c->stats_stream ? c->stats_stream : c->stream, src,
                               src_pitch, h, w, oh, ow, owg, inv_area, num_type, normed ? 1 : 0, want_t, want_sum2, tp[0],
                               sum2, sq, st.pitch, rsq, sb0, blk, st.blk_pitch, lay);
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
                           st.pitch, 0);
    } else if (c->dtype == MTM_U16 && c->chans == 1 && w <= 768 && (double)w * h * 65535.0 < 4294967296.0 && c->fuse_stats) {
        // single-channel uint16: the fused kernel over the two byte planes (the ones the MFMA kernel reads)
        const int owg = stats_u8_owg(w);
        const int nsb = (oh + kStatBand4 - 1) / kStatBand4;
        const int b1 = sb1 < 0 ? nsb : std::min(sb1, nsb);
        const uint8_t* hib = c->slot[c->cur].u8b.as<uint8_t>();
        double* blk = nullptr;
        if (normed && rk == MTM_KERNEL_MFMA16) {    // statistic ranges per 16-pixel column block: the kernel's hits-only screen
            st.blk_pitch = (st.pitch + 15) / 16;
            MTMC(c->stats_blk.ensure(sizeof(double) * 4 * (size_t)st.blk_pitch * oh));
            blk = c->stats_blk.as<double>();
            st.blk = blk;
        }
        if (b1 > sb0) {
            const dim3 gs((ow + owg - 1) / owg, b1 - sb0);
            hipLaunchKernelGGL(stats_u16_kernel, gs, dim3(256), 0, c->stats_stream ? c->stats_stream : c->stream, hib,
                               hib + img.u8_plane, img.u8_pitch, h, w, oh, ow, owg, inv_area, num_type, normed ? 1 : 0, want_t, 1,
                               tp[0], sum2, sq, st.pitch, blk, st.blk_pitch, sb0);
        }
    } else {
        // Round 6, banded float32 uploads (run_score_banded): c->lay_r0 .. lay_r1 are the image rows that have just arrived
        // (their horizontal sums), sb0 .. sb1 the output rows whose windows they complete, in blocks of kStatBand4 rows that
        // make whole vsum bands (a band restarts its column sums: the planes are bit for bit those of one launch)
        const bool part = c->lay_r1 > c->lay_r0 && w <= 1024;
        const hipStream_t ss = c->stats_stream ? c->stats_stream : c->stream;
        if (!part) MTMC(ensure_f32_plane(c));
        if (w <= 1024) {    // (the same sums in the same order, the row staged through LDS: mtm_k_stats.hip.h)
            const dim3 g1p(g1.x, part ? c->lay_r1 - c->lay_r0 : c->rows, c->chans);
            hipLaunchKernelGGL(hsum_lds_kernel, g1p, dim3(256), hsum_lds_bytes(w), ss, img.f32, img.f32_pitch,
                               img.f32_plane, c->rows, w, ow, c->hs1.as<double>(), c->hs2.as<double>(), hs_pitch,
                               hs_plane, part ? c->lay_r0 : 0);
        } else
            hipLaunchKernelGGL(hsum_kernel<double>, g1, dim3(256), 0, ss, img.f32, img.f32_pitch,
                               img.f32_plane, c->rows, w, ow, c->hs1.as<double>(), c->hs2.as<double>(), hs_pitch,
                               hs_plane);
        const int o0 = part ? sb0 * kStatBand4 : 0, o1 = part && sb1 >= 0 ? std::min(oh, sb1 * kStatBand4) : oh;
        if (o1 > o0) {
            const dim3 g2p(g2.x, (o1 - o0 + kVsumBand - 1) / kVsumBand);
            hipLaunchKernelGGL((vsum_stats_kernel<double, double>), g2p, dim3(256), 0, ss,
                               c->hs1.as<double>(), c->hs2.as<double>(), hs_pitch, hs_plane, c->chans, h, oh, ow,
                               inv_area, num_type, normed ? 1 : 0, want_t, tp[0], tp[1], tp[2], tp[3], sum2, sq, st.pitch,
                               o0 / kVsumBand);
        }
    }
    HIPC(hipGetLastError());
    for (int k = 0; k < kMaxChans; ++k) st.t[k] = tp[k];
    st.sum2 = sum2;
    st.sq = sq;
    if (masked_mfma && sc.mask_rm_off >= 0 && fused_stats) {
        // sum I^2 * M over every window on the matrix cores (see square_planes_kernel): two row-multiplexed
        // raw correlations of the byte planes of I^2 with the mask, combined into the sum2 plane
        const size_t plane_bytes = (size_t)img.u8_plane;
        MTMC(ensure_square_planes(c));
        const int map_pitch = (int)round_up((size_t)ow, 4);
        const long long raw_map = (long long)oh * map_pitch;
        const bool fused_sq = c->masksq_fused != 0;
        if (!fused_sq) MTMC(c->raw16.ensure(sizeof(int) * (size_t)(2 * raw_map)));
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
        const uint8_t* ap = c->apacks.as<uint8_t>() + sc.mask_rm_off + (long long)16 * p.nb * 1024;
        const int grid = ((p.n_work + 7) / 8) * 8;
        const double km257 = 257.0 * 128.0 * sc.mask_ones;      // the mask operand is not biased
        // the pass is part of the masked class's score work: its own event pair (bench.py: roofline.masked_stat)
        if ((int)c->sq_ev.size() <= c->timing.sq_launches) {
            hipEvent_t a, b;
            HIPC(hipEventCreate(&a));
            HIPC(hipEventCreate(&b));
            c->sq_ev.emplace_back(a, b);
        }
        auto& sqe = c->sq_ev[(size_t)c->timing.sq_launches];
        HIPC(hipEventRecord(sqe.first, c->stream));
        if (fused_sq) {
            // ONE launch: the byte planes of I^2 are its two channels, the epilogue writes sum2 and the block minima
            p.chans = 2;
            p.img = c->sq_planes.as<uint8_t>();
            p.plane = (long long)plane_bytes;
            p.sq_fused = 1;
            p.sq_k = km257;
            st.sum2 = sum2;             // (the kernel's output planes travel in the plane table)
            hipLaunchKernelGGL(mfma_raw_fn(true, false), dim3(grid), dim3(256), lds, c->stream, p,
                               c->td.as<TemplDev>(), c->tlist.as<int>(), ap, st, c->maps.as<float>());
        } else {
            for (int x = 0; x < 2; ++x) {
                p.img = c->sq_planes.as<uint8_t>() + (size_t)x * plane_bytes;
                p.raw_out = c->raw16.as<int>() + (size_t)x * raw_map;
                hipLaunchKernelGGL(mfma_raw_fn(true, false), dim3(grid), dim3(256), lds, c->stream, p,
                                   c->td.as<TemplDev>(), c->tlist.as<int>(), ap, st, c->maps.as<float>());
            }
            hipLaunchKernelGGL(masksq_combine_kernel, dim3((ow + 255) / 256, oh), dim3(256), 0, c->stream, c->raw16.as<int>(),
                               c->raw16.as<int>() + raw_map, map_pitch, sum2, st.pitch, km257, oh, ow,
                               const_cast<double*>(st.blk), st.blk_pitch);
        }
        HIPC(hipGetLastError());
        HIPC(hipEventRecord(sqe.second, c->stream));
        c->timing.sq_launches++;
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

// Masked float32 class on the bf16 matrix cores (mtm_maskf32.hip.h): J = I^2 and the window sums of I and J, two raw
// launches of ncc_bf16_kernel (I x U = T M^2, J x V = M^2) into scratch maps, the combine pass (placeholders + the list of
// outputs that could pass the threshold), exact re-scoring of the list into the score maps.  *done = false: the list
// overflowed (a threshold that passes nearly everything) - the caller runs the float64 kernel.
static int launch_masked_bf16(mtm_ctx* c, const SizeClass& sc, float* maps, bool* done) {
    *done = false;
    const int h = sc.h, w = sc.w, oh = c->rows - h + 1, ow = c->cols - w + 1, n_all = (int)sc.members.size();
    const ImageDev img = image_dev(c);
    const size_t plane_floats = (size_t)img.f32_plane;
    MTMC(c->f32_sq.ensure(sizeof(float) * plane_floats));
    if (!c->f32_sq_valid) {
        hipLaunchKernelGGL(square_f32_kernel, dim3((unsigned)((plane_floats / 4 + 255) / 256)), dim3(256), 0, c->stream, img.f32,
                           c->f32_sq.as<float>(), plane_floats / 4);
        c->f32_sq_valid = true;
    }
    // window sums (float64): S1, S2 of I and of J - the two-pass kernels of the float32 statistics, on both planes
    const int st_pitch = (int)round_up((size_t)ow, 4);
    const size_t st_plane = (size_t)st_pitch * oh;
    MTMC(c->mbf_stats.ensure(sizeof(double) * 4 * st_plane));
    double* s1i = c->mbf_stats.as<double>();
    double* s2i = s1i + st_plane;
    double* s1j = s2i + st_plane;
    double* s2j = s1j + st_plane;
    {
        const int hs_pitch = (int)round_up((size_t)ow, 4);
        const size_t hs_plane = (size_t)hs_pitch * c->rows;
        MTMC(c->hs1.ensure(sizeof(double) * hs_plane));
        MTMC(c->hs2.ensure(sizeof(double) * hs_plane));
        const dim3 g1((ow + 256 * kHsumSeg - 1) / (256 * kHsumSeg), c->rows, 1);
        const dim3 g2((ow + 255) / 256, (oh + kVsumBand - 1) / kVsumBand);
        for (int pl = 0; pl < 2; ++pl) {
            const float* src = pl == 0 ? img.f32 : c->f32_sq.as<float>();
            hipLaunchKernelGGL(hsum_lds_kernel, g1, dim3(256), hsum_lds_bytes(w), c->stream, src, img.f32_pitch, img.f32_plane, c->rows,
                               w, ow, c->hs1.as<double>(), c->hs2.as<double>(), hs_pitch, (long long)hs_plane, 0);
            hipLaunchKernelGGL((vsum_stats_kernel<double, double>), g2, dim3(256), 0, c->stream, c->hs1.as<double>(),
                               c->hs2.as<double>(), hs_pitch, (long long)hs_plane, 1, h, oh, ow, 1.0 / ((double)h * w), 0, 0, 1,
                               pl == 0 ? s1i : s1j, (double*)nullptr, (double*)nullptr, (double*)nullptr, pl == 0 ? s2i : s2j,
                               (double*)nullptr, st_pitch, 0);
        }
    }
    // the two raw launches
    MTMC(c->mbf_maps.ensure(sizeof(float) * 2 * std::max<size_t>(4, c->maps_floats)));
    const int mb = n_all > 16 ? 2 : 1;
    Bf16Params p{};
    p.pitch = img.f32_pitch;
    p.plane = img.f32_plane;
    p.chans = 1;
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
    p.method = MTM_TM_CCORR;                                    // raw sums: acc + centre * S1
    p.group_bytes = sc.mbf_group_bytes;
    p.piece_bytes = sc.mbf_group_bytes * mfma_groups_alloc(n_all);
    p.only_li = -1;
    p.n_work = p.nseg * p.nyb * p.ntg;
    const size_t n_tiles = (size_t)p.nseg * p.nyb;
    MTMC(c->mbf_mu.ensure(sizeof(float) * 2 * n_tiles));
    const size_t lds = bf16_lds_bytes(p.chunk_h, p.lds_cols);
    const int grid = ((p.n_work + 7) / 8) * 8;
    const int* tl_k = c->tlist.as<int>() + sc.tlist_off;
    StatPlanes st{};
    st.pitch = st_pitch;
    const unsigned long long cap = (unsigned long long)std::min<int64_t>(std::max<int64_t>(c->hit_cap, 1 << 18), 4096LL * 256);
    MTMC(c->mbf_list.ensure(16 + sizeof(mtm_hit) * (size_t)cap));
    MaskF32Params q{};
    unsigned long long count = 0;
    // Round 6: ONE piece product first (ncc_bf16_kernel<MB, 1>: bounds 2^-7 instead of 2^-15 of the norms' products - the
    // combine pass states them with the eps of the launch that ran); a list that overflows repeats the screen with three
    for (int np = c->bf16_np_now == 1 ? 1 : 3;; np = 3) {
    for (int pl = 0; pl < 2; ++pl) {
        p.img = pl == 0 ? img.f32 : c->f32_sq.as<float>();
        p.mu_out = c->mbf_mu.as<float>() + (size_t)pl * n_tiles;
        st.t[0] = pl == 0 ? s1i : s1j;
        st.sum2 = pl == 0 ? s2i : s2j;
        st.sq = st.sum2;                                        // (not read by a raw-sum launch)
        const uint8_t* ap = c->apacks.as<uint8_t>() + (pl == 0 ? sc.mbf_off_u : sc.mbf_off_v);
        const TemplDev* tdp = (pl == 0 ? c->td_u : c->td_v).as<TemplDev>();
        hipLaunchKernelGGL(bf16_kernel(mb, np), dim3(grid), dim3(256), lds, c->stream, p, tdp, tl_k, ap, st, c->mbf_maps.as<float>());
    }
    c->timing.f32_pieces = np;
    // combine: placeholders into the score maps, the rest listed
    HIPC(hipMemsetAsync(c->mbf_list.p, 0, 16, c->stream));
    q = MaskF32Params{};
    q.m1 = q.m2 = c->mbf_maps.as<float>();
    q.td = c->td.as<TemplDev>();
    q.td_u = c->td_u.as<TemplDev>();
    q.td_v = c->td_v.as<TemplDev>();
    q.tlist = tl_k;
    q.s1i = s1i;
    q.s2i = s2i;
    q.s1j = s1j;
    q.s2j = s2j;
    q.st_pitch = st_pitch;
    q.mu_i = c->mbf_mu.as<float>();
    q.mu_j = c->mbf_mu.as<float>() + n_tiles;
    q.nseg = p.nseg;
    q.method = c->method;
    q.mode_min = c->method == MTM_TM_SQDIFF ? 1 : 0;
    q.thr = c->mbf_thr;
    q.eps = bf16_rig_eps(1, h, p.nkb, np);
    q.h = h;
    q.w = w;
    q.oh = oh;
    q.ow = ow;
    q.maps = maps;
    q.list = reinterpret_cast<mtm_hit*>(c->mbf_list.as<uint8_t>() + 16);
    q.counter = c->mbf_list.as<unsigned long long>();
    q.cap = cap;
    if (c->mbf_global) {
        // N_object == 1: the templates' best lower bounds first (ordered-float keys behind the list's header + records), then
        // everything that reaches them
        MTMC(c->mbf_best.ensure(sizeof(unsigned int) * c->templs.size()));
        HIPC(hipMemsetAsync(c->mbf_best.p, 0, sizeof(unsigned int) * c->templs.size(), c->stream));
        hipLaunchKernelGGL(maskf32_best_kernel, dim3((ow + 255) / 256, oh), dim3(256), 0, c->stream, q, n_all, c->mbf_best.as<unsigned int>());
        hipLaunchKernelGGL(maskf32_list_best_kernel, dim3((ow + 255) / 256, oh), dim3(256), 0, c->stream, q, n_all,
                           c->mbf_best.as<unsigned int>());
    } else {
        hipLaunchKernelGGL(maskf32_combine_kernel, dim3((ow + 255) / 256, oh), dim3(256), 0, c->stream, q, n_all);
    }
    HIPC(hipGetLastError());
    // how many?  (one small read-back: this path is milliseconds long, and an overflowing list changes the route)
    HIPC(hipMemcpyAsync(&count, c->mbf_list.p, sizeof(count), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    if (count <= cap) {
        if (np == 1) c->np1_backoff_len = 16;
        break;
    }
    if (np == 3) return MTM_OK;                                 // *done stays false: float64 kernel
    c->bf16_np_now = 3;                                         // (this call's other classes and the next calls start with three)
    c->np1_backoff = c->np1_backoff_len;
    c->np1_backoff_len = std::min(2 * c->np1_backoff_len, 1024);
    }
    if (count > 0) {
        RefineParams r{};
        r.img = img;
        r.td = c->td.as<TemplDev>();
        r.weights = c->weights.as<double>();
        r.method = c->method;
        r.cls = c->td_host[(size_t)sc.members[0]].cls;
        r.ring = 0;
        r.list = q.list;
        r.count = q.counter;
        r.cap = cap;
        r.maps = maps;
        const unsigned rgrid = (unsigned)std::min<unsigned long long>(count, 16384ull);
        hipLaunchKernelGGL(refine_rescore_masked_kernel, dim3(rgrid), dim3(64), 0, c->stream, r);
        HIPC(hipGetLastError());
    }
    c->mbf_used = true;
    *done = true;
    return MTM_OK;
}

// Score maps of `n_list` templates of class `sc` (device list at tlist + list_off).
// `yb0`, `yb1`: range of output row blocks (MFMA kernel only; banded image upload), yb1 < 0 = all.
int launch_ncc(mtm_ctx* c, const SizeClass& sc, int list_off, int n_list, const StatPlanes& st, int only_li, int yb0,
               int yb1) {
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
    hipStream_t ncc_s = c->stream;
    // (The events are stream commands of their own.  Handing them to the launch itself - hipExtLaunchKernelGGL's start /
    // stop events - was measured in round 4: the gaps around the launches stayed, the call got 14 us SLOWER;
    // profiles/r04_r04w.)
    bool ev_own = false;                // a branch below records the pair itself (on the stream its launch goes to)
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
        const bool rmr = sc.slab_R > 0;
        // The slabs' launches are independent.  One of them is (output rows / 8R) x (columns / 256) work items - 91 for the
        // reference's own benchmark shape (2048^2 image, one 414 x 400 template: four slabs), against 512 resident
        // work-group slots: launches that do not fill the chip twice over run side by side on up to slab_concurrency
        // streams, forked from and joined into the score stream.
        int n_side = 1;
        {
            const int rows_per_item = rmr ? 8 * sc.slab_R : kMfRows;
            const long long items = (long long)((ow + kMfSeg - 1) / kMfSeg) * ((oh + rows_per_item - 1) / rows_per_item) *
                                    (rmr ? 1 : (n_all + 31) / 32);
            const int cus = c->n_cus > 0 ? c->n_cus : 256;
            if (items < 4LL * cus) n_side = (int)std::min<long long>(std::min(S, kSlabStreams), (4LL * cus + items - 1) / items);
        }
        // MTM_SLAB_MERGE (default 1): slabs of equal height and block count go out as ONE launch (MfmaParams::n_slab) - the
        // hardware fills the chip from one grid and back-fills as work-groups finish, where several launches on several
        // streams share four hardware queues (the fourth of four side-by-side slab launches started when the first had
        // ended: profiles/r04_r04v_slab) - on one side stream, so that the statistics pass still runs under it.
        bool merged = rmr && S > 1;
        for (int k = 1; k < S && merged; ++k) {
            const SizeClass::Slab &a = sc.slabs[0], &b = sc.slabs[(size_t)k];
            merged = (b.r1 - b.r0) == (a.r1 - a.r0) && (b.c1 - b.c0 + 63) / 64 == (a.c1 - a.c0 + 63) / 64 &&
                     b.apack_off - a.apack_off == (long long)k * rm_pack_bytes(a.r1 - a.r0, a.c1 - a.c0, sc.slab_R);
        }
        if (merged) n_side = c->slab_fork_early ? 2 : 1;        // (2: "side streams in use"; only the first one is)
        if (n_side > 1) {
            while ((int)c->slab_streams.size() < n_side) {
                hipStream_t s2;
                hipEvent_t e2;
                HIPC(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
                HIPC(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
                c->slab_streams.push_back(s2);
                c->slab_done.push_back(e2);
            }
            if (!c->slab_fork) HIPC(hipEventCreateWithFlags(&c->slab_fork, hipEventDisableTiming));
            if (!c->slab_fork_early) HIPC(hipEventRecord(c->slab_fork, ncc_s));      // (else: recorded ahead of the statistics)
            for (int i = 0; i < n_side; ++i) HIPC(hipStreamWaitEvent(c->slab_streams[(size_t)i], c->slab_fork, 0));
        }
        for (int k = 0; k < (merged ? 1 : S); ++k) {
            const SizeClass::Slab& sl = sc.slabs[(size_t)k];
            const int hs = sl.r1 - sl.r0, ws = sl.c1 - sl.c0;
            hipStream_t slab_s = n_side > 1 ? c->slab_streams[(size_t)(merged ? 0 : k % n_side)] : ncc_s;
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
            if (merged) {                       // every slab of the class: slab 0's geometry (the widest), S times the items
                int ncb = 0;
                for (const auto& q : sc.slabs) ncb += (q.ch == sl.ch && q.r0 == sl.r0) ? 1 : 0;
                p.n_slab = S;
                p.slab_ncb = ncb;
                p.slab_nrb = S / (ncb * c->chans);
                p.slab_cw = ncb > 1 ? sc.slabs[1].c0 - sl.c0 : 0;
                p.slab_rh = hs;
                p.slab_ap_step = rm_pack_bytes(hs, ws, sc.slab_R);
                p.slab_raw_step = (long long)n_all * raw_map;
                p.n_work *= S;
            }
            const size_t lds_main = (std::max<size_t>((size_t)tile_rows * p.lds_pitch, (size_t)kMfRows * kMfEpiBytesPerWave) + 15) &
                                    ~(size_t)15;
            p.tc_off = (int)lds_main;
            p.st_off = (int)((lds_main + sizeof(MfTemplConst) * 32 + kMfItemBytes + 15) & ~(size_t)15);
            const size_t lds = (size_t)p.st_off + (rmr ? 0 : (size_t)kMfRows * kMfStatBytesPerWave);
            const int grid = ((p.n_work + 7) / 8) * 8;
            const uint8_t* ap = c->apacks.as<uint8_t>() + sl.apack_off + (rmr ? (long long)sc.slab_R * p.nb * 1024 : 0);
            // (merged launch on a side stream: the timing pair goes there too - on the score stream it would bracket
            // the fork and the join, not the launch)
            if (merged && slab_s != ncc_s) HIPC(hipEventRecord(evp.first, slab_s));
            hipLaunchKernelGGL(mfma_raw_fn(rmr, false), dim3(grid), dim3(256), lds, slab_s, p, td,
                               c->tlist.as<int>() + sc.tlist_off, ap, st, maps);
            if (merged && slab_s != ncc_s) {
                HIPC(hipEventRecord(evp.second, slab_s));
                ev_own = true;
            }
        }
        for (int i = 0; i < (merged ? 1 : n_side) && n_side > 1; ++i) {
            HIPC(hipEventRecord(c->slab_done[(size_t)i], c->slab_streams[(size_t)i]));
            HIPC(hipStreamWaitEvent(ncc_s, c->slab_done[(size_t)i], 0));
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
        if (c->ext_now && only_li < 0) {              // fused global extremum: keys instead of maps / candidates
            q.ext_on = 1;
            q.ext_best = c->counters.as<unsigned long long>();
            q.cand_on = 0;
            q.hits_only = 1;
        }
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
        p.cand_on = (c->cand_on && only_li < 0) ? 1 : 0;
        p.hits_only = (p.cand_on && c->hits_only_now) ? 1 : 0;
        p.cand_thr_lo = (double)c->cand_thr - 1e-6 * std::max(1.0, std::fabs((double)c->cand_thr));
        p.screen_hi = std::min(p.cand_thr_lo, 0.999999) - 1e-6;
        p.sq_floor = 0.99 / std::sqrt((double)w * (double)h);
        p.screen_l1 = c->screen_l1;
        p.cand_min = c->cand_min ? 1 : 0;
        p.cand_thr = c->cand_thr;
        if (c->sparse_now && only_li < 0) {     // maps in memory + a flag per row segment that holds something above the threshold
            p.seg_flags = c->seg_flags.as<uint8_t>();
            p.flag_tstride = c->flag_tstride;
            p.flag_rstride = c->flag_rstride;
            const bool normed_m = c->method == MTM_TM_SQDIFF_NORMED || c->method == MTM_TM_CCORR_NORMED || c->method == MTM_TM_CCOEFF_NORMED;
            p.seg_skip = (c->seg_skip && !sc.masked && normed_m) ? 1 : 0;
            c->seg_skip_used = c->seg_skip_used || p.seg_skip != 0;
        }
        p.cand_cap = (unsigned long long)c->hit_cap;
        p.cand_counter = c->cands.as<unsigned long long>();
        p.cand_hits = reinterpret_cast<mtm_hit*>(c->cands.as<uint8_t>() + 16);
        // the 8 spare bytes of the candidate header carry the shader clock the kernel measured (fetched with it)
        p.clk_out = (p.cand_on && c->cands.p) ? reinterpret_cast<float*>(c->cands.as<uint8_t>() + 8) : nullptr;
        if (c->cand_pin_now && p.cand_on && c->pinned) {       // the head of the list also into the host's landing buffer
            p.cand_pin = reinterpret_cast<mtm_hit*>(static_cast<uint8_t*>(c->pinned) + 16);
            p.cand_pin_n = (unsigned long long)c->cand_pin_n;
            p.clk_out = reinterpret_cast<float*>(static_cast<uint8_t*>(c->pinned) + 8);
        }
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
        if (p.cand_on && !ext) {             // wave-private candidate staging (see emit_at)
            lds = (lds + 15) & ~(size_t)15;
            p.cs_off = (int)lds;
            lds += (size_t)kMfRows * kMfCandStageBytes;
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
        sel.masked = sc.masked;
        sel.rm = rm;
        sel.ch = (c->chans == 3 && !sc.masked) ? 3 : 1;
        sel.method = (c->chans == 1 || sel.ch == 3) ? c->method : -1;     // other channel counts: the generic epilogue
        sel.ext = ext;
        // IEEE division in the epilogue (the default since round 5: measured free on the hits-only path, profiles/r05*).  The
        // fused global extremum of MASKED classes only exists with the reciprocal normalisation (<= 1 ulp(float32) on ~1e-8
        // of the outputs); exact_div == 2 (strict) sends those calls through maps + extremum_kernel instead (fm_begin).
        sel.exact_div = c->exact_div != 0 && !(ext && sc.masked);
        sel.r2 = r2;
        sel.kp = sc.kp_nseg > 0;
        const MfmaFn fn = mfma_kernel(sel);
        if (!fn) {
            set_error("internal: no ncc_mfma_kernel instantiation for this class");
            return MTM_E_STATE;
        }
        hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, ncc_s, p, td, tl_k, ap, st, maps);
        c->timing.kernel_used = MTM_KERNEL_MFMA;
    } else if (kernel == MTM_KERNEL_MFMA16) {
        // uint16: ONE launch over the image's two byte planes (the "channels" of the launch) x [T_hi | T_lo] of 16 templates
        // per work item; the high-byte partial sums stay in registers while the low-byte plane is walked, and the epilogue
        // finishes the exact 16-bit correlation + normalisation (kMfU16, mtm_mfma.hip.h).
        const int n_all = (int)sc.members.size(), n_pad = sc.n_pad;
        MfmaParams p{};
        p.pitch = img.u8_pitch;
        p.plane = img.u8_plane;
        p.chans = 2;                                            // byte planes: high, low
        p.h = h;
        p.w = w;
        p.oh = oh;
        p.ow = ow;
        p.nb = (w + 63) / 64;
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
        int tg0 = 0;
        if (only_li >= 0) {                 // one template: just its group of 16
            tg0 = only_li / 16;
            p.ntg = 1;
        }
        if (yb1 >= 0) {                     // banded launch: row blocks yb0 .. yb1 - 1
            p.yb0 = yb0;
            p.nyb = std::min(yb1, p.nyb) - yb0;
        }
        p.n_work = p.nseg * p.nyb * p.ntg;
        const size_t lds_main = (std::max<size_t>((size_t)(std::min(h, kMfChunkH) + kMfRows - 1) * p.lds_pitch,
                                                  (size_t)kMfRows * kMfU16EpiBytesPerWave) + 15) & ~(size_t)15;
        p.tc_off = (int)lds_main;
        p.st_off = (int)((lds_main + sizeof(MfTemplConst) * 32 + kMfItemBytes + 15) & ~(size_t)15);
        size_t lds = (size_t)p.st_off;                          // (no statistics prefetch: the epilogue reads them from memory)
        const int grid = ((p.n_work + 7) / 8) * 8;
        const uint8_t* ap = c->apacks.as<uint8_t>() + sc.apack_off + (long long)tg0 * 2 * sc.group_bytes;
        const int* tl_k = c->tlist.as<int>() + sc.tlist_off + tg0 * 16;
        p.img = c->slot[c->cur].u8b.as<uint8_t>();             // plane 0: high bytes, plane 1: low bytes
        p.kp_nseg = sc.kp_nseg;
        p.kp_blocks = sc.kp_nseg ? kp_blocks(h, sc.kp_nseg) : 0;
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
        if (c->cand_pin_now && p.cand_on && c->pinned) {
            p.cand_pin = reinterpret_cast<mtm_hit*>(static_cast<uint8_t*>(c->pinned) + 16);
            p.cand_pin_n = (unsigned long long)c->cand_pin_n;
        }
        p.cand_thr_lo = (double)c->cand_thr - 1e-6 * std::max(1.0, std::fabs((double)c->cand_thr));
        p.screen_hi = std::min(p.cand_thr_lo, 0.999999) - 1e-6;
        p.sq_floor = 0.99 / std::sqrt((double)w * (double)h);
        p.screen_l1 = c->screen_l1;
        const bool ext = c->ext_now && only_li < 0;      // fused global extremum (find_matches_impl checked the classes)
        if (ext) {
            p.ext_off = (int)lds;                         // 4 waves x 32 keys
            lds += (size_t)kMfRows * 32 * sizeof(unsigned long long);
            p.ext_best = c->counters.as<unsigned long long>();
            p.cand_on = 1;
            p.hits_only = 1;
        }
        if (p.cand_on && !ext) {
            lds = (lds + 15) & ~(size_t)15;
            p.cs_off = (int)lds;
            lds += (size_t)kMfRows * kMfCandStageBytes;
        }
        MfmaSel sel16;
        sel16.method = kMfU16;
        sel16.kp = sc.kp_nseg > 0;
        sel16.ext = ext;
        sel16.exact_div = c->exact_div != 0;
        hipLaunchKernelGGL(mfma_kernel(sel16), dim3(grid), dim3(256), lds, ncc_s, p, td, tl_k, ap, st, maps);
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
        if (yb1 >= 0) {                             // a band of row blocks (run_score_banded, float32 uploads)
            p.yb0 = yb0;
            p.nyb = std::min(yb1, p.nyb) - yb0;
        }
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
        const bool raw_m = c->method == MTM_TM_SQDIFF || c->method == MTM_TM_CCORR || c->method == MTM_TM_CCOEFF;
        // piece products of this launch: one where only a list leaves the kernel and every listing decision rests on a
        // bound that knows it (below: rig 1 / 2, the refined extremum by bounds); three everywhere else
        int np = 3;
        if (c->bf16_np_now == 1 && c->refine_now && only_li < 0 && (p.hits_only || c->ext_now) &&
            (!raw_m || c->ext_now || c->raw_rig_now))
            np = 1;
        if (c->refine_now && !raw_m && only_li < 0) {
            // Round 5: the listing decisions of the refined routes by the rigorous per-output bound (Bf16Params::rig)
            p.rig = 1;
            p.rig_eps = bf16_rig_eps(c->chans, h, p.nkb, np);
            p.rig_thr = c->rig_thr;
            p.list_all = c->cand_min ? (c->rig_thr < -1.0f ? 1 : 0) : (c->rig_thr < 0.0f ? 1 : 0);
            if (c->refine_scan_now) {           // map mode: the scan's tolerances hold while no bound exceeds the cap
                p.rig_cap = c->rig_cap;
                p.rig_flag = reinterpret_cast<unsigned int*>(c->cands.as<uint8_t>() + 8);
            }
        }
        if (c->refine_now && raw_m && c->raw_rig_now && !c->ext_now && only_li < 0) {
            // raw sums with a threshold (round 5): everything whose UPPER bound passes is listed and re-scored exactly
            p.rig = 2;
            p.rig_eps = bf16_rig_eps(c->chans, h, p.nkb, np);
            p.rig_thr = c->rig_thr;
            p.list_all = 0;
        }
        if (c->ext_now && only_li < 0) {                  // fused global extremum (find_matches_impl checked the classes)
            p.ext_on = 1;
            p.ext_best = c->counters.as<unsigned long long>();
            p.cand_on = 1;
            p.hits_only = 1;
            p.ext_margin = c->refine_now ? kRefineThrMargin : 0.0f;
            if (raw_m && c->refine_now) {
                // rigorous bounds instead of a relative margin (Bf16Params::ext_raw): 2^-15 for the dropped piece products
                // and the two 16-bit representations, 2^-24 per float32 accumulation (three MFMAs per 32-tap block)
                p.ext_raw = 1;
                p.ext_eps = bf16_ext_eps(c->chans, h, p.nkb, np);
            } else if (p.rig) {
                p.ext_raw = 1;                              // (bounds of the quality instead of scores: the rig branch of the epilogue)
                p.list_all = 0;
            }
        }
        const size_t lds = bf16_lds_bytes(p.chunk_h, p.lds_cols);
        const int grid = ((p.n_work + 7) / 8) * 8;
        const uint8_t* ap = c->apacks.as<uint8_t>() + sc.apack_off + (long long)tg0 * mb * sc.group_bytes;
        const int* tl_k = c->tlist.as<int>() + sc.tlist_off + tg0 * 16 * mb;
        // (a launch that is neither listing by a bound nor the bound-keeping extremum must not run the screen: its scores
        // would be taken at face value)
        if (np == 1 && !(p.rig != 0 || (p.ext_on && p.ext_raw))) np = 3;
        if (np == 1 && !p.hits_only) np = 3;
        if (c->f32_mfma == 4) np = 1;               // (diagnostic: the screen's scores as they are)
        hipLaunchKernelGGL(bf16_kernel(mb, np), dim3(grid), dim3(256), lds, c->stream, p, td, tl_k, ap, st, maps);
        c->timing.kernel_used = MTM_KERNEL_MFMA_F32;
        c->timing.f32_pieces = np;
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
        // masked float32 class, local extrema against a threshold: two raw launches of the bf16 kernel as a screen + exact
        // re-scoring of everything that could pass (mtm_maskf32.hip.h); the float64 kernel only if that list overflows
        bool screened = false;
        if (sc.masked && sc.mask_bf16 && c->mbf_thr_on && only_li < 0 && n_list == (int)sc.members.size() && kernel == MTM_KERNEL_AUTO)
            MTMC(launch_masked_bf16(c, sc, maps, &screened));
        if (screened) {
            c->timing.kernel_used = MTM_KERNEL_MFMA_F32;
            c->timing.f32_route = 4;
        } else if (sc.masked)
            hipLaunchKernelGGL(ncc_f64_kernel<true>, grd, dim3(256), 0, c->stream, img, td, tl,
                               c->weights.as<double>(), st, c->method, maps, ntx);
        else
            hipLaunchKernelGGL(ncc_f64_kernel<false>, grd, dim3(256), 0, c->stream, img, td, tl,
                               c->weights.as<double>(), st, c->method, maps, ntx);
        if (c->timing.kernel_used == 0) c->timing.kernel_used = MTM_KERNEL_AUTO;
    }
    HIPC(hipGetLastError());
    if (!ev_own) HIPC(hipEventRecord(evp.second, ncc_s));
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

// side lanes 1 .. n of multi-class calls (mtm_ctx::Lane): a stream + its join event each
int ensure_lanes(mtm_ctx* c, int n) {
    while ((int)c->lanes.size() < n) {
        mtm_ctx::Lane L;
        HIPC(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
        HIPC(hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
        c->lanes.push_back(L);
    }
    return MTM_OK;
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

// refined global extremum: the keys the score kernel kept are approximate - rebuild them from the re-scored list
static int launch_refine_extremum(mtm_ctx* c) {
    const size_t n = c->templs.size();
    HIPC(hipMemsetAsync(c->counters.p, 0, sizeof(unsigned long long) * 2 * n, c->stream));
    const unsigned long long cap = (unsigned long long)std::min<int64_t>(c->hit_cap, 4096LL * 256);
    hipLaunchKernelGGL(refine_extremum_kernel, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, c->stream,
                       reinterpret_cast<const mtm_hit*>(c->cands.as<uint8_t>() + 16), c->cands.as<unsigned long long>(), cap,
                       c->td.as<TemplDev>(), c->cand_min ? 1 : 0, c->counters.as<unsigned long long>());
    HIPC(hipGetLastError());
    return MTM_OK;
}

// the map scan of the refined route: potential peaks of class `sc` (approximate maps in memory) -> candidate buffer
int launch_refine_scan(mtm_ctx* c, const SizeClass& sc) {
    const int oh = c->rows - sc.h + 1, ow = c->cols - sc.w + 1;
    const dim3 grd((ow + kPkCols - 1) / kPkCols, (oh + 4 * kPkRows - 1) / (4 * kPkRows), (unsigned)sc.members.size());
    hipLaunchKernelGGL(refine_scan_kernel, grd, dim3(256), 0, c->stream, c->maps.as<float>(), c->td.as<TemplDev>(),
                       c->tlist.as<int>() + sc.tlist_off, c->cand_min ? 1 : 0, c->scan_thr,
                       2.0f * c->rig_cap, c->opt_border,
                       reinterpret_cast<mtm_hit*>(c->cands.as<uint8_t>() + 16),
                       (unsigned long long)std::min<int64_t>(c->hit_cap, 4096LL * 256), c->cands.as<unsigned long long>());
    HIPC(hipGetLastError());
    return MTM_OK;
}

namespace {
// The context's stream and per-class scratch exchanged with a lane's for the time a class is queued (everything below
// launch_stats / launch_ncc keeps using c->stream, c->stats ...); swapped back on every path out.
struct LaneScope {
    mtm_ctx* c;
    mtm_ctx::Lane* L;
    void swap_all() {
        std::swap(c->stream, L->stream);
        std::swap(c->stats, L->stats);
        std::swap(c->stats_rsq, L->stats_rsq);
        std::swap(c->stats_blk, L->stats_blk);
        std::swap(c->hs1, L->hs1);
        std::swap(c->hs2, L->hs2);
        std::swap(c->raw16, L->raw16);
        std::swap(c->slab_raw, L->slab_raw);
        std::swap(c->stats_hi, L->stats_hi);
        std::swap(c->mask_td, L->mask_td);          // (the scratch template record of the dot4 sum I^2 M pass)
        std::swap(c->slab_streams, L->slab_streams);    // (side streams + fork / join events of a slab class)
        std::swap(c->slab_done, L->slab_done);
        std::swap(c->slab_fork, L->slab_fork);
    }
    LaneScope(mtm_ctx* c_, mtm_ctx::Lane* L_) : c(c_), L(L_) { if (L) swap_all(); }
    ~LaneScope() { if (L) swap_all(); }
    LaneScope(const LaneScope&) = delete;
    LaneScope& operator=(const LaneScope&) = delete;
};
}  // namespace

// Does anything class `sc` launches read the float32 plane of the image (the float64 / naive score kernels, the two-pass
// statistics)?  A banded uint8 upload leaves that plane out (ensure_f32_plane rebuilds it on demand).
static bool class_reads_f32(const mtm_ctx* c, const SizeClass& sc) {
    const int rk = resolved_kernel(c, sc);
    if (rk == MTM_KERNEL_NAIVE || rk == MTM_KERNEL_AUTO) return true;               // ncc_naive_kernel / ncc_f64_kernel
    if (c->dtype == MTM_U8) {
        const bool fused1 = c->chans == 1 && sc.w <= 768 && (double)sc.w * sc.h * 65025.0 < 4294967296.0 && c->fuse_stats;
        const bool fused3 = c->chans == 3 && sc.w <= 768 && 3.0 * sc.w * sc.h * 65025.0 < 4294967296.0 && c->fuse_stats;
        return !fused1 && !fused3 && c->cols > 8191;                                 // hsum_kernel on the float32 plane
    }
    return false;       // uint16 / float32 images are uploaded with their float32 plane
}

// Statistics + score launches of every size class but `skip` (the class a banded call has already queued; -1: none).
// `fork`: the event the lanes start behind instead of the present end of c->stream (banded calls: the last band's event -
// the image is complete - so that the next class runs under the banded class's last launch).
static int run_score_classes(mtm_ctx* c, int skip, hipEvent_t fork) {
    if (!c->hits_only_now) MTMC(ensure_maps(c));
    // several size classes: alternate them over lanes (see mtm_ctx::Lane).  Not while the float32 refinement is on: its
    // re-scoring kernels walk the candidate list of the class that just ran.
    int n_lanes = 1;
    const size_t n_todo = c->classes.size() - (skip >= 0 ? 1 : 0);
    bool mbf_any = false;                  // (masked float32 classes screened on the bf16 cores share their scratch buffers)
    for (const SizeClass& sc : c->classes) mbf_any = mbf_any || (sc.mask_bf16 && c->mbf_thr_on);
    if (c->classes.size() > 1 && n_todo > 0 && c->class_lanes > 1 && !c->refine_now && !c->refine_scan_now && !c->f32_exact_now && !mbf_any)
        n_lanes = (int)std::min<size_t>(skip >= 0 ? n_todo + 1 : n_todo, (size_t)c->class_lanes);
    if (n_lanes > 1) {
        MTMC(ensure_lanes(c, n_lanes - 1));
        if (!c->lane_fork) HIPC(hipEventCreateWithFlags(&c->lane_fork, hipEventDisableTiming));
        MTMC(ensure_square_planes(c));                  // shared by the masked classes of every lane: before the fork
        if (fork == nullptr || c->sq_valid) {           // (the planes of I^2 were just queued on c->stream: fork behind them)
            HIPC(hipEventRecord(c->lane_fork, c->stream));
            fork = c->lane_fork;
        } else {                                        // banded call: behind its set-up (lane_fork, recorded by
            for (int i = 0; i + 1 < n_lanes; ++i)       // run_score_banded) and behind the last band (fork)
                HIPC(hipStreamWaitEvent(c->lanes[(size_t)i].stream, c->lane_fork, 0));
        }
        for (int i = 0; i + 1 < n_lanes; ++i) HIPC(hipStreamWaitEvent(c->lanes[(size_t)i].stream, fork, 0));
        // The float32 plane a banded upload left out is built ONCE, ahead of every lane that may read it: built lazily it
        // would be queued on whichever lane asks first while the host already marks it valid, and a class on another
        // lane could read it before it is written.  On the first lane's stream (behind the complete image, not behind
        // the banded class's last score launch on c->stream); every other stream of the call waits for it.
        if (!c->slot[c->cur].f32_valid) {
            bool any = false;
            int k = -1;
            for (const SizeClass& sc : c->classes)
                if (++k != skip) any = any || class_reads_f32(c, sc);
            if (any) {
                mtm_ctx::Lane& L0 = c->lanes[0];
                if (!c->f32_built) HIPC(hipEventCreateWithFlags(&c->f32_built, hipEventDisableTiming));
                std::swap(c->stream, L0.stream);
                const int rc = ensure_f32_plane(c);
                std::swap(c->stream, L0.stream);
                MTMC(rc);
                HIPC(hipEventRecord(c->f32_built, L0.stream));
                HIPC(hipStreamWaitEvent(c->stream, c->f32_built, 0));
                for (int i = 1; i + 1 < n_lanes; ++i) HIPC(hipStreamWaitEvent(c->lanes[(size_t)i].stream, c->f32_built, 0));
            }
        }
    }
    int k_cls = skip >= 0 ? 1 : 0;                      // the banded class went to lane 0 (c->stream)
    int idx = -1;
    for (const SizeClass& sc : c->classes) {
        if (++idx == skip) continue;
        const int lane = n_lanes > 1 ? k_cls % n_lanes : 0;
        ++k_cls;
        LaneScope scope(c, lane > 0 ? &c->lanes[(size_t)(lane - 1)] : nullptr);
        StatPlanes st;
        // slabs: the raw launches read no statistics (slab_combine_kernel does) - their side streams fork HERE, ahead of
        // the statistics pass, which then runs under them (2048^2 x 414x400: hsum + vsum took 0.25 of the call's 1.16 ms)
        c->slab_fork_early = false;
        if (!sc.slabs.empty() && resolved_kernel(c, sc) == MTM_KERNEL_MFMA) {
            if (!c->slab_fork) HIPC(hipEventCreateWithFlags(&c->slab_fork, hipEventDisableTiming));
            HIPC(hipEventRecord(c->slab_fork, c->stream));
            c->slab_fork_early = true;
        }
        const int rc_st = launch_stats(c, sc, &st);
        const int rc_ncc = rc_st == MTM_OK ? launch_ncc(c, sc, sc.tlist_off, (int)sc.members.size(), st) : rc_st;
        c->slab_fork_early = false;
        MTMC(rc_ncc);
        if (c->refine_now && !c->f32_exact_now && resolved_kernel(c, sc) == MTM_KERNEL_MFMA_F32) {
            if (c->refine_scan_now) {
                MTMC(launch_refine_scan(c, sc));
                MTMC(launch_refine(c, sc, st, true, true));
            } else {
                MTMC(launch_refine(c, sc, st, false, !c->hits_only_now));
            }
        }
    }
    for (int i = 0; i + 1 < n_lanes; ++i) {             // join: everything after the score pass is queued on c->stream
        HIPC(hipEventRecord(c->lanes[(size_t)i].done, c->lanes[(size_t)i].stream));
        HIPC(hipStreamWaitEvent(c->stream, c->lanes[(size_t)i].done, 0));
    }
    if (c->refine_now && !c->f32_exact_now && c->ext_now) MTMC(launch_refine_extremum(c));
    return MTM_OK;
}

int run_score_all(mtm_ctx* c) { return run_score_classes(c, -1, nullptr); }

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
    float sq_ms = 0.f;
    for (int i = 0; i < c->timing.sq_launches; ++i) {
        float d = 0.f;
        HIPC(hipEventElapsedTime(&d, c->sq_ev[(size_t)i].first, c->sq_ev[(size_t)i].second));
        sq_ms += d;
    }
    c->timing.masked_stat_ms = sq_ms;
    return MTM_OK;
}

// Asynchronous half of mtm_find_matches: statistics, score kernels, the upload of the next image and (usual
// case) the copy of the candidate list into pinned memory are queued; nothing waits for the GPU.
// Can the image of a fused call arrive in row bands (copy / layout / statistics of band k+1 under the score
// kernel of band k)?  One unmasked single-channel uint8 size class on the MFMA kernel with the fused
// statistics kernel; anything else uploads the image in one piece (still without a round trip to the host).
static bool class_bandable(const mtm_ctx* c, const ImageArgs& a, const SizeClass& sc, double* work, bool any_fill = false);

// Several size classes: the class with the most multiply-accumulates among those that qualify is the one that runs under
// the upload (c->banded_cls); the others follow on the complete image (run_score_banded).
bool banded_ok(mtm_ctx* c, const ImageArgs& a) {
    c->banded_cls = -1;
    double best = 0.0;
    for (size_t i = 0; i < c->classes.size(); ++i) {
        double work = 0.0;
        if (class_bandable(c, a, c->classes[i], &work) && work > best) {
            best = work;
            c->banded_cls = (int)i;
        }
    }
    c->single_band_now = false;
    if (c->banded_cls < 0 && c->classes.size() == 1 && a.dtype == MTM_U8) {
        // Round 5: a call too small to be worth two score launches (1080p x 8 templates) still takes the banded path's
        // kernels, as ONE band - layout conversion inside the statistics launch, the candidate header cleared there: two
        // launches and a fill command fewer than the plain upload path
        double work = 0.0;
        if (class_bandable(c, a, c->classes[0], &work, true)) {
            c->banded_cls = 0;
            c->single_band_now = true;
        }
    }
    return c->banded_cls >= 0;
}

static bool class_bandable(const mtm_ctx* c, const ImageArgs& a, const SizeClass& sc, double* work, bool any_fill) {
    const bool u16 = a.dtype == MTM_U16;
    if (a.dtype == MTM_F32) {
        // Round 6: single-channel float32 images whose one size class runs the bf16 kernel - 33 MB cross PCIe at 4K, 0.7 ms
        // next to a 1.9 ms score launch since the one-product screen; two bands, the second under the first's score launch
        if (any_fill || c->upload_bands.size() < 2 || a.chans != 1 || c->classes.size() != 1 || sc.masked || sc.w > 1024 ||
            resolved_kernel(c, sc) != MTM_KERNEL_MFMA_F32)
            return false;
        const int oh = a.rows - sc.h + 1;
        if (!((size_t)a.rows * a.cols >= ((size_t)1 << 20) && oh >= 8 * kVsumBand)) return false;
        *work = (double)oh * (a.cols - sc.w + 1) * sc.h * sc.w * (double)sc.members.size();
        return true;
    }
    if (c->upload_bands.size() < 2 || (a.dtype != MTM_U8 && !u16) || a.chans != 1) return false;
    if (sc.masked || !c->fuse_stats || !sc.slabs.empty() || resolved_kernel(c, sc) != (u16 ? MTM_KERNEL_MFMA16 : MTM_KERNEL_MFMA))
        return false;
    if (!(sc.w <= 768 && (double)sc.w * sc.h * (u16 ? 65535.0 : 65025.0) < 4294967296.0)) return false;   // the fused statistics
    if (!any_fill && !((size_t)a.rows * a.cols >= ((size_t)1 << 20) && a.rows - sc.h + 1 >= 256)) return false;
    if (any_fill) return true;
    // A band's score launch must still fill the chip: two work-groups per CU are resident, and a launch of fewer than a
    // couple of such generations runs at the latency of its last one.  1080p x 8 templates is 512 work items in all -
    // banded 0.26 ms per call (two launches of 55 us for 62 us of work), in one piece 0.22 ms.
    const int oh = a.rows - sc.h + 1, ow = a.cols - sc.w + 1, n = (int)sc.members.size();
    const int rows_per_item = u16 ? kMfRows : sc.rm_R > 0 ? 8 * sc.rm_R : (sc.r2 ? sc.r2 * kMfRows : kMfRows);
    const int tg = u16 ? (n + 15) / 16 : sc.rm_R > 0 ? 1 : (n + (sc.r2 ? 16 : 32) - 1) / (sc.r2 ? 16 : 32);
    const long long items = (long long)((ow + kMfSeg - 1) / kMfSeg) * ((oh + rows_per_item - 1) / rows_per_item) * tg;
    const int cus = c->n_cus > 0 ? c->n_cus : 256;
    *work = (double)oh * ow * sc.h * sc.w * n;
    // MTM_BAND_MIN_FILL (default 1: the first band's launch is at least one full generation of resident work-groups; 0:
    // always band).  Measured at 4K (tools/probes/two_class_probe.py): 2 x 16 templates 48x64 (first band = 0.97
    // generations) 0.819 ms unbanded, 0.795 banded; 2 x 8 templates (0.49) 0.591 / 0.582; 1080p x 8 (0.125) 0.22 / 0.26.
    return (double)items * c->upload_bands[0] >= 0.97 * c->band_min_fill * (2.0 * cus);
}

// The score pass of a fused call with a banded upload.  copy_stream: per band the rows' copy, their layout
// conversion and the window statistics of the output rows that became computable; c->stream: the score kernel
// over the row blocks whose statistics exist, behind the band's event.  With a pageable source every copy call
// blocks the host while its rows are staged - the kernels queued before it run meanwhile.
// ... of a float32 image (round 6): two bands; the first ends where ~30 % of the output rows are complete, at a whole number of
// vsum bands (their column sums restart per band: the statistics planes are those of one launch) - the second band's 23 MB
// cross PCIe under the first band's score launch.  The rows go straight into the padded float32 plane.
static int run_score_banded_f32(mtm_ctx* c, const ImageArgs& a) {
    const SizeClass& sc = c->classes[(size_t)c->banded_cls];
    if (!c->hits_only_now) MTMC(ensure_maps(c));
    MTMC(ensure_copy_stream(c));
    mtm_ctx::ImageSlot& sl = c->slot[c->cur];
    SlotGeom g{};
    MTMC(prepare_slot(c, sl, a.rows, a.cols, 1, a.dtype, c->copy_stream, 1, &g));
    while ((int)c->band_ev.size() < 2) {
        hipEvent_t e;
        HIPC(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->band_ev.push_back(e);
    }
    const hipStream_t bs0 = c->copy_stream;
    if (c->f32_exact_now) {
        // (this call's method / mode sends the class to the float64 kernel after all - decided behind the banding decision:
        // the whole image in one piece, then the plain score pass)
        MTMC(upload_rows_f32c1(sl, g, a.px, a.stride, 0, a.rows, bs0));
        HIPC(hipEventRecord(c->band_ev[0], bs0));
        HIPC(hipStreamWaitEvent(c->stream, c->band_ev[0], 0));
        HIPC(hipEventRecord(c->ev[0], c->stream));
        return run_score_all(c);
    }
    const int h = sc.h, oh = a.rows - h + 1;
    const int nyb = (oh + kBfRows - 1) / kBfRows;
    const double frac = std::min(0.6, std::max(0.1, c->upload_bands[0] + 0.05));
    const int o_split = std::max(kVsumBand, ((int)(frac * oh) / kVsumBand) * kVsumBand);      // < oh: class_bandable asked for 8 bands
    const hipStream_t bs = c->copy_stream;
    StatPlanes st;
    int r_done = 0;
    for (int k = 0; k < 2; ++k) {
        const bool last = k == 1;
        const int r1 = last ? a.rows : o_split + h - 1;
        MTMC(upload_rows_f32c1(sl, g, a.px, a.stride, r_done, r1, bs));
        if (k == 0) HIPC(hipEventRecord(c->ev[0], c->stream));          // (see fm_begin)
        c->lay_r0 = r_done;
        c->lay_r1 = r1;
        c->stats_stream = bs;
        const int rc = launch_stats(c, sc, &st, last ? o_split / kStatBand4 : 0, last ? -1 : o_split / kStatBand4);
        c->stats_stream = nullptr;
        c->lay_r0 = c->lay_r1 = 0;
        MTMC(rc);
        r_done = r1;
        HIPC(hipEventRecord(c->band_ev[(size_t)k], bs));
        (void)hipStreamQuery(bs);
        HIPC(hipStreamWaitEvent(c->stream, c->band_ev[(size_t)k], 0));
        MTMC(launch_ncc(c, sc, sc.tlist_off, (int)sc.members.size(), st, -1, last ? o_split / kBfRows : 0, last ? nyb : o_split / kBfRows));
        (void)hipStreamQuery(c->stream);
    }
    // what run_score_all does behind a bf16 class's launch: the exact re-scoring of what the screen listed
    if (c->refine_now && !c->f32_exact_now) {
        if (c->refine_scan_now) {
            MTMC(launch_refine_scan(c, sc));
            MTMC(launch_refine(c, sc, st, true, true));
        } else {
            MTMC(launch_refine(c, sc, st, false, !c->hits_only_now));
        }
        if (c->ext_now) MTMC(launch_refine_extremum(c));
    }
    return MTM_OK;
}

int run_score_banded(mtm_ctx* c, const ImageArgs& a) {
    if (a.dtype == MTM_F32) return run_score_banded_f32(c, a);
    const SizeClass& sc = c->classes[(size_t)c->banded_cls];
    host_trace(c, 16);
    if (!c->hits_only_now) MTMC(ensure_maps(c));
    MTMC(ensure_copy_stream(c));
    host_trace(c, 17);
    mtm_ctx::ImageSlot& sl = c->slot[c->cur];
    SlotGeom g{};
    const bool u16 = a.dtype == MTM_U16;
    static const std::vector<double> kOneBand{1.0};
    const std::vector<double>& bands = c->single_band_now ? kOneBand : c->upload_bands;
    const int nb = (int)bands.size();
    // (Round 4 measured three other layouts of the same work against this one and round 5 removed their code: the first
    // band on the score stream itself, consecutive bands on two copy-side streams, score launches alternating between two
    // streams - all within +-2 %, docs/HISTORY.md.)
    MTMC(prepare_slot(c, sl, a.rows, a.cols, 1, a.dtype, c->copy_stream, 1, &g));
    host_trace(c, 13);
    while ((int)c->band_ev.size() < nb) {
        hipEvent_t e;
        HIPC(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->band_ev.push_back(e);
    }
    int k_prev = -1;                        // the band queued before this one (bands without rows are skipped)
    const int h = sc.h, oh = a.rows - h + 1;
    const int RB = u16 ? kMfRows : sc.rm_R > 0 ? 8 * sc.rm_R : (sc.r2 ? sc.r2 * kMfRows : kMfRows);   // output rows per score-kernel row block
    const int nyb = (oh + RB - 1) / RB, nsb = (oh + kStatBand4 - 1) / kStatBand4;
    int r_done = 0, sb_done = 0, yb_done = 0;
    // Round 5 (MTM_BAND_ALIGN, default 1): a band ends where its score launch is a whole number of generations of resident
    // work-groups (two per CU): the configured fractions are moved to the nearest such boundary, at least one generation
    // per launch.  (A part-filled last generation runs as long as a full one; on some boxes a first band of 0.25 of the
    // rows - 3.46 generations at 4K x 32 templates - cost 11 us of kernel time against 0.28 = 3.93, profiles/r05b.)
    double gen_blocks = 0.0;                // output row blocks per generation
    {
        const int ow = a.cols - sc.w + 1, n = (int)sc.members.size();
        const int tg = u16 ? (n + 15) / 16 : sc.rm_R > 0 ? 1 : (n + (sc.r2 ? 16 : 32) - 1) / (sc.r2 ? 16 : 32);
        const int per_block = ((ow + kMfSeg - 1) / kMfSeg) * tg;
        const int cus = c->n_cus > 0 ? c->n_cus : 256;
        gen_blocks = 2.0 * cus / (double)per_block;
    }
    int yb_target = 0;
    const hipStream_t bs = c->copy_stream;
    for (int k = 0; k < nb; ++k) {
        bool last = k == nb - 1;
        int r1 = last ? a.rows : std::min(a.rows, (int)(bands[(size_t)k] * a.rows) & ~7);
        if (!last && gen_blocks > 0.0) {
            const double want = bands[(size_t)k] * nyb - yb_target;                  // row blocks of this band's launch
            const int gens = std::max(1, (int)std::floor(want / gen_blocks + 0.5));
            yb_target = std::min(nyb, yb_target + std::max(1, (int)std::floor(gens * gen_blocks)));
            // the rows that complete those blocks' windows (and their statistics blocks of kStatBand4 rows)
            const int need = ((yb_target * RB + kStatBand4 - 1) / kStatBand4) * kStatBand4 + h - 1;
            r1 = std::min(a.rows, (need + 7) & ~7);
            if (yb_target >= nyb) r1 = a.rows;
        }
        if (r1 <= r_done) continue;
        last = last || r1 >= a.rows;            // (a band that reaches the end of the image is the last one: every block left)
        if (r_done == 0) host_trace(c, 14);
        // Round 5 (MTM_FUSE_LAYOUT, default 1): a band whose rows complete new statistics blocks gets ONE kernel for layout
        // conversion + window statistics instead of two launches with a kernel boundary between them (stats_u8_kernel's
        // StatLayout; row lengths that are multiples of 4, no float32 plane asked for)
        const int avail = r1 - h + 1;                            // output rows whose windows are complete
        const int sb1 = last ? nsb : std::max(sb_done, avail > 0 ? avail / kStatBand4 : 0);
        const bool fuse_lay = c->fuse_layout != 0 && !u16 && (a.cols % 4) == 0 && sb1 > sb_done;
        if (u16)
            MTMC(upload_rows_u16c1(sl, g, a.px, a.stride, r_done, r1, bs));
        else
            MTMC(upload_rows_u8c1(sl, g, a.px, a.stride, r_done, r1, bs, true, nullptr, nullptr, !fuse_lay));
        c->lay_r0 = fuse_lay ? r_done : 0;
        c->lay_r1 = fuse_lay ? r1 : 0;
        if (r_done == 0) {
            HIPC(hipEventRecord(c->ev[0], c->stream));           // (see fm_begin)
            if (c->classes.size() > 1) {                         // the lanes of the other classes start behind the call's set-up too
                if (!c->lane_fork) HIPC(hipEventCreateWithFlags(&c->lane_fork, hipEventDisableTiming));
                HIPC(hipEventRecord(c->lane_fork, c->stream));
            }
        }
        r_done = r1;
        host_trace(c, k == 0 ? 4 : 7);                           // the band's copy call returned
        StatPlanes st;
        c->stats_stream = bs;
        const int rc = launch_stats(c, sc, &st, sb_done, sb1);
        c->stats_stream = nullptr;
        c->lay_r0 = c->lay_r1 = 0;
        MTMC(rc);
        sb_done = sb1;
        HIPC(hipEventRecord(c->band_ev[(size_t)k], bs));
        (void)hipStreamQuery(bs);                                // submit now (the runtime batches commands)
        k_prev = k;
        if (k == 0) host_trace(c, 5);                            // layout conversion + statistics of band 0 submitted
        int yb1 = last ? nyb : (sb1 * kStatBand4) / RB;
        if (!last && gen_blocks > 0.0) yb1 = std::min(yb1, yb_target);      // (exactly the whole generations, not the rows' rounding on top)
        if (yb1 > yb_done) {
            if (c->zero_pending) {          // (no statistics launch took the clearing of the candidate header along)
                HIPC(hipMemsetAsync(c->cands.p, 0, 16, c->stream));
                c->zero_pending = false;
            }
            HIPC(hipStreamWaitEvent(c->stream, c->band_ev[(size_t)k], 0));
            const int rc2 = launch_ncc(c, sc, sc.tlist_off, (int)sc.members.size(), st, -1, yb_done, yb1);
            MTMC(rc2);
            (void)hipStreamQuery(c->stream);
            host_trace(c, k == 0 ? 6 : 8);                       // the band's score launch is submitted
            yb_done = yb1;
        }
    }
    // the other size classes, on the complete image: the first of them on a lane behind the last band's event - under
    // the banded class's last launch - the rest alternating as in run_score_all
    // (k_prev: the last band that was queued - trailing bands without rows of their own record no event)
    if (c->classes.size() > 1) MTMC(run_score_classes(c, c->banded_cls, c->band_ev[(size_t)(k_prev >= 0 ? k_prev : nb - 1)]));
    return MTM_OK;
}

}  // namespace mtmi
