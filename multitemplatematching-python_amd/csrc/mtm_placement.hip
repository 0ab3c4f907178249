// libmtm_hip.so - template sets: pixels -> size classes, device-resident sources and unit views, operand packs for every
// kernel family, per-template constants (place_templates).
#include <functional>
#include "mtm_ctx.h"

using namespace mtm;
using namespace mtmi;
#include "mtm_templates.hip.h"

namespace {

// Packs one uint8 template for ncc_dot4_kernel (layout documented in mtm_k_score.hip.h).
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
std::vector<SizeClass::Slab> slab_layout(const mtm_ctx* c, const SizeClass& sc) {
    std::vector<SizeClass::Slab> out;
    if (sc.masked || !sc.all_u8 || c->dtype != MTM_U8) return out;
    for (int m : sc.members)
        if (!c->templs[(size_t)m].on_device) return out;
    auto layout = [&](int cw) {
        std::vector<SizeClass::Slab> v;
        const int rows_max = 131071 / std::min(cw, ((sc.w + 63) / 64) * 64);
        const int nrb = (sc.h + rows_max - 1) / rows_max, rh = (sc.h + nrb - 1) / nrb;
        for (int ch = 0; ch < c->chans; ++ch)
            for (int r0 = 0; r0 < sc.h; r0 += rh)
                for (int c0 = 0; c0 < sc.w; c0 += cw)
                    v.push_back(SizeClass::Slab{r0, std::min(sc.h, r0 + rh), c0, std::min(sc.w, c0 + cw), ch, 0, 0});
        if (v.size() > 24) v.clear();            // absurdly large: the VALU kernel takes it
        return v;
    };
    if (sc.members.size() > 16) return layout(256);
    // <= 16 templates (row-multiplexed raw launches): column blocks of 128 or 64 taps.  A slab launch is few, long work
    // items - 91 of 128 output rows x 256 columns for 2048^2 x 414x400 - whose waves cannot be split: with 128-tap slabs
    // that shape is 1.24 waves' worth of work per SIMD and the launches take as long as the SIMDs that got two.  Half as
    // wide slabs are twice as many waves of half the length (and more raw planes for slab_combine_kernel to add): the
    // width whose greedy schedule over the CUs ends first is taken.
    std::vector<SizeClass::Slab> best = layout(128);
    if (!c->have_image || c->rows < sc.h || c->cols < sc.w) return best;
    const int oh = c->rows - sc.h + 1, ow = c->cols - sc.w + 1;
    const int cus = c->n_cus > 0 ? c->n_cus : 256;
    auto makespan = [&](const std::vector<SizeClass::Slab>& v) {
        if (v.empty()) return 1e300;
        int hs = 0, ws = 0;
        for (const auto& sl : v) {
            hs = std::max(hs, sl.r1 - sl.r0);
            ws = std::max(ws, sl.c1 - sl.c0);
        }
        int nt = 1;
        while (nt < (int)sc.members.size()) nt <<= 1;
        const size_t lds_pitch = (size_t)(16 + 4 * ((ws + 63) / 64) + 1) * 16;
        auto tile_bytes = [&](int R) { return (size_t)(std::min(hs + 2 * R - 1, kMfChunkH) + 6 * R) * lds_pitch; };
        while (nt < 16 && tile_bytes(16 / nt) > 72 * 1024) nt <<= 1;
        const int R = 16 / nt;
        const long long items = (long long)((ow + kMfSeg - 1) / kMfSeg) * ((oh + 8 * R - 1) / (8 * R));
        std::vector<double> cost;                         // one entry per work-group, longest first
        for (const auto& sl : v)
            cost.insert(cost.end(), (size_t)items, (double)((sl.c1 - sl.c0 + 63) / 64) * (sl.r1 - sl.r0 + 2 * R - 1));
        std::sort(cost.begin(), cost.end(), std::greater<double>());
        std::vector<double> load((size_t)cus, 0.0);       // min-heap of CU loads
        auto cmp = std::greater<double>();
        for (double x : cost) {
            std::pop_heap(load.begin(), load.end(), cmp);
            load.back() += x;
            std::push_heap(load.begin(), load.end(), cmp);
        }
        // every slab's raw plane is written once and read once by the combine pass: ~5 us per 4 B x 2 x oh x ow at
        // 2048^2, against ~0.2 us per (block x template row) step of a work-group - in units of steps
        return *std::max_element(load.begin(), load.end()) + 25.0 * (double)v.size() * ((double)oh * ow / 2.7e6);
    };
    std::vector<SizeClass::Slab> narrow = layout(64);
    if (makespan(narrow) < 0.9 * makespan(best)) best.swap(narrow);
    return best;
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
                out[(((size_t)sp * nb + b) * 64 + (16 * q + i)) * 16 + byte] = v;           // int8 0 / 1: the mask is not biased
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

// float32 image + float32 templates, no mask: bfloat16 pieces on the bf16 matrix cores.  The normalised methods' outputs
// are O(1) and stay within ~1e-5 of the float64 result; the raw sums (TM_SQDIFF, TM_CCORR, TM_CCOEFF) can cancel to values
// far smaller than the sums they are built from (an exact copy: SQDIFF = 0), so no RELATIVE margin holds for them - their
// lists are decided by an absolute per-output bound and exact re-scoring (mtm_api.hip), their maps by the float64 kernel.
bool bf16_class_ok(const mtm_ctx* c, const SizeClass& sc) {
    // All six methods (round 4).  The raw-sum methods only take this kernel for the refined global extremum
    // (N_object == 1; mtm_api.hip decides per call) - everything else they do runs the float64 kernel, for which a
    // bf16 class carries its float64 weights anyway.
    // (1-D and 1x1 score maps go through scipy's find_peaks on the host, which has no refinement: float64 kernel)
    return c->f32_mfma && c->dtype == MTM_F32 && sc.all_f32 && !sc.masked && sc.w <= kBfMaxW &&
           c->rows > sc.h && c->cols > sc.w;
}
// float32 class with masks (float weights or binary), one channel, TM_SQDIFF / TM_CCORR_NORMED - the methods the reference
// lets masks through for (MTM/__init__.py:78): two raw bf16 correlations as a screen + exact re-scoring (mtm_maskf32.hip.h)
bool masked_bf16_class_ok(const mtm_ctx* c, const SizeClass& sc) {
    return f32_refined(c) && c->dtype == MTM_F32 && sc.all_f32 && sc.masked && c->chans == 1 && sc.w <= kBfMaxW &&
           c->rows > sc.h && c->cols > sc.w && (c->method == MTM_TM_SQDIFF || c->method == MTM_TM_CCORR_NORMED);
}
inline int bf16_nkb(int w) { return (w + 31) / 32; }
long long bf16_group_bytes(int h, int w, int chans) { return (long long)chans * h * bf16_nkb(w) * 1024; }

// A packs of a float32 class for ncc_bf16_kernel: [piece 0 | piece 1][group of 16][ch][dy][32-tap block][lane = 16 q + i]
// [8 bf16]: lane (i, q) holds taps 32 kb + 8 q .. + 7 of template i, centred by its channel mean and split
// v = v0 + v1 (bfloat16, round to nearest even).  centre[] receives the means (TemplDev::centre).
// `which`: 0 = the template's pixels; 1 = U = T M^2, 2 = V = M^2 (masked classes, single channel: mtm_maskf32.hip.h) - then
// td_host is the U / V copy of the table: centre and centred_sum2 of the packed values go there
void pack_class_bf16(const mtm_ctx* c, const SizeClass& sc, uint8_t* out, std::vector<TemplDev>& td_host, int which = 0) {
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
        auto val = [&](size_t k) {
            if (which == 0) return t.px[k];
            const double m2 = t.mask[k] * t.mask[k];
            return which == 1 ? t.px[k] * m2 : m2;
        };
        if (which) d.centred_sum2 = 0.0;
        for (int ch = 0; ch < chans; ++ch) {
            double mean = 0.0;
            for (size_t k = 0; k < plane; ++k) mean += val(ch * plane + k);
            mean /= (double)plane;
            d.centre[ch] = mean;
            if (which)
                for (size_t k = 0; k < plane; ++k) d.centred_sum2 += (val(ch * plane + k) - mean) * (val(ch * plane + k) - mean);
            for (int dy = 0; dy < h; ++dy)
                for (int dx = 0; dx < w; ++dx) {
                    const float v = (float)(val(ch * plane + (size_t)dy * w + dx) - mean);
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

}  // namespace

namespace mtmi {

int place_templates(mtm_ctx* c) {
    if (!c->have_image || !c->have_templ) {
        set_error("set the image and the templates first");
        return MTM_E_STATE;
    }
    if (c->placed) return MTM_OK;
    if (c->place_pending) {             // the previous placement's tables are released below: their copies first
        HIPC(hipStreamSynchronize(c->stream));
        c->place_pending = false;
    }
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
        // packed K: uint8 classes (one channel, masked or not; RGB) on the plain or row-multiplexed tiling whose width
        // leaves part of the last 64-tap block empty.  Replaces the two-row variant where both apply (that one saves template loads,
        // this one whole MFMA steps).
        sc.kp_nseg = 0;
        {
            const int nseg = (sc.w + 15) / 16;
            const bool normed = c->method == MTM_TM_SQDIFF_NORMED || c->method == MTM_TM_CCORR_NORMED ||
                                c->method == MTM_TM_CCOEFF_NORMED;       // the instantiated variants (ncc_mfma_kernel<.., KP>)
            if (class_kernel[k] == MTM_KERNEL_MFMA && sc.slabs.empty() && nseg % 4 != 0 && normed &&
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
        d.centred_sum2 = t.st.centred_sum2;
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
    bool any_mbf = false;
    for (size_t k = 0; k < classes.size(); ++k) {
        SizeClass& sc = classes[k];
        sc.mask_bf16 = class_kernel[k] == MTM_KERNEL_AUTO && masked_bf16_class_ok(c, sc);      // (AUTO: the float64 kernel)
        sc.mbf_off_u = sc.mbf_off_v = -1;
        if (!sc.mask_bf16) continue;
        any_mbf = true;
        sc.mbf_group_bytes = bf16_group_bytes(sc.h, sc.w, 1);
        const size_t set_bytes = (size_t)(2 * sc.mbf_group_bytes * mfma_groups_alloc((int)sc.members.size()));
        sc.mbf_off_u = (long long)a_off;
        a_off += set_bytes;
        sc.mbf_off_v = (long long)a_off;
        a_off += set_bytes;
    }
    size_t ts_off = 0;
    for (size_t k = 0; k < classes.size(); ++k) {
        SizeClass& sc = classes[k];
        sc.tsum_off = -1;
        if (class_kernel[k] != MTM_KERNEL_MFMA16) continue;
        {   // packed K for the two byte-plane passes (widths that are not multiples of 64), as for uint8 classes
            const int nseg = (sc.w + 15) / 16;
            sc.kp_nseg = (nseg % 4 != 0) ? nseg : 0;
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
    any_host_pack = any_host_pack || any_mbf;
    std::vector<uint8_t> apacks(any_host_pack ? a_off : 0);
    // masked float32 classes: the U = T M^2 / V = M^2 packs and their copies of the template table (centre, centred
    // energy, and where the approximate c1 / c2 maps go: the real maps' layout, twice, in a scratch arena of their own)
    std::vector<TemplDev> td_u, td_v;
    if (any_mbf) {
        td_u = td_host;
        td_v = td_host;
        for (size_t k = 0; k < classes.size(); ++k) {
            if (!classes[k].mask_bf16) continue;
            SizeClass u = classes[k];
            u.group_bytes = classes[k].mbf_group_bytes;
            pack_class_bf16(c, u, apacks.data() + classes[k].mbf_off_u, td_u, 1);
            pack_class_bf16(c, u, apacks.data() + classes[k].mbf_off_v, td_v, 2);
            for (int m : classes[k].members) {
                td_u[(size_t)m].all_ones = td_v[(size_t)m].all_ones = 0;
                td_v[(size_t)m].map_off += (long long)map_off;       // second half of the scratch arena
            }
        }
    }
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
    // From here on copies from this function's own vectors are in flight: an error return below drains the stream before
    // they go out of scope (the normal path hands the long-lived ones to the context and says so in place_pending).
    struct DrainGuard {
        hipStream_t s;
        bool armed = true;
        ~DrainGuard() {
            if (armed) (void)hipStreamSynchronize(s);
        }
    } drain{c->stream};
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
    if (any_mbf) {
        MTMC(c->td_u.ensure(sizeof(TemplDev) * n));
        MTMC(c->td_v.ensure(sizeof(TemplDev) * n));
        HIPC(hipMemcpyAsync(c->td_u.p, td_u.data(), sizeof(TemplDev) * n, hipMemcpyHostToDevice, c->stream));
        HIPC(hipMemcpyAsync(c->td_v.p, td_v.data(), sizeof(TemplDev) * n, hipMemcpyHostToDevice, c->stream));
    }
    MTMC(c->tsum.ensure(sizeof(double) * std::max<size_t>(2, ts_off)));
    if (ts_off) HIPC(hipMemcpyAsync(c->tsum.p, tsums.data(), sizeof(double) * ts_off, hipMemcpyHostToDevice, c->stream));
    // device-side packing: gathers the A operands straight from the unit views (the template list is in place)
    for (size_t k = 0; k < classes.size(); ++k)
        if (dev_pack[k]) MTMC(pack_class_on_device(c, classes[k]));
    // host staging vectors go out of scope; the tables that stay in the context (td_host, tlist_host) need no wait
    // (set_templates_device)
    const bool local_sources = any_host_pack || any_mbf || w_off || p_off || ts_off || units_all.size() > c->usrc_units;
    if (local_sources) HIPC(hipStreamSynchronize(c->stream));
    c->place_pending = !local_sources;
    drain.armed = false;
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

}  // namespace mtmi

extern "C" {

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
    // The host image of the arena prefix (the bases), the unit table and - place_templates - the constants tables are
    // copied from context-owned vectors that live until the next template set: no host wait for those copies here, the
    // template work queues ahead of the call's image upload and runs under it.  Before the vectors are rewritten the
    // stream is drained (idle by then in any ordinary sequence of calls).
    if (c->stage_pending) {
        HIPC(hipStreamSynchronize(c->stream));
        c->stage_pending = false;
    }
    std::vector<uint8_t>& stage = c->tstage;
    stage.clear();
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
                if (t.chans == 1 && !mp) {       // the common case: a row is a run of bytes
                    std::memcpy(o, rp, (size_t)t.cols);
                    u8_run_sums(rp, (size_t)t.cols, &s, &sq);
                    continue;
                }
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
    // (set BEFORE the copy is queued: an error return further down must not leave a copy from `tstage` in flight unrecorded)
    c->stage_pending = true;
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
    c->usrc_units = units.size();
    c->usrc_host.swap(units);                   // (the previous table - now `units` - was drained above)
    c->stage_pending = true;
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

}  // extern "C"
