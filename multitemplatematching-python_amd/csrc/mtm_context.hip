// libmtm_hip.so - the context: creation, options, the device copies of an image (upload, layout conversion).
// gfx950 (MI355X / CDNA4) only; built by multitemplatematching-python_amd/build.py.
#include <cstdio>
#include "mtm_ctx.h"

using namespace mtm;
using namespace mtmi;
#include "mtm_k_image.hip.h"

namespace mtmi {

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
// `copy_done` (optional): recorded right behind the copy; `before_kernels` (optional): the conversion kernels wait for it
// (banded uploads on two streams: the next band's copy starts behind this one's copy, not behind its kernels).
int upload_rows_u8c1(mtm_ctx::ImageSlot& sl, const SlotGeom& g, const void* src, int64_t src_stride, int r0, int r1,
                     hipStream_t stream, bool skip_f32, hipEvent_t copy_done, hipEvent_t before_kernels, bool convert) {
    const int cols = g.cols, nrows = r1 - r0;
    if (nrows <= 0) return MTM_OK;
    uint8_t* raw = sl.raw.as<uint8_t>() + (size_t)r0 * cols;
    HIPC(hipMemcpy2DAsync(raw, (size_t)cols, (const uint8_t*)src + (size_t)r0 * src_stride, (size_t)src_stride, (size_t)cols,
                          nrows, hipMemcpyHostToDevice, stream));
    if (copy_done) HIPC(hipEventRecord(copy_done, stream));
    if (before_kernels) HIPC(hipStreamWaitEvent(stream, before_kernels, 0));
    if (!convert) {             // (the band's statistics launch converts the rows: stats_u8_kernel's StatLayout)
        sl.f32_valid = false;
        return MTM_OK;
    }
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

// Rows r0 .. r1 - 1 of a single-channel float32 image (round 6: banded float32 uploads): straight into the padded plane -
// what planarize_f32_kernel makes of a one-channel image is a copy.
int upload_rows_f32c1(mtm_ctx::ImageSlot& sl, const SlotGeom& g, const void* src, int64_t src_stride, int r0, int r1,
                      hipStream_t stream) {
    const int nrows = r1 - r0;
    if (nrows <= 0) return MTM_OK;
    float* dst = sl.f32.as<float>() + (size_t)r0 * g.pitch;
    HIPC(hipMemcpy2DAsync(dst, sizeof(float) * (size_t)g.pitch, (const uint8_t*)src + (size_t)r0 * src_stride, (size_t)src_stride,
                          sizeof(float) * (size_t)g.cols, nrows, hipMemcpyHostToDevice, stream));
    sl.f32_valid = true;
    return MTM_OK;
}

// The same for rows r0 .. r1 - 1 of a single-channel uint16 image: the three byte planes and the float32 plane.
int upload_rows_u16c1(mtm_ctx::ImageSlot& sl, const SlotGeom& g, const void* src, int64_t src_stride, int r0, int r1,
                      hipStream_t stream, hipEvent_t copy_done, hipEvent_t before_kernels) {
    const int cols = g.cols, nrows = r1 - r0;
    if (nrows <= 0) return MTM_OK;
    uint16_t* raw = sl.raw.as<uint16_t>() + (size_t)r0 * cols;
    HIPC(hipMemcpy2DAsync(raw, (size_t)cols * 2, (const uint8_t*)src + (size_t)r0 * src_stride, (size_t)src_stride,
                          (size_t)cols * 2, nrows, hipMemcpyHostToDevice, stream));
    if (copy_done) HIPC(hipEventRecord(copy_done, stream));
    if (before_kernels) HIPC(hipStreamWaitEvent(stream, before_kernels, 0));
    const size_t o = (size_t)r0 * g.pitch;
    sl.f32_valid = true;
    hipLaunchKernelGGL(planarize_u16_kernel, dim3((cols + 255) / 256, nrows), dim3(256), 0, stream, raw, cols, 1, 1, nrows, cols,
                       sl.u8.as<uint8_t>() + o, sl.u8b.as<uint8_t>() + o, sl.u8b.as<uint8_t>() + g.u8_bytes + o, g.pitch,
                       sl.f32.as<float>() + o, g.pitch, (long long)g.pitch * g.rows_alloc);
    HIPC(hipGetLastError());
    return MTM_OK;
}

// Upload one image into `sl` and build its planar padded planes on `stream`.  `src` has tightly
// packed rows when `src_stride` == cols * chans * elem size or any larger stride.
int upload_image(mtm_ctx* c, mtm_ctx::ImageSlot& sl, const void* src, int64_t src_stride, int src_rows, int src_cols,
                 int chans, int dtype, hipStream_t stream, int factor) {
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
    c->f32_sq_valid = false;
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

// The producer side of the upload pipelines (copies, layout conversion, window statistics of a band): its short
// kernels must not queue behind the score kernel's work-groups for a free CU, hence the highest stream priority.
int ensure_copy_stream(mtm_ctx* c) {
    if (c->copy_stream) return MTM_OK;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    HIPC(hipStreamCreateWithPriority(&c->copy_stream, hipStreamNonBlocking, hi));
    return MTM_OK;
}

}  // namespace mtmi

extern "C" {

int mtm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void* mtm_host_alloc(size_t bytes) {
    void* p = nullptr;
    // portable: a device group stages ONE image there and every device of the process copies from it (mtm_group.cpp)
    const hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable | hipHostMallocMapped);
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
    if (const char* v = std::getenv("MTM_CAND_PINNED")) c->cand_pinned = std::atoi(v);
    if (const char* v = std::getenv("MTM_FUSE_LAYOUT")) c->fuse_layout = std::atoi(v);
    if (const char* v = std::getenv("MTM_SEG_SKIP")) c->seg_skip = std::atoi(v);
    if (const char* v = std::getenv("MTM_BAND_MIN_FILL")) c->band_min_fill = std::atof(v);
    if (const char* v = std::getenv("MTM_TEMPL_ON_DEVICE")) c->templ_on_device = std::atoi(v);
    if (const char* v = std::getenv("MTM_ROW_MUX")) c->row_mux = std::atoi(v);
    if (const char* v = std::getenv("MTM_FUSE_STATS")) c->fuse_stats = std::atoi(v);
    if (const char* v = std::getenv("MTM_MFMA_R2")) c->mfma_r2 = std::atoi(v) != 0;
    if (const char* v = std::getenv("MTM_F32_MFMA")) c->f32_mfma = std::atoi(v);
    if (const char* v = std::getenv("MTM_MASKSQ_FUSED")) c->masksq_fused = std::atoi(v);
    if (const char* v = std::getenv("MTM_SPARSE_MAPS")) c->sparse_maps = std::atoi(v);
    if (const char* v = std::getenv("MTM_NMS_DEVICE_MIN")) c->nms_device_min = std::atoll(v) < 0 ? (1ll << 60) : std::max(1, std::atoi(v));   // < 0: never
    if (const char* v = std::getenv("MTM_SCREEN_L1")) c->screen_l1 = std::atoi(v);
    if (const char* v = std::getenv("MTM_HOST_TRACE")) c->host_trace = std::atoi(v) != 0;
    if (const char* v = std::getenv("MTM_CLASS_LANES")) c->class_lanes = std::max(1, std::min(8, std::atoi(v)));
    if (const char* v = std::getenv("MTM_COMM_TIMEOUT_S")) c->comm_timeout_s = std::atof(v);
    if (const char* v = std::getenv("MTM_HITS_ONLY")) c->hits_only = std::atoi(v);
    if (const char* v = std::getenv("MTM_EXACT_DIV")) c->exact_div = std::atoi(v);
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c->device) == hipSuccess) c->n_cus = prop.multiProcessorCount;
    }
    // Round 5: the copy-side stream exists from the start.  The runtime maps a process's streams onto four hardware queues
    // in creation order; created lazily by the first banded call it came AFTER the streams RCCL makes for a communicator
    // (mtm_group_comm_init) and shared a hardware queue with them - the band's statistics launch then started 67 us
    // after its copy ended instead of 8, every call (profiles/r05c: group + RCCL 1.04 ms against 0.87 with the host merge).
    if (ensure_copy_stream(c) != MTM_OK) c->copy_stream = nullptr;
    // (the first side lane of multi-class calls likewise: third in the rotation, a hardware queue of its own)
    if (c->class_lanes > 1) (void)ensure_lanes(c, 1);
    *out = c;
    return MTM_OK;
}

void mtm_ctx_destroy(mtm_ctx* c) {
    if (!c) return;
    if (c->host_trace) {
        static const char* kPhase[24] = {"entry", "args checked", "templates placed", "call set up", "band 0 copy queued",
                                         "band 0 layout + statistics queued", "band 0 score queued", "last band copy queued",
                                         "last band score queued", "score pass queued", "stream synchronised", "hits verified",
                                         "hits sorted", "(banded: image slot prepared)", "(banded: before the first copy call)", "return",
                                         "(banded: score pass entered)", "(banded: copy stream ready)", "(fm_begin: decisions taken)", "", "", "", "", ""};
        for (int k = 0; k < 24; ++k)
            if (c->trace_n[k] > 0)
                std::fprintf(stderr, "[mtm host trace] %-36s %9.1f us after entry (mean of %lld)\n", kPhase[k],
                             c->trace_acc[k] / (double)c->trace_n[k], c->trace_n[k]);
    }
    (void)hipSetDevice(c->device);
    mtm_comm_destroy(c);
    (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    for (auto& sl : c->slot)
        for (DevBuf* b : {&sl.raw, &sl.u8, &sl.u8b, &sl.f32}) b->release();
    for (DevBuf* b : {&c->tsrc, &c->usrc_dev, &c->tsums_dev, &c->tgather, &c->slab_raw, &c->seg_flags, &c->hits_t, &c->nms_buf, &c->td_u, &c->td_v,
                      &c->mbf_maps, &c->f32_sq, &c->mbf_stats, &c->mbf_mu, &c->mbf_list, &c->mbf_best}) b->release();
    for (DevBuf* b : {&c->td, &c->tlist, &c->weights, &c->packs, &c->apacks, &c->maps, &c->hs1, &c->hs2, &c->stats, &c->hits,
                      &c->counters, &c->sched, &c->cands, &c->mask_td, &c->chash, &c->raw16, &c->stats_hi, &c->tsum, &c->stats_rsq, &c->stats_blk, &c->sq_planes, &c->comm_send,
                      &c->comm_recv})
        b->release();
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->pin_small) (void)hipHostFree(c->pin_small);
    if (c->comm_pin) (void)hipHostFree(c->comm_pin);
    if (c->next_ready) (void)hipEventDestroy(c->next_ready);
    for (hipEvent_t e : c->band_ev) (void)hipEventDestroy(e);
    if (c->lane_fork) (void)hipEventDestroy(c->lane_fork);
    if (c->f32_built) (void)hipEventDestroy(c->f32_built);
    for (auto& L : c->lanes) {
        if (L.stream) {
            (void)hipStreamSynchronize(L.stream);
            (void)hipStreamDestroy(L.stream);
        }
        if (L.done) (void)hipEventDestroy(L.done);
        if (L.slab_fork) (void)hipEventDestroy(L.slab_fork);
        for (hipEvent_t e : L.slab_done) (void)hipEventDestroy(e);
        for (hipStream_t s2 : L.slab_streams) {
            (void)hipStreamSynchronize(s2);
            (void)hipStreamDestroy(s2);
        }
        for (DevBuf* b : {&L.stats, &L.stats_rsq, &L.stats_blk, &L.hs1, &L.hs2, &L.raw16, &L.slab_raw, &L.stats_hi, &L.mask_td, &L.sched})
            b->release();
    }
    if (c->slab_fork) (void)hipEventDestroy(c->slab_fork);
    for (hipEvent_t e : c->slab_done) (void)hipEventDestroy(e);
    for (hipStream_t s2 : c->slab_streams) {
        (void)hipStreamSynchronize(s2);
        (void)hipStreamDestroy(s2);
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (auto* evs : {&c->ncc_ev, &c->sq_ev})
        for (auto& p : *evs) {
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
            c->exact_div = value < 0 ? 0 : value > 2 ? 2 : (int)value;
            return MTM_OK;
        case MTM_OPT_HITS_ONLY:
            c->hits_only = value ? 1 : 0;
            c->fuse_backoff = 0;
            c->backoff_len = 16;
            c->np1_backoff = 0;
            c->np1_backoff_len = 16;
            return MTM_OK;
        case MTM_OPT_F32_MFMA:
            if (value < 0 || value > 4) break;
            if ((c->f32_mfma != 0) != (value != 0)) c->placed = false;     // the packs follow the kernel
            c->f32_mfma = (int)value;
            c->np1_backoff = 0;
            c->np1_backoff_len = 16;
            return MTM_OK;
        case MTM_OPT_DOT4_VARIANT:
            if (!dot_variant_ok(value)) break;
            c->dot_variant = (int)value;
            return MTM_OK;
        default: break;
    }
    set_error("mtm_set_option: bad option or value");
    return MTM_E_INVALID;
}

int mtm_get_option(mtm_ctx* c, int option, int64_t* value) {
    if (!c || !value) return MTM_E_INVALID;
    switch (option) {
        case MTM_OPT_KERNEL: *value = c->opt_kernel; return MTM_OK;
        case MTM_OPT_PEAK_BORDER: *value = c->opt_border; return MTM_OK;
        case MTM_OPT_HIT_CAPACITY: *value = c->hit_cap; return MTM_OK;
        case MTM_OPT_EXACT_DIV: *value = c->exact_div; return MTM_OK;
        case MTM_OPT_HITS_ONLY: *value = c->hits_only; return MTM_OK;
        case MTM_OPT_F32_MFMA: *value = c->f32_mfma; return MTM_OK;
        case MTM_OPT_DOT4_VARIANT: *value = c->dot_variant; return MTM_OK;
        default: break;
    }
    set_error("mtm_get_option: bad option");
    return MTM_E_INVALID;
}

// ---- test support: memory a kernel must not depend on, filled with a pattern (mtm_debug_poison) -------------------------
// Per-lane private memory: 160 dwords through a volatile pointer (a real scratch array, larger than any product kernel's
// spill area), held while the wave sleeps so that the launch occupies every wave slot of the chip at once - scratch is
// handed out per wave SLOT, a kernel that reloads a slot it has not stored sees what the previous tenant left there.
__global__ __launch_bounds__(256) void poison_scratch_kernel(uint32_t pat, int sleeps, uint32_t* sink) {
    uint32_t buf[160];
    volatile uint32_t* vb = buf;
    for (int i = 0; i < 160; ++i) vb[i] = pat;
    for (int i = 0; i < sleeps; ++i) __builtin_amdgcn_s_sleep(127);
    uint32_t acc = 0;
    for (int i = 0; i < 160; ++i) acc += vb[(i * 7 + (int)threadIdx.x) % 160] != pat;
    if (acc) atomicAdd(sink, acc);                  // (never: keeps the array alive)
}
// LDS: 40 KiB per work-group, four of them fill a CU's 160 KiB while they sleep side by side
constexpr int kPoisonLdsBytes = 40 * 1024;
__global__ __launch_bounds__(256) void poison_lds_kernel(uint32_t pat, int sleeps, uint32_t* sink) {
    __shared__ uint32_t lds[kPoisonLdsBytes / 4];
    for (int i = threadIdx.x; i < kPoisonLdsBytes / 4; i += 256) lds[i] = pat;
    __syncthreads();
    for (int i = 0; i < sleeps; ++i) __builtin_amdgcn_s_sleep(127);
    if (lds[(threadIdx.x * 13) % (kPoisonLdsBytes / 4)] != pat) atomicAdd(sink, 1u);
}

int mtm_debug_poison(mtm_ctx* c, int pattern_byte, int what) {
    if (!c) return MTM_E_INVALID;
    MTM_NOT_IN_FLIGHT(c, "mtm_debug_poison");
    HIPC(hipSetDevice(c->device));
    const uint32_t b = (uint32_t)(pattern_byte & 0xFF), pat = b | (b << 8) | (b << 16) | (b << 24);
    if (c->n_cus == 0) {
        hipDeviceProp_t prop;
        HIPC(hipGetDeviceProperties(&prop, c->device));
        c->n_cus = prop.multiProcessorCount;
    }
    MTMC(c->sched.ensure(sizeof(unsigned int) * (1 + 4096 + 8 * 32)));
    uint32_t* sink = c->sched.as<uint32_t>() + 4096;        // (a word no launch reads before clearing the block)
    // ~8128 cycles per s_sleep(127): a few tens of microseconds per work-group, every slot taken several times over
    if (what & MTM_POISON_SCRATCH)
        hipLaunchKernelGGL(poison_scratch_kernel, dim3(c->n_cus * 8 * 3), dim3(256), 0, c->stream, pat, 4, sink);
    if (what & MTM_POISON_LDS)
        hipLaunchKernelGGL(poison_lds_kernel, dim3(c->n_cus * 4 * 3), dim3(256), 0, c->stream, pat, 4, sink);
    if (what & MTM_POISON_ARENAS) {
        // buffers a call writes before it reads them: window statistics (planes, reciprocals, block ranges, row sums),
        // raw partial maps, the score maps, the byte planes of I^2 - of the context and of every class lane.  Buffers with
        // an invariant kept between calls (image padding, candidate header, hash tables, flags, packs) are left alone.
        auto fill = [&](mtm_ctx::DevBuf& d) -> int {
            if (d.p && d.cap) HIPC(hipMemsetAsync(d.p, (int)b, d.cap, c->stream));
            return MTM_OK;
        };
        for (mtm_ctx::DevBuf* d : {&c->stats, &c->stats_rsq, &c->stats_blk, &c->hs1, &c->hs2, &c->raw16, &c->slab_raw,
                                   &c->stats_hi, &c->maps, &c->sq_planes, &c->mbf_maps, &c->f32_sq, &c->mbf_stats, &c->mbf_mu})
            MTMC(fill(*d));
        for (auto& ln : c->lanes)
            for (mtm_ctx::DevBuf* d : {&ln.stats, &ln.stats_rsq, &ln.stats_blk, &ln.hs1, &ln.hs2, &ln.raw16, &ln.slab_raw, &ln.stats_hi})
                MTMC(fill(*d));
        c->maps_valid = false;
        c->sq_valid = false;
        c->f32_sq_valid = false;
    }
    HIPC(hipStreamSynchronize(c->stream));
    HIPC(hipGetLastError());
    return MTM_OK;
}

// ---- test support: quotient_as_float against the IEEE division it replaces (mtm_debug_quotient_check) -------------------
// Operands shaped like the epilogue's: sq = sqrt of an integer-valued window energy, templ_norm = sqrt of one, num an
// integer-valued or fractional numerator with |num| <~ tt.  Even cases are plain random draws; odd cases are adversarial -
// the numerator is chosen so that the quotient lands within a few ulp(double) of a float32 rounding boundary (the middle
// between two neighbouring floats, or - every fourth of those - between two float32 denormals' neighbours of a tiny quotient), the
// place where a quotient that is a few ulp off rounds to the other float.  out[0] cases, out[1] results that differ from
// (float)(num / tt) in any bit, out[2] cases that took the division, out[3] the largest distance in ulp(double) between
// num * rr and num / tt seen where both are normal.
__device__ __forceinline__ uint64_t dq_mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void quotient_check_kernel(uint64_t n_per_thread, uint64_t seed, unsigned long long* out) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bad = 0, div = 0, far = 0;
    for (uint64_t k = 0; k < n_per_thread; ++k) {
        const uint64_t idx = gid * n_per_thread + k;
        const uint64_t r0 = dq_mix(seed ^ (idx * 4u)), r1 = dq_mix(seed ^ (idx * 4u + 1u)), r2 = dq_mix(seed ^ (idx * 4u + 2u)),
                       r3 = dq_mix(seed ^ (idx * 4u + 3u));
        // window energy and template energy: integers of 1 .. 44 bits (what 4 channels of 16-bit pixels under 2^14 taps reach)
        const double e_w = (double)(1ull + (r0 >> (20 + (r3 & 31)))), e_t = (double)(1ull + (r1 >> (20 + ((r3 >> 5) & 31))));
        const double sq = sqrt(e_w), tn = sqrt(e_t);
        const double tt = sq * tn;
        const double rr = (1.0 / sq) * (1.0 / tn);
        double num;
        if ((idx & 1u) == 0u) {
            const double u = (double)(int64_t)(r2 >> 11) * 0x1p-52 - 1.0;                    // [-1, 1)
            num = u * tt * 1.2;
            if (r3 & (1ull << 40)) num = rint(num);                                            // integer-valued numerators
        } else {
            // a float32 rounding boundary: a random float in [2^-20, 1) and the double half way to its successor
            uint32_t fb = 0x35800000u + (uint32_t)(r2 % (0x3f800000u - 0x35800000u));
            if ((idx & 7u) == 7u) fb = 0x00000001u + (uint32_t)(r2 % 0x02000000u);          // denormal / tiny floats
            const double lo_f = (double)__uint_as_float(fb), hi_f = (double)__uint_as_float(fb + 1u);
            const double mid = 0.5 * (lo_f + hi_f);
            num = mid * tt;
            // a few ulp either side of it (the product above is already rounded: the quotient straddles the boundary)
            const int step = (int)((r3 >> 44) % 9u) - 4;
            num = __longlong_as_double(__double_as_longlong(num) + (long long)step);
            if (r3 & (1ull << 41)) num = -num;
        }
        const double q0 = num * rr;
        const double qr = num / tt;
        const float ref = (float)qr;
        const float got = quotient_as_float(num, tt, rr);
        bad += __float_as_uint(ref) != __float_as_uint(got);
        // the epilogues' instantiation (no test for tiny quotients) on everything but the denormal-range cases
        if ((idx & 7u) != 7u) bad += __float_as_uint(ref) != __float_as_uint(quotient_as_float<false>(num, tt, rr));
        div += quotient_needs_division(q0);
        if (fabs(qr) > 0x1p-1000 && fabs(qr) < 0x1p1000) {
            const long long d = __double_as_longlong(fabs(q0)) - __double_as_longlong(fabs(qr));
            const unsigned long long ad = (unsigned long long)(d < 0 ? -d : d);
            far = ad > far ? ad : far;
        }
    }
    atomicAdd(&out[1], bad);
    atomicAdd(&out[2], div);
    atomicMax(&out[3], far);
}

int mtm_debug_quotient_check(mtm_ctx* c, uint64_t n_cases, uint64_t seed, uint64_t* out4) {
    if (!c || !out4) return MTM_E_INVALID;
    MTM_NOT_IN_FLIGHT(c, "mtm_debug_quotient_check");
    HIPC(hipSetDevice(c->device));
    const int blocks = 2048, threads = 256;
    const uint64_t per = (n_cases + (uint64_t)blocks * threads - 1) / ((uint64_t)blocks * threads);
    MTMC(c->sched.ensure(sizeof(unsigned int) * (1 + 4096 + 8 * 32)));
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(c->sched.as<uint32_t>() + 4096 + 16);   // (behind mtm_debug_poison's word)
    HIPC(hipMemsetAsync(acc, 0, 4 * sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(quotient_check_kernel, dim3(blocks), dim3(threads), 0, c->stream, per, seed, acc);
    unsigned long long h[4] = {0, 0, 0, 0};
    HIPC(hipMemcpyAsync(h, acc, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    HIPC(hipGetLastError());
    out4[0] = per * (uint64_t)blocks * threads;
    out4[1] = h[1];
    out4[2] = h[2];
    out4[3] = h[3];
    return MTM_OK;
}

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

}  // extern "C"
