// mtm_group: one process, several GPUs.  One context + one worker thread per device; a search shards the units
// (templates / rotations / scales) over the devices by longest-processing-time-first on their multiply-accumulate
// cost, every device uploads the image and searches its shard concurrently (mtm_set_templates +
// mtm_find_matches_image on its own context and streams), and the per-device hit lists are merged on the host in
// template order.  The reference's equivalent is its thread pool over templates (MTM/__init__.py:172-175); the
// independence of the units is the same, the workers are GPUs.
//   * Hit exchange (round 4).  north_star / SURVEY 8e name the exchange: a single process, ncclCommInitAll, one stream
//     per device, one all-gather of fixed-size slots of hit records inside ncclGroupStart / ncclGroupEnd.
//     mtm_group_comm_init() builds those communicators; with MTM_GROUP_EXCHANGE_RCCL selected the per-device lists
//     travel device-side through that all-gather and rank 0's gathered list is what gets merged - the list the host
//     merge (every list is in host memory when its worker returns; the default without communicators, and the
//     fallback when a device is listed twice) produces from the workers' buffers.
//   * Image staging (round 4).  Every worker used to hand the caller's pageable image to its own upload (N staging
//     copies of the same pixels through the runtime's bounce buffers).  Now the workers copy one slice each into ONE
//     page-locked buffer, wait for each other, and every device's (banded) upload is a plain DMA from it.
// Host-only C++ on top of the C ABI.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "mtm_internal.h"

using namespace mtm;

namespace {

struct Worker {
    mtm_ctx* ctx = nullptr;
    int device = 0;
    std::thread th;
    // job inputs (written by the caller before the generation counter moves)
    std::vector<mtm_templ> templs;
    std::vector<int> global_idx;          // list position of every unit of this shard
    // job outputs
    std::vector<mtm_hit> hits;
    int rc = MTM_OK;
    std::string err;
};

struct Job {
    int method = 0, mode = 0;
    double thr = 0.0;
    const void* px = nullptr;
    int rows = 0, cols = 0, chans = 0, dtype = 0;
    int64_t stride = 0;
};

}  // namespace

struct mtm_group {
    std::vector<Worker> workers;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    unsigned long long generation = 0;
    bool stop = false;
    // Round 5: the same hand-off through atomics the workers and the caller SPIN on before they sleep on the condition
    // variables (a wake-up through the futex is 20-50 us, twice per search and on the critical path of every device)
    std::atomic<unsigned long long> gen_a{0};
    std::atomic<int> pending_a{0};
    std::atomic<bool> stop_a{false};
    long long spin_us = 300;                    // MTM_GROUP_SPIN_US: how long a worker polls for the next job before it sleeps; the
                                                // caller polls for the searches' end for at most 1 ms.  0 = no polling at all:
                                                // the setting for shared / CPU-constrained hosts (N + 1 busy cores otherwise)
    Job job;
    std::vector<mtm_hit> last_hits;
    // hit exchange
    int exchange = MTM_GROUP_EXCHANGE_HOST;     // what the next search uses
    int exchange_used = MTM_GROUP_EXCHANGE_HOST;   // what the last search used
    bool comm_ready = false;
    // shared page-locked staging of the image (small images and failed allocations: every worker uploads the caller's buffer)
    void* pin = nullptr;
    size_t pin_cap = 0;
    size_t row_bytes = 0;                       // bytes of one image row in the staging buffer (tight)
    bool staged_job = false;                    // this job's image goes through the staging buffer
    std::atomic<int> staged{0};                 // workers that have copied their slice of this job's image
};

namespace {

void run_job(mtm_group* g, Worker& w) {
    const Job& j = g->job;
    w.hits.clear();
    w.rc = MTM_OK;
    w.err.clear();
    const void* px = j.px;
    int64_t stride = j.stride;
    if (g->staged_job) {
        // my slice of the rows -> the shared page-locked buffer (every worker takes part, also one without units), then
        // wait for the others: the searches below read all of it
        const int nd = (int)g->workers.size(), wi = (int)(&w - g->workers.data());
        const int r0 = (int)((long long)j.rows * wi / nd), r1 = (int)((long long)j.rows * (wi + 1) / nd);
        uint8_t* dst = static_cast<uint8_t*>(g->pin);
        const uint8_t* src = static_cast<const uint8_t*>(j.px);
        if ((size_t)j.stride == g->row_bytes) {
            std::memcpy(dst + (size_t)r0 * g->row_bytes, src + (size_t)r0 * g->row_bytes, (size_t)(r1 - r0) * g->row_bytes);
        } else {
            for (int r = r0; r < r1; ++r) std::memcpy(dst + (size_t)r * g->row_bytes, src + (long long)r * j.stride, g->row_bytes);
        }
        g->staged.fetch_add(1, std::memory_order_acq_rel);
        for (int spins = 0; g->staged.load(std::memory_order_acquire) < nd; ++spins)
            if (spins > 200) std::this_thread::yield();
        px = g->pin;
        stride = (int64_t)g->row_bytes;
    }
    if (w.templs.empty()) return;                     // more devices than units
    int rc = mtm_set_templates(w.ctx, w.templs.data(), (int)w.templs.size(), j.method);
    if (rc == MTM_OK) {
        int64_t n = 0;
        w.hits.resize(4096);
        rc = mtm_find_matches_image(w.ctx, px, j.rows, j.cols, j.chans, j.dtype, stride, j.mode, j.thr, w.hits.data(),
                                    (int64_t)w.hits.size(), &n);
        if (rc == MTM_E_OVERFLOW) {
            w.hits.resize((size_t)n);
            rc = mtm_last_hits(w.ctx, w.hits.data(), n, &n);
        }
        if (rc == MTM_OK) {
            w.hits.resize((size_t)n);
            for (mtm_hit& h : w.hits) h.templ_idx = w.global_idx[(size_t)h.templ_idx];
        }
    }
    if (rc != MTM_OK) {
        w.rc = rc;
        w.err = std::string("device ") + std::to_string(w.device) + ": " + mtm_last_error();   // this thread's message
        w.hits.clear();
    }
}

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield" ::: "memory");
#else
    std::this_thread::yield();
#endif
}

void worker_main(mtm_group* g, int wi) {
    unsigned long long seen = 0;
    for (;;) {
        // poll for the next job first (a loop over images calls again within ~100 us), then sleep
        bool got = false;
        if (g->spin_us > 0) {
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0;; ++i) {
                if (g->stop_a.load(std::memory_order_acquire) || g->gen_a.load(std::memory_order_acquire) != seen) {
                    got = true;
                    break;
                }
                cpu_relax();
                if ((i & 255) == 255 &&
                    std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > g->spin_us)
                    break;
            }
        }
        if (!got) {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv_job.wait(lk, [&] { return g->stop || g->generation != seen; });
        }
        if (g->stop_a.load(std::memory_order_acquire)) return;
        seen = g->gen_a.load(std::memory_order_acquire);
        run_job(g, g->workers[(size_t)wi]);
        if (g->pending_a.fetch_sub(1, std::memory_order_acq_rel) == 1) {
            std::lock_guard<std::mutex> lk(g->mu);      // (the caller may be asleep on cv_done)
            g->cv_done.notify_one();
        }
    }
}

// multiply-accumulates of the direct method for one unit (the partition cost; MTM/distributed.py::unit_cost)
double unit_cost(const mtm_templ& t, int rows, int cols, int method) {
    const double oh = std::max(rows - t.rows + 1, 0), ow = std::max(cols - t.cols + 1, 0);
    const bool masked = t.mask != nullptr && (method == MTM_TM_SQDIFF || method == MTM_TM_CCORR_NORMED);
    return oh * ow * (double)t.rows * (double)t.cols * (double)t.chans * (masked ? 2.0 : 1.0);
}

}  // namespace

extern "C" {

int mtm_group_create(mtm_group** out, const int* device_ids, int n_devices) {
    if (!out || n_devices < 1 || !device_ids) {
        set_error("mtm_group_create: bad arguments");
        return MTM_E_INVALID;
    }
    mtm_group* g = new mtm_group();
    g->workers.resize((size_t)n_devices);
    for (int i = 0; i < n_devices; ++i) {
        g->workers[(size_t)i].device = device_ids[i];
        const int rc = mtm_ctx_create(&g->workers[(size_t)i].ctx, device_ids[i]);
        if (rc != MTM_OK) {
            for (int k = 0; k < i; ++k) mtm_ctx_destroy(g->workers[(size_t)k].ctx);
            delete g;
            return rc;
        }
    }
    if (const char* v = std::getenv("MTM_GROUP_SPIN_US")) g->spin_us = std::atoll(v);
    for (int i = 0; i < n_devices; ++i) g->workers[(size_t)i].th = std::thread(worker_main, g, i);
    *out = g;
    return MTM_OK;
}

int mtm_group_comm_init(mtm_group* g) {
    if (!g) {
        set_error("mtm_group_comm_init: null group");
        return MTM_E_INVALID;
    }
    std::vector<mtm_ctx*> ctxs;
    for (Worker& w : g->workers) ctxs.push_back(w.ctx);
    g->comm_ready = false;
    const int rc = mtm_comm_init_all(ctxs.data(), (int)ctxs.size());
    if (rc != MTM_OK) {
        g->exchange = MTM_GROUP_EXCHANGE_HOST;
        return rc;
    }
    g->comm_ready = true;
    g->exchange = MTM_GROUP_EXCHANGE_RCCL;
    return MTM_OK;
}

int mtm_group_comm_ranks(mtm_group* g) {
    if (!g || !g->comm_ready || g->workers.empty()) return 0;
    return mtm_comm_count(g->workers[0].ctx);
}

int mtm_group_set_exchange(mtm_group* g, int kind) {
    if (!g || (kind != MTM_GROUP_EXCHANGE_HOST && kind != MTM_GROUP_EXCHANGE_RCCL)) {
        set_error("mtm_group_set_exchange: bad arguments");
        return MTM_E_INVALID;
    }
    if (kind == MTM_GROUP_EXCHANGE_RCCL && !g->comm_ready) {
        set_error("mtm_group_set_exchange: no communicators (mtm_group_comm_init first)");
        return MTM_E_STATE;
    }
    g->exchange = kind;
    return MTM_OK;
}

int mtm_group_exchange_used(const mtm_group* g) { return g ? g->exchange_used : MTM_GROUP_EXCHANGE_HOST; }

void mtm_group_destroy(mtm_group* g) {
    if (!g) return;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->stop = true;
        g->stop_a.store(true, std::memory_order_release);
    }
    g->cv_job.notify_all();
    for (Worker& w : g->workers)
        if (w.th.joinable()) w.th.join();
    for (Worker& w : g->workers) mtm_ctx_destroy(w.ctx);        // (also destroys the context's communicator)
    if (g->pin) mtm_host_free(g->pin);
    delete g;
}

int mtm_group_size(const mtm_group* g) { return g ? (int)g->workers.size() : 0; }

mtm_ctx* mtm_group_ctx(mtm_group* g, int i) {
    if (!g || i < 0 || i >= (int)g->workers.size()) return nullptr;
    return g->workers[(size_t)i].ctx;
}

int mtm_group_set_option(mtm_group* g, int option, int64_t value) {
    if (!g) return MTM_E_INVALID;
    for (Worker& w : g->workers) {
        const int rc = mtm_set_option(w.ctx, option, value);
        if (rc != MTM_OK) return rc;
    }
    return MTM_OK;
}

int mtm_group_shards(const mtm_group* g, const mtm_templ* templs, int n_templ, int method, int rows, int cols,
                     int32_t* device_of_unit) {
    if (!g || n_templ < 0 || (n_templ > 0 && (!templs || !device_of_unit))) {
        set_error("mtm_group_shards: bad arguments");
        return MTM_E_INVALID;
    }
    // longest-processing-time-first, deterministic: units by descending cost (ties: list order), each to the
    // least-loaded device (ties: lowest index)
    const int nd = (int)g->workers.size();
    std::vector<double> cost((size_t)n_templ);
    for (int i = 0; i < n_templ; ++i) cost[(size_t)i] = unit_cost(templs[i], rows, cols, method);
    std::vector<int> order((size_t)n_templ);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[(size_t)a] > cost[(size_t)b]; });
    std::vector<double> load((size_t)nd, 0.0);
    for (int i : order) {
        int best = 0;
        for (int d = 1; d < nd; ++d)
            if (load[(size_t)d] < load[(size_t)best]) best = d;
        device_of_unit[i] = best;
        load[(size_t)best] += cost[(size_t)i];
    }
    return MTM_OK;
}

}  // extern "C"

namespace {
struct GroupNms {
    bool on = false;
    double score_threshold = 0.0, max_overlap = 0.0;
    int64_t n_object = -1;
};
}  // namespace

static int group_search(mtm_group* g, const mtm_templ* templs, int n_templ, int method, const void* px, int rows,
                        int cols, int chans, int dtype, int64_t row_stride_bytes, int mode, double score_threshold,
                        const GroupNms& nms, mtm_hit* out, int64_t capacity, int64_t* n_out) {
    if (!g || n_templ < 0 || (n_templ > 0 && !templs) || !px || !n_out || capacity < 0 || (capacity > 0 && !out)) {
        set_error("mtm_group_find_matches: bad arguments");
        return MTM_E_INVALID;
    }
    const int nd = (int)g->workers.size();
    std::vector<int32_t> dev((size_t)n_templ, 0);
    const int rc0 = mtm_group_shards(g, templs, n_templ, method, rows, cols, dev.data());
    if (rc0 != MTM_OK) return rc0;
    for (Worker& w : g->workers) {
        w.templs.clear();
        w.global_idx.clear();
    }
    for (int i = 0; i < n_templ; ++i) {          // list order inside every shard
        Worker& w = g->workers[(size_t)dev[(size_t)i]];
        w.templs.push_back(templs[i]);
        w.global_idx.push_back(i);
    }
    // the image through ONE page-locked buffer (groups of several devices, images of a megabyte and more)
    {
        const size_t esz = dtype == MTM_U8 ? 1 : dtype == MTM_U16 ? 2 : 4;
        const size_t row_bytes = (size_t)cols * (size_t)chans * esz, need = row_bytes * (size_t)rows;
        g->staged_job = nd > 1 && need >= ((size_t)1 << 20) && row_stride_bytes >= (int64_t)row_bytes;
        if (g->staged_job && g->pin_cap < need) {
            if (g->pin) mtm_host_free(g->pin);
            g->pin = mtm_host_alloc(need);
            g->pin_cap = g->pin ? need : 0;
            if (!g->pin) g->staged_job = false;       // (no page-locked memory: every worker uploads the caller's buffer)
        }
        g->row_bytes = row_bytes;
        g->staged.store(0, std::memory_order_release);
    }
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->job = Job{method, mode, score_threshold, px, rows, cols, chans, dtype, row_stride_bytes};
        g->pending_a.store(nd, std::memory_order_release);
        ++g->generation;
        g->gen_a.store(g->generation, std::memory_order_release);
    }
    g->cv_job.notify_all();
    {
        // the searches take about a millisecond: poll for their end (this thread has nothing else to do), sleep only
        // when they take much longer than that
        bool done = false;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0;; ++i) {
            if (g->pending_a.load(std::memory_order_acquire) == 0) {
                done = true;
                break;
            }
            cpu_relax();
            if ((i & 255) == 255 &&
                std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >
                    std::min<long long>(20 * g->spin_us, 1000))
                break;
        }
        if (!done) {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv_done.wait(lk, [&] { return g->pending_a.load(std::memory_order_acquire) == 0; });
        }
    }
    for (Worker& w : g->workers)
        if (w.rc != MTM_OK) {
            set_error(w.err);
            return w.rc;
        }
    // merge in the single-device order: template index, then each device's own order within a template
    std::vector<mtm_hit> all;
    g->exchange_used = MTM_GROUP_EXCHANGE_HOST;
    if (g->exchange == MTM_GROUP_EXCHANGE_RCCL && g->comm_ready) {
        // device-side: one all-gather per device inside ncclGroupStart / End; rank 0's gathered list (rank order, each
        // rank's own order inside) is the concatenation the host merge would build
        std::vector<mtm_ctx*> ctxs;
        std::vector<const mtm_hit*> local;
        std::vector<int64_t> n_local, counts((size_t)nd, 0);
        size_t total = 0;
        for (Worker& w : g->workers) {
            ctxs.push_back(w.ctx);
            local.push_back(w.hits.data());
            n_local.push_back((int64_t)w.hits.size());
            total += w.hits.size();
        }
        all.resize(total);
        int64_t n_all = 0;
        const int rcx = mtm_comm_allgather_hits_all(ctxs.data(), nd, local.data(), n_local.data(), all.data(), (int64_t)all.size(),
                                                    counts.data(), &n_all);
        if (rcx == MTM_OK) {
            all.resize((size_t)n_all);
            g->exchange_used = MTM_GROUP_EXCHANGE_RCCL;
        } else {
            // every worker's list is in host memory already: a failed collective (time-out, aborted communicators) costs
            // the exchange, not the search - this and the following searches merge on the host
            g->comm_ready = false;
            g->exchange = MTM_GROUP_EXCHANGE_HOST;
            all.clear();
        }
    }
    if (g->exchange_used == MTM_GROUP_EXCHANGE_HOST)
        for (Worker& w : g->workers) all.insert(all.end(), w.hits.begin(), w.hits.end());
    std::stable_sort(all.begin(), all.end(), [](const mtm_hit& a, const mtm_hit& b) { return a.templ_idx < b.templ_idx; });
    if (nms.on) {
        // MTM.NMS on the merged list, as mtm_find_matches_image_nms runs it on a single context's (MTM/NMS.py:53-84): a list
        // of one hit is returned as it is, else cv2.dnn.NMSBoxes' selection in its order, then the first N_object
        const bool ascending = method == MTM_TM_SQDIFF || method == MTM_TM_SQDIFF_NORMED;
        if (all.size() > 1) {
            const float thr_s = (float)(ascending ? (1.0 - nms.score_threshold) : nms.score_threshold);
            std::vector<int32_t> keep;
            nms_select(all.data(), (int64_t)all.size(), ascending ? 1 : 0, thr_s, (float)nms.max_overlap, keep);
            std::vector<mtm_hit> kept(keep.size());
            for (size_t i = 0; i < keep.size(); ++i) kept[i] = all[(size_t)keep[i]];
            all.swap(kept);
        }
        if (nms.n_object >= 0 && (int64_t)all.size() > nms.n_object) all.resize((size_t)nms.n_object);
    }
    *n_out = (int64_t)all.size();
    g->last_hits.swap(all);
    if ((int64_t)g->last_hits.size() > capacity) {
        set_error("mtm_group_find_matches: output capacity too small (fetch the result with mtm_group_last_hits)");
        return MTM_E_OVERFLOW;
    }
    if (!g->last_hits.empty()) std::memcpy(out, g->last_hits.data(), sizeof(mtm_hit) * g->last_hits.size());
    return MTM_OK;
}

extern "C" {

int mtm_group_find_matches(mtm_group* g, const mtm_templ* templs, int n_templ, int method, const void* px, int rows,
                           int cols, int chans, int dtype, int64_t row_stride_bytes, int mode, double score_threshold,
                           mtm_hit* out, int64_t capacity, int64_t* n_out) {
    return group_search(g, templs, n_templ, method, px, rows, cols, chans, dtype, row_stride_bytes, mode, score_threshold,
                        GroupNms{}, out, capacity, n_out);
}

int mtm_group_find_matches_nms(mtm_group* g, const mtm_templ* templs, int n_templ, int method, const void* px, int rows,
                               int cols, int chans, int dtype, int64_t row_stride_bytes, double score_threshold,
                               double max_overlap, int64_t n_object, mtm_hit* out, int64_t capacity, int64_t* n_out) {
    GroupNms nms;
    nms.on = true;
    nms.score_threshold = score_threshold;
    nms.max_overlap = max_overlap;
    nms.n_object = n_object;
    return group_search(g, templs, n_templ, method, px, rows, cols, chans, dtype, row_stride_bytes, MTM_PEAKS_LOCAL,
                        score_threshold, nms, out, capacity, n_out);
}

int mtm_group_last_hits(mtm_group* g, mtm_hit* out, int64_t capacity, int64_t* n_out) {
    if (!g || !n_out || capacity < 0 || (capacity > 0 && !out)) {
        set_error("mtm_group_last_hits: bad arguments");
        return MTM_E_INVALID;
    }
    *n_out = (int64_t)g->last_hits.size();
    if ((int64_t)g->last_hits.size() > capacity) {
        set_error("mtm_group_last_hits: output capacity too small");
        return MTM_E_OVERFLOW;
    }
    if (!g->last_hits.empty()) std::memcpy(out, g->last_hits.data(), sizeof(mtm_hit) * g->last_hits.size());
    return MTM_OK;
}

}  // extern "C"
