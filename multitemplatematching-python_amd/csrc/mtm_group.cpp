// mtm_group: one process, several GPUs.  One context + one worker thread per device; a search shards the units
// (templates / rotations / scales) over the devices by longest-processing-time-first on their multiply-accumulate
// cost, every device uploads the image and searches its shard concurrently (mtm_set_templates +
// mtm_find_matches_image on its own context and streams), and the per-device hit lists are merged on the host in
// template order.  The reference's equivalent is its thread pool over templates (MTM/__init__.py:172-175); the
// independence of the units is the same, the workers are GPUs.  No collective is needed in a single process:
// every hit list is already in host memory when its worker finishes (the one-process-per-GPU form with the RCCL
// all-gather is mtm_comm_*).  Host-only C++ on top of the C ABI.
#include <algorithm>
#include <condition_variable>
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
    int pending = 0;
    bool stop = false;
    Job job;
    std::vector<mtm_hit> last_hits;
};

namespace {

void run_job(mtm_group* g, Worker& w) {
    const Job& j = g->job;
    w.hits.clear();
    w.rc = MTM_OK;
    w.err.clear();
    if (w.templs.empty()) return;                     // more devices than units
    int rc = mtm_set_templates(w.ctx, w.templs.data(), (int)w.templs.size(), j.method);
    if (rc == MTM_OK) {
        int64_t n = 0;
        w.hits.resize(4096);
        rc = mtm_find_matches_image(w.ctx, j.px, j.rows, j.cols, j.chans, j.dtype, j.stride, j.mode, j.thr, w.hits.data(),
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

void worker_main(mtm_group* g, int wi) {
    unsigned long long seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv_job.wait(lk, [&] { return g->stop || g->generation != seen; });
            if (g->stop) return;
            seen = g->generation;
        }
        run_job(g, g->workers[(size_t)wi]);
        {
            std::lock_guard<std::mutex> lk(g->mu);
            if (--g->pending == 0) g->cv_done.notify_one();
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
    for (int i = 0; i < n_devices; ++i) g->workers[(size_t)i].th = std::thread(worker_main, g, i);
    *out = g;
    return MTM_OK;
}

void mtm_group_destroy(mtm_group* g) {
    if (!g) return;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->stop = true;
    }
    g->cv_job.notify_all();
    for (Worker& w : g->workers)
        if (w.th.joinable()) w.th.join();
    for (Worker& w : g->workers) mtm_ctx_destroy(w.ctx);
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

int mtm_group_find_matches(mtm_group* g, const mtm_templ* templs, int n_templ, int method, const void* px, int rows,
                           int cols, int chans, int dtype, int64_t row_stride_bytes, int mode, double score_threshold,
                           mtm_hit* out, int64_t capacity, int64_t* n_out) {
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
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->job = Job{method, mode, score_threshold, px, rows, cols, chans, dtype, row_stride_bytes};
        g->pending = nd;
        ++g->generation;
    }
    g->cv_job.notify_all();
    {
        std::unique_lock<std::mutex> lk(g->mu);
        g->cv_done.wait(lk, [&] { return g->pending == 0; });
    }
    for (Worker& w : g->workers)
        if (w.rc != MTM_OK) {
            set_error(w.err);
            return w.rc;
        }
    // merge in the single-device order: template index, then each device's own order within a template
    std::vector<mtm_hit> all;
    for (Worker& w : g->workers) all.insert(all.end(), w.hits.begin(), w.hits.end());
    std::stable_sort(all.begin(), all.end(), [](const mtm_hit& a, const mtm_hit& b) { return a.templ_idx < b.templ_idx; });
    *n_out = (int64_t)all.size();
    g->last_hits.swap(all);
    if ((int64_t)g->last_hits.size() > capacity) {
        set_error("mtm_group_find_matches: output capacity too small (fetch the result with mtm_group_last_hits)");
        return MTM_E_OVERFLOW;
    }
    if (!g->last_hits.empty()) std::memcpy(out, g->last_hits.data(), sizeof(mtm_hit) * g->last_hits.size());
    return MTM_OK;
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
