// Sanitizer driver for the host-only C++ of libmtm_hip.so (no GPU, no HIP): mtm_host.cpp (NMS, 1-D peaks, hit
// sorting, template statistics) and mtm_group.cpp (worker threads, generation counter, LPT shards, host merge) are
// compiled as they are, with -fsanitize=address,undefined and again with -fsanitize=thread; the per-device context
// API the group drives (mtm_ctx_create, mtm_set_templates, mtm_find_matches_image, ...) is replaced by a fake that
// returns deterministic hits - so the group's threading protocol runs thousands of jobs under the sanitizers.
// Built and run by tests/test_native_sanitizers_cpu.py.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../multitemplatematching-python_amd/csrc/mtm_internal.h"
#include "../../multitemplatematching-python_amd/csrc/mtm_nms_core.h"

using namespace mtm;

// ---- fake per-device contexts (what mtm_context / _placement / _api .hip provide in the real library)
struct mtm_ctx {
    int device = 0;
    std::vector<mtm_templ> templs;
    int method = 0;
    std::vector<mtm_hit> last;
    int64_t opt = 0;
};
static std::atomic<int> g_live_ctx{0};

extern "C" {
int mtm_ctx_create(mtm_ctx** out, int device_id) {
    if (device_id < 0) {
        set_error("fake: no such device");
        return MTM_E_NO_DEVICE;
    }
    *out = new mtm_ctx();
    (*out)->device = device_id;
    ++g_live_ctx;
    return MTM_OK;
}
void mtm_ctx_destroy(mtm_ctx* c) {
    if (c) --g_live_ctx;
    delete c;
}
int mtm_set_option(mtm_ctx* c, int, int64_t v) {
    c->opt = v;
    return MTM_OK;
}
int mtm_set_templates(mtm_ctx* c, const mtm_templ* t, int n, int method) {
    c->templs.assign(t, t + n);
    c->method = method;
    return MTM_OK;
}
// every template yields (rows % 7) hits whose coordinates encode (device-independent) facts about it; a template with
// cols == 13 makes the call fail (error propagation through the worker)
int mtm_find_matches_image(mtm_ctx* c, const void* px, int rows, int cols, int, int, int64_t, int, double thr, mtm_hit* out,
                           int64_t cap, int64_t* n_out) {
    c->last.clear();
    for (size_t i = 0; i < c->templs.size(); ++i) {
        const mtm_templ& t = c->templs[i];
        if (t.cols == 13) {
            set_error("fake: template refused");
            return MTM_E_INVALID;
        }
        for (int k = 0; k < t.rows % 7; ++k) {
            mtm_hit h;
            h.templ_idx = (int)i;
            h.x = t.cols * 100 + k;
            h.y = rows - t.rows + (px ? 1 : 0);
            h.w = t.cols;
            h.h = t.rows;
            h.score = (float)thr + (float)k * 0.001f + (float)cols * 1e-6f;
            c->last.push_back(h);
        }
    }
    *n_out = (int64_t)c->last.size();
    if ((int64_t)c->last.size() > cap) {
        set_error("fake: capacity");
        return MTM_E_OVERFLOW;
    }
    if (!c->last.empty()) std::memcpy(out, c->last.data(), sizeof(mtm_hit) * c->last.size());
    return MTM_OK;
}
// page-locked memory of the group's shared image staging: plain heap here
void* mtm_host_alloc(size_t bytes) { return std::malloc(bytes ? bytes : 1); }
void mtm_host_free(void* p) { std::free(p); }
// the in-process communicator calls: rank i = context i; the "all-gather" concatenates the lists in rank order
static std::atomic<int> g_comm_calls{0};
static std::atomic<bool> g_comm_fail{false};
int mtm_comm_init_all(mtm_ctx* const* ctxs, int n) {
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < i; ++k)
            if (ctxs[i]->device == ctxs[k]->device) {
                set_error("fake: device listed twice");
                return MTM_E_COMM;
            }
    return MTM_OK;
}
int mtm_comm_count(mtm_ctx*) { return 1; }
int mtm_comm_allgather_hits_all(mtm_ctx* const*, int n, const mtm_hit* const* local, const int64_t* n_local, mtm_hit* out,
                                int64_t cap, int64_t* counts, int64_t* n_out) {
    ++g_comm_calls;
    if (g_comm_fail.load()) {
        set_error("fake: the collective timed out");
        return MTM_E_COMM;
    }
    int64_t o = 0;
    for (int r = 0; r < n; ++r) {
        counts[r] = n_local[r];
        if (o + n_local[r] > cap) return MTM_E_OVERFLOW;
        if (n_local[r]) std::memcpy(out + o, local[r], sizeof(mtm_hit) * (size_t)n_local[r]);
        o += n_local[r];
    }
    *n_out = o;
    return MTM_OK;
}
int mtm_last_hits(mtm_ctx* c, mtm_hit* out, int64_t cap, int64_t* n_out) {
    *n_out = (int64_t)c->last.size();
    if ((int64_t)c->last.size() > cap) return MTM_E_OVERFLOW;
    if (!c->last.empty()) std::memcpy(out, c->last.data(), sizeof(mtm_hit) * c->last.size());
    return MTM_OK;
}
}

#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #cond, __LINE__); \
            std::exit(2);                                                        \
        }                                                                        \
    } while (0)

static void test_group(std::mt19937& rng) {
    for (int nd : {1, 2, 3, 8}) {
        std::vector<int> devs(nd);
        for (int i = 0; i < nd; ++i) devs[i] = i % 3;
        mtm_group* g = nullptr;
        CHECK(mtm_group_create(&g, devs.data(), nd) == MTM_OK);
        CHECK(mtm_group_size(g) == nd && mtm_group_ctx(g, nd) == nullptr && mtm_group_ctx(g, 0) != nullptr);
        CHECK(mtm_group_set_option(g, 3, 77) == MTM_OK);
        // in-process communicators: one rank per device - refused (host merge kept) when a device is listed twice
        const bool unique_devs = nd <= 3;
        CHECK(mtm_group_set_exchange(g, MTM_GROUP_EXCHANGE_RCCL) == MTM_E_STATE);
        CHECK(mtm_group_comm_init(g) == (unique_devs ? MTM_OK : MTM_E_COMM));
        CHECK(mtm_group_comm_ranks(g) == (unique_devs ? 1 : 0));
        const int calls0 = g_comm_calls.load();
        static const uint8_t pixel = 0;
        for (int job = 0; job < 400; ++job) {
            const int n = (int)(rng() % 70);
            std::vector<mtm_templ> t((size_t)n);
            long long expect = 0;
            bool refuse = false;
            for (int i = 0; i < n; ++i) {
                std::memset(&t[i], 0, sizeof(mtm_templ));
                t[i].px = &pixel;
                t[i].rows = 1 + (int)(rng() % 40);
                t[i].cols = 1 + (int)(rng() % 40);
                if (job % 50 == 49 && i == n / 2) t[i].cols = 13;
                refuse = refuse || t[i].cols == 13;
                t[i].chans = 1;
                expect += t[i].rows % 7;
            }
            std::vector<int32_t> dev((size_t)std::max(n, 1));
            CHECK(mtm_group_shards(g, t.data(), n, 5, 500, 600, dev.data()) == MTM_OK);
            for (int i = 0; i < n; ++i) CHECK(dev[i] >= 0 && dev[i] < nd);
            const int64_t cap = job % 7 == 0 ? 3 : 4096;            // small capacity: the overflow protocol
            if (unique_devs) CHECK(mtm_group_set_exchange(g, job % 3 != 2 ? MTM_GROUP_EXCHANGE_RCCL : MTM_GROUP_EXCHANGE_HOST) == MTM_OK);
            std::vector<mtm_hit> out((size_t)cap);
            int64_t got = -1;
            int rc = mtm_group_find_matches(g, t.data(), n, 5, &pixel, 500, 600, 1, MTM_U8, 600, 0, 0.5, out.data(), cap, &got);
            if (refuse) {
                CHECK(rc == MTM_E_INVALID && std::string(mtm_last_error()).find("template refused") != std::string::npos);
                continue;
            }
            if (expect > cap) {
                CHECK(rc == MTM_E_OVERFLOW && got == expect);
                out.resize((size_t)got);
                rc = mtm_group_last_hits(g, out.data(), got, &got);
            }
            CHECK(rc == MTM_OK && got == expect);
            CHECK(mtm_group_exchange_used(g) == (unique_devs && job % 3 != 2 ? MTM_GROUP_EXCHANGE_RCCL : MTM_GROUP_EXCHANGE_HOST));
            // merged in template order, global indices, each template's own hits in the order its device produced them
            int prev = -1, k = 0;
            for (int64_t i = 0; i < got; ++i) {
                const mtm_hit& h = out[(size_t)i];
                CHECK(h.templ_idx >= prev && h.templ_idx < n);
                k = h.templ_idx == prev ? k + 1 : 0;
                prev = h.templ_idx;
                CHECK(h.w == t[(size_t)h.templ_idx].cols && h.h == t[(size_t)h.templ_idx].rows && h.x == h.w * 100 + k);
            }
        }
        CHECK(unique_devs ? g_comm_calls.load() > calls0 : g_comm_calls.load() == calls0);
        // mtm_group_find_matches_nms == mtm_nms over mtm_group_find_matches' list (overlapping boxes: the fake's hits of one
        // template sit one pixel apart), for both exchanges, finite N_object and an overflowing capacity
        for (int job = 0; job < 60; ++job) {
            const int n = 1 + (int)(rng() % 24);
            std::vector<mtm_templ> t((size_t)n);
            for (int i = 0; i < n; ++i) {
                std::memset(&t[i], 0, sizeof(mtm_templ));
                t[i].px = &pixel;
                t[i].rows = 1 + (int)(rng() % 40);
                t[i].cols = 14 + (int)(rng() % 3);          // few distinct widths: boxes of different templates coincide
                t[i].chans = 1;
            }
            if (unique_devs) CHECK(mtm_group_set_exchange(g, job & 1 ? MTM_GROUP_EXCHANGE_RCCL : MTM_GROUP_EXCHANGE_HOST) == MTM_OK);
            const int method = job % 3 == 0 ? 1 : 5;
            const double thr = 0.5, ov = (job % 4) * 0.25;
            const int64_t nobj = job % 5 == 0 ? 2 : -1;
            std::vector<mtm_hit> all(4096), fused(4096);
            int64_t n_all = -1, n_fused = -1;
            CHECK(mtm_group_find_matches(g, t.data(), n, method, &pixel, 500, 600, 1, MTM_U8, 600, 0, thr, all.data(), 4096, &n_all) == MTM_OK);
            const int64_t cap = job % 7 == 0 ? 1 : 4096;
            int rc = mtm_group_find_matches_nms(g, t.data(), n, method, &pixel, 500, 600, 1, MTM_U8, 600, thr, ov, nobj, fused.data(), cap, &n_fused);
            if (rc == MTM_E_OVERFLOW) {
                CHECK(n_fused > cap);
                rc = mtm_group_last_hits(g, fused.data(), 4096, &n_fused);
            }
            CHECK(rc == MTM_OK);
            std::vector<int32_t> keep((size_t)std::max<int64_t>(n_all, 1));
            int64_t nk = 0;
            if (n_all > 1) {
                CHECK(mtm_nms(all.data(), n_all, thr, method == 1, nobj, ov, keep.data(), &nk) == MTM_OK);
            } else {                                            // MTM/NMS.py:53-55: a list of one hit is returned as it is
                nk = n_all;
                keep[0] = 0;
                if (nobj >= 0 && nk > nobj) nk = nobj;
            }
            CHECK(nk == n_fused);
            for (int64_t i = 0; i < nk; ++i) CHECK(std::memcmp(&all[(size_t)keep[(size_t)i]], &fused[(size_t)i], sizeof(mtm_hit)) == 0);
        }
        // a failing collective costs the exchange, not the search: the lists are merged on the host, now and afterwards
        if (unique_devs) {
            CHECK(mtm_group_set_exchange(g, MTM_GROUP_EXCHANGE_RCCL) == MTM_OK);
            g_comm_fail.store(true);
            std::vector<mtm_templ> t(3);
            for (int i = 0; i < 3; ++i) {
                std::memset(&t[i], 0, sizeof(mtm_templ));
                t[i].px = &pixel;
                t[i].rows = 5 + i;
                t[i].cols = 20;
                t[i].chans = 1;
            }
            std::vector<mtm_hit> out(4096);
            int64_t got = -1;
            CHECK(mtm_group_find_matches(g, t.data(), 3, 5, &pixel, 500, 600, 1, MTM_U8, 600, 0, 0.5, out.data(), 4096, &got) == MTM_OK);
            CHECK(got == 5 + 6 + 0 && mtm_group_exchange_used(g) == MTM_GROUP_EXCHANGE_HOST);
            g_comm_fail.store(false);
            CHECK(mtm_group_set_exchange(g, MTM_GROUP_EXCHANGE_RCCL) == MTM_E_STATE);      // until mtm_group_comm_init runs again
            CHECK(mtm_group_find_matches(g, t.data(), 3, 5, &pixel, 500, 600, 1, MTM_U8, 600, 0, 0.5, out.data(), 4096, &got) == MTM_OK && got == 11);
            CHECK(mtm_group_comm_init(g) == MTM_OK);
        }
        // an image of a megabyte and more goes through the group's shared page-locked staging buffer: every worker copies
        // its slice of the rows (strided and contiguous sources), waits for the others, searches from the buffer
        if (nd > 1) {
            const int rows = 1030, cols = 1100;
            for (int64_t stride : {(int64_t)cols, (int64_t)cols + 52}) {
                std::vector<uint8_t> img((size_t)rows * (size_t)stride, 7);
                for (int job = 0; job < 40; ++job) {
                    const int n = 1 + (int)(rng() % 12);
                    std::vector<mtm_templ> t((size_t)n);
                    long long expect = 0;
                    for (int i = 0; i < n; ++i) {
                        std::memset(&t[i], 0, sizeof(mtm_templ));
                        t[i].px = img.data();
                        t[i].rows = 1 + (int)(rng() % 40);
                        t[i].cols = 14 + (int)(rng() % 30);
                        expect += t[i].rows % 7;
                    }
                    std::vector<mtm_hit> out(4096);
                    int64_t got = -1;
                    CHECK(mtm_group_find_matches(g, t.data(), n, 5, img.data(), rows, cols, 1, MTM_U8, stride, 0, 0.5, out.data(), 4096,
                                                 &got) == MTM_OK && got == expect);
                }
            }
        }
        mtm_group_destroy(g);
        CHECK(g_live_ctx.load() == 0);
    }
    mtm_group* bad = nullptr;
    const int neg = -1;
    CHECK(mtm_group_create(&bad, &neg, 1) == MTM_E_NO_DEVICE && bad == nullptr && g_live_ctx.load() == 0);
}

static void test_host(std::mt19937& rng) {
    std::uniform_real_distribution<float> uf(0.f, 1.f);
    for (int rep = 0; rep < 300; ++rep) {
        const int n = (int)(rng() % 600);
        std::vector<mtm_hit> hits((size_t)n);
        for (auto& h : hits) {
            h.templ_idx = (int)(rng() % 5);
            h.x = (int)(rng() % 900);
            h.y = (int)(rng() % 700);
            h.w = 1 + (int)(rng() % 90);
            h.h = 1 + (int)(rng() % 90);
            h.score = rep % 11 == 0 ? 0.5f : uf(rng);               // all-equal scores: the stable-sort paths
            if (rep % 17 == 0 && (rng() % 9) == 0) h.score = NAN;
        }
        std::vector<int32_t> keep((size_t)std::max(n, 1));
        int64_t nk = -1;
        CHECK(mtm_nms(hits.data(), n, uf(rng), rep & 1, rep % 5 == 0 ? 3 : -1, uf(rng), keep.data(), &nk) == MTM_OK);
        CHECK(nk >= 0 && nk <= n);
        for (int64_t i = 0; i < nk; ++i) CHECK(keep[(size_t)i] >= 0 && keep[(size_t)i] < n);
        std::vector<mtm_hit> s = hits;
        sort_hits(s, (rep & 1) != 0);
        CHECK(s.size() == hits.size());
        for (size_t i = 1; i < s.size(); ++i) CHECK(s[i - 1].templ_idx <= s[i].templ_idx);
        // 1-D peaks on lines with plateaus and NaNs at the ends
        const int len = 1 + (int)(rng() % 300);
        std::vector<float> line((size_t)len);
        for (auto& v : line) v = (float)(rng() % 7) * 0.1f;
        const std::vector<int> pk = find_peaks_1d(line.data(), len, 1, 0.25f, (rep & 2) != 0);
        for (int p : pk) CHECK(p > 0 && p < len - 1);
        // template constants from pixels and from sums agree in size and do not read out of bounds
        const int th = 1 + (int)(rng() % 20), tw = 1 + (int)(rng() % 20), tc = 1 + (int)(rng() % 3);
        std::vector<double> px((size_t)th * tw * tc), mk((size_t)th * tw * tc);
        for (auto& v : px) v = (double)(rng() % 256);
        for (auto& v : mk) v = (double)(rng() % 2);
        for (int method = 0; method < 6; ++method) {
            const TemplStats a = compute_templ_stats(px.data(), nullptr, th, tw, tc, method, true);
            const TemplStats b = compute_templ_stats(px.data(), mk.data(), th, tw, tc, method, true);
            CHECK(a.inv_area > 0.0 && b.templ2_mask2_sum >= 0.0);
        }
    }
    // the device's NMS is built from two decisions (mtm_nms_core.h): "a precedes b" and "a suppresses b".  Greedy NMS written
    // with them, over hit lists in the order mtm_find_matches returns (sort_hits), must select what mtm_nms selects - ties in
    // the transformed score (few distinct scores, several templates), maxima and minima methods, every overlap threshold
    for (int rep = 0; rep < 200; ++rep) {
        const int n = 2 + (int)(rng() % 400), asc = rep & 1;
        std::vector<mtm_hit> hits((size_t)n);
        for (auto& h : hits) {
            h.templ_idx = (int)(rng() % 4);
            h.w = 20 + 10 * (h.templ_idx & 1);
            h.h = 24;
            h.x = (int)(rng() % 160);
            h.y = (int)(rng() % 120);
            h.score = rep % 3 == 0 ? 0.25f * (float)(rng() % 5) : uf(rng);
        }
        // (a pixel is listed once per template)
        std::sort(hits.begin(), hits.end(), [](const mtm_hit& a, const mtm_hit& b) {
            return std::tie(a.templ_idx, a.y, a.x) < std::tie(b.templ_idx, b.y, b.x); });
        hits.erase(std::unique(hits.begin(), hits.end(), [](const mtm_hit& a, const mtm_hit& b) {
            return a.templ_idx == b.templ_idx && a.y == b.y && a.x == b.x; }), hits.end());
        sort_hits(hits, asc != 0);
        const double thr = 0.1 * (double)(rng() % 8), ov = 0.1 * (double)(rng() % 11);
        std::vector<int32_t> keep(hits.size());
        int64_t nk = 0;
        CHECK(mtm_nms(hits.data(), (int64_t)hits.size(), thr, asc, -1, ov, keep.data(), &nk) == MTM_OK);
        std::vector<int> order;
        const float thr_s = (float)(asc ? 1.0 - thr : thr);
        for (int i = 0; i < (int)hits.size(); ++i)
            if (nms_score(hits[(size_t)i], asc) > thr_s) order.push_back(i);
        std::sort(order.begin(), order.end(), [&](int a, int b) { return nms_earlier(hits[(size_t)a], hits[(size_t)b], asc); });
        std::vector<int> kept;
        for (int i : order) {
            bool ok = true;
            for (int k : kept) ok = ok && nms_rect_overlap(hits[(size_t)i], hits[(size_t)k]) <= (float)ov;
            if (ok) kept.push_back(i);
        }
        CHECK((int64_t)kept.size() == nk);
        for (size_t i = 0; i < kept.size(); ++i) CHECK(kept[i] == keep[i]);
        // nms_select (mtm_find_matches_image_nms): the same hits in the same order from the list in ANY order
        std::vector<mtm_hit> shuffled = hits;
        std::shuffle(shuffled.begin(), shuffled.end(), rng);
        std::vector<int32_t> sel;
        nms_select(shuffled.data(), (int64_t)shuffled.size(), asc, thr_s, (float)ov, sel);
        CHECK((int64_t)sel.size() == nk);
        for (size_t i = 0; i < sel.size(); ++i)
            CHECK(std::memcmp(&shuffled[(size_t)sel[i]], &hits[(size_t)keep[i]], sizeof(mtm_hit)) == 0);
        // ... and what the device does first (mtm_k_nms.hip.h): "champions" - candidates no earlier candidate overlaps
        // beyond the limit - are kept for certain, what a champion overlaps beyond the limit is dropped; the selection from
        // [champions | undecided rest], champions untested, is the same again
        std::vector<mtm_hit> champs, rest;
        std::vector<char> is_champ(hits.size(), 0), doomed(hits.size(), 0);
        for (int i : order) {
            bool c = true;
            for (int k : order) {
                if (k == i) break;                              // `order` is sorted: everything before i is earlier
                if (nms_rect_overlap(hits[(size_t)i], hits[(size_t)k]) > (float)ov) { c = false; break; }
            }
            is_champ[(size_t)i] = c;
        }
        for (int i : order)
            for (int k : order)
                if (is_champ[(size_t)k] && k != i && nms_rect_overlap(hits[(size_t)i], hits[(size_t)k]) > (float)ov) doomed[(size_t)i] = 1;
        for (int i : order) {
            if (is_champ[(size_t)i]) champs.push_back(hits[(size_t)i]);
            else if (!doomed[(size_t)i]) rest.push_back(hits[(size_t)i]);
        }
        if (ov >= 0.0) {
            std::vector<mtm_hit> pruned = champs;
            std::shuffle(rest.begin(), rest.end(), rng);
            pruned.insert(pruned.end(), rest.begin(), rest.end());
            nms_select(pruned.data(), (int64_t)pruned.size(), asc, thr_s, (float)ov, sel, (int64_t)champs.size());
            CHECK((int64_t)sel.size() == nk);
            for (size_t i = 0; i < sel.size(); ++i)
                CHECK(std::memcmp(&pruned[(size_t)sel[i]], &hits[(size_t)keep[i]], sizeof(mtm_hit)) == 0);
        }
    }
    // byte-run sums (the SSE2 pass over fresh template bytes): exact at every length and alignment, incl. all-255 runs
    // long enough to wrap a 32-bit lane if the block length were wrong
    std::vector<uint8_t> bytes((size_t)(16 * 8192 * 3 + 77));
    for (int rep = 0; rep < 40; ++rep) {
        for (auto& v : bytes) v = rep == 0 ? 255 : (uint8_t)(rng() & 255);
        const size_t off = rep == 0 ? 0 : rng() % 33, n = rep < 2 ? bytes.size() - off : rng() % 5000;
        unsigned long long s = 7, q = 9, rs = 7, rq = 9;
        u8_run_sums(bytes.data() + off, n, &s, &q);
        for (size_t i = 0; i < n; ++i) {
            rs += bytes[off + i];
            rq += (unsigned long long)bytes[off + i] * bytes[off + i];
        }
        CHECK(s == rs && q == rq);
    }
    int64_t nk = 0;
    CHECK(mtm_nms(nullptr, 5, 0.5, 0, -1, 0.5, nullptr, &nk) == MTM_E_INVALID && std::strlen(mtm_last_error()) > 0);
}

int main() {
    std::mt19937 rng(12345);
    test_host(rng);
    test_group(rng);
    // two groups driven from two caller threads at once (each group is single-caller; the library must not share state)
    std::thread a([] { std::mt19937 r(1); test_group(r); }), b([] { std::mt19937 r(2); test_host(r); });
    a.join();
    b.join();
    std::puts("sanitize_host: ok");
    return 0;
}
