// libmtm_hip.so - hit exchange between one-process-per-GPU ranks: RCCL all-gather of fixed-size slots of hit records
// (librccl is dlopen'ed; the package's own TCP store carries the unique id).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "mtm_ctx.h"

using namespace mtm;
using namespace mtmi;

namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    // one process, several devices (mtm_comm_init_all / mtm_comm_allgather_hits_all)
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
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
    g_rccl.CommInitAll = reinterpret_cast<decltype(g_rccl.CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(dlsym(lib, "ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
    g_rccl.CommCount = reinterpret_cast<decltype(g_rccl.CommCount)>(dlsym(lib, "ncclCommCount"));
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

extern "C" {

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

}  // extern "C"

namespace mtmi {
// The exchange itself.  `my_flag` travels in the spare bytes of this rank's slot header and comes back for every rank in
// `flags_out` (n_ranks entries): a rank whose LOCAL step failed still takes part - with zero records and its error code -
// so that the others neither hang in the collective nor return a list with that rank's hits silently missing
// (mtm_find_matches_image_sharded_nms; advisor finding of round 5).
int comm_allgather_hits_flagged(mtm_ctx* c, const mtm_hit* local, int64_t n_local, int32_t my_flag, mtm_hit* out,
                                int64_t capacity, int64_t* counts_out, int64_t* n_out, std::vector<int32_t>* flags_out) {
    if (!c || !c->comm || n_local < 0 || (n_local > 0 && !local) || !counts_out || !n_out) {
        set_error("mtm_comm_allgather_hits: bad arguments or communicator not initialised");
        return MTM_E_INVALID;
    }
    MTM_NOT_IN_FLIGHT(c, "mtm_comm_allgather_hits");
    HIPC(hipSetDevice(c->device));
    const int R = c->n_ranks;
    if (flags_out) flags_out->assign((size_t)R, 0);
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
        std::memcpy(mine + 8, &my_flag, sizeof(my_flag));
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
            if (flags_out) std::memcpy(&(*flags_out)[(size_t)r], all + slot * r + 8, sizeof(int32_t));
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
}  // namespace mtmi

extern "C" {

int mtm_comm_allgather_hits(mtm_ctx* c, const mtm_hit* local, int64_t n_local, mtm_hit* out,
                            int64_t capacity, int64_t* counts_out, int64_t* n_out) {
    std::vector<int32_t> flags;
    const int rc = mtmi::comm_allgather_hits_flagged(c, local, n_local, 0, out, capacity, counts_out, n_out, &flags);
    if (rc != MTM_OK && rc != MTM_E_OVERFLOW) return rc;
    for (size_t r = 0; r < flags.size(); ++r)
        if (flags[r] != 0) {            // (a rank inside mtm_find_matches_image_sharded_nms whose local search failed)
            set_error("mtm_comm_allgather_hits: rank " + std::to_string(r) + " reported a failed local step (code " +
                      std::to_string(flags[r]) + "); its hits are missing from the gathered list");
            return MTM_E_COMM;
        }
    return rc;
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

// ---- one process, one communicator per device of a group (single process, ncclCommInitAll, one stream per
// device, the all-gather of every device inside ncclGroupStart / ncclGroupEnd, issued by the calling thread)
int mtm_comm_init_all(mtm_ctx* const* ctxs, int n) {
    if (!ctxs || n < 1) {
        set_error("mtm_comm_init_all: bad arguments");
        return MTM_E_INVALID;
    }
    std::vector<int> devs((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) {
            set_error("mtm_comm_init_all: null context");
            return MTM_E_INVALID;
        }
        devs[(size_t)i] = ctxs[i]->device;
        for (int k = 0; k < i; ++k)
            if (devs[(size_t)k] == devs[(size_t)i]) {
                // RCCL refuses two ranks of one communicator on one device; such a group (several contexts aliased
                // onto one GPU) merges its hit lists on the host
                set_error("mtm_comm_init_all: device " + std::to_string(devs[(size_t)i]) +
                          " is listed twice - one RCCL rank per device (the group keeps the host merge)");
                return MTM_E_COMM;
            }
    }
    MTMC(load_rccl());
    if (!g_rccl.CommInitAll || !g_rccl.GroupStart || !g_rccl.GroupEnd) {
        set_error("librccl.so lacks ncclCommInitAll / ncclGroupStart / ncclGroupEnd");
        return MTM_E_COMM;
    }
    for (int i = 0; i < n; ++i) (void)mtm_comm_destroy(ctxs[i]);
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    NCCLC(g_rccl.CommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; ++i) {
        ctxs[i]->comm = comms[(size_t)i];
        ctxs[i]->n_ranks = n;
        ctxs[i]->rank = i;
        ctxs[i]->comm_slot_hits = 512;
    }
    return MTM_OK;
}

int mtm_comm_count(mtm_ctx* c) {
    if (!c || !c->comm) return 0;
    int n = c->n_ranks;
    if (g_rccl.CommCount && g_rccl.CommCount(c->comm, &n) != ncclSuccess) return 0;
    return n;
}

int mtm_comm_allgather_hits_all(mtm_ctx* const* ctxs, int n, const mtm_hit* const* local, const int64_t* n_local, mtm_hit* out,
                                int64_t capacity, int64_t* counts_out, int64_t* n_out) {
    if (!ctxs || n < 1 || !local || !n_local || !counts_out || !n_out || capacity < 0 || (capacity > 0 && !out)) {
        set_error("mtm_comm_allgather_hits_all: bad arguments");
        return MTM_E_INVALID;
    }
    long long mx = 1;
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i] || !ctxs[i]->comm || ctxs[i]->n_ranks != n || ctxs[i]->rank != i || n_local[i] < 0 ||
            (n_local[i] > 0 && !local[i])) {
            set_error("mtm_comm_allgather_hits_all: contexts are not the ranks 0 .. n-1 of one mtm_comm_init_all");
            return MTM_E_INVALID;
        }
        MTM_NOT_IN_FLIGHT(ctxs[i], "mtm_comm_allgather_hits_all");
        mx = std::max<long long>(mx, n_local[i]);
    }
    // one process: every count is known before the exchange, so the slot is sized once ([count | records], 512 records
    // or the next power of two that holds the longest list) and ONE all-gather per device does it
    long long slot_hits = 512;
    while (slot_hits < mx) slot_hits <<= 1;
    const size_t slot = 16 + sizeof(mtm_hit) * (size_t)slot_hits;
    for (int i = 0; i < n; ++i) {
        mtm_ctx* c = ctxs[i];
        HIPC(hipSetDevice(c->device));
        MTMC(c->comm_send.ensure(slot));
        MTMC(c->comm_recv.ensure(slot * n));
        const size_t pin_bytes = slot * (size_t)(n + 1);
        if (c->comm_pin_cap < pin_bytes) {
            if (c->comm_pin) (void)hipHostFree(c->comm_pin);
            c->comm_pin = nullptr;
            c->comm_pin_cap = 0;
            HIPC(hipHostMalloc(&c->comm_pin, pin_bytes, hipHostMallocDefault));
            c->comm_pin_cap = pin_bytes;
        }
        uint8_t* mine = static_cast<uint8_t*>(c->comm_pin);
        std::memset(mine, 0, 16);
        const long long cnt = n_local[i];
        std::memcpy(mine, &cnt, sizeof(cnt));
        if (cnt > 0) std::memcpy(mine + 16, local[i], sizeof(mtm_hit) * (size_t)cnt);
        HIPC(hipMemcpyAsync(c->comm_send.p, mine, 16 + sizeof(mtm_hit) * (size_t)cnt, hipMemcpyHostToDevice, c->stream));
    }
    NCCLC(g_rccl.GroupStart());
    for (int i = 0; i < n; ++i) {
        mtm_ctx* c = ctxs[i];
        const ncclResult_t r = g_rccl.AllGather(c->comm_send.p, c->comm_recv.p, slot, ncclInt8, c->comm, c->stream);
        if (r != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            set_error(std::string("ncclAllGather: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error"));
            return MTM_E_COMM;
        }
    }
    NCCLC(g_rccl.GroupEnd());
    // rank 0's gathered slots feed the caller (the global NMS runs once, MTM/NMS.py:78); the other devices only have to
    // finish their part of the collective
    mtm_ctx* c0 = ctxs[0];
    HIPC(hipSetDevice(c0->device));
    uint8_t* gathered = static_cast<uint8_t*>(c0->comm_pin) + slot;
    HIPC(hipMemcpyAsync(gathered, c0->comm_recv.p, slot * n, hipMemcpyDeviceToHost, c0->stream));
    const double limit = c0->comm_timeout_s;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) {
        mtm_ctx* c = ctxs[i];
        hipError_t qs;
        int spins = 0;
        while ((qs = hipStreamQuery(c->stream)) == hipErrorNotReady) {
            if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
            if (limit > 0.0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
                for (int k = 0; k < n; ++k) {
                    if (g_rccl.CommAbort) (void)g_rccl.CommAbort(ctxs[k]->comm);
                    ctxs[k]->comm = nullptr;
                    ctxs[k]->n_ranks = 1;
                    ctxs[k]->rank = 0;
                }
                set_error("mtm_comm_allgather_hits_all: the exchange did not finish within the deadline (MTM_COMM_TIMEOUT_S); "
                          "communicators aborted");
                return MTM_E_COMM;
            }
        }
        HIPC(qs);
    }
    std::vector<long long> counts((size_t)n, 0);
    for (int r = 0; r < n; ++r) std::memcpy(&counts[(size_t)r], gathered + slot * r, sizeof(long long));
    c0->comm_last_counts = counts;
    c0->comm_last_slot = slot;
    return mtm_comm_last_gather(c0, out, capacity, counts_out, n_out);
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
