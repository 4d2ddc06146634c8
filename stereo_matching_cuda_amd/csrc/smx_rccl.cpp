// smx_rccl.cpp -- libsmx_rccl.so: the one exchange step of the D-sharded stereo path (RCCL over xGMI)
// and a single-process multi-GPU driver on top of the device-pointer C-ABI of libsmx_hip.so.
// Host code only (no kernels here); see include/smx_rccl.h.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "smx_rccl.h"

namespace {

thread_local char g_msg[512];

int fail(int code, const char* fmt, const char* what, const char* detail, int line) {
    snprintf(g_msg, sizeof(g_msg), fmt, what, detail, line);
    fprintf(stderr, "smx_rccl: %s\n", g_msg);
    return code;
}

#define RC_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess) return fail(SMX_E_HIP, "%s -> %s (line %d)", #call, hipGetErrorString(e__), __LINE__); \
    } while (0)
#define RC_NCCL(call)                                                                                  \
    do {                                                                                               \
        ncclResult_t e__ = (call);                                                                     \
        if (e__ != ncclSuccess) return fail(SMX_E_HIP, "%s -> %s (line %d)", #call, ncclGetErrorString(e__), __LINE__); \
    } while (0)
#define RC_SMX(call)                                                                                   \
    do {                                                                                               \
        int e__ = (call);                                                                              \
        if (e__ != SMX_OK) return fail(e__, "%s -> %s (line %d)", #call, smx_last_error(), __LINE__);  \
    } while (0)

// everything one device owns, for the lifetime of the context
struct Shard {
    int dev = -1;
    hipStream_t st = nullptr;       // aggregation
    hipStream_t cst = nullptr;      // exchange (so that it can overlap the aggregation of the other view)
    hipEvent_t ev_view[2] = {nullptr, nullptr};   // keys of view v complete on `st`
    hipEvent_t ev_comm = nullptr;                 // exchange complete on `cst`
    ncclComm_t comm = nullptr;
    uint8_t *L = nullptr, *R = nullptr, *mean = nullptr;
    int64_t* keys = nullptr;
    void* ws = nullptr;
    size_t ws_bytes = 0;
    int s0 = 0, s1 = 0;             // slice range
    ~Shard() {
        if (dev < 0) return;
        (void)hipSetDevice(dev);
        if (st) (void)hipStreamSynchronize(st);
        if (cst) (void)hipStreamSynchronize(cst);
        if (comm) (void)ncclCommDestroy(comm);
        for (void* q : {(void*)L, (void*)R, (void*)mean, (void*)keys, ws})
            if (q) (void)hipFree(q);
        for (hipEvent_t e : {ev_view[0], ev_view[1], ev_comm})
            if (e) (void)hipEventDestroy(e);
        if (st) (void)hipStreamDestroy(st);
        if (cst) (void)hipStreamDestroy(cst);
    }
};

// One host thread per device, alive as long as the context: a pair is ~12 kernel launches, two uploads and one collective
// per device, and ONE thread feeding eight devices in turn puts ~100 serial runtime calls in front of a 0.3 ms shard
// (round-4 review).  A worker sets its device once, then runs the jobs the owner hands it; the error text of a failing
// job (thread-local in both libraries) travels back with its return code.
struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = true, quit = false;
    int rc = SMX_OK;
    char msg[512] = {0};
    void start(int dev) {
        th = std::thread([this, dev] {
            (void)hipSetDevice(dev);
            std::unique_lock<std::mutex> lk(m);
            for (;;) {
                cv.wait(lk, [this] { return has_job || quit; });
                if (quit) return;
                has_job = false;
                lk.unlock();
                const int r = job();
                lk.lock();
                rc = r;
                snprintf(msg, sizeof(msg), "%s", r == SMX_OK ? "" : g_msg);
                done = true;
                cv.notify_all();
            }
        });
    }
    void post(std::function<int()> f) {
        std::lock_guard<std::mutex> lk(m);
        job = std::move(f);
        has_job = true;
        done = false;
        cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [this] { return done; });
        return rc;
    }
    ~Worker() {
        if (!th.joinable()) return;
        { std::lock_guard<std::mutex> lk(m); quit = true; cv.notify_all(); }
        th.join();
    }
};

// all workers of a pair meet here between the aggregation and the exchange: a collective is only started when EVERY
// rank has come this far without an error (a rank that failed earlier would leave the others' collectives hanging)
struct Rendezvous {
    std::mutex m;
    std::condition_variable cv;
    int n = 0, arrived = 0, generation = 0;
    std::atomic<int> failed{0};
    void arrive(bool ok) {
        if (!ok) failed.store(1);
        std::unique_lock<std::mutex> lk(m);
        const int gen = generation;
        if (++arrived == n) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen; });
    }
};

}  // namespace

struct smx_sharded_ctx {
    smx_params p;
    int w = 0, h = 0, size_d = 0, ngpu = 0, flags = 0;
    size_t n = 0;
    std::vector<Shard> sh;
    std::vector<std::unique_ptr<Worker>> workers;     // one per device (declared behind `sh`: joined before the shards go)
    Rendezvous meet;
    // decode + LR check + filling on device 0
    float *best = nullptr, *map = nullptr, *occ = nullptr, *fil = nullptr;
    ~smx_sharded_ctx() {
        if (!sh.empty() && sh[0].dev >= 0) {
            (void)hipSetDevice(sh[0].dev);
            for (float* q : {best, map, occ, fil})
                if (q) (void)hipFree(q);
        }
    }
};

extern "C" {

int smx_wta_allreduce(int64_t* d_keys, int64_t n, void* nccl_comm, void* stream) {
    if (!d_keys || n <= 0 || !nccl_comm) return fail(SMX_E_ARG, "%s: %s (line %d)", "smx_wta_allreduce", "bad argument", __LINE__);
    RC_NCCL(ncclAllReduce(d_keys, d_keys, (size_t)n, ncclInt64, ncclMin, (ncclComm_t)nccl_comm, (hipStream_t)stream));
    return SMX_OK;
}

int smx_wta_reduce(int64_t* d_keys, int64_t n, int root, void* nccl_comm, void* stream) {
    if (!d_keys || n <= 0 || !nccl_comm || root < 0) return fail(SMX_E_ARG, "%s: %s (line %d)", "smx_wta_reduce", "bad argument", __LINE__);
    RC_NCCL(ncclReduce(d_keys, d_keys, (size_t)n, ncclInt64, ncclMin, root, (ncclComm_t)nccl_comm, (hipStream_t)stream));
    return SMX_OK;
}

int smx_sharded_create(const smx_params* p, int w, int h, int size_d, int ngpu, int flags, smx_sharded_ctx** out) {
    if (!p || !out || w < 2 || h < 1 || size_d < 1 || ngpu < 1)
        return fail(SMX_E_ARG, "%s: %s (line %d)", "smx_sharded_create", "bad argument", __LINE__);
    *out = nullptr;
    int ndev = 0;
    RC_HIP(hipGetDeviceCount(&ndev));
    if (ngpu > ndev) return fail(SMX_E_ARG, "%s: %s (line %d)", "smx_sharded_create", "more shards than devices", __LINE__);
    smx_sharded_ctx* c = new (std::nothrow) smx_sharded_ctx;
    if (!c) return fail(SMX_E_HIP, "%s: %s (line %d)", "smx_sharded_create", "out of host memory", __LINE__);
    struct Guard { smx_sharded_ctx* c; ~Guard() { delete c; } } guard{c};
    c->p = *p; c->w = w; c->h = h; c->size_d = size_d; c->ngpu = ngpu; c->flags = flags;
    c->n = (size_t)w * h;
    const size_t n = c->n;
    c->sh.resize(ngpu);
    std::vector<int> devs(ngpu);
    std::vector<ncclComm_t> comms(ngpu, nullptr);
    for (int g = 0; g < ngpu; ++g) devs[g] = g;
    // the communicators go to their owners before anything else can fail: the Shard destructors release them
    {
        ncclResult_t r = ncclCommInitAll(comms.data(), ngpu, devs.data());
        for (int g = 0; g < ngpu; ++g) { c->sh[g].dev = g; c->sh[g].comm = comms[g]; }
        if (r != ncclSuccess) return fail(SMX_E_HIP, "%s -> %s (line %d)", "ncclCommInitAll", ncclGetErrorString(r), __LINE__);
    }
    int max_slices = 1;
    for (int g = 0; g < ngpu; ++g) {
        c->sh[g].s0 = (int)((int64_t)g * size_d / ngpu);
        c->sh[g].s1 = (int)((int64_t)(g + 1) * size_d / ngpu);
        if (c->sh[g].s1 - c->sh[g].s0 > max_slices) max_slices = c->sh[g].s1 - c->sh[g].s0;
    }
    // cap the slices in flight at ~4 GiB of workspace per device
    int in_flight = max_slices;
    while (in_flight > 1 && 2 * smx_agg_workspace_bytes(w, h, in_flight) > ((size_t)4 << 30)) in_flight = (in_flight + 1) / 2;
    for (int g = 0; g < ngpu; ++g) {
        Shard& s = c->sh[g];
        RC_HIP(hipSetDevice(g));
        RC_HIP(hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking));
        RC_HIP(hipStreamCreateWithFlags(&s.cst, hipStreamNonBlocking));
        for (hipEvent_t* e : {&s.ev_view[0], &s.ev_view[1], &s.ev_comm}) RC_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
        s.ws_bytes = 2 * smx_agg_workspace_bytes(w, h, in_flight);
        RC_HIP(hipMalloc((void**)&s.L, n));
        RC_HIP(hipMalloc((void**)&s.R, n));
        RC_HIP(hipMalloc((void**)&s.mean, 2 * n));
        RC_HIP(hipMalloc((void**)&s.keys, 2 * n * sizeof(int64_t)));
        RC_HIP(hipMalloc(&s.ws, s.ws_bytes));
    }
    RC_HIP(hipSetDevice(0));
    const size_t fb = n * sizeof(float);
    RC_HIP(hipMalloc((void**)&c->best, 2 * fb));
    RC_HIP(hipMalloc((void**)&c->map, 2 * fb));
    RC_HIP(hipMalloc((void**)&c->occ, fb));
    RC_HIP(hipMalloc((void**)&c->fil, fb));
    c->meet.n = ngpu;
    for (int g = 0; g < ngpu; ++g) {
        c->workers.emplace_back(new (std::nothrow) Worker);
        if (!c->workers.back()) return fail(SMX_E_HIP, "%s: %s (line %d)", "smx_sharded_create", "out of host memory", __LINE__);
        c->workers.back()->start(g);
    }
    guard.c = nullptr;
    *out = c;
    return SMX_OK;
}

int smx_sharded_destroy(smx_sharded_ctx* c) {
    if (!c) return SMX_OK;
    int dev = -1;
    (void)hipGetDevice(&dev);
    delete c;
    if (dev >= 0) (void)hipSetDevice(dev);
    return SMX_OK;
}

// What device g does for one pair, on its own host thread: upload, aggregation + running WTA of its slice range, the
// exchange step on the exchange stream, (device 0: decode + LR check + filling + download), synchronise, status.
static int shard_pair(smx_sharded_ctx* c, int g, const uint8_t* gray_l, const uint8_t* gray_r, int dminl, int dminr,
                      const smx_pair_out* out) {
    const smx_params* p = &c->p;
    const int w = c->w, h = c->h;
    const size_t n = c->n;
    const bool per_view = (c->flags & SMX_SHARDED_OVERLAP_VIEWS) != 0;
    const bool all_ranks = (c->flags & SMX_SHARDED_ALLREDUCE) != 0;
    Shard& s = c->sh[g];
    auto enqueue = [&]() -> int {
        RC_HIP(hipSetDevice(g));
        RC_HIP(hipMemcpyAsync(s.L, gray_l, n, hipMemcpyHostToDevice, s.st));
        RC_HIP(hipMemcpyAsync(s.R, gray_r, n, hipMemcpyHostToDevice, s.st));
        RC_SMX(smx_dev_init_keys(s.keys, (int64_t)(2 * n), s.st));
        if (per_view) {
            // one launch per view: the exchange of the left keys runs under the aggregation of the right volume
            RC_SMX(smx_dev_aggregate_wta(p, s.L, s.R, nullptr, w, h, dminl, s.s0, s.s1, s.keys, s.mean, nullptr, s.ws,
                                         s.ws_bytes, s.st));
            RC_HIP(hipEventRecord(s.ev_view[0], s.st));
            RC_SMX(smx_dev_aggregate_wta(p, s.R, s.L, nullptr, w, h, dminr, s.s0, s.s1, s.keys + n, s.mean + n, nullptr, s.ws,
                                         s.ws_bytes, s.st));
            RC_HIP(hipEventRecord(s.ev_view[1], s.st));
        } else {
            RC_SMX(smx_dev_aggregate_wta_pair(p, s.L, s.R, w, h, dminl, dminr, s.s0, s.s1, s.keys, s.mean, nullptr, s.ws,
                                              s.ws_bytes, s.st));
            RC_HIP(hipEventRecord(s.ev_view[1], s.st));
        }
        return SMX_OK;
    };
    const int rc_a = enqueue();
    c->meet.arrive(rc_a == SMX_OK);
    if (c->meet.failed.load()) {
        (void)hipStreamSynchronize(s.st);
        return rc_a != SMX_OK ? rc_a : fail(SMX_E_HIP, "%s: %s (line %d)", "smx_sharded_run", "another rank failed before the exchange", __LINE__);
    }
    // the one exchange step, on the exchange stream (every rank calls its collective from its own thread: no group).
    // Only device 0 needs the reassembled maps (ncclReduce) unless the caller asked for them on every rank.
    const int nex = per_view ? 2 : 1;
    for (int x = 0; x < nex; ++x) {
        const int64_t cnt = per_view ? (int64_t)n : (int64_t)(2 * n);
        int64_t* off = s.keys + (per_view ? (size_t)x * n : 0);
        RC_HIP(hipStreamWaitEvent(s.cst, s.ev_view[per_view ? x : 1], 0));
        if (all_ranks) RC_SMX(smx_wta_allreduce(off, cnt, s.comm, s.cst));
        else RC_SMX(smx_wta_reduce(off, cnt, 0, s.comm, s.cst));
    }
    RC_HIP(hipEventRecord(s.ev_comm, s.cst));
    if (g == 0) {
        // decode + LR check + filling on device 0 (n-sized, microseconds): main.cu:112-118, 140-155
        RC_HIP(hipStreamWaitEvent(s.st, s.ev_comm, 0));
        RC_SMX(smx_dev_finish_pair(p, s.keys, w, h, dminl, dminr, dminl - 100, (float)dminl, c->best, c->map, c->occ, c->fil, s.st));
        const size_t fb = n * sizeof(float);
        struct { void* dst; const void* src; size_t b; } copies[] = {
            {out->best_l, c->best, fb}, {out->best_r, c->best + n, fb}, {out->dmap_l, c->map, fb}, {out->dmap_r, c->map + n, fb},
            {out->mean_l, s.mean, n}, {out->mean_r, s.mean + n, n}, {out->occlusion, c->occ, fb}, {out->filled, c->fil, fb},
        };
        for (auto& cp : copies)
            if (cp.dst) RC_HIP(hipMemcpyAsync(cp.dst, cp.src, cp.b, hipMemcpyDeviceToHost, s.st));
    }
    RC_HIP(hipStreamSynchronize(s.cst));
    RC_HIP(hipStreamSynchronize(s.st));
    RC_SMX(smx_dev_agg_status(s.ws));
    return SMX_OK;
}

int smx_sharded_run(smx_sharded_ctx* c, const uint8_t* gray_l, const uint8_t* gray_r, int dminl, int dminr,
                    const smx_pair_out* out) {
    if (!c || !gray_l || !gray_r || !out)
        return fail(SMX_E_ARG, "%s: %s (line %d)", "smx_sharded_run", "bad argument", __LINE__);
    if (out->cost_l || out->cost_r || out->agg_l || out->agg_r)
        return fail(SMX_E_ARG, "%s: %s (line %d)", "smx_sharded_run", "cost / agg outputs are not available in the sharded driver", __LINE__);
    c->meet.failed.store(0);
    for (int g = 0; g < c->ngpu; ++g)
        c->workers[g]->post([=] { return shard_pair(c, g, gray_l, gray_r, dminl, dminr, out); });
    int rc = SMX_OK;
    for (int g = 0; g < c->ngpu; ++g) {
        const int r = c->workers[g]->wait();
        if (r != SMX_OK && rc == SMX_OK) {
            rc = r;
            snprintf(g_msg, sizeof(g_msg), "device %d: %s", g, c->workers[g]->msg);
        }
    }
    return rc;
}

int smx_stereo_pair_sharded(const smx_params* p, const uint8_t* gray_l, const uint8_t* gray_r, int w, int h,
                            int size_d, int dminl, int dminr, int ngpu, const smx_pair_out* out) {
    smx_sharded_ctx* c = nullptr;
    int rc = smx_sharded_create(p, w, h, size_d, ngpu, 0, &c);
    if (rc) return rc;
    rc = smx_sharded_run(c, gray_l, gray_r, dminl, dminr, out);
    (void)smx_sharded_destroy(c);
    return rc;
}

}  // extern "C"
