// smx_rccl.cpp -- libsmx_rccl.so: the one exchange step of the D-sharded stereo path (RCCL over xGMI)
// and a single-process multi-GPU driver on top of the device-pointer C-ABI of libsmx_hip.so.
// Host code only (no kernels here); see include/smx_rccl.h.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

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

// everything one device owns for a sharded pair
struct Shard {
    int dev = -1;
    hipStream_t st = nullptr;
    ncclComm_t comm = nullptr;
    uint8_t *L = nullptr, *R = nullptr, *mean = nullptr;
    int64_t* keys = nullptr;
    void* ws = nullptr;
    size_t ws_bytes = 0;
    ~Shard() {
        if (dev < 0) return;
        (void)hipSetDevice(dev);
        if (comm) (void)ncclCommDestroy(comm);
        for (void* q : {(void*)L, (void*)R, (void*)mean, (void*)keys, ws})
            if (q) (void)hipFree(q);
        if (st) (void)hipStreamDestroy(st);
    }
};

}  // namespace

extern "C" {

int smx_wta_allreduce(int64_t* d_keys, int64_t n, void* nccl_comm, void* stream) {
    if (!d_keys || n <= 0 || !nccl_comm) return fail(SMX_E_ARG, "%s: %s (line %d)", "smx_wta_allreduce", "bad argument", __LINE__);
    RC_NCCL(ncclAllReduce(d_keys, d_keys, (size_t)n, ncclInt64, ncclMin, (ncclComm_t)nccl_comm, (hipStream_t)stream));
    return SMX_OK;
}

int smx_stereo_pair_sharded(const smx_params* p, const uint8_t* gray_l, const uint8_t* gray_r, int w, int h,
                            int size_d, int dminl, int dminr, int ngpu, const smx_pair_out* out) {
    if (!p || !gray_l || !gray_r || !out || w < 2 || h < 1 || size_d < 1 || ngpu < 1)
        return fail(SMX_E_ARG, "%s: %s (line %d)", "smx_stereo_pair_sharded", "bad argument", __LINE__);
    if (out->cost_l || out->cost_r || out->agg_l || out->agg_r)
        return fail(SMX_E_ARG, "%s: %s (line %d)", "smx_stereo_pair_sharded", "cost / agg outputs are not available in the sharded driver", __LINE__);
    int ndev = 0;
    RC_HIP(hipGetDeviceCount(&ndev));
    if (ngpu > ndev) return fail(SMX_E_ARG, "%s: %s (line %d)", "smx_stereo_pair_sharded", "more shards than devices", __LINE__);
    const size_t n = (size_t)w * h;
    std::vector<Shard> sh(ngpu);
    std::vector<int> devs(ngpu);
    std::vector<ncclComm_t> comms(ngpu);
    for (int g = 0; g < ngpu; ++g) devs[g] = g;
    RC_NCCL(ncclCommInitAll(comms.data(), ngpu, devs.data()));
    int max_slices = 0;
    for (int g = 0; g < ngpu; ++g) {
        const int s0 = (int)((int64_t)g * size_d / ngpu), s1 = (int)((int64_t)(g + 1) * size_d / ngpu);
        if (s1 - s0 > max_slices) max_slices = s1 - s0;
    }
    if (max_slices < 1) max_slices = 1;
    // cap the slices in flight at ~4 GiB of workspace per device
    int in_flight = max_slices;
    while (in_flight > 1 && 2 * smx_agg_workspace_bytes(w, h, in_flight) > ((size_t)4 << 30)) in_flight = (in_flight + 1) / 2;
    for (int g = 0; g < ngpu; ++g) {
        Shard& s = sh[g];
        s.dev = g;
        s.comm = comms[g];
        RC_HIP(hipSetDevice(g));
        RC_HIP(hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking));
        s.ws_bytes = 2 * smx_agg_workspace_bytes(w, h, in_flight);
        RC_HIP(hipMalloc((void**)&s.L, n));
        RC_HIP(hipMalloc((void**)&s.R, n));
        RC_HIP(hipMalloc((void**)&s.mean, 2 * n));
        RC_HIP(hipMalloc((void**)&s.keys, 2 * n * sizeof(int64_t)));
        RC_HIP(hipMalloc(&s.ws, s.ws_bytes));
        RC_HIP(hipMemcpyAsync(s.L, gray_l, n, hipMemcpyHostToDevice, s.st));
        RC_HIP(hipMemcpyAsync(s.R, gray_r, n, hipMemcpyHostToDevice, s.st));
    }
    // local aggregation + running WTA of every device's slice range (asynchronous, all devices busy)
    for (int g = 0; g < ngpu; ++g) {
        Shard& s = sh[g];
        const int s0 = (int)((int64_t)g * size_d / ngpu), s1 = (int)((int64_t)(g + 1) * size_d / ngpu);
        RC_HIP(hipSetDevice(g));
        RC_SMX(smx_dev_init_keys(s.keys, (int64_t)(2 * n), s.st));
        RC_SMX(smx_dev_aggregate_wta_pair(p, s.L, s.R, w, h, dminl, dminr, s0, s1, s.keys, s.mean, nullptr, s.ws,
                                          s.ws_bytes, s.st));
    }
    // the one exchange step: grouped because one thread drives all ranks
    RC_NCCL(ncclGroupStart());
    for (int g = 0; g < ngpu; ++g) {
        RC_HIP(hipSetDevice(g));
        RC_SMX(smx_wta_allreduce(sh[g].keys, (int64_t)(2 * n), sh[g].comm, sh[g].st));
    }
    RC_NCCL(ncclGroupEnd());
    // decode + LR check + filling on device 0 (n-sized, microseconds)
    RC_HIP(hipSetDevice(0));
    hipStream_t st = sh[0].st;
    float *best = nullptr, *map = nullptr, *occ = nullptr, *fil = nullptr;
    const size_t fb = n * sizeof(float);
    RC_HIP(hipMalloc((void**)&best, 2 * fb));
    RC_HIP(hipMalloc((void**)&map, 2 * fb));
    RC_HIP(hipMalloc((void**)&occ, fb));
    RC_HIP(hipMalloc((void**)&fil, fb));
    int rc = SMX_OK;
    do {
        if ((rc = smx_dev_init_wta(best, map, (int64_t)(2 * n), st))) break;
        if ((rc = smx_dev_apply_keys(sh[0].keys, (int64_t)n, dminl, best, map, st))) break;
        if ((rc = smx_dev_apply_keys(sh[0].keys + n, (int64_t)n, dminr, best + n, map + n, st))) break;
        if (hipMemcpyAsync(occ, map, fb, hipMemcpyDeviceToDevice, st) != hipSuccess) { rc = SMX_E_HIP; break; }
        if ((rc = smx_dev_detect_occlusion(p, occ, map + n, dminl - 100, w, h, st))) break;      // main.cu:149
        if (hipMemcpyAsync(fil, occ, fb, hipMemcpyDeviceToDevice, st) != hipSuccess) { rc = SMX_E_HIP; break; }
        if ((rc = smx_dev_fill_occlusion(fil, w, h, (float)dminl, st))) break;                   // main.cu:154
        for (int g = 0; g < ngpu && rc == SMX_OK; ++g) {
            if (hipSetDevice(g) != hipSuccess || hipStreamSynchronize(sh[g].st) != hipSuccess) { rc = SMX_E_HIP; break; }
            rc = smx_dev_agg_status(sh[g].ws);
        }
        if (rc) break;
        (void)hipSetDevice(0);
        struct { void* dst; const void* src; size_t b; } copies[] = {
            {out->best_l, best, fb}, {out->best_r, best + n, fb}, {out->dmap_l, map, fb}, {out->dmap_r, map + n, fb},
            {out->mean_l, sh[0].mean, n}, {out->mean_r, sh[0].mean + n, n}, {out->occlusion, occ, fb}, {out->filled, fil, fb},
        };
        for (auto& c : copies)
            if (c.dst && hipMemcpy(c.dst, c.src, c.b, hipMemcpyDeviceToHost) != hipSuccess) { rc = SMX_E_HIP; break; }
    } while (0);
    (void)hipSetDevice(0);
    for (float* q : {best, map, occ, fil}) (void)hipFree(q);
    if (rc) return fail(rc, "%s: %s (line %d)", "smx_stereo_pair_sharded", smx_last_error(), __LINE__);
    return SMX_OK;
}

}  // extern "C"
