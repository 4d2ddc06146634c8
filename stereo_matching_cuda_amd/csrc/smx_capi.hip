// smx_capi.hip -- the C-ABI of include/smx.h: argument checks, workspace carving, stage
// orchestration, and the host-pointer wrappers that mirror the reference's per-stage functions.
#include <string.h>

#include <new>
#include <string>
#include <vector>

#include "smx_launch.h"

namespace smx {

static thread_local std::string g_err;
static thread_local int g_timing = 0;        // 0 off, 1 the last call, 2 cumulative over calls
static thread_local int g_launches = 0;
static thread_local int g_in_ctx = 0;        // inside a persistent-context entry: the entry marks ST_BEGIN itself, nested calls do not
static thread_local int g_marks_dropped = 0;
// Stage timing of the calling thread (smx_set_timing / smx_stage_times): a list of (stage, event) marks of the
// last timed call; the time between two consecutive marks belongs to the stage of the later one.  Events are taken
// from a pool per device (an event records on the device that was current when it was created).
struct StageTimer {
    struct Mark { int stage; hipEvent_t ev; };
    std::vector<Mark> marks;
    std::vector<std::vector<hipEvent_t>> pool;   // [device] -> events
    std::vector<size_t> used;                    // [device] -> events handed out for the current call
    void begin() { marks.clear(); for (auto& u : used) u = 0; }
    hipEvent_t get() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return nullptr;
        if ((size_t)dev >= pool.size()) { pool.resize(dev + 1); used.resize(dev + 1, 0); }
        if (used[dev] == pool[dev].size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            pool[dev].push_back(e);
        }
        return pool[dev][used[dev]++];
    }
};
static thread_local StageTimer g_timer;
#ifndef SMX_DEFAULT_AGG_PATH
#define SMX_DEFAULT_AGG_PATH 0
#endif
// Aggregation path of the calling thread's smx_dev_* / host-pointer calls (smx_set_agg_path); a persistent
// context carries its own (smx_ctx_set_agg_path).  0 auto, 1 multi-kernel, 2 fused (walker chosen by the call),
// 3 fused with the ring walker (smx_agg_v4.hip), 4 fused FAST (not bit-exact), 5 fused with the comb walker
// (smx_agg_v5.hip; an error where it does not apply)
static thread_local int g_agg_path = SMX_DEFAULT_AGG_PATH;
static thread_local int g_last_path = 0;
static thread_local int g_max_chunk = 0;     // smx_set_max_slices_per_launch
static thread_local int g_keys_fresh = 0;    // smx_set_keys_fresh
static thread_local AggInfo g_last_info;     // smx_last_agg_chunk

// smx_agg_v4.hip (host orchestration of both fused walkers)
bool v4_supported(const smx_params* p);
size_t v4_workspace_bytes(int w, int h, int nslices);
int aggregate_v4(const smx_params* p, int nviews, const uint8_t* const* d_guide,
                 const uint8_t* const* d_other, const float* const* d_cost, int w, int h,
                 const int* dmin, int s_begin, int s_end, int64_t* const* d_keys,
                 uint8_t* const* d_mean_u8, float* const* d_agg, void* d_ws, size_t ws_bytes,
                 hipStream_t st, const AggOpts& opt, AggInfo* info);
size_t v5_fix_bytes(int w, int h, int nviews);
int v4_read_status(const void* d_ws, unsigned* out, int nwords);
void v4_geometry(int* ow, int* bh);
// smx_agg_v5.hip
bool v5_supported(const smx_params* p);
void v5_geometry(int* ow, int* bh);
void v5_slots(int h, int K, int* bands, int* q_last, int* period);

// the fused aggregation; reports the path that ran: 2 = ring walker, 4 = FAST, 5 = comb walker
static int aggregate_fused(int path, const smx_params* p, int nviews, const uint8_t* const* d_guide,
                           const uint8_t* const* d_other, const float* const* d_cost, int w, int h,
                           const int* dmin, int s_begin, int s_end, int64_t* const* d_keys,
                           uint8_t* const* d_mean_u8, float* const* d_agg, void* d_ws, size_t ws_bytes,
                           hipStream_t st, int* launches) {
    if (g_keys_fresh && s_end <= s_begin) {
        // nothing to aggregate: the promise "the call presets the keys" still holds
        for (int v = 0; v < nviews; ++v) { int rk = launch_init_keys(d_keys[v], (int64_t)w * h, st); if (rk) return rk; }
    }
    AggOpts opt;
    opt.keys_fresh = g_keys_fresh != 0;
    opt.fast = path == 4;
    opt.walker = path == 3 ? 4 : path == 5 ? 5 : 0;
    opt.max_chunk = g_max_chunk;
    AggInfo info;
    int rc = aggregate_v4(p, nviews, d_guide, d_other, d_cost, w, h, dmin, s_begin, s_end, d_keys, d_mean_u8,
                          d_agg, d_ws, ws_bytes, st, opt, &info);
    if (rc) return rc;
    g_last_info = info;
    if (launches) *launches = info.launches;
    g_last_path = path == 4 ? 4 : (info.walker_used == 5 ? 5 : 2);
    return SMX_OK;
}

void stage_mark(int stage, hipStream_t st) {
    if (!g_timing) return;
    if (stage == ST_BEGIN && g_in_ctx > 1) return;       // (a device-pointer call nested in a context entry: one call, one ST_BEGIN)
    if (stage == ST_BEGIN && g_timing == 1) { g_timer.begin(); g_marks_dropped = 0; }
    if (g_timer.marks.size() >= 32768) { ++g_marks_dropped; return; }
    hipEvent_t e = g_timer.get();
    if (!e || hipEventRecord(e, st) != hipSuccess) return;
    g_timer.marks.push_back({stage, e});
}

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// RAII device allocation for the host-pointer wrappers.
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T> T* as() { return (T*)p; }
};

constexpr size_t WS_ALIGN = 256;

static size_t plane_bytes(int w, int h) { return align_up((size_t)w * h * sizeof(float), WS_ALIGN); }

}  // namespace smx

using namespace smx;

extern "C" {

void smx_default_params(smx_params* p) {
    if (!p) return;
    p->r_w = 0.299; p->g_w = 0.587; p->b_w = 0.0721;
    p->alpha = 0.9; p->th_color = 7; p->th_grad = 2;
    p->radius = 9; p->eps = 6.5025; p->d_lr = 0;
}

const char* smx_last_error(void) { return g_err.c_str(); }

const char* smx_version(void) { return "smx-hip gfx950 0.9 (comb walker: items pipelined across tickets, q scratch in row pairs, cost volumes on the comb walker; WTA winner in the float domain; guidance in three launches; two 512-thread workgroups per CU)"; }

int smx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int64_t smx_pack_key(float cost, uint32_t slice) { return pack_key(cost, slice); }
void smx_unpack_key(int64_t key, float* cost, uint32_t* slice) {
    float c; uint32_t s;
    unpack_key(key, &c, &s);
    if (cost) *cost = c;
    if (slice) *slice = s;
}

int smx_set_timing(int mode) {
    if (mode < 0 || mode > 2) return fail(SMX_E_ARG, "smx_set_timing: mode must be 0, 1 or 2");
    g_timing = mode;
    g_timer.begin();
    g_marks_dropped = 0;
    return SMX_OK;
}

int smx_stage_times(smx_stage_ms* out) {
    SMX_ARG(out);
    memset(out, 0, sizeof(*out));
    if (g_timer.marks.size() < 2) return fail(SMX_E_ARG, "smx_stage_times: no timed call recorded (smx_set_timing(1) first)");
    SMX_HIP(hipEventSynchronize(g_timer.marks.back().ev));
    float acc[ST_COUNT] = {0};
    int calls = g_timer.marks.front().stage == ST_BEGIN ? 1 : 0;
    for (size_t i = 1; i < g_timer.marks.size(); ++i) {
        if (g_timer.marks[i].stage == ST_BEGIN) { ++calls; continue; }     // (the gap in front of a call is nobody's)
        float t = 0;
        SMX_HIP(hipEventElapsedTime(&t, g_timer.marks[i - 1].ev, g_timer.marks[i].ev));
        acc[g_timer.marks[i].stage] += t;
    }
    out->calls = calls;
    out->dropped = g_marks_dropped;
    out->upload = acc[ST_UPLOAD]; out->guidance = acc[ST_GUIDANCE]; out->aggregation = acc[ST_WALK];
    out->wta = acc[ST_WTA]; out->finish = acc[ST_FINISH]; out->download = acc[ST_DOWNLOAD];
    out->total = out->upload + out->guidance + out->aggregation + out->wta + out->finish + out->download;
    return SMX_OK;
}

int smx_last_agg_ms(float* ms, int* launches) {
    smx_stage_ms t;
    int rc = smx_stage_times(&t);
    if (rc) return rc;
    if (ms) *ms = t.guidance + t.aggregation + t.wta;
    if (launches) *launches = g_launches;
    return SMX_OK;
}

/* ------------------------------------------------------------------------------------------
 * device-pointer API
 * ---------------------------------------------------------------------------------------- */

int smx_dev_rgb_to_grayscale(const smx_params* p, const uint8_t* d_rgb, int64_t n, int channels,
                             uint8_t* d_gray, void* stream) {
    SMX_ARG(p && d_rgb && d_gray && n > 0 && channels >= 3);
    return launch_gray(p, d_rgb, n, channels, d_gray, (hipStream_t)stream);
}

int smx_dev_cost_volume(const smx_params* p, const uint8_t* d_i1, const uint8_t* d_i2, float* d_cost,
                        int w1, int w2, int h, int dmin, int s_begin, int s_end, void* stream) {
    SMX_ARG(p && d_i1 && d_i2 && d_cost);
    SMX_ARG(w1 >= 2 && h >= 1 && w1 == w2 && s_begin >= 0 && s_end >= s_begin);
    return launch_cost(p, d_i1, d_i2, d_cost, w1, h, dmin + s_begin, s_end - s_begin,
                       (hipStream_t)stream);
}

int smx_dev_integral(const float* d_in, float* d_out, int w, int h, int nplanes, void* stream) {
    SMX_ARG(d_in && d_out && w >= 1 && h >= 1 && nplanes >= 0);
    return launch_integral(0, d_in, nullptr, d_out, nullptr, w, h, nplanes, (hipStream_t)stream);
}

size_t smx_agg_workspace_bytes(int w, int h, int nslices) {
    if (w < 1 || h < 1 || nslices < 1) return 0;
    // v1 path: guidance im, mean_im, cinv, S_im, S_sq ; per slice in flight: cost, T0, T1, A, B
    const size_t v1 = plane_bytes(w, h) * (5 + 5 * (size_t)nslices) + 2 * WS_ALIGN;
    const size_t f = v4_workspace_bytes(w, h, nslices);
    return v1 > f ? v1 : f;
}

size_t smx_agg_workspace_bytes_for(const smx_params* p, int w, int h, int nslices) {
    if (!p || w < 1 || h < 1 || nslices < 1) return 0;
    // radius <= 9 runs a fused walker (unless the multi-kernel path is forced: smx_set_agg_path(1) callers size with
    // smx_agg_workspace_bytes): image / guidance planes, per slice ONE q plane + the hand-off records
    if (v4_supported(p)) return v4_workspace_bytes(w, h, nslices);
    return smx_agg_workspace_bytes(w, h, nslices);
}

int smx_set_agg_path(int path) {
    if (path < 0 || path > 5) return fail(SMX_E_ARG, "smx_set_agg_path: path must be 0 .. 5");
    g_agg_path = path;
    return SMX_OK;
}

int smx_last_agg_path(void) { return g_last_path; }

int smx_set_keys_fresh(int on) {
    g_keys_fresh = on ? 1 : 0;
    return SMX_OK;
}

int smx_set_max_slices_per_launch(int n) {
    if (n < 0) return fail(SMX_E_ARG, "smx_set_max_slices_per_launch: n must be >= 0 (0 = as many as the workspace holds)");
    g_max_chunk = n;
    return SMX_OK;
}

int smx_last_agg_chunk(int* slices_per_launch, int* walker_launches) {
    if (slices_per_launch) *slices_per_launch = g_last_info.chunk;
    if (walker_launches) *walker_launches = g_last_info.walker_launches;
    return SMX_OK;
}

// (dev / test hook, not in smx.h: the size of the region the comb walker addresses through one 32-bit-offset descriptor)
__attribute__((visibility("default"))) int smx_debug_v5_fix_bytes(int w, int h, int nviews, uint64_t* bytes) {
    SMX_ARG(bytes && w >= 2 && h >= 1 && (nviews == 1 || nviews == 2));
    *bytes = (uint64_t)v5_fix_bytes(w, h, nviews);
    return SMX_OK;
}

// (dev / test hook, not in smx.h: WtaRun, the float-domain winner of a run of ascending slices as the WTA kernels form it)
__attribute__((visibility("default"))) int smx_debug_wta_run(const float* q, int n, uint32_t slice0, int64_t* key) {
    SMX_ARG(q && key && n >= 0);
    WtaRun r;
    for (int i = 0; i < n; ++i) r.step(q[i], slice0 + (uint32_t)i);
    *key = r.key();
    return SMX_OK;
}

// (dev / test hook, not in smx.h: the comb walker's slot geometry for an image of h rows in K strips -- bands per item, the
// last stage-2 slot, the period between the starts of two items of a workgroup -- for tools/v5_protocol_sim.py)
__attribute__((visibility("default"))) int smx_debug_v5_period(int h, int K, int* bands, int* q_last, int* period) {
    SMX_ARG(h >= 1 && K >= 1);
    int b = 0, q = 0, pd = 0;
    v5_slots(h, K, &b, &q, &pd);
    if (bands) *bands = b;
    if (q_last) *q_last = q;
    if (period) *period = pd;
    return SMX_OK;
}

int smx_agg_geometry(int radius, int* strip_cols, int* band_rows, int* tile_cols) {
    int ow = 0, bh = 0;
    v4_geometry(&ow, &bh);
    smx_params p;
    smx_default_params(&p);
    p.radius = radius;
    if (g_agg_path != 3 && g_agg_path != 4 && v5_supported(&p)) v5_geometry(&ow, &bh);   // the comb walker
    if (strip_cols) *strip_cols = ow;
    if (band_rows) *band_rows = bh;
    if (tile_cols) *tile_cols = ow + 2 * radius + 1;
    return SMX_OK;
}

int smx_dev_agg_status(const void* d_workspace) {
    SMX_ARG(d_workspace);
    unsigned st = 0;
    int rc = v4_read_status(d_workspace, &st, 1);
    if (rc) return rc;
    if (st != 0)
        return fail(SMX_E_HIP, "fused aggregation: hand-off wait of work item %u timed out (results invalid)",
                    st - 1);
    return SMX_OK;
}

int smx_dev_agg_fallback(const void* d_workspace, int* ring_walker_reran) {
    SMX_ARG(d_workspace && ring_walker_reran);
    unsigned st[2] = {0, 0};
    int rc = v4_read_status(d_workspace, st, 2);
    if (rc) return rc;
    *ring_walker_reran = st[1] != 0;
    return SMX_OK;
}

// The kernels are launched on the CURRENT device: a workspace that lives on another one is a caller bug
// that would otherwise fault on the GPU.
static int check_same_device(const void* d_ws, const char* who) {
    hipPointerAttribute_t a;
    int dev = -1;
    if (hipPointerGetAttributes(&a, d_ws) != hipSuccess) {
        (void)hipGetLastError();
        return SMX_OK;   // not a pointer the runtime knows (e.g. a sub-allocation it cannot resolve): let it through
    }
    SMX_HIP(hipGetDevice(&dev));
    if (a.type == hipMemoryTypeDevice && a.device != dev)
        return fail(SMX_E_ARG, "%s: workspace lives on device %d but the current device is %d "
                               "(hipSetDevice to the workspace's device before the call)", who, a.device, dev);
    return SMX_OK;
}

int smx_dev_init_keys(int64_t* d_keys, int64_t n, void* stream) {
    SMX_ARG(d_keys && n > 0);
    return launch_init_keys(d_keys, n, (hipStream_t)stream);
}

int smx_dev_init_wta(float* d_best, float* d_dmap, int64_t n, void* stream) {
    SMX_ARG(d_best && d_dmap && n > 0);
    return launch_init_wta(d_best, d_dmap, n, (hipStream_t)stream);
}

int smx_dev_apply_keys(const int64_t* d_keys, int64_t n, int dmin, float* d_best, float* d_dmap,
                       void* stream) {
    SMX_ARG(d_keys && d_best && d_dmap && n > 0);
    return launch_apply_keys(d_keys, n, dmin, d_best, d_dmap, (hipStream_t)stream);
}

int smx_dev_detect_occlusion(const smx_params* p, float* d_dL, const float* d_dR, int dOcclusion,
                             int w, int h, void* stream) {
    SMX_ARG(p && d_dL && d_dR && w >= 1 && h >= 1);
    return launch_detect_occlusion(p, d_dL, d_dR, dOcclusion, w, h, (hipStream_t)stream);
}

int smx_dev_fill_occlusion(float* d_disp, int w, int h, float vMin, void* stream) {
    SMX_ARG(d_disp && w >= 1 && h >= 1);
    return launch_fill_occlusion(d_disp, d_disp, w, h, vMin, (hipStream_t)stream);
}

// main.cu:112-155 behind the aggregation, both views, in three launches (presets + winning slices + copy of
// the left map; LR check; filling out of place) instead of the seven of the per-call sequence
int smx_dev_finish_pair(const smx_params* p, const int64_t* d_keys, int w, int h, int dminl, int dminr,
                        int dOcclusion, float vMin, float* d_best, float* d_dmap, float* d_occlusion,
                        float* d_filled, void* stream) {
    SMX_ARG(p && d_keys && d_best && d_dmap && d_occlusion && d_filled && w >= 1 && h >= 1);
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)w * h;
    int rc;
    // one launch (a row per workgroup) where the row fits the LDS three times, else the three kernels
    if (finish_pair_row_supported(w) && d_filled != d_occlusion) {
        rc = launch_finish_pair_row(p, d_keys, w, h, dminl, dminr, dOcclusion, vMin, d_best, d_dmap, d_occlusion,
                                    d_filled, st);
    } else {
        if ((rc = launch_finish_keys(d_keys, n, dminl, dminr, d_best, d_dmap, d_occlusion, st))) return rc;
        if ((rc = launch_detect_occlusion(p, d_occlusion, d_dmap + n, dOcclusion, w, h, st))) return rc;
        rc = launch_fill_occlusion(d_occlusion, d_filled, w, h, vMin, st);
    }
    stage_mark(ST_FINISH, st);
    return rc;
}

int smx_dev_aggregate_wta(const smx_params* p, const uint8_t* d_guide, const uint8_t* d_other,
                          const float* d_cost, int w, int h, int dmin, int s_begin, int s_end,
                          int64_t* d_keys, uint8_t* d_mean_u8, float* d_agg, void* d_workspace,
                          size_t workspace_bytes, void* stream) {
    SMX_ARG(p && d_guide && d_keys && d_workspace);
    SMX_ARG(d_cost || d_other);
    SMX_ARG(w >= 2 && h >= 1 && s_begin >= 0 && s_end >= s_begin && p->radius >= 0);
    { int rcd = check_same_device(d_workspace, "smx_dev_aggregate_wta"); if (rcd) return rcd; }
    hipStream_t st = (hipStream_t)stream;
    stage_mark(ST_BEGIN, st);
    // fused path: radius <= 9; cost built on the fly or read from d_cost
    const bool can_fuse = v4_supported(p);
    if (g_agg_path >= 2 && !can_fuse)
        return fail(SMX_E_ARG, "smx_dev_aggregate_wta: fused path forced but radius > 9");
    if (can_fuse && g_agg_path != 1) {
        g_launches = 0;
        int rc2 = aggregate_fused(g_agg_path, p, 1, &d_guide, &d_other, &d_cost, w, h, &dmin, s_begin, s_end, &d_keys,
                               &d_mean_u8, &d_agg, d_workspace, workspace_bytes, st, &g_launches);
        if (rc2) return rc2;
        return SMX_OK;
    }
    g_last_path = 1;
    if (g_keys_fresh) { int rk = launch_init_keys(d_keys, (int64_t)w * h, st); if (rk) return rk; }
    const size_t pb = plane_bytes(w, h);
    const int64_t n = (int64_t)w * h;
    char* base = (char*)align_up((size_t)d_workspace, WS_ALIGN);
    size_t avail = workspace_bytes > (size_t)(base - (char*)d_workspace) + WS_ALIGN
                       ? workspace_bytes - (size_t)(base - (char*)d_workspace) - WS_ALIGN : 0;
    if (avail >= pb * 10) {
        // first 256 B: status word of the call (smx_dev_agg_status); this path cannot time out
        SMX_HIP(hipMemsetAsync(base, 0, WS_ALIGN, st));
        base += WS_ALIGN;
    }
    if (workspace_bytes < WS_ALIGN || avail < pb * 10)
        return fail(SMX_E_WS, "smx_dev_aggregate_wta: workspace %zu B < %zu B needed for one slice",
                    workspace_bytes, smx_agg_workspace_bytes(w, h, 1));
    const int per_slice = d_cost ? 4 : 5;
    const int total = s_end - s_begin;
    const size_t slice_b = (size_t)n * sizeof(float);
    // largest chunk c with 5 guidance planes + per_slice volumes of c slices inside `avail`
    size_t c_fit = (avail - 5 * pb) / per_slice / slice_b;
    while (c_fit > 1 && 5 * pb + per_slice * align_up(c_fit * slice_b, WS_ALIGN) > avail) --c_fit;
    int chunk = c_fit > (size_t)total ? total : (int)c_fit;
    if (chunk < 1) chunk = 1;
    float* im = (float*)(base + 0 * pb);
    float* mean_im = (float*)(base + 1 * pb);
    float* cinv = (float*)(base + 2 * pb);
    float* g0 = (float*)(base + 3 * pb);
    float* g1 = (float*)(base + 4 * pb);
    char* cb = base + 5 * pb;
    // chunk volumes are packed with plane stride n floats (kernels index planes as z*n)
    const size_t vol = align_up((size_t)chunk * slice_b, WS_ALIGN);
    float* T0 = (float*)(cb + 0 * vol);
    float* T1 = (float*)(cb + 1 * vol);
    float* A = (float*)(cb + 2 * vol);
    float* B = (float*)(cb + 3 * vol);
    float* C = d_cost ? nullptr : (float*)(cb + 4 * vol);

    int rc;
    g_launches = 0;
    // guidance statistics (guidedFilter.cu:58-123)
    if ((rc = launch_guid_prep(d_guide, im, g1, n, st))) return rc;
    if ((rc = launch_integral(2, im, g1, g0, g1, w, h, 1, st))) return rc;
    if ((rc = launch_guid_finish(p, g0, g1, mean_im, cinv, d_mean_u8, w, h, st))) return rc;
    g_launches += 4;
    stage_mark(ST_GUIDANCE, st);
    // slice loop (guidedFilter.cu:171-238), `chunk` slices per pass
    for (int s0 = s_begin; s0 < s_end; s0 += chunk) {
        const int cnt = (s_end - s0) < chunk ? (s_end - s0) : chunk;
        const float* cost = d_cost ? d_cost + (int64_t)(s0 - s_begin) * n : C;
        if (!d_cost) {
            if ((rc = launch_cost(p, d_guide, d_other, C, w, h, dmin + s0, cnt, st))) return rc;
            ++g_launches;
        }
        if ((rc = launch_integral(1, cost, im, T0, T1, w, h, cnt, st))) return rc;
        if ((rc = launch_ab(p, T0, T1, mean_im, cinv, A, B, w, h, cnt, st))) return rc;
        if ((rc = launch_integral(2, A, B, A, B, w, h, cnt, st))) return rc;
        float* agg = d_agg ? d_agg + (int64_t)(s0 - s_begin) * n : nullptr;
        if ((rc = launch_q_wta(p, A, B, im, d_keys, agg, w, h, cnt, s0, st))) return rc;
        g_launches += 6;
        stage_mark(ST_WALK, st);      // (this path folds the WTA into its last pass)
    }
    return SMX_OK;
}

static int aggregate_pair(const char* who, const smx_params* p, const uint8_t* d_left, const uint8_t* d_right,
                          const float* d_cost_l, const float* d_cost_r, int w, int h, int dminl, int dminr, int s_begin,
                          int s_end, int64_t* d_keys, uint8_t* d_mean_u8, float* d_agg, void* d_workspace,
                          size_t workspace_bytes, void* stream) {
    SMX_ARG(p && d_left && d_right && d_keys && d_workspace);
    SMX_ARG(w >= 2 && h >= 1 && s_begin >= 0 && s_end >= s_begin && p->radius >= 0);
    SMX_ARG((d_cost_l != nullptr) == (d_cost_r != nullptr));
    { int rcd = check_same_device(d_workspace, who); if (rcd) return rcd; }
    hipStream_t st = (hipStream_t)stream;
    stage_mark(ST_BEGIN, st);
    const int64_t n = (int64_t)w * h;
    const int64_t vol = n * (s_end - s_begin);
    if (v4_supported(p) && g_agg_path != 1) {
        const uint8_t* guide[2] = {d_left, d_right};
        const uint8_t* other[2] = {d_right, d_left};
        const float* cost[2] = {d_cost_l, d_cost_r};
        const int dmin[2] = {dminl, dminr};
        int64_t* keys[2] = {d_keys, d_keys + n};
        uint8_t* mean[2] = {d_mean_u8, d_mean_u8 ? d_mean_u8 + n : nullptr};
        float* agg[2] = {d_agg, d_agg ? d_agg + vol : nullptr};
        g_launches = 0;
        int rc2 = aggregate_fused(g_agg_path, p, 2, guide, other, d_cost_l ? cost : nullptr, w, h, dmin, s_begin, s_end, keys,
                               d_mean_u8 ? mean : nullptr, d_agg ? agg : nullptr, d_workspace,
                               workspace_bytes, st, &g_launches);
        if (rc2) return rc2;
        return SMX_OK;
    }
    if (g_agg_path >= 2) return fail(SMX_E_ARG, "%s: fused path forced but radius > 9", who);
    int rc = smx_dev_aggregate_wta(p, d_left, d_right, d_cost_l, w, h, dminl, s_begin, s_end, d_keys,
                                   d_mean_u8, d_agg, d_workspace, workspace_bytes, stream);
    if (rc) return rc;
    return smx_dev_aggregate_wta(p, d_right, d_left, d_cost_r, w, h, dminr, s_begin, s_end, d_keys + n,
                                 d_mean_u8 ? d_mean_u8 + n : nullptr, d_agg ? d_agg + vol : nullptr,
                                 d_workspace, workspace_bytes, stream);
}

int smx_dev_aggregate_wta_pair(const smx_params* p, const uint8_t* d_left, const uint8_t* d_right,
                               int w, int h, int dminl, int dminr, int s_begin, int s_end,
                               int64_t* d_keys, uint8_t* d_mean_u8, float* d_agg, void* d_workspace,
                               size_t workspace_bytes, void* stream) {
    return aggregate_pair("smx_dev_aggregate_wta_pair", p, d_left, d_right, nullptr, nullptr, w, h, dminl, dminr, s_begin, s_end,
                          d_keys, d_mean_u8, d_agg, d_workspace, workspace_bytes, stream);
}

int smx_dev_aggregate_wta_pair_cost(const smx_params* p, const uint8_t* d_left, const uint8_t* d_right,
                                    const float* d_cost_l, const float* d_cost_r, int w, int h, int dminl, int dminr,
                                    int s_begin, int s_end, int64_t* d_keys, uint8_t* d_mean_u8, float* d_agg,
                                    void* d_workspace, size_t workspace_bytes, void* stream) {
    SMX_ARG(d_cost_l && d_cost_r);
    return aggregate_pair("smx_dev_aggregate_wta_pair_cost", p, d_left, d_right, d_cost_l, d_cost_r, w, h, dminl, dminr, s_begin,
                          s_end, d_keys, d_mean_u8, d_agg, d_workspace, workspace_bytes, stream);
}

/* ------------------------------------------------------------------------------------------
 * host-pointer stage API (reference L2 wrappers: allocate, upload, run, download, free)
 * ---------------------------------------------------------------------------------------- */

int smx_rgb_to_grayscale(const smx_params* p, const uint8_t* h_rgb, int64_t n, int channels,
                         uint8_t* h_gray) {
    SMX_ARG(p && h_rgb && h_gray && n > 0 && channels >= 3);
    DevBuf rgb, gray;
    SMX_HIP(rgb.alloc((size_t)n * channels));
    SMX_HIP(gray.alloc((size_t)n));
    SMX_HIP(hipMemcpy(rgb.p, h_rgb, (size_t)n * channels, hipMemcpyHostToDevice));
    int rc = smx_dev_rgb_to_grayscale(p, rgb.as<uint8_t>(), n, channels, gray.as<uint8_t>(), nullptr);
    if (rc) return rc;
    SMX_HIP(hipDeviceSynchronize());
    SMX_HIP(hipMemcpy(h_gray, gray.p, (size_t)n, hipMemcpyDeviceToHost));
    return SMX_OK;
}

int smx_compute_cost(const smx_params* p, const uint8_t* i1, const uint8_t* i2, float* cost, int w1,
                     int w2, int h1, int h2, int size_d, int dmin) {
    SMX_ARG(p && i1 && i2 && cost && size_d >= 1);
    SMX_ARG(w1 >= 2 && w1 == w2 && h1 >= 1 && h1 == h2);
    const size_t n = (size_t)w1 * h1;
    DevBuf d1, d2, dc;
    SMX_HIP(d1.alloc(n));
    SMX_HIP(d2.alloc(n));
    SMX_HIP(dc.alloc(n * size_d * sizeof(float)));
    SMX_HIP(hipMemcpy(d1.p, i1, n, hipMemcpyHostToDevice));
    SMX_HIP(hipMemcpy(d2.p, i2, n, hipMemcpyHostToDevice));
    int rc = smx_dev_cost_volume(p, d1.as<uint8_t>(), d2.as<uint8_t>(), dc.as<float>(), w1, w2, h1,
                                 dmin, 0, size_d, nullptr);
    if (rc) return rc;
    SMX_HIP(hipDeviceSynchronize());
    SMX_HIP(hipMemcpy(cost, dc.p, n * size_d * sizeof(float), hipMemcpyDeviceToHost));
    return SMX_OK;
}

int smx_integral(const float* image, float* integral, int width, int height) {
    SMX_ARG(image && integral && width >= 1 && height >= 1);
    const size_t bytes = (size_t)width * height * sizeof(float);
    DevBuf d;
    SMX_HIP(d.alloc(bytes));
    SMX_HIP(hipMemcpy(d.p, image, bytes, hipMemcpyHostToDevice));
    int rc = smx_dev_integral(d.as<float>(), d.as<float>(), width, height, 1, nullptr);
    if (rc) return rc;
    SMX_HIP(hipDeviceSynchronize());
    SMX_HIP(hipMemcpy(integral, d.p, bytes, hipMemcpyDeviceToHost));
    return SMX_OK;
}

static size_t pick_ws_bytes(int w, int h, int size_d) {
    // keep at most ~2 GiB of slices in flight for the host-pointer wrappers
    const size_t one = smx_agg_workspace_bytes(w, h, 1);
    const size_t all = smx_agg_workspace_bytes(w, h, size_d);
    const size_t cap = (size_t)2 << 30;
    if (all <= cap) return all;
    return one > cap ? one : cap;
}

int smx_compute_guided_filter(const smx_params* p, const uint8_t* i, const float* cost,
                              float* filter_cost, float* disp_map, uint8_t* mean, float* agg, int w,
                              int h, int size_d, int dmin) {
    SMX_ARG(p && i && cost && filter_cost && disp_map && size_d >= 1 && w >= 2 && h >= 1);
    const size_t n = (size_t)w * h;
    const size_t ws_bytes = pick_ws_bytes(w, h, size_d);
    DevBuf dI, dC, dBest, dMap, dMean, dKeys, dAgg, ws;
    SMX_HIP(dI.alloc(n));
    SMX_HIP(dC.alloc(n * size_d * sizeof(float)));
    SMX_HIP(dBest.alloc(n * sizeof(float)));
    SMX_HIP(dMap.alloc(n * sizeof(float)));
    SMX_HIP(dMean.alloc(n));
    SMX_HIP(dKeys.alloc(n * sizeof(int64_t)));
    if (agg) SMX_HIP(dAgg.alloc(n * size_d * sizeof(float)));
    SMX_HIP(ws.alloc(ws_bytes));
    SMX_HIP(hipMemcpy(dI.p, i, n, hipMemcpyHostToDevice));
    SMX_HIP(hipMemcpy(dC.p, cost, n * size_d * sizeof(float), hipMemcpyHostToDevice));
    SMX_HIP(hipMemcpy(dBest.p, filter_cost, n * sizeof(float), hipMemcpyHostToDevice));
    SMX_HIP(hipMemcpy(dMap.p, disp_map, n * sizeof(float), hipMemcpyHostToDevice));
    int rc;
    if ((rc = smx_dev_init_keys(dKeys.as<int64_t>(), (int64_t)n, nullptr))) return rc;
    if ((rc = smx_dev_aggregate_wta(p, dI.as<uint8_t>(), nullptr, dC.as<float>(), w, h, dmin, 0,
                                    size_d, dKeys.as<int64_t>(), dMean.as<uint8_t>(),
                                    agg ? dAgg.as<float>() : nullptr, ws.p, ws_bytes, nullptr)))
        return rc;
    if ((rc = smx_dev_apply_keys(dKeys.as<int64_t>(), (int64_t)n, dmin, dBest.as<float>(),
                                 dMap.as<float>(), nullptr)))
        return rc;
    SMX_HIP(hipDeviceSynchronize());
    if ((rc = smx_dev_agg_status(ws.p))) return rc;
    SMX_HIP(hipMemcpy(filter_cost, dBest.p, n * sizeof(float), hipMemcpyDeviceToHost));
    SMX_HIP(hipMemcpy(disp_map, dMap.p, n * sizeof(float), hipMemcpyDeviceToHost));
    if (mean) SMX_HIP(hipMemcpy(mean, dMean.p, n, hipMemcpyDeviceToHost));
    if (agg) SMX_HIP(hipMemcpy(agg, dAgg.p, n * size_d * sizeof(float), hipMemcpyDeviceToHost));
    return SMX_OK;
}

int smx_detect_occlusion(const smx_params* p, float* disparityLeft, const float* disparityRight,
                         int dOcclusion, int w, int h) {
    SMX_ARG(p && disparityLeft && disparityRight && w >= 1 && h >= 1);
    const size_t bytes = (size_t)w * h * sizeof(float);
    DevBuf dL, dR;
    SMX_HIP(dL.alloc(bytes));
    SMX_HIP(dR.alloc(bytes));
    SMX_HIP(hipMemcpy(dL.p, disparityLeft, bytes, hipMemcpyHostToDevice));
    SMX_HIP(hipMemcpy(dR.p, disparityRight, bytes, hipMemcpyHostToDevice));
    int rc = smx_dev_detect_occlusion(p, dL.as<float>(), dR.as<float>(), dOcclusion, w, h, nullptr);
    if (rc) return rc;
    SMX_HIP(hipDeviceSynchronize());
    SMX_HIP(hipMemcpy(disparityLeft, dL.p, bytes, hipMemcpyDeviceToHost));
    return SMX_OK;
}

int smx_fill_occlusion(float* disparity, int w, int h, float vMin) {
    SMX_ARG(disparity && w >= 1 && h >= 1);
    const size_t bytes = (size_t)w * h * sizeof(float);
    DevBuf d;
    SMX_HIP(d.alloc(bytes));
    SMX_HIP(hipMemcpy(d.p, disparity, bytes, hipMemcpyHostToDevice));
    int rc = smx_dev_fill_occlusion(d.as<float>(), w, h, vMin, nullptr);
    if (rc) return rc;
    SMX_HIP(hipDeviceSynchronize());
    SMX_HIP(hipMemcpy(disparity, d.p, bytes, hipMemcpyDeviceToHost));
    return SMX_OK;
}

int smx_dev_filter(const smx_params* p, const uint8_t* d_image, int w, int h, uint8_t* d_mean,
                   float* d_var, void* stream) {
    SMX_ARG(p && d_image && d_mean && d_var && w >= 1 && h >= 1 && p->radius >= 0);
    return launch_filter(p, d_image, d_mean, d_var, w, h, (hipStream_t)stream);
}

int smx_filter(const smx_params* p, const uint8_t* image, int w, int h, uint8_t* mean, float* var) {
    SMX_ARG(p && image && mean && var && w >= 1 && h >= 1 && p->radius >= 0);
    const size_t n = (size_t)w * h;
    DevBuf dI, dM, dV;
    SMX_HIP(dI.alloc(n));
    SMX_HIP(dM.alloc(n));
    SMX_HIP(dV.alloc(n * sizeof(float)));
    SMX_HIP(hipMemcpy(dI.p, image, n, hipMemcpyHostToDevice));
    int rc = smx_dev_filter(p, dI.as<uint8_t>(), w, h, dM.as<uint8_t>(), dV.as<float>(), nullptr);
    if (rc) return rc;
    SMX_HIP(hipDeviceSynchronize());
    SMX_HIP(hipMemcpy(mean, dM.p, n, hipMemcpyDeviceToHost));
    SMX_HIP(hipMemcpy(var, dV.p, n * sizeof(float), hipMemcpyDeviceToHost));
    return SMX_OK;
}

// ---- persistent context of the host-pointer pair entry ---------------------------------------------
struct smx_ctx {
    smx_params p;
    int w = 0, h = 0, size_d = 0, dev = -1;
    int agg_path = 0;      // this context's aggregation path (smx_ctx_set_agg_path); starts as the creating thread's
    size_t n = 0, ws_bytes = 0;
    hipStream_t st = nullptr;
    // keys / best / dmap / mean: left view first, right view behind it (one buffer each)
    DevBuf dL, dR, keys, best, map, mean, occ, fil, ws, costL, costR, aggLR;
    // pipelined entry (smx_ctx_stereo_pair_async): two slots of device inputs / results and pinned host staging, created
    // on first use.  Staging of a slot: [gray_l | gray_r] going up; [best_l best_r dmap_l dmap_r occlusion filled | mean_l
    // mean_r | status word] coming down.
    struct Slot {
        DevBuf in, res, mean;
        uint8_t* h_in = nullptr;
        char* h_out = nullptr;
        hipEvent_t up = nullptr, done = nullptr, down = nullptr;
        int dminl = 0, dminr = 0;
        bool busy = false;
    } slot[2];
    hipStream_t st_up = nullptr, st_dn = nullptr;
    uint64_t submitted = 0, waited = 0;
    ~smx_ctx() {
        for (Slot& sl : slot) {
            if (sl.h_in) (void)hipHostFree(sl.h_in);
            if (sl.h_out) (void)hipHostFree(sl.h_out);
            for (hipEvent_t e : {sl.up, sl.done, sl.down})
                if (e) (void)hipEventDestroy(e);
        }
        for (hipStream_t x : {st, st_up, st_dn})
            if (x) (void)hipStreamDestroy(x);
    }
};

int smx_create(const smx_params* p, int w, int h, int size_d, smx_ctx** out) {
    SMX_ARG(p && out && w >= 2 && h >= 1 && size_d >= 1 && p->radius >= 0);
    *out = nullptr;
    smx_ctx* c = new (std::nothrow) smx_ctx;
    if (!c) return fail(SMX_E_HIP, "smx_create: out of host memory");
    struct Guard { smx_ctx* c; ~Guard() { delete c; } } guard{c};
    c->p = *p; c->w = w; c->h = h; c->size_d = size_d;
    c->agg_path = g_agg_path;
    c->n = (size_t)w * h;
    const size_t n = c->n, fb = n * sizeof(float);
    SMX_HIP(hipGetDevice(&c->dev));
    SMX_HIP(hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking));
    c->ws_bytes = 2 * pick_ws_bytes(w, h, size_d);     // both views per launch
    SMX_HIP(c->dL.alloc(n)); SMX_HIP(c->dR.alloc(n));
    SMX_HIP(c->keys.alloc(2 * n * 8));
    SMX_HIP(c->best.alloc(2 * fb)); SMX_HIP(c->map.alloc(2 * fb));
    SMX_HIP(c->mean.alloc(2 * n));
    SMX_HIP(c->occ.alloc(fb)); SMX_HIP(c->fil.alloc(fb));
    SMX_HIP(c->ws.alloc(c->ws_bytes));
    guard.c = nullptr;
    *out = c;
    return SMX_OK;
}

int smx_ctx_set_agg_path(smx_ctx* c, int path) {
    SMX_ARG(c);
    if (path < 0 || path > 5) return fail(SMX_E_ARG, "smx_ctx_set_agg_path: path must be 0 .. 5");
    c->agg_path = path;
    return SMX_OK;
}

int smx_destroy(smx_ctx* c) {
    if (!c) return SMX_OK;
    int dev = -1;
    (void)hipGetDevice(&dev);
    if (c->dev >= 0 && dev != c->dev) (void)hipSetDevice(c->dev);
    for (hipStream_t x : {c->st_up, c->st, c->st_dn})
        if (x) (void)hipStreamSynchronize(x);
    delete c;
    if (dev >= 0) (void)hipSetDevice(dev);
    return SMX_OK;
}

// The path of one pair on the context's stream: device images in, the eight result planes out (+ the optional volumes
// of the context).  Shared by the synchronous and the pipelined host-pointer entry.
static int ctx_enqueue(smx_ctx* c, const uint8_t* dL, const uint8_t* dR, int dminl, int dminr, bool want_cost, bool want_agg,
                       float* bestL, float* mapL, uint8_t* mean, float* occ, float* fil) {
    const smx_params* p = &c->p;
    const int w = c->w, h = c->h, size_d = c->size_d;
    const size_t n = c->n;
    hipStream_t st = c->st;
    int rc;
    const int64_t nn = (int64_t)n;
    int64_t* keysL = c->keys.as<int64_t>(); int64_t* keysR = keysL + n;
    if (c->agg_path >= 2 && !v4_supported(p))
        return fail(SMX_E_ARG, "smx_ctx_stereo_pair: fused path forced but radius > 9");
    struct Nest { Nest() { g_in_ctx += 2; } ~Nest() { g_in_ctx -= 2; } } nest;     // (nested smx_dev_* calls do not restart the stage marks)
    // cost volumes are materialised only when the caller asks for them (main.cu:80-82) and then feed
    // the aggregation like in the reference; otherwise the slices are built on the fly inside it.
    if (want_cost) {
        if ((rc = smx_dev_cost_volume(p, dL, dR, c->costL.as<float>(), w, w, h, dminl, 0, size_d, st))) return rc;
        if ((rc = smx_dev_cost_volume(p, dR, dL, c->costR.as<float>(), w, w, h, dminr, 0, size_d, st))) return rc;
    }
    if ((rc = smx_dev_init_keys(keysL, 2 * nn, st))) return rc;
    // main.cu:133-134, both views per kernel launch
    if (v4_supported(p) && c->agg_path != 1) {
        const uint8_t* guide[2] = {dL, dR};
        const uint8_t* other[2] = {dR, dL};
        const float* cost[2] = {c->costL.as<float>(), c->costR.as<float>()};
        const int dmin[2] = {dminl, dminr};
        int64_t* kv[2] = {keysL, keysR};
        uint8_t* mv[2] = {mean, mean + n};
        float* av[2] = {c->aggLR.as<float>(), want_agg ? c->aggLR.as<float>() + (size_t)size_d * n : nullptr};
        g_launches = 0;
        if ((rc = aggregate_fused(c->agg_path, p, 2, guide, other, want_cost ? cost : nullptr, w, h, dmin, 0, size_d, kv, mv,
                                  want_agg ? av : nullptr, c->ws.p, c->ws_bytes, st, &g_launches)))
            return rc;
    } else {
        const int saved = g_agg_path;
        g_agg_path = 1;
        if ((rc = smx_dev_aggregate_wta(p, dL, dR, want_cost ? c->costL.as<float>() : nullptr, w, h, dminl, 0, size_d,
                                        keysL, mean, want_agg ? c->aggLR.as<float>() : nullptr, c->ws.p,
                                        c->ws_bytes, st)))
            { g_agg_path = saved; return rc; }
        if ((rc = smx_dev_aggregate_wta(p, dR, dL, want_cost ? c->costR.as<float>() : nullptr, w, h, dminr, 0, size_d,
                                        keysR, mean + n,
                                        want_agg ? c->aggLR.as<float>() + (size_t)size_d * n : nullptr, c->ws.p,
                                        c->ws_bytes, st)))
            { g_agg_path = saved; return rc; }
        g_agg_path = saved;
    }
    // main.cu:112-118 presets, winning slices, main.cu:140-155
    return smx_dev_finish_pair(p, keysL, w, h, dminl, dminr, dminl - 100, (float)dminl, bestL, mapL, occ, fil, st);
}

static int ctx_check_device(smx_ctx* c, const char* who) {
    int dev = -1;
    SMX_HIP(hipGetDevice(&dev));
    if (dev != c->dev) return fail(SMX_E_ARG, "%s: the context lives on device %d, current device is %d", who, c->dev, dev);
    return SMX_OK;
}

int smx_ctx_stereo_pair(smx_ctx* c, const uint8_t* gray_l, const uint8_t* gray_r, int dminl, int dminr,
                        const smx_pair_out* out) {
    SMX_ARG(c && gray_l && gray_r && out);
    const int size_d = c->size_d;
    const size_t n = c->n, fb = n * sizeof(float), vb = fb * size_d;
    int rc;
    if ((rc = ctx_check_device(c, "smx_ctx_stereo_pair"))) return rc;
    if (c->submitted != c->waited) return fail(SMX_E_ARG, "smx_ctx_stereo_pair: pipelined pairs are still in flight (smx_ctx_wait)");
    hipStream_t st = c->st;
    const bool want_cost = out->cost_l || out->cost_r;
    const bool want_agg = out->agg_l || out->agg_r;
    if (want_cost && !c->costL.p) { SMX_HIP(c->costL.alloc(vb)); SMX_HIP(c->costR.alloc(vb)); }
    if (want_agg && !c->aggLR.p) SMX_HIP(c->aggLR.alloc(2 * vb));
    uint8_t* dL = c->dL.as<uint8_t>(); uint8_t* dR = c->dR.as<uint8_t>();
    stage_mark(ST_BEGIN, st);
    SMX_HIP(hipMemcpyAsync(dL, gray_l, n, hipMemcpyHostToDevice, st));
    SMX_HIP(hipMemcpyAsync(dR, gray_r, n, hipMemcpyHostToDevice, st));
    stage_mark(ST_UPLOAD, st);
    float* bestL = c->best.as<float>(); float* bestR = bestL + n;
    float* mapL = c->map.as<float>();   float* mapR = mapL + n;
    if ((rc = ctx_enqueue(c, dL, dR, dminl, dminr, want_cost, want_agg, bestL, mapL, c->mean.as<uint8_t>(), c->occ.as<float>(),
                          c->fil.as<float>())))
        return rc;
    struct { void* dst; const void* src; size_t b; } copies[] = {
        {out->best_l, bestL, fb}, {out->best_r, bestR, fb}, {out->dmap_l, mapL, fb},
        {out->dmap_r, mapR, fb},  {out->mean_l, c->mean.p, n}, {out->mean_r, c->mean.as<uint8_t>() + n, n},
        {out->occlusion, c->occ.p, fb}, {out->filled, c->fil.p, fb},  {out->cost_l, c->costL.p, vb},
        {out->cost_r, c->costR.p, vb}, {out->agg_l, c->aggLR.p, vb},
        {out->agg_r, want_agg ? (const void*)(c->aggLR.as<float>() + (size_t)size_d * n) : nullptr, vb},
    };
    for (auto& cp : copies)
        if (cp.dst && cp.src) SMX_HIP(hipMemcpyAsync(cp.dst, cp.src, cp.b, hipMemcpyDeviceToHost, st));
    stage_mark(ST_DOWNLOAD, st);
    SMX_HIP(hipStreamSynchronize(st));
    return smx_dev_agg_status(c->ws.p);
}

// ---- pipelined host-pointer entry ---------------------------------------------------------------------------------
// Pair k uses slot k % 2.  Three streams: uploads, the path, downloads; events chain a pair through them, so that the
// upload of pair k+1 and the download of pair k-1 run under the aggregation of pair k.
static size_t slot_out_bytes(size_t n) { return 6 * n * sizeof(float) + 2 * n + 256; }

static int ctx_async_setup(smx_ctx* c) {
    if (c->st_up) return SMX_OK;
    const size_t n = c->n;
    SMX_HIP(hipStreamCreateWithFlags(&c->st_up, hipStreamNonBlocking));
    SMX_HIP(hipStreamCreateWithFlags(&c->st_dn, hipStreamNonBlocking));
    for (smx_ctx::Slot& sl : c->slot) {
        SMX_HIP(sl.in.alloc(2 * n));
        SMX_HIP(sl.res.alloc(6 * n * sizeof(float)));
        SMX_HIP(sl.mean.alloc(2 * n));
        SMX_HIP(hipHostMalloc((void**)&sl.h_in, 2 * n, hipHostMallocDefault));
        SMX_HIP(hipHostMalloc((void**)&sl.h_out, slot_out_bytes(n), hipHostMallocDefault));
        SMX_HIP(hipEventCreateWithFlags(&sl.up, hipEventDisableTiming));
        SMX_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        SMX_HIP(hipEventCreateWithFlags(&sl.down, hipEventDisableTiming));
    }
    return SMX_OK;
}

int smx_ctx_stereo_pair_async(smx_ctx* c, const uint8_t* gray_l, const uint8_t* gray_r, int dminl, int dminr) {
    SMX_ARG(c && gray_l && gray_r);
    int rc;
    if ((rc = ctx_check_device(c, "smx_ctx_stereo_pair_async"))) return rc;
    if (c->submitted - c->waited >= 2)
        return fail(SMX_E_ARG, "smx_ctx_stereo_pair_async: two pairs are in flight already (smx_ctx_wait takes the older one)");
    if ((rc = ctx_async_setup(c))) return rc;
    // no stage marks in the pipelined entry: pairs overlap on three streams, so "the stages of the last call" has no meaning
    // here, and an event record per stage is a bubble on the queue the pipeline exists to keep full
    struct Pause { int saved; Pause() : saved(g_timing) { g_timing = 0; } ~Pause() { g_timing = saved; } } pause;
    const size_t n = c->n, fb = n * sizeof(float);
    smx_ctx::Slot& sl = c->slot[c->submitted & 1];
    // the caller's images into the slot's pinned staging: the caller's buffers are free again when this call returns
    memcpy(sl.h_in, gray_l, n);
    memcpy(sl.h_in + n, gray_r, n);
    sl.dminl = dminl; sl.dminr = dminr;
    uint8_t* dL = sl.in.as<uint8_t>(); uint8_t* dR = dL + n;
    SMX_HIP(hipMemcpyAsync(dL, sl.h_in, 2 * n, hipMemcpyHostToDevice, c->st_up));
    SMX_HIP(hipEventRecord(sl.up, c->st_up));
    SMX_HIP(hipStreamWaitEvent(c->st, sl.up, 0));
    float* r = sl.res.as<float>();                       // best_l best_r dmap_l dmap_r occlusion filled
    if ((rc = ctx_enqueue(c, dL, dR, dminl, dminr, false, false, r, r + 2 * n, sl.mean.as<uint8_t>(), r + 4 * n, r + 5 * n)))
        return rc;
    // status word of this pair's aggregation (the next pair's launch clears it): behind the planes in the staging
    char* status_h = sl.h_out + 6 * fb + 2 * n;
    SMX_HIP(hipMemcpyAsync(status_h, (const char*)align_up((size_t)c->ws.p, 256), sizeof(unsigned), hipMemcpyDeviceToHost, c->st));
    SMX_HIP(hipEventRecord(sl.done, c->st));
    SMX_HIP(hipStreamWaitEvent(c->st_dn, sl.done, 0));
    SMX_HIP(hipMemcpyAsync(sl.h_out, r, 6 * fb, hipMemcpyDeviceToHost, c->st_dn));
    SMX_HIP(hipMemcpyAsync(sl.h_out + 6 * fb, sl.mean.p, 2 * n, hipMemcpyDeviceToHost, c->st_dn));
    SMX_HIP(hipEventRecord(sl.down, c->st_dn));
    sl.busy = true;
    ++c->submitted;
    return SMX_OK;
}

int smx_ctx_wait(smx_ctx* c, smx_pair_out* staged, const smx_pair_out* copy_to) {
    SMX_ARG(c);
    if (c->submitted == c->waited) return fail(SMX_E_ARG, "smx_ctx_wait: no pair in flight");
    int rc;
    if ((rc = ctx_check_device(c, "smx_ctx_wait"))) return rc;
    smx_ctx::Slot& sl = c->slot[c->waited & 1];
    SMX_HIP(hipEventSynchronize(sl.down));
    sl.busy = false;
    ++c->waited;
    const size_t n = c->n, fb = n * sizeof(float);
    float* f = (float*)sl.h_out;
    uint8_t* m = (uint8_t*)(sl.h_out + 6 * fb);
    smx_pair_out v;
    memset(&v, 0, sizeof(v));
    v.best_l = f; v.best_r = f + n; v.dmap_l = f + 2 * n; v.dmap_r = f + 3 * n; v.occlusion = f + 4 * n; v.filled = f + 5 * n;
    v.mean_l = m; v.mean_r = m + n;
    if (staged) *staged = v;
    if (copy_to) {
        struct { void* dst; const void* src; size_t b; } copies[] = {
            {copy_to->best_l, v.best_l, fb}, {copy_to->best_r, v.best_r, fb}, {copy_to->dmap_l, v.dmap_l, fb},
            {copy_to->dmap_r, v.dmap_r, fb}, {copy_to->occlusion, v.occlusion, fb}, {copy_to->filled, v.filled, fb},
            {copy_to->mean_l, v.mean_l, n}, {copy_to->mean_r, v.mean_r, n},
        };
        for (auto& cp : copies)
            if (cp.dst) memcpy(cp.dst, cp.src, cp.b);
    }
    unsigned status = 0;
    memcpy(&status, sl.h_out + 6 * fb + 2 * n, sizeof(status));
    if (status != 0)
        return fail(SMX_E_HIP, "fused aggregation: hand-off wait of work item %u timed out (results invalid)", status - 1);
    return SMX_OK;
}

int smx_stereo_pair(const smx_params* p, const uint8_t* gray_l, const uint8_t* gray_r, int w, int h,
                    int size_d, int dminl, int dminr, const smx_pair_out* out) {
    SMX_ARG(p && gray_l && gray_r && out && w >= 2 && h >= 1 && size_d >= 1);
    smx_ctx* c = nullptr;
    int rc = smx_create(p, w, h, size_d, &c);
    if (rc) return rc;
    rc = smx_ctx_stereo_pair(c, gray_l, gray_r, dminl, dminr, out);
    (void)smx_destroy(c);
    return rc;
}

}  // extern "C"
