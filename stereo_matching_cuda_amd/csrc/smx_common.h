// Internal helpers shared by the HIP translation units of libsmx_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <mutex>

#include "smx.h"

namespace smx {

int fail(int code, const char* fmt, ...);

#define SMX_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e__ = (call);                                                        \
        if (e__ != hipSuccess)                                                          \
            return ::smx::fail(SMX_E_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, \
                               hipGetErrorString(e__));                                 \
    } while (0)

#define SMX_ARG(cond)                                                                       \
    do {                                                                                    \
        if (!(cond)) return ::smx::fail(SMX_E_ARG, "%s: bad argument: %s", __func__, #cond); \
    } while (0)

// Float constants derived from smx_params exactly as the reference kernels derive them from the
// macros (costVolume.cu:169-171,184): all in f32, each operation rounded on its own.
struct CostConst {
    float alpha;      // 1.0f*ALPHA
    float oma;        // 1.0f - alpha
    float th_color;   // 1.0f*TH_color
    float th_grad;    // 1.0f*TH_grad
    float border;     // (1 - alpha)*th_color + 1.0f*alpha*th_grad
};

inline CostConst make_cost_const(const smx_params* p) {
    CostConst c;
    c.alpha = 1.0f * p->alpha;
    c.oma = 1.0f - c.alpha;
    c.th_color = 1.0f * p->th_color;
    c.th_grad = 1.0f * p->th_grad;
    float a = c.oma * c.th_color;
    float b = 1.0f * c.alpha * c.th_grad;
    c.border = a + b;
    return c;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// A kernel's dynamic-LDS limit, raised ONCE per device to the most its launcher ever asks for (hipFuncSetAttribute is a
// per-device setting and a runtime call: it does not belong in front of every launch).  One static instance per kernel.
struct LdsLimitOnce {
    std::once_flag done[64];
    hipError_t err[64];
    hipError_t ensure(const void* fn, int bytes) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const int i = dev & 63;
        std::call_once(done[i], [&] { err[i] = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); });
        return err[i];
    }
};
// developer overrides from the environment, read once per process (never per call)
inline int env_int_once(const char* name, int absent) {
    const char* e = getenv(name);
    return e ? atoi(e) : absent;
}

// Stage boundary of the calling thread's timed call (smx_set_timing(1) / smx_stage_times): records a HIP event of the
// CURRENT device on `st`; a no-op when timing is off.  Stage ids: smx_capi.hip.
enum StageId { ST_BEGIN = 0, ST_UPLOAD, ST_GUIDANCE, ST_WALK, ST_WTA, ST_FINISH, ST_DOWNLOAD, ST_COUNT };
void stage_mark(int stage, hipStream_t st);

// Options / report of one fused aggregation call (smx_agg_v4.hip aggregate_v4; set and read through the C-ABI:
// smx_set_agg_path, smx_set_max_slices_per_launch, smx_last_agg_path, smx_last_agg_chunk)
struct AggOpts {
    bool fast = false;       // FAST mode (not bit-exact)
    int walker = 0;          // 0 choose, 4 ring walker forced, 5 comb walker forced (an error where it does not apply)
    int max_chunk = 0;       // upper bound on the slices of one walker launch; 0 = as many as the workspace holds
    bool keys_fresh = false; // the keys hold nothing yet: the first WTA pass starts from the identity (no smx_dev_init_keys needed)
};
struct AggInfo {
    int walker_used = 0;     // 4 ring walker, 5 comb walker
    int chunk = 0;           // slices per walker launch (of the first = largest launch)
    int walker_launches = 0;
    int launches = 0;        // all kernel launches + memsets of the call
};

// ---- packed WTA key ---------------------------------------------------------------------
// key = sord(cost) << 32 | (0xFFFFFFFF - slice), compared as SIGNED 64-bit integers: sord = monotone
// f32 -> i32 (-0 folded to +0), so that the per-pixel reduction of the shards is a plain int64 MIN
// (RCCL ncclInt64 / torch.int64) with no re-encoding.  Smallest key = smallest cost, and among equal
// costs the LARGEST slice: the reference's `best >= q` rule with ascending slices.
// A NaN cost (only from degenerate parameters, e.g. var + eps == 0) never wins, like the reference's
// `best >= q`, which is false for NaN: it maps to the identity key INT64_MAX.
constexpr int64_t KEY_IDENTITY = 0x7FFFFFFFFFFFFFFFll;
__host__ __device__ inline int64_t pack_key(float cost, uint32_t slice) {
    if (cost != cost) return KEY_IDENTITY;
    if (cost == 0.0f) cost = 0.0f;
    uint32_t u = __builtin_bit_cast(uint32_t, cost);
    u = (u & 0x80000000u) ? (~u ^ 0x80000000u) : u;      // negative floats: reverse their order
    return (int64_t)(((uint64_t)u << 32) | (uint64_t)(0xFFFFFFFFu - slice));
}

// The winner of a run of ASCENDING slices in the float domain: smallest cost, among equal costs (-0 == +0) the LAST slice, a NaN
// never -- `take = q <= m` from m = +inf with no slice yet.  That is the order of the packed keys restricted to ascending
// slices; key() packs the winner once (the identity if nothing was taken), and the caller merges it with the key a pixel already
// holds by the integer min.  (Packing every candidate costs ten vector instructions per element, a step costs three.)
struct WtaRun {
    float m = __builtin_inff();
    uint32_t z = 0xFFFFFFFFu;
    __host__ __device__ inline void step(float q, uint32_t slice) {
        const bool take = q <= m;
        m = take ? q : m;
        z = take ? slice : z;
    }
    __host__ __device__ inline int64_t key() const { return z == 0xFFFFFFFFu ? KEY_IDENTITY : pack_key(m, z); }
};

__host__ __device__ inline void unpack_key(int64_t key, float* cost, uint32_t* slice) {
    uint32_t u = (uint32_t)((uint64_t)key >> 32);
    u = (u & 0x80000000u) ? ~(u ^ 0x80000000u) : u;
    *cost = __builtin_bit_cast(float, u);
    *slice = 0xFFFFFFFFu - (uint32_t)((uint64_t)key & 0xFFFFFFFFu);
}

}  // namespace smx
