// Internal helpers shared by the HIP translation units of libsmx_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

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

// Stage boundary of the calling thread's timed call (smx_set_timing(1) / smx_stage_times): records a HIP event of the
// CURRENT device on `st`; a no-op when timing is off.  Stage ids: smx_capi.hip.
enum StageId { ST_BEGIN = 0, ST_UPLOAD, ST_GUIDANCE, ST_WALK, ST_WTA, ST_FINISH, ST_DOWNLOAD, ST_COUNT };
void stage_mark(int stage, hipStream_t st);

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

__host__ __device__ inline void unpack_key(int64_t key, float* cost, uint32_t* slice) {
    uint32_t u = (uint32_t)((uint64_t)key >> 32);
    u = (u & 0x80000000u) ? ~(u ^ 0x80000000u) : u;
    *cost = __builtin_bit_cast(float, u);
    *slice = 0xFFFFFFFFu - (uint32_t)((uint64_t)key & 0xFFFFFFFFu);
}

}  // namespace smx
