// smx_agg_dev.h -- device helpers shared by the fused aggregation kernels (smx_agg_v4.hip, smx_agg_v5.hip):
// vector types, buffer descriptors, the hand-off accesses (sc1 form of the guide), the LDS-only workgroup
// barrier, the matching cost of one cell and the exhaustive-checked division by a small integer area.
// gfx950 only; every translation unit that includes this must be compiled with -ffp-contract=off.
#pragma once
#include "smx_common.h"

namespace smx {
namespace aggdev {

typedef _Float16 fg_t __attribute__((ext_vector_type(2)));   // (pixel value, x-derivative), exact in fp16
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

constexpr int PADX = 4;                 // sentinel columns on either side of an image plane of k_v4_prep

// RN(1/d) for d = 0 .. 361 (box areas are (xmax-xmin)*(ymax-ymin) <= 19*19)
constexpr int RCP_N = 19 * 19 + 1;
struct RcpTable {
    float v[RCP_N];
    constexpr RcpTable() : v() {
        v[0] = 0.0f;
        for (int i = 1; i < RCP_N; ++i) v[i] = 1.0f / (float)i;
    }
};
static __constant__ RcpTable kRcp = RcpTable();

// x / d, correctly rounded, for an integer-valued d in [1, 361] with r = RN(1/d): one residual correction
// step (Markstein).  Bit-identical to IEEE division for |x| >= 2^-100 (exhaustive over the significand for
// every area: tools/check_fastdiv.c); callers route smaller |x| (incl. -0, whose sign the correction would
// lose; +0 is exact) and non-finite x to the true division.
__device__ __forceinline__ float div_small_int(float x, float d, float r) {
    float q = x * r;
    float e = __builtin_fmaf(-q, d, x);
    return __builtin_fmaf(e, r, q);
}
// the same for both components of a cell at once: v_pk_mul_f32 + 2 v_pk_fma_f32 (elementwise, each rounded once:
// bit-identical to the scalar form)
__device__ __forceinline__ f2 div_small_int2(f2 x, float d, float r) {
    const f2 d2 = {d, d}, r2 = {r, r};
    f2 q = x * r2;
    f2 e = __builtin_elementwise_fma(-q, d2, x);
    return __builtin_elementwise_fma(e, r2, q);
}
__device__ __forceinline__ bool div_needs_exact(float x) {
    const float ax = fabsf(x);
    return !(ax >= 0x1p-100f && ax < __builtin_inff());   // tiny, zero, inf or NaN
}

// Guidance statistics of one pixel from the integral images S0 (of I) and S1 (of I*I): mean_I = box(S0),
// var = box(S1) - mean_I*mean_I, 1/(var + eps) in double as the reference's compute_ak_and_bk (guidedFilter.cu:350).
// Returns (mean_I, 1/(var_I + eps)).
__device__ __forceinline__ f2 guid_point(const float* __restrict__ S0, const float* __restrict__ S1, int x, int y, int w, int h,
                                         int R, double eps) {
    const int ymin = max(-1, y - R - 1), ymax = min(h - 1, y + R);
    const int xmin = max(-1, x - R - 1), xmax = min(w - 1, x + R);
    const bool hx = xmin >= 0, hy = ymin >= 0;
    const size_t i11 = (size_t)ymax * w + xmax, i10 = (size_t)ymax * w + (hx ? xmin : 0);
    const size_t i01 = (size_t)(hy ? ymin : 0) * w + xmax, i00 = (size_t)(hy ? ymin : 0) * w + (hx ? xmin : 0);
    const float area = (float)((xmax - xmin) * (ymax - ymin));
    auto box = [&](const float* __restrict__ S) {      // computeBoxFilterOnGPU guidedFilter.cu:305-318
        float val = S[i11];
        if (hx) val -= S[i10];
        if (hy) val -= S[i01];
        if (hx && hy) val += S[i00];
        return 1.0f * val / area;
    };
    const float m = box(S0), sq = box(S1);
    const float m2 = m * m;                             // pixelMultOnGPU(mean, mean) :112
    const float var = sq - m2;                          // pixelSousOnGPU :121
    return (f2){m, (float)(1.0f / ((double)var + eps))};
}
__device__ __forceinline__ uint8_t mean_to_u8(float m) {   // flToChOnGPU guidedFilter.cu:451-458
    const int ci = (int)m;
    return (ci > 255) ? 255 : (uint8_t)ci;
}

// p = (1-alpha)*min(|I1 - I2|, 7) + alpha*min(|g1 - g2|, 2) and I1*p  (costVolume.cu:187,
// guidedFilter.cu:209).  The halves convert exactly, so the f32 operations equal the reference's; the
// sentinel 60000 of an out-of-range partner saturates both terms = the border constant (:184).
// The two differences and the two products as packed instructions, the two selects as v_min_f32 (no operand is
// ever a NaN: the inputs are finite halves, so min(|d|, th) == (|d| < th ? |d| : th)).
__device__ __forceinline__ f2 cost_pair(fg_t q1, fg_t q2, const CostConst& cc) {
    const f2 v1 = {(float)q1.x, (float)q1.y}, v2 = {(float)q2.x, (float)q2.y};
    const f2 d = v1 - v2;
    const f2 m = {__builtin_fminf(__builtin_fabsf(d.x), cc.th_color), __builtin_fminf(__builtin_fabsf(d.y), cc.th_grad)};
    const f2 xz = (f2){cc.oma, cc.alpha} * m;
    f2 r;
    r.x = xz.x + xz.y;
    r.y = v1.x * r.x;
    return r;
}

// ---- hand-off accesses: sc1 (bypass this CU's L1, write through the XCD's L2) ----------------
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int AUX_SC1 = 16;
#ifndef SMX_Q_ST_AUX
#define SMX_Q_ST_AUX 2     // (A/B: 0 plain, 2 nt)
#endif
constexpr int AUX_NT = SMX_Q_ST_AUX;   // q stores and the WTA's q loads are nt: measured best of plain / sc1 / nt
__device__ __forceinline__ rsrc_t mk_rsrc(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0,
                                             (int)(bytes > 0xFFFFFFFFull ? 0xFFFFFFFFull : bytes),
                                             0x00020000);
}
__device__ __forceinline__ f4 ld16_sc1(rsrc_t r, unsigned byteoff) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byteoff, 0, AUX_SC1));
}
__device__ __forceinline__ void st16_sc1(rsrc_t r, unsigned byteoff, f4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, (int)byteoff, 0, AUX_SC1);
}
__device__ __forceinline__ unsigned ldu(rsrc_t r, unsigned voff, int soff) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, soff, 0);
}
typedef __attribute__((address_space(1))) unsigned gu32;
__device__ __forceinline__ unsigned flag_load(unsigned* p) {
    return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void flag_store(unsigned* p, unsigned v) {
    __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Workgroup barrier that orders LDS only: __syncthreads() would also wait for every outstanding global
// load and store, which is exactly the latency the cross-phase prefetches hide.
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

constexpr unsigned FLAG_DONE = 0x7fffffffu;
constexpr unsigned OOB = 0x80000000u;   // a buffer offset beyond every plane: loads return 0, stores are dropped

// A value the compiler must re-derive where it is used (keeps per-lane constants out of long live ranges)
__device__ __forceinline__ int opaque(int x) { asm volatile("" : "+v"(x)); return x; }
typedef __attribute__((address_space(3))) const char lds_cc;
__device__ __forceinline__ unsigned lds_off(const void* p) { return (unsigned)(size_t)(lds_cc*)p; }

}  // namespace aggdev
}  // namespace smx
