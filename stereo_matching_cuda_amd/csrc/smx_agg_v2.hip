// smx_agg_v2.hip -- fused guided-filter aggregation for gfx950 ("carry prepass + strip walker").
//
// Data flow per call (reference: guidedFilter.cu:58-238; cost: costVolume.cu:163-190); every launch
// covers both views of the pair (blockIdx.z / work-item order):
//
//   prep      u8 images -> one plane of half2 (value, x-gradient) per image, transposed and band-
//             blocked ([band][x][64 rows]), with one sentinel column (60000) on each side: out-of-
//             range disparities saturate both truncated terms -> exactly the reference's border
//             constant.
//   events    segment start columns sorted ascending (8 sub-strips per strip interleave).
//   carry<M>  one wave per (slice, 64-row band), LANE = ROW: walks the full row left->right with the
//             reference's sequential f32 adds and stores the running row sums at every sub-strip
//             start ("carries").  No LDS, output is ~6 % of a plane.
//   walk<M>   one 512-thread workgroup per (slice, strip of TW = 104 columns), walking 64-row bands
//             top->bottom in a 96-row LDS ring:
//               phase R  LANE = ROW   : each of the 8 waves continues the row scan of one 13-column
//                                       sub-strip from its carry, writing R into the ring (strided,
//                                       odd pitch -> conflict-free)
//               phase C  LANE = COLUMN: sequential column scan down the band, S kept in a register
//                                       per column across bands, R -> S in place in the ring
//               phase B  LANE = ROW   : box means from the ring (4 taps, reference order, division
//                                       bit-identical to IEEE) + the per-pixel arithmetic of the
//                                       stage; outputs are written in the band-blocked transposed
//                                       layout so the next stage's LANE=ROW readers are coalesced
//             The addition order of every prefix sum is exactly the reference's (integral.cu:82-86,
//             124-128); strips overlap by 2R+1 recomputed columns instead of exchanging halos.
//   modes     GUID: (I, I*I)      -> mean_I, 1/(var+eps)      (guidedFilter.cu:58-123)
//             S1  : (p, I*p)      -> a_k, b_k   with p built on the fly (costVolume.cu:184-189,
//                                                               guidedFilter.cu:198-223)
//             S2  : (a_k, b_k)    -> q = mean(a)*I + mean(b)    (guidedFilter.cu:224-233)
//   wta       one lane per pixel over the q planes of the chunk, packed-key min
//             (dispSelectOnGPU guidedFilter.cu:403-411).
//
// Must be compiled with -ffp-contract=off.  SMX_EXP selects timing-experiment variants whose
// results are wrong by design (tools/exp_build.sh); the product build is SMX_EXP == 0.
#include <string.h>

#ifndef SMX_EXP
#define SMX_EXP 0   // timing experiments (tools/exp_build.sh); 0 = product build
#endif
#ifndef SMX_AB_LOAD_AUX
#define SMX_AB_LOAD_AUX 2   // nt: a,b planes are read once per kernel (carry<S2> -23 %, walk<S2> -7 % vs default)
#endif
#if SMX_EXP == 4     // timing experiment: no workgroup barriers in the walker
#define WALK_SYNC() ((void)0)
#else
#define WALK_SYNC() __syncthreads()
#endif

#include "smx_common.h"
#include "smx_launch.h"

namespace smx {
namespace v2 {

constexpr int TW = 104;            // columns of S computed per strip
constexpr int NSUB = 8;            // sub-strips per strip = waves per workgroup
constexpr int SUBW = TW / NSUB;    // 13
constexpr int NTHREADS = NSUB * 64;
constexpr int BH = 64;             // band height = wave width
constexpr int RMAX = 9;            // largest supported box radius
constexpr int RR = 96;             // ring rows >= BH + 2*RMAX + 2; multiple of 32 so that the wrap
                                   // of a band inside the ring does not shift LDS banks
constexpr int PITCH = TW + 1;      // odd pitch: LANE=ROW accesses hit distinct banks
constexpr int PB = 4;              // output columns per load batch in phase B
constexpr int CB = 16;             // columns per load batch of the carry prepass
static_assert(RR >= BH + 2 * RMAX + 2 && RR % 32 == 0, "ring too small");
static_assert(TW % NSUB == 0 && 2 * TW <= NTHREADS, "tile geometry");

enum Mode { GUID = 0, S1 = 1, S2 = 2 };

// (pixel value, x-derivative) packed as two IEEE halves: values are integers in [0, 255] and
// multiples of 0.5 in [-127.5, 127.5], both exact in fp16; the sentinel 60000 saturates both
// truncated cost terms like the reference's out-of-range branch (costVolume.cu:184).
typedef _Float16 fg_t __attribute__((ext_vector_type(2)));

// Band-blocked transposed plane layout ("[band][x][64 rows]"): element (x, y) of a plane that is
// Wp columns wide lives at  ((y >> 6) * Wp + x) * 64 + (y & 63).  A wave (LANE = ROW, 64 rows of one
// band) touches 256 contiguous bytes per column and consecutive columns are adjacent, so a band
// tile of a strip is one contiguous run in HBM.  bcol() is the wave-uniform part, blane() the
// per-lane part (fits 32 bits: a plane is < 4 GiB).
__host__ __device__ __forceinline__ size_t bcol(int x) { return (size_t)x * 64; }
__host__ __device__ __forceinline__ unsigned blane(int y, int Wp) {
    return ((unsigned)(y >> 6) * (unsigned)Wp) * 64u + (unsigned)(y & 63);
}

// Buffer-descriptor accesses for the walker: 128-bit descriptor in SGPRs (built from kernel
// arguments and blockIdx only, so it is provably wave-uniform), per-lane byte offset in ONE VGPR
// that is constant across the columns of a band, per-column byte offset in an SGPR / immediate.
// No per-access VALU address arithmetic; out-of-range accesses are dropped by the hardware.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t mk_rsrc(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0,
                                             (int)(bytes > 0xFFFFFFFFull ? 0xFFFFFFFFull : bytes),
                                             0x00020000);
}
__device__ __forceinline__ uint32_t bld(rsrc_t r, unsigned voff, unsigned soff) {
#if SMX_EXP == 3   // timing experiment: no global loads in the walker
    return voff + soff;
#else
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0);
#endif
}
// streamed-once inputs (a, b planes): non-temporal so they do not displace the re-read image planes
__device__ __forceinline__ uint32_t bld_nt(rsrc_t r, unsigned voff, unsigned soff) {
#if SMX_EXP == 3
    return voff + soff;
#else
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, SMX_AB_LOAD_AUX);
#endif
}
__device__ __forceinline__ float bldf(rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
// Cache policy of the streamed outputs, measured on KITTI shape (ms per pair, sc1 / plain / nt):
// a,b stores of stage 1: walk<S1> 0.76 / 0.70 / 0.85, the following carry<S2> 0.41 / 0.38 / 0.32;
// q stores of stage 2: walk<S2> 0.80 / 0.77 / 0.75.  Plain for a,b and nt for q is the best total.
template <int AUX>
__device__ __forceinline__ void bstf(rsrc_t r, unsigned voff, unsigned soff, float v) {
#if SMX_EXP == 1   // timing experiment: no output stores (keep the value alive)
    asm volatile("" ::"v"(v));
#else
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, (int)voff, (int)soff, AUX);
#endif
}

struct Args {
    int w, h, hp, R, ow, nstrips, nsegs;
    const fg_t* FG1;                    // this view: (value, x-gradient) as half2, padded transposed
    const fg_t* FG2;                    // other view (S1 only)
    const float* meanT; const float* cinvT;  // guidance statistics [x*hp + y] (S1 reads, GUID writes)
    const float* srcA; const float* srcB;    // S2 sources: aT, bT [slice][x*hp + y]
    float* dstA; float* dstB;           // GUID: meanT, cinvT; S1: aT, bT; S2: qT (dstB unused)
    uint8_t* mean_u8;                   // GUID optional, row-major [y*w + x]
    float* carry;                       // [(slice*2 + i)*nsegs + g][hp]
    const int* ev_start;                // segment start columns, ascending (k_v2_events)
    const int* ev_seg;                  // segment id g of each event
    int d0;                             // disparity of local slice 0
    CostConst cc;
    double eps;
};

// Both views of a pair travel in one launch: blockIdx.z selects the view.
struct Launch {
    Args v[2];
    int nslices, nviews;
};

__device__ __forceinline__ int seg_c0(const Args& a, int g) {
    return (g / NSUB) * a.ow - (a.R + 1) + (g % NSUB) * SUBW;
}

// x / d, correctly rounded, for an integer-valued d in [1, 361] with r = RN(1/d): one residual
// correction step (Markstein).  Bit-identical to IEEE division for |x| >= 2^-100 (checked on 5.7e9
// samples over all window areas, tools/check_fastdiv.c); callers route smaller |x| (incl. +-0, whose
// sign the correction would lose) to the true division.
__device__ __forceinline__ float div_small_int(float x, float d, float r) {
    float q = x * r;
    float e = __builtin_fmaf(-q, d, x);
    return __builtin_fmaf(e, r, q);
}

// p = (1-alpha)*min(|I1 - I2|, 7) + alpha*min(|g1 - g2|, 2) and I1*p  (costVolume.cu:187,
// guidedFilter.cu:209).  The halves convert exactly, so the f32 operations equal the reference's.
__device__ __forceinline__ void cost_pair(fg_t q1, fg_t q2, const CostConst& cc, float& p, float& ip) {
    const float a1 = (float)q1.x, b1 = (float)q1.y, a2 = (float)q2.x, b2 = (float)q2.y;
    float t1 = fabsf(a1 - a2);
    float t2 = fabsf(b1 - b2);
    float m1 = t1 < cc.th_color ? t1 : cc.th_color;
    float m2 = t2 < cc.th_grad ? t2 : cc.th_grad;
    float x = cc.oma * m1;
    float z = cc.alpha * m2;
    p = x + z;
    ip = a1 * p;
}

// The two scanned quantities of NB consecutive columns of row y (LANE = ROW; every load is a 256-B
// coalesced row of a transposed plane).  All global loads of a batch are issued before the first
// use, so one batch exposes one memory latency instead of NB.  Columns outside [0, w) are clamped
// (their values are never accumulated).
template <int MODE>
struct Source {
    rsrc_t img, ra, rb;          // both image planes; S2: this slice's a / b planes
    unsigned fg1o, fg2o;         // byte offsets of this view's / the other view's plane inside img
    unsigned yfg, ypl;           // lane byte offsets in image planes / a,b planes
    int w, d;
    CostConst cc;
    __device__ __forceinline__ Source(const Args& a, int slice, int y) {
        w = a.w; d = a.d0 + slice; cc = a.cc;
        const size_t plane = (size_t)a.w * a.hp;
        const size_t fgbytes = (size_t)(a.w + 2) * a.hp * sizeof(fg_t);
        const fg_t* lo = a.FG1 < a.FG2 ? a.FG1 : a.FG2;
        const fg_t* hi = a.FG1 < a.FG2 ? a.FG2 : a.FG1;
        if (MODE == S1) {
            img = mk_rsrc(lo, (size_t)((const char*)hi - (const char*)lo) + fgbytes);
            fg1o = (unsigned)((const char*)a.FG1 - (const char*)lo);
            fg2o = (unsigned)((const char*)a.FG2 - (const char*)lo);
        } else {
            img = mk_rsrc(a.FG1, fgbytes);
            fg1o = 0; fg2o = 0;
        }
        ra = mk_rsrc(MODE == S2 ? a.srcA + (size_t)slice * plane : (const float*)a.FG1, plane * sizeof(float));
        rb = mk_rsrc(MODE == S2 ? a.srcB + (size_t)slice * plane : (const float*)a.FG1, plane * sizeof(float));
        yfg = blane(y, a.w + 2) * 4u;
        ypl = blane(y, a.w) * 4u;
    }
    // raw operands of NB columns (issue only; nothing waits here).  Columns >= w read past the
    // descriptor or a neighbouring column; their values are never accumulated.
    template <int NB>
    struct Raw {
        uint32_t a[NB], b[NB];
    };
    template <int NB>
    __device__ __forceinline__ void fetch(int c0, Raw<NB>& r) const {
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            const int c = c0 + t;
            if (MODE == GUID) {
                r.a[t] = bld(img, yfg, (unsigned)(c + 1) * 256u);
                r.b[t] = 0;
            } else if (MODE == S1) {
                int xx = c + d;
                xx = xx < -1 ? -1 : (xx > w ? w : xx);  // sentinel columns at -1 and w
                r.a[t] = bld(img, yfg, fg1o + (unsigned)(c + 1) * 256u);
                r.b[t] = bld(img, yfg, fg2o + (unsigned)(xx + 1) * 256u);
            } else {
                r.a[t] = bld_nt(ra, ypl, (unsigned)c * 256u);
                r.b[t] = bld_nt(rb, ypl, (unsigned)c * 256u);
            }
        }
    }
    template <int NB>
    __device__ __forceinline__ void eval(const Raw<NB>& r, int t, float& v0, float& v1) const {
        if (MODE == GUID) {
            v0 = (float)__builtin_bit_cast(fg_t, r.a[t]).x;      // chToFlOnGPU
            v1 = v0 * v0;                                        // pixelMultOnGPU(d_im, d_im)
        } else if (MODE == S1) {
            cost_pair(__builtin_bit_cast(fg_t, r.a[t]), __builtin_bit_cast(fg_t, r.b[t]), cc, v0, v1);
        } else {
            v0 = __builtin_bit_cast(float, r.a[t]);
            v1 = __builtin_bit_cast(float, r.b[t]);
        }
    }
};

// ---------------------------------------------------------------------------------------------
// prep: u8 image [h][w] -> F_T, G_T [(x+1)*hp + y], sentinel columns x = -1 and x = w.
// grid (ceil(hp/64), w + 2, nimages), block 64.
// ---------------------------------------------------------------------------------------------
struct PrepArgs {
    const uint8_t* I[2];
    fg_t* FG[2];
};

__global__ __launch_bounds__(64) void k_v2_prep(PrepArgs pa, int w, int h, int hp) {
    const uint8_t* __restrict__ I = pa.I[blockIdx.z];
    fg_t* __restrict__ FG = pa.FG[blockIdx.z];
    const int y = blockIdx.x * 64 + threadIdx.x;
    const int x = (int)blockIdx.y - 1;
    if (y >= hp) return;
    float f = 0.0f, g = 0.0f;
    if (x < 0 || x >= w) {
        f = 60000.0f; g = 60000.0f;
    } else if (y < h) {
        const uint8_t* row = I + (size_t)y * w;
        f = 1.0f * (float)(int)row[x];
        int c1, c2;  // x_derivativeOnGPU costVolume.cu:358-381
        if (x - 1 >= 0 && x + 1 < w) { c1 = row[x + 1]; c2 = row[x - 1]; }
        else if (x + 1 >= w)         { c1 = row[x];     c2 = row[x - 1]; }
        else                         { c1 = row[x + 1]; c2 = row[x];     }
        g = 1.0f * (float)(c2 - c1) / 2;
    }
    fg_t v;
    v.x = (_Float16)f;
    v.y = (_Float16)g;
    FG[blane(y, w + 2) + bcol(x + 1)] = v;
}

// ---------------------------------------------------------------------------------------------
// carry prepass.  grid (nbands, nslices, nviews), block 64 (LANE = ROW).
// ---------------------------------------------------------------------------------------------
// Segment g = (strip k, sub-strip q) starts at column seg_c0(g).  With 8 sub-strips per strip the
// starts are not monotone in g (the last sub-strips of strip k lie right of the first of strip k+1),
// so the prepass walks an event list sorted by start column.  One block, built once per call.
__global__ void k_v2_events(int nsegs, int ow, int R, int* __restrict__ ev_start, int* __restrict__ ev_seg) {
    for (int g = threadIdx.x; g < nsegs; g += blockDim.x) {
        const int c = (g / NSUB) * ow - (R + 1) + (g % NSUB) * SUBW;
        int rank = 0;
        for (int o = 0; o < nsegs; ++o) {
            const int co = (o / NSUB) * ow - (R + 1) + (o % NSUB) * SUBW;
            rank += (co < c) || (co == c && o < g);
        }
        ev_start[rank] = c;
        ev_seg[rank] = g;
    }
}

template <int MODE>
__global__ __launch_bounds__(64) void k_v2_carry(Launch L) {
    const Args& a = L.v[blockIdx.z];
    const int slice = blockIdx.y;
    const int y = blockIdx.x * 64 + threadIdx.x;
    if (y >= a.h) return;
    Source<MODE> src(a, slice, y);
    float* c0p = a.carry + ((size_t)(slice * 2 + 0) * a.nsegs) * a.hp + y;
    float* c1p = a.carry + ((size_t)(slice * 2 + 1) * a.nsegs) * a.hp + y;
    const int* __restrict__ evs = a.ev_start;
    const int* __restrict__ evg = a.ev_seg;
    float acc0 = -0.0f, acc1 = -0.0f;
    int e = 0;
    // segments that start at or left of column 0 begin with the additive identity
    while (e < a.nsegs && evs[e] <= 0) {
        const int g = evg[e];
        c0p[(size_t)g * a.hp] = acc0;
        c1p[(size_t)g * a.hp] = acc1;
        ++e;
    }
    int next = e < a.nsegs ? evs[e] : 0x7fffffff;
    // one batch of CB columns at a time: all loads of the batch issue before the first use.
    // (Double-buffering the batches was measured slower: +20 % on k_v2_carry<S1>.)
    for (int cb = 0; cb < a.w; cb += CB) {
        typename Source<MODE>::template Raw<CB> cur;
        src.template fetch<CB>(cb, cur);
#pragma unroll
        for (int t = 0; t < CB; ++t) {
            const int c = cb + t;
            if (c < a.w) {
                while (c == next) {   // wave-uniform: carry of segment g = row sum left of column c
                    const int g = evg[e];
                    c0p[(size_t)g * a.hp] = acc0;
                    c1p[(size_t)g * a.hp] = acc1;
                    ++e;
                    next = e < a.nsegs ? evs[e] : 0x7fffffff;
                }
                float v0, v1;
                src.template eval<CB>(cur, t, v0, v1);
                acc0 = v0 + acc0;
                acc1 = v1 + acc1;
            }
        }
    }
    for (; e < a.nsegs; ++e) {     // segments starting at or beyond column w are never read
        const int g = evg[e];
        c0p[(size_t)g * a.hp] = acc0;
        c1p[(size_t)g * a.hp] = acc1;
    }
}

// ---------------------------------------------------------------------------------------------
// strip walker.  grid (nstrips, nslices, nviews), block 256.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int ring_row(int y) { return y % RR; }

// descriptors of the planes a walker workgroup touches (all wave-uniform)
// (few descriptors on purpose: each costs 4 SGPRs for the whole kernel; planes that are neighbours
// in the workspace share one and are told apart by a scalar byte offset)
struct Planes {
    rsrc_t img;             // both padded image planes; fg1o / fg2o select this view's / the other
    rsrc_t stat;            // guidance statistics mean_I, then 1/(var+eps) at +cinvo   (S1 reads)
    rsrc_t srcA, srcB;      // S2 sources (this slice's plane)
    rsrc_t dstA, dstB;      // outputs (this slice's plane)
    rsrc_t car;             // this wave's carry column of integral 0; integral 1 at +car1o
    unsigned fg1o, fg2o, cinvo, car1o;
};

// per-pixel arithmetic of the stage on the two box means m0, m1; the outputs go to column xo
// (byte offset cob = xo * 256) at the lane's row (byte offset yb) of the slice's output planes
template <int MODE>
__device__ __forceinline__ void stage_out(const Args& a, const Planes& P, float m0, float m1, float ga,
                                          float gb, int xo, unsigned yo, unsigned yb, unsigned cob) {
    if (MODE == GUID) {
        float mm = m0 * m0;          // pixelMultOnGPU(mean, mean) guidedFilter.cu:112
        float var = m1 - mm;         // pixelSousOnGPU :121
        float c = (float)(1.0f / ((double)var + a.eps));   // :350
        bstf<0>(P.dstA, yb, cob, m0);
        bstf<0>(P.dstB, yb, cob, c);
        if (a.mean_u8) {             // flToChOnGPU :451-458
            int ci8 = (int)m0;
            a.mean_u8[(size_t)yo * a.w + xo] = (ci8 > 255) ? 255 : (uint8_t)ci8;
        }
    } else if (MODE == S1) {
        float mI = ga;
        float c = gb;
        float mm = mI * m0;          // compute_ak_and_bk guidedFilter.cu:345-354
        float ak = 1.0f * (m1 - mm) * c;
        float mb2 = 1.0f * mI * ak;
        float bk = 1.0f * m0 - mb2;
        bstf<0>(P.dstA, yb, cob, ak);
        bstf<0>(P.dstB, yb, cob, bk);
    } else {
        float tq = m0 * ga;          // compute_q guidedFilter.cu:363-369
        bstf<2>(P.dstA, yb, cob, tq + m1);
    }
}

constexpr int NCB = (TW - 1 + NSUB - 1) / NSUB;   // max output columns per wave (R = 0): 26
constexpr int NBATCH = (NCB + PB - 1) / PB;         // 7

// raw operands of one sub-strip row (SUBW columns), loaded one band ahead of their use:
// one dword per column and plane (S1: two half2 planes; S2: a and b; GUID: one half2 plane)
template <int MODE>
struct RawRow {
    uint32_t u[MODE == GUID ? 1 : 2][SUBW];
};

// FAST: every column of the sub-strip (and, for S1, its disparity-shifted partner) is inside the
// image, so the byte offsets are base + t*256 with t a compile-time constant: they fold into the
// instructions' immediate offsets and the 13 loads cost no scalar arithmetic at all.
template <int MODE, bool FAST>
__device__ __forceinline__ void raw_load(const Planes& P, int w, int d, unsigned yfg, unsigned ypl, int c0,
                                         RawRow<MODE>& r) {
    // yfg / ypl: lane byte offsets blane(y, w + 2) * 4 (image planes) and blane(y, w) * 4 (a, b)
    if (FAST) {
        const unsigned o1 = P.fg1o + (unsigned)(c0 + 1) * 256u;
        const unsigned o2 = P.fg2o + (unsigned)(c0 + d + 1) * 256u;
        const unsigned oa = (unsigned)c0 * 256u;
#pragma unroll
        for (int t = 0; t < SUBW; ++t) {
            if (MODE == GUID) r.u[0][t] = bld(P.img, yfg, o1 + t * 256u);
            if (MODE == S1) { r.u[0][t] = bld(P.img, yfg, o1 + t * 256u); r.u[1][t] = bld(P.img, yfg, o2 + t * 256u); }
            if (MODE == S2) { r.u[0][t] = bld_nt(P.srcA, ypl, oa + t * 256u); r.u[1][t] = bld_nt(P.srcB, ypl, oa + t * 256u); }
        }
        return;
    }
    if (MODE == GUID) {
#pragma unroll
        for (int t = 0; t < SUBW; ++t) {
            int c = c0 + t;
            c = c < 0 ? 0 : (c >= w ? w - 1 : c);
            r.u[0][t] = bld(P.img, yfg, P.fg1o + (unsigned)(c + 1) * 256u);
        }
    } else if (MODE == S1) {
#pragma unroll
        for (int t = 0; t < SUBW; ++t) {
            int c = c0 + t;
            c = c < 0 ? 0 : (c >= w ? w - 1 : c);
            int xx = c + d;
            xx = xx < -1 ? -1 : (xx > w ? w : xx);  // sentinel columns at -1 and w
            r.u[0][t] = bld(P.img, yfg, P.fg1o + (unsigned)(c + 1) * 256u);
            r.u[1][t] = bld(P.img, yfg, P.fg2o + (unsigned)(xx + 1) * 256u);
        }
    } else {
#pragma unroll
        for (int t = 0; t < SUBW; ++t) {
            int c = c0 + t;
            c = c < 0 ? 0 : (c >= w ? w - 1 : c);
            r.u[0][t] = bld_nt(P.srcA, ypl, (unsigned)c * 256u);
            r.u[1][t] = bld_nt(P.srcB, ypl, (unsigned)c * 256u);
        }
    }
}

// the two scanned quantities of column t from the raw operands (same arithmetic as Source::load)
template <int MODE>
__device__ __forceinline__ void raw_eval(const RawRow<MODE>& r, int t, const CostConst& cc, float& v0,
                                         float& v1) {
    if (MODE == GUID) {
        v0 = (float)__builtin_bit_cast(fg_t, r.u[0][t]).x;
        v1 = v0 * v0;
    } else if (MODE == S1) {
        cost_pair(__builtin_bit_cast(fg_t, r.u[0][t]), __builtin_bit_cast(fg_t, r.u[1][t]), cc, v0, v1);
    } else {
        v0 = __builtin_bit_cast(float, r.u[0][t]);
        v1 = __builtin_bit_cast(float, r.u[1][t]);
    }
}

template <int MODE>
__global__ __launch_bounds__(NTHREADS, 4) void k_v2_walk(Launch L) {
    // flat allocation with a small tail pad: the batched phase-B reads of masked-off columns may
    // run up to PB*NSUB + 2R + 1 floats past the last ring row
    __shared__ float ring_s[2 * RR * PITCH + 64];
    float (*ring)[RR][PITCH] = reinterpret_cast<float (*)[RR][PITCH]>(ring_s);
    // XCD-aware work order (speed only, never correctness): workgroups are dealt round-robin over
    // the 8 XCDs, so give each XCD a contiguous range of the strip-major item order.  The workgroups
    // resident on one XCD then sit in one or two strips and re-read the same image / guidance columns
    // from that XCD's L2 instead of thrashing it with all strips (bijective remap of the guide, T1).
    const int nslices = L.nslices, nviews = L.nviews;
    const int T = (int)gridDim.x;
    const int q8 = T >> 3, r8 = T & 7;
    const int xcd = (int)blockIdx.x & 7, jx = (int)blockIdx.x >> 3;
    const int item = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + jx;
    const int per_strip = nslices * nviews;
    const int k = item / per_strip;
    const int rem = item - k * per_strip;
    const int slice = rem / nviews;
    const Args& a = L.v[rem - slice * nviews];
    // readfirstlane: the wave index is uniform but the compiler cannot know -- without it every
    // per-column pointer and bounds test below is computed per lane in VGPRs
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int R = a.R, w = a.w, h = a.h, hp = a.hp, ow = a.ow;
    const int xs_ = k * ow, cs_ = xs_ - (R + 1);
    const int dj = 2 * R + 1;
    const size_t plane = (size_t)w * hp;
    const int nbands = (h + BH - 1) / BH;
    const CostConst cc = a.cc;
    // strips whose every output column has an unclipped window in x take the fast phase B
    const bool x_interior = (cs_ >= 0) && (xs_ + ow - 1 + R <= w - 1);
    // phase C ownership: thread -> (integral, column of the tile)
    const int ci = tid / TW, cj = tid - ci * TW;
    float S = -0.0f;
    const size_t pbytes = plane * sizeof(float);
    const size_t fgbytes = (size_t)(w + 2) * hp * sizeof(fg_t);
    Planes P;
    {
        const fg_t* lo = a.FG1 < a.FG2 ? a.FG1 : a.FG2;
        const fg_t* hi = a.FG1 < a.FG2 ? a.FG2 : a.FG1;
        P.img = mk_rsrc(lo, (size_t)((const char*)hi - (const char*)lo) + fgbytes);
        P.fg1o = (unsigned)((const char*)a.FG1 - (const char*)lo);
        P.fg2o = (unsigned)((const char*)a.FG2 - (const char*)lo);
    }
    P.stat = mk_rsrc(MODE == S1 ? a.meanT : a.dstA,
                     MODE == S1 ? (size_t)((const char*)a.cinvT - (const char*)a.meanT) + pbytes : pbytes);
    P.cinvo = MODE == S1 ? (unsigned)((const char*)a.cinvT - (const char*)a.meanT) : 0u;
    P.srcA = mk_rsrc(MODE == S2 ? a.srcA + (size_t)slice * plane : a.dstA, pbytes);
    P.srcB = mk_rsrc(MODE == S2 ? a.srcB + (size_t)slice * plane : a.dstA, pbytes);
    P.dstA = mk_rsrc(MODE == GUID ? a.dstA : a.dstA + (size_t)slice * plane, pbytes);
    P.dstB = mk_rsrc(MODE == GUID ? a.dstB : (MODE == S1 ? a.dstB + (size_t)slice * plane : a.dstA), pbytes);
    // this wave's row-carry columns (sub-strip `wave` of strip k): integral 0, integral 1 behind it
    P.car1o = (unsigned)((size_t)a.nsegs * hp * sizeof(float));
    P.car = mk_rsrc(a.carry + ((size_t)(slice * 2 + 0) * a.nsegs + NSUB * k + wave) * hp,
                    (size_t)P.car1o + (size_t)hp * 4);
    const int dsl = a.d0 + slice;
    const int j0 = wave * SUBW;
    const int cbeg_ = cs_ + j0;

    // sub-strip (and for S1 its shifted partner window) entirely inside the image: straight-line
    // phase R with immediate offsets (wave-uniform)
    const bool fastR = (cbeg_ >= 0) && (cbeg_ + SUBW <= w) &&
                       (MODE != S1 || ((cbeg_ + dsl >= -1) && (cbeg_ + SUBW - 1 + dsl <= w)));
    // operands of phase R are loaded one band ahead (their latency hides behind phase B)
    RawRow<MODE> raw;
    float cin0, cin1;
    {
        const int y = min(lane, h - 1);
        cin0 = bldf(P.car, (unsigned)y * 4u, 0);
        cin1 = bldf(P.car, (unsigned)y * 4u, P.car1o);
        if (fastR) raw_load<MODE, true>(P, w, dsl, blane(y, w + 2) * 4u, blane(y, w) * 4u, cbeg_, raw);
        else raw_load<MODE, false>(P, w, dsl, blane(y, w + 2) * 4u, blane(y, w) * 4u, cbeg_, raw);
    }

    for (int b = 0; b < nbands; ++b) {
        const int y0 = b * BH;
        const int rows = min(BH, h - y0);
        // The unrolled phases below have ~100 wave-uniform per-column values (bounds tests, byte
        // offsets).  They are loop invariant, so the compiler would keep them all live across the
        // band loop and spill SGPRs to VGPR lanes (v_writelane/v_readlane in the hot loop).  Passing
        // the column origins through an empty asm once per band makes them cheap to recompute instead.
        int cbeg = cbeg_; int xs = xs_; int cs = cs_;
        asm volatile("" : "+s"(cbeg), "+s"(xs), "+s"(cs));
        // ---------------- phase R: LANE = ROW, wave -> sub-strip --------------------------------
        if (lane < rows) {
            const int rr = ring_row(y0 + lane);
            float* r0 = &ring[0][rr][j0];
            float* r1 = &ring[1][rr][j0];
            float acc0 = cin0, acc1 = cin1;
            if (fastR) {
#pragma unroll
                for (int t = 0; t < SUBW; ++t) {
                    float v0, v1;
                    raw_eval<MODE>(raw, t, cc, v0, v1);
                    acc0 = v0 + acc0;
                    acc1 = v1 + acc1;
                    r0[t] = acc0;
                    r1[t] = acc1;
                }
            } else {
#pragma unroll
                for (int t = 0; t < SUBW; ++t) {
                    const int c = cbeg + t;
                    if (c >= 0 && c < w) {
                        float v0, v1;
                        raw_eval<MODE>(raw, t, cc, v0, v1);
                        acc0 = v0 + acc0;
                        acc1 = v1 + acc1;
                        r0[t] = acc0;
                        r1[t] = acc1;
                    }
                }
            }
        }
        // phase B geometry of this lane (first 64 output rows of the band)
        const int ylo = (b == 0) ? 0 : y0 - R;
        const int yhi = (b == nbands - 1) ? h : y0 + BH - R;
        // global operands of phase B for all of this wave's columns: in flight during phase C
        float gA[NCB], gB[NCB];
        if (x_interior && MODE != GUID) {
            const int yoc = min(ylo + lane, yhi - 1);
            const unsigned yob = blane(yoc, w) * 4u, yof = blane(yoc, w + 2) * 4u;
#pragma unroll
            for (int i = 0; i < NCB; ++i) {
                const int m = wave + i * NSUB;
                const unsigned cob = (unsigned)(xs + (m < ow ? m : 0)) * 256u;   // uniform
                if (MODE == S1) { gA[i] = bldf(P.stat, yob, cob); gB[i] = bldf(P.stat, yob, P.cinvo + cob); }
                if (MODE == S2) {
                    gA[i] = (float)__builtin_bit_cast(fg_t, bld(P.img, yof, P.fg1o + cob + 256u)).x;
                    gB[i] = 0.0f;
                }
            }
        }
        WALK_SYNC();
        // ---------------- phase C: LANE = COLUMN, R -> S in place ------------------------------
        if (ci < 2 && SMX_EXP != 5) {
            // 64 dependent adds per column in batches of 8 rows.  y0 and RR are multiples of 8, so a
            // batch never straddles the ring wrap: one base address per batch, immediate offsets for
            // its rows.  Where registers allow (not in S1, which would spill) two batches ping-pong:
            // the LDS reads of the next batch are issued before the adds of the current one.
            constexpr int CBT = 8;
            constexpr bool AHEAD = MODE != S1;
            float* colp = &ring[ci][0][cj];
            int rr = ring_row(y0);
            const int nb = rows / CBT;
            auto next_batch = [&]() {
                float* bp = colp + rr * PITCH;
                rr = (rr + CBT >= RR) ? rr + CBT - RR : rr + CBT;
                return bp;
            };
            auto rd = [&](float* bp, float (&v)[CBT]) {
#pragma unroll
                for (int t = 0; t < CBT; ++t) v[t] = bp[t * PITCH];
            };
            auto acc = [&](float* bp, const float (&v)[CBT]) {
#pragma unroll
                for (int t = 0; t < CBT; ++t) {
                    S = v[t] + S;
                    bp[t * PITCH] = S;
                }
            };
            float va[CBT], vb[CBT];
            if (AHEAD) {
                float *pa = nullptr, *pb = nullptr;
                if (nb > 0) { pa = next_batch(); rd(pa, va); }
                for (int i = 0; i < nb; i += 2) {
                    if (i + 1 < nb) { pb = next_batch(); rd(pb, vb); }
                    acc(pa, va);
                    if (i + 2 < nb) { pa = next_batch(); rd(pa, va); }
                    if (i + 1 < nb) acc(pb, vb);
                }
            } else {
                for (int i = 0; i < nb; ++i) {
                    float* pa = next_batch();
                    rd(pa, va);
                    acc(pa, va);
                }
            }
            for (int r = nb * CBT; r < rows; ++r) {
                float v = colp[rr * PITCH];
                S = v + S;
                colp[rr * PITCH] = S;
                rr = (rr + 1 == RR) ? 0 : rr + 1;
            }
        }
        // next band's phase-R operands: in flight during phase B
        if (b + 1 < nbands) {
            const int y = min(y0 + BH + lane, h - 1);
            cin0 = bldf(P.car, (unsigned)y * 4u, 0);
            cin1 = bldf(P.car, (unsigned)y * 4u, P.car1o);
            if (fastR) raw_load<MODE, true>(P, w, dsl, blane(y, w + 2) * 4u, blane(y, w) * 4u, cbeg, raw);
            else raw_load<MODE, false>(P, w, dsl, blane(y, w + 2) * 4u, blane(y, w) * 4u, cbeg, raw);
        }
        WALK_SYNC();
        // ---------------- phase B: LANE = ROW, box means + stage arithmetic --------------------
#if SMX_EXP == 2   // timing experiment: no phase B
        if (gA[0] == 12345.0f)
#endif
        for (int yo = ylo + lane, pass = 0; yo < yhi; yo += 64, ++pass) {
            const int ymax = min(h - 1, yo + R);
            const int ymin = yo - R - 1;
            const bool hy = ymin >= 0;
            const int ych = ymax - (hy ? ymin : -1);
            const int rr1 = ring_row(ymax);
            const int rr0 = ring_row(hy ? ymin : 0);
            if (x_interior) {
                // window width is 2R+1 for every column of the strip: per-lane area and reciprocal
                const float area = (float)(dj * ych);
                const float rarea = 1.0f / area;
                const float* b1 = &ring[0][rr1][wave];
                const float* b0 = &ring[0][rr0][wave];
                const unsigned cob0 = (unsigned)(xs + wave) * 256u;   // uniform
                const unsigned uyo = (unsigned)yo;
                const unsigned ypl = blane(yo, w) * 4u, yfg = blane(yo, w + 2) * 4u;
#pragma unroll
                for (int ib = 0; ib < NBATCH; ++ib) {
                    const int mb = wave + ib * PB * NSUB;
                    if (mb < ow) {
                        const float* p1 = b1 + ib * PB * NSUB;
                        const float* p0 = b0 + ib * PB * NSUB;
                        float s11[PB], s10[PB], s01[PB], s00[PB], u11[PB], u10[PB], u01[PB], u00[PB];
#pragma unroll
                        for (int t = 0; t < PB; ++t) {
                            s10[t] = p1[t * NSUB];                s11[t] = p1[t * NSUB + dj];
                            s00[t] = p0[t * NSUB];                s01[t] = p0[t * NSUB + dj];
                            u10[t] = p1[t * NSUB + RR * PITCH];   u11[t] = p1[t * NSUB + dj + RR * PITCH];
                            u00[t] = p0[t * NSUB + RR * PITCH];   u01[t] = p0[t * NSUB + dj + RR * PITCH];
                        }
#pragma unroll
                        for (int t = 0; t < PB; ++t) {
                            const int i = ib * PB + t;
                            if (i < NCB && mb + t * NSUB < ow) {
                                float val0 = s11[t] - s10[t];
                                float val1 = u11[t] - u10[t];
                                float t0 = val0 - s01[t], t1 = val1 - u01[t];
                                val0 = hy ? t0 : val0;  val1 = hy ? t1 : val1;
                                t0 = val0 + s00[t];     t1 = val1 + u00[t];
                                val0 = hy ? t0 : val0;  val1 = hy ? t1 : val1;
                                float m0 = div_small_int(val0, area, rarea);
                                float m1 = div_small_int(val1, area, rarea);
                                const bool tiny = (fabsf(val0) < 0x1p-100f) || (fabsf(val1) < 0x1p-100f);
                                if (__any(tiny)) {
                                    // rare (window sums that are exactly 0 or denormal-small).  The empty
                                    // asm keeps this a real wave-uniform branch: without it hipcc
                                    // if-converts and runs both IEEE divisions for every column.
                                    asm volatile("; exact-division slow path");
                                    m0 = 1.0f * val0 / area;
                                    m1 = 1.0f * val1 / area;
                                }
                                const unsigned cob = cob0 + (unsigned)(i * NSUB) * 256u;   // uniform
                                float ga = 0.0f, gb = 0.0f;
                                if (MODE != GUID) {
                                    if (pass == 0) { ga = gA[i < NCB ? i : 0]; gb = gB[i < NCB ? i : 0]; }
                                    else if (MODE == S1) { ga = bldf(P.stat, ypl, cob); gb = bldf(P.stat, ypl, P.cinvo + cob); }
                                    else { ga = (float)__builtin_bit_cast(fg_t, bld(P.img, yfg, P.fg1o + cob + 256u)).x; }
                                }
                                stage_out<MODE>(a, P, m0, m1, ga, gb, xs + mb + t * NSUB, uyo, ypl, cob);
                            }
                        }
                    }
                    // keep the batches apart: without this the scheduler hoists every batch's LDS
                    // reads to the top of the unrolled loop and spills
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                const unsigned ypl = blane(yo, w) * 4u, yfg = blane(yo, w + 2) * 4u;
                for (int mb = wave; mb < ow; mb += NSUB * PB) {
                    float ga[PB], gb[PB];
#pragma unroll
                    for (int t = 0; t < PB; ++t) {
                        const int m = mb + t * NSUB;
                        int xo = xs + m;
                        xo = (m < ow && xo < w) ? xo : (w - 1);
                        const unsigned cob = (unsigned)xo * 256u;   // uniform
                        ga[t] = 0.0f; gb[t] = 0.0f;
                        if (MODE == S1) { ga[t] = bldf(P.stat, ypl, cob); gb[t] = bldf(P.stat, ypl, P.cinvo + cob); }
                        if (MODE == S2) { ga[t] = (float)__builtin_bit_cast(fg_t, bld(P.img, yfg, P.fg1o + cob + 256u)).x; }
                    }
#pragma unroll
                    for (int t = 0; t < PB; ++t) {
                        const int m = mb + t * NSUB;
                        const int xo = xs + m;
                        if (m < ow && xo < w) {
                            const int xmax = min(w - 1, xo + R);
                            const int jmax = xmax - cs;
                            const bool hx = (xo - R - 1) >= 0;
                            const int jmin = m;  // (xo - R - 1) - cs
                            const int xcw = xmax - (hx ? xo - R - 1 : -1);
                            const float area = (float)(xcw * ych);
                            float val0 = ring[0][rr1][jmax];
                            float val1 = ring[1][rr1][jmax];
                            if (hx) { val0 -= ring[0][rr1][jmin]; val1 -= ring[1][rr1][jmin]; }
                            if (hy) { val0 -= ring[0][rr0][jmax]; val1 -= ring[1][rr0][jmax]; }
                            if (hx && hy) { val0 += ring[0][rr0][jmin]; val1 += ring[1][rr0][jmin]; }
                            const float m0 = 1.0f * val0 / area;
                            const float m1 = 1.0f * val1 / area;
                            stage_out<MODE>(a, P, m0, m1, ga[t], gb[t], xo, (unsigned)yo, ypl,
                                            (unsigned)xo * 256u);
                        }
                    }
                }
            }
        }
        WALK_SYNC();
    }
}

// ---------------------------------------------------------------------------------------------
// WTA over the chunk's q planes (transposed).  One lane per pixel, y fastest.
// grid (ceil(h/64), w, nviews)
// ---------------------------------------------------------------------------------------------
struct WtaArgs {
    const float* qT[2];
    uint64_t* keys[2];
};

__global__ __launch_bounds__(64) void k_v2_wta(WtaArgs wa, int w, int h, int hp, int count,
                                               int slice0) {
    const int y = blockIdx.x * 64 + threadIdx.x;
    const int x = blockIdx.y;
    if (y >= h) return;
    const size_t plane = (size_t)w * hp;
    const float* __restrict__ q = wa.qT[blockIdx.z] + blane(y, w) + bcol(x);
    uint64_t* keys = wa.keys[blockIdx.z];
    const size_t id = (size_t)y * w + x;
    uint64_t key = keys[id];
    int z = 0;
    for (; z + 8 <= count; z += 8) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = __builtin_nontemporal_load(&q[(size_t)(z + t) * plane]);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            uint64_t kk = pack_key(v[t], (uint32_t)(slice0 + z + t));
            key = kk < key ? kk : key;
        }
    }
    for (; z < count; ++z) {
        uint64_t kk = pack_key(__builtin_nontemporal_load(&q[(size_t)z * plane]), (uint32_t)(slice0 + z));
        key = kk < key ? kk : key;
    }
    keys[id] = key;
}

// qT [slice][x*hp + y] -> agg [slice][y*w + x]   (only when the caller asks for the volume)
__global__ void k_v2_untranspose(const float* __restrict__ qT, float* __restrict__ agg, int w, int h,
                                 int hp) {
    __shared__ float t[32][33];
    const size_t planeT = (size_t)w * hp, plane = (size_t)w * h;
    const int z = blockIdx.z;
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
    for (int i = ty; i < 32; i += 8) {
        int x = x0 + i, y = y0 + tx;
        if (x < w && y < h) t[i][tx] = qT[(size_t)z * planeT + blane(y, w) + bcol(x)];
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        int y = y0 + i, x = x0 + tx;
        if (x < w && y < h) agg[(size_t)z * plane + (size_t)y * w + x] = t[tx][i];
    }
}

}  // namespace v2

// =============================================================================================
// host orchestration
// =============================================================================================
static inline unsigned cdivu(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

struct V2Layout {
    int hp, ow, nstrips, nsegs;
    size_t padded_plane;   // (w+2)*hp floats
    size_t plane;          // w*hp floats
    size_t carry_slice;    // 2*nsegs*hp floats
};

static V2Layout v2_layout(int w, int h, int R) {
    V2Layout L;
    L.hp = (h + 63) / 64 * 64;
    L.ow = v2::TW - (2 * R + 1);
    L.nstrips = (w + L.ow - 1) / L.ow;
    L.nsegs = v2::NSUB * L.nstrips;
    L.padded_plane = (size_t)(w + 2) * L.hp;
    L.plane = (size_t)w * L.hp;
    L.carry_slice = (size_t)2 * L.nsegs * L.hp;
    return L;
}

bool v2_supported(const smx_params* p) { return p->radius >= 0 && p->radius <= v2::RMAX; }

// bytes for ONE view with `nslices` slices in flight
size_t v2_workspace_bytes(int w, int h, int R, int nslices) {
    (void)R;  // the largest radius has the narrowest strips, i.e. the most carry segments
    V2Layout L9 = v2_layout(w, h, v2::RMAX);
    size_t fl = 2 * L9.padded_plane + 2 * L9.plane + L9.carry_slice + 2 * (size_t)L9.nsegs +
                (size_t)nslices * (3 * L9.plane + L9.carry_slice);
    return fl * sizeof(float) + 24 * 256;
}

// side stream used to overlap the (latency-bound, single-slice) guidance stage with the stage-1
// carry prepass; fork/join with events so the caller's stream semantics are preserved
static thread_local hipStream_t g_side = nullptr;
static thread_local hipEvent_t g_fork = nullptr, g_join = nullptr;

// Slice sub-chunks of one call are software-pipelined over NPIPE streams: the latency-bound carry
// prepasses of one sub-chunk run under the issue-bound walkers of another (a walker workgroup leaves
// half of a CU's wave slots free).  Fork/join by events, so the caller still sees one stream.
constexpr int NPIPE = 3;
constexpr int MAXSUB = 32;
static thread_local hipStream_t g_pipe[NPIPE] = {nullptr, nullptr, nullptr};
static thread_local hipEvent_t g_ev_wta[MAXSUB], g_ev_done[NPIPE], g_ev_fork2 = nullptr;
static int g_pipeline_subchunks = 1;   // 0/1 = no pipelining (measured: pipelining 2-8 sub-chunks is 6-37 % slower on KITTI shape)

static int ensure_side_stream() {
    if (!g_side) {
        SMX_HIP(hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking));
        SMX_HIP(hipEventCreateWithFlags(&g_fork, hipEventDisableTiming));
        SMX_HIP(hipEventCreateWithFlags(&g_join, hipEventDisableTiming));
        for (int i = 0; i < NPIPE; ++i) {
            SMX_HIP(hipStreamCreateWithFlags(&g_pipe[i], hipStreamNonBlocking));
            SMX_HIP(hipEventCreateWithFlags(&g_ev_done[i], hipEventDisableTiming));
        }
        for (int i = 0; i < MAXSUB; ++i) SMX_HIP(hipEventCreateWithFlags(&g_ev_wta[i], hipEventDisableTiming));
        SMX_HIP(hipEventCreateWithFlags(&g_ev_fork2, hipEventDisableTiming));
    }
    return SMX_OK;
}

void v2_set_pipeline(int subchunks) { g_pipeline_subchunks = subchunks < 0 ? 0 : (subchunks > MAXSUB ? MAXSUB : subchunks); }

template <int MODE>
static int launch_carry(const v2::Launch& L, int nslices, int nviews, hipStream_t st) {
    const v2::Args& a = L.v[0];
    hipLaunchKernelGGL(v2::k_v2_carry<MODE>, dim3((a.h + 63) / 64, nslices, nviews), dim3(64), 0, st, L);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

template <int MODE>
static int launch_walk(const v2::Launch& L, int nslices, int nviews, hipStream_t st) {
    const v2::Args& a = L.v[0];
    v2::Launch LL = L;
    LL.nslices = nslices;
    LL.nviews = nviews;
    hipLaunchKernelGGL(v2::k_v2_walk<MODE>, dim3((unsigned)(a.nstrips * nslices * nviews)), dim3(v2::NTHREADS),
                       0, st, LL);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

// Aggregation + WTA of slices [s_begin, s_end) of `nviews` (1 or 2) views with the cost built on
// the fly.  View v uses d_img[v] as guidance and d_img[v ^ 1] (nviews == 2) or d_other as the
// matching image; keys/mean/agg pointers are per view.
int aggregate_v2(const smx_params* p, int nviews, const uint8_t* const* d_guide,
                 const uint8_t* const* d_other, int w, int h, const int* dmin, int s_begin, int s_end,
                 uint64_t* const* d_keys, uint8_t* const* d_mean_u8, float* const* d_agg, void* d_ws,
                 size_t ws_bytes, hipStream_t st, int* launches) {
    const int R = p->radius;
    V2Layout L = v2_layout(w, h, R);
    char* base = (char*)align_up((size_t)d_ws, 256);
    size_t avail = ws_bytes > (size_t)(base - (char*)d_ws) ? ws_bytes - (size_t)(base - (char*)d_ws) : 0;
    bool oom = false;
    auto carve = [&](size_t nfloat) {
        float* r = (float*)base;
        size_t b = align_up(nfloat * sizeof(float), 256);
        if (b > avail) { oom = true; b = avail; }
        base += b;
        avail -= b;
        return r;
    };
    const size_t per_slice = (3 * L.plane + L.carry_slice) * sizeof(float);
    // fixed planes: the two images' F/G are shared by both views of a pair
    const int nimg = 2;
    v2::fg_t* FG[2];
    float *meanT[2], *cinvT[2], *gcarry[2];
    for (int i = 0; i < nimg; ++i) FG[i] = (v2::fg_t*)carve(L.padded_plane);
    for (int v = 0; v < nviews; ++v) {
        meanT[v] = carve(L.plane); cinvT[v] = carve(L.plane); gcarry[v] = carve(L.carry_slice);
    }
    int* ev_start = (int*)carve((size_t)L.nsegs);
    int* ev_seg = (int*)carve((size_t)L.nsegs);
    const int total = s_end - s_begin;
    size_t fit = avail > 8 * 256 ? (avail - 8 * 256) / (per_slice * nviews) : 0;
    if (oom || fit < 1)
        return fail(SMX_E_WS, "aggregate_v2: workspace %zu B too small (need >= %zu B per view)",
                    ws_bytes, v2_workspace_bytes(w, h, R, 1));
    int chunk = fit > (size_t)total ? total : (int)fit;
    if (chunk < 1) chunk = 1;
    float *aT[2], *bT[2], *qT[2], *carry[2];
    for (int v = 0; v < nviews; ++v) {
        aT[v] = carve((size_t)chunk * L.plane);
        bT[v] = carve((size_t)chunk * L.plane);
        qT[v] = carve((size_t)chunk * L.plane);
        carry[v] = carve((size_t)chunk * L.carry_slice);
    }
    if (oom) return fail(SMX_E_WS, "aggregate_v2: workspace carve overflow");

    int nl = 0, rc;
    // image 0 = guide of view 0; image 1 = the other image (guide of view 1 in a pair)
    v2::PrepArgs pa;
    pa.I[0] = d_guide[0]; pa.I[1] = nviews == 2 ? d_guide[1] : d_other[0];
    pa.FG[0] = FG[0]; pa.FG[1] = FG[1];
    hipLaunchKernelGGL(v2::k_v2_prep, dim3(L.hp / 64, w + 2, 2), dim3(64), 0, st, pa, w, h, L.hp);
    SMX_HIP(hipGetLastError());
    hipLaunchKernelGGL(v2::k_v2_events, dim3(1), dim3(256), 0, st, L.nsegs, L.ow, R, ev_start, ev_seg);
    SMX_HIP(hipGetLastError());
    nl += 2;

    v2::Launch base_l;
    memset(&base_l, 0, sizeof(base_l));
    for (int v = 0; v < nviews; ++v) {
        v2::Args& a = base_l.v[v];
        a.w = w; a.h = h; a.hp = L.hp; a.R = R; a.ow = L.ow; a.nstrips = L.nstrips; a.nsegs = L.nsegs;
        a.FG1 = FG[v]; a.FG2 = FG[v ^ 1];
        a.cc = make_cost_const(p);
        a.eps = p->eps;
        a.ev_start = ev_start; a.ev_seg = ev_seg;
    }
    // ---- guidance statistics on the side stream, overlapped with the first stage-1 carry prepass
    if ((rc = ensure_side_stream())) return rc;
    SMX_HIP(hipEventRecord(g_fork, st));
    SMX_HIP(hipStreamWaitEvent(g_side, g_fork, 0));
    {
        v2::Launch g = base_l;
        for (int v = 0; v < nviews; ++v) {
            g.v[v].dstA = meanT[v]; g.v[v].dstB = cinvT[v];
            g.v[v].mean_u8 = d_mean_u8 ? d_mean_u8[v] : nullptr;
            g.v[v].carry = gcarry[v];
        }
        if ((rc = launch_carry<v2::GUID>(g, 1, nviews, g_side))) return rc;
        if ((rc = launch_walk<v2::GUID>(g, 1, nviews, g_side))) return rc;
        nl += 2;
    }
    SMX_HIP(hipEventRecord(g_join, g_side));
    bool joined = false;
    for (int v = 0; v < nviews; ++v) { base_l.v[v].meanT = meanT[v]; base_l.v[v].cinvT = cinvT[v]; }

    for (int s0 = s_begin; s0 < s_end; s0 += chunk) {
        const int cnt = (s_end - s0) < chunk ? (s_end - s0) : chunk;
        // sub-chunks of this workspace chunk, pipelined over the side streams
        int nsub = g_pipeline_subchunks;
        if (nsub > cnt / 8) nsub = cnt / 8;          // keep >= 8 slices per sub-chunk
        if (nsub < 2) nsub = 1;
        const bool piped = nsub > 1;
        if (piped) SMX_HIP(hipEventRecord(g_ev_fork2, st));
        bool used[NPIPE] = {false, false, false};
        for (int j = 0; j < nsub; ++j) {
            const int t0 = s0 + (int)((int64_t)cnt * j / nsub);
            const int t1 = s0 + (int)((int64_t)cnt * (j + 1) / nsub);
            const int tc = t1 - t0;
            const size_t off = (size_t)(t0 - s0);
            hipStream_t sj = piped ? g_pipe[j % NPIPE] : st;
            if (piped && !used[j % NPIPE]) {
                SMX_HIP(hipStreamWaitEvent(sj, g_ev_fork2, 0));
                SMX_HIP(hipStreamWaitEvent(sj, g_join, 0));     // guidance statistics ready
                used[j % NPIPE] = true;
            }
            v2::Launch s1 = base_l, s2 = base_l;
            v2::WtaArgs wa;
            for (int v = 0; v < 2; ++v) {
                const int vv = v < nviews ? v : 0;
                float* av = aT[vv] + off * L.plane;
                float* bv = bT[vv] + off * L.plane;
                float* qv = qT[vv] + off * L.plane;
                float* cv = carry[vv] + off * L.carry_slice;
                if (v < nviews) {
                    s1.v[v].d0 = dmin[v] + t0; s1.v[v].dstA = av; s1.v[v].dstB = bv; s1.v[v].carry = cv;
                    s2.v[v].srcA = av; s2.v[v].srcB = bv; s2.v[v].dstA = qv; s2.v[v].carry = cv;
                }
                wa.qT[v] = qv;
                wa.keys[v] = d_keys[vv];
            }
            if ((rc = launch_carry<v2::S1>(s1, tc, nviews, sj))) return rc;
            if (!piped && !joined) {
                SMX_HIP(hipStreamWaitEvent(st, g_join, 0));
                joined = true;
            }
            if ((rc = launch_walk<v2::S1>(s1, tc, nviews, sj))) return rc;
            if ((rc = launch_carry<v2::S2>(s2, tc, nviews, sj))) return rc;
            if ((rc = launch_walk<v2::S2>(s2, tc, nviews, sj))) return rc;
            // the key update is a read-modify-write of the same buffer: keep sub-chunks in order
            if (piped && j > 0) SMX_HIP(hipStreamWaitEvent(sj, g_ev_wta[j - 1], 0));
            hipLaunchKernelGGL(v2::k_v2_wta, dim3(cdivu(h, 64), w, nviews), dim3(64), 0, sj, wa, w, h,
                               L.hp, tc, t0);
            SMX_HIP(hipGetLastError());
            if (piped) SMX_HIP(hipEventRecord(g_ev_wta[j], sj));
            nl += 5;
            for (int v = 0; v < nviews; ++v) {
                if (d_agg && d_agg[v]) {
                    dim3 tg(cdivu(w, 32), cdivu(h, 32), tc);
                    hipLaunchKernelGGL(v2::k_v2_untranspose, tg, dim3(256), 0, sj, wa.qT[v],
                                       d_agg[v] + (size_t)(t0 - s_begin) * w * h, w, h, L.hp);
                    SMX_HIP(hipGetLastError());
                    ++nl;
                }
            }
        }
        if (piped) {
            joined = true;   // every pipeline stream waited for the guidance join
            for (int i = 0; i < NPIPE; ++i) {
                if (!used[i]) continue;
                SMX_HIP(hipEventRecord(g_ev_done[i], g_pipe[i]));
                SMX_HIP(hipStreamWaitEvent(st, g_ev_done[i], 0));
            }
        }
    }
    if (!joined) SMX_HIP(hipStreamWaitEvent(st, g_join, 0));
    if (launches) *launches = nl;
    return SMX_OK;
}

}  // namespace smx
