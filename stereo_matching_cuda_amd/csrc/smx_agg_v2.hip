// smx_agg_v2.hip -- fused guided-filter aggregation for gfx950 ("carry prepass + strip walker").
//
// Data flow per view (reference: guidedFilter.cu:58-238; cost: costVolume.cu:163-190):
//
//   prep      u8 images -> transposed f32 planes F_T[x][y], G_T[x][y] (value, x-gradient) with one
//             sentinel column on each side (out-of-range disparities read 1e9 -> both truncated
//             terms saturate -> exactly the reference's border constant).
//   carry<M>  one wave per (slice, 64-row band), LANE = ROW: walks the full row left->right with the
//             reference's sequential f32 adds and stores the running row sums at every sub-strip
//             start ("carries").  No LDS, output is ~3 % of a plane.
//   walk<M>   one 256-thread workgroup per (slice, strip of TW columns), walking 64-row bands top->
//             bottom entirely in LDS:
//               phase R  LANE = ROW   : each wave continues the row scan of one SUBW-column sub-strip
//                                       from its carry, writing R into the LDS ring (strided, odd
//                                       pitch -> conflict-free)
//               phase C  LANE = COLUMN: sequential column scan down the band, S kept in a register
//                                       per column across bands, R -> S in place in the ring
//               phase B  LANE = ROW   : box means from the ring (4 taps, reference order, IEEE
//                                       division) + the per-pixel arithmetic of the stage; outputs
//                                       are written transposed ([x][y]) so the next stage's LANE=ROW
//                                       readers are coalesced
//             The addition order of every prefix sum is exactly the reference's (integral.cu:82-86,
//             124-128); strips overlap by 2R+1 recomputed columns instead of exchanging halos.
//   modes     GUID: (I, I*I)      -> mean_I, 1/(var+eps)      (guidedFilter.cu:58-123)
//             S1  : (p, I*p)      -> a_k, b_k   with p built on the fly (costVolume.cu:184-189,
//                                                               guidedFilter.cu:198-223)
//             S2  : (a_k, b_k)    -> q = mean(a)*I + mean(b)    (guidedFilter.cu:224-233)
//   wta       one lane per pixel over the q planes of the chunk, packed-key min
//             (dispSelectOnGPU guidedFilter.cu:403-411).
//
// Must be compiled with -ffp-contract=off.
#include <string.h>

#include "smx_common.h"
#include "smx_launch.h"

namespace smx {
namespace v2 {

constexpr int TW = 104;            // columns of S computed per strip
constexpr int NSUB = 4;            // sub-strips per strip = waves per workgroup
constexpr int SUBW = TW / NSUB;    // 26
constexpr int BH = 64;             // band height = wave width
constexpr int RMAX = 9;            // largest supported box radius
constexpr int RR = 96;             // ring rows >= BH + 2*RMAX + 2; multiple of 32 so that the wrap
                                   // of a band inside the ring does not shift LDS banks
constexpr int PITCH = TW + 1;      // odd pitch: LANE=ROW accesses hit distinct banks
constexpr int RBATCH = 13;         // columns per load batch in phase R (SUBW = 2 batches)
constexpr int PB = 4;              // output columns per load batch in phase B
constexpr int CB = 16;             // columns per load batch of the carry prepass
static_assert(RR >= BH + 2 * RMAX + 2 && RR % 32 == 0, "ring too small");
static_assert(SUBW % RBATCH == 0 && 2 * TW <= 256, "tile geometry");

enum Mode { GUID = 0, S1 = 1, S2 = 2 };

struct Args {
    int w, h, hp, R, ow, nstrips, nsegs;
    const float* F1; const float* G1;   // this view: value / gradient, padded transposed planes
    const float* F2; const float* G2;   // other view (S1 only)
    const float* meanT; const float* cinvT;  // guidance statistics [x*hp + y] (S1 reads, GUID writes)
    const float* srcA; const float* srcB;    // S2 sources: aT, bT [slice][x*hp + y]
    float* dstA; float* dstB;           // GUID: meanT, cinvT; S1: aT, bT; S2: qT (dstB unused)
    uint8_t* mean_u8;                   // GUID optional, row-major [y*w + x]
    float* carry;                       // [(slice*2 + i)*nsegs + g][hp]
    int d0;                             // disparity of local slice 0
    CostConst cc;
    double eps;
};

// Both views of a pair travel in one launch: blockIdx.z selects the view.
struct Launch {
    Args v[2];
};

__device__ __forceinline__ int seg_c0(const Args& a, int g) {
    return (g >> 2) * a.ow - (a.R + 1) + (g & 3) * SUBW;
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

// The two scanned quantities of NB consecutive columns of row y (LANE = ROW; every load is a 256-B
// coalesced row of a transposed plane).  All global loads of a batch are issued before the first
// use, so one batch exposes one memory latency instead of NB.  Columns outside [0, w) are clamped
// (their values are never accumulated).
template <int MODE>
struct Source {
    const float *f1, *g1, *f2, *g2, *sa, *sb;
    int hp, w, d;
    CostConst cc;
    __device__ __forceinline__ Source(const Args& a, int slice, int y) {
        hp = a.hp; w = a.w; d = a.d0 + slice; cc = a.cc;
        f1 = a.F1 + y; g1 = a.G1 + y; f2 = a.F2 + y; g2 = a.G2 + y;
        const size_t plane = (size_t)a.w * a.hp;
        sa = MODE == S2 ? a.srcA + (size_t)slice * plane + y : nullptr;
        sb = MODE == S2 ? a.srcB + (size_t)slice * plane + y : nullptr;
    }
    template <int NB>
    __device__ __forceinline__ void load(int c0, float (&v0)[NB], float (&v1)[NB]) const {
        if (MODE == GUID) {
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                int c = c0 + t;
                c = c < 0 ? 0 : (c >= w ? w - 1 : c);
                v0[t] = f1[(size_t)(c + 1) * hp];      // chToFlOnGPU
            }
#pragma unroll
            for (int t = 0; t < NB; ++t) v1[t] = v0[t] * v0[t];   // pixelMultOnGPU(d_im, d_im)
        } else if (MODE == S1) {
            float a2[NB], b1[NB], b2[NB];
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                int c = c0 + t;
                c = c < 0 ? 0 : (c >= w ? w - 1 : c);
                int xx = c + d;
                xx = xx < -1 ? -1 : (xx > w ? w : xx);  // sentinel columns at -1 and w
                const size_t o1 = (size_t)(c + 1) * hp, o2 = (size_t)(xx + 1) * hp;
                v1[t] = f1[o1]; b1[t] = g1[o1]; a2[t] = f2[o2]; b2[t] = g2[o2];
            }
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                float t1 = fabsf(v1[t] - a2[t]);
                float t2 = fabsf(b1[t] - b2[t]);
                float m1 = t1 < cc.th_color ? t1 : cc.th_color;
                float m2 = t2 < cc.th_grad ? t2 : cc.th_grad;
                float x = cc.oma * m1;
                float z = cc.alpha * m2;
                float p = x + z;        // costVolume.cu:187
                v0[t] = p;
                v1[t] = v1[t] * p;      // pixelMultOnGPU(d_im, d_pki) guidedFilter.cu:209
            }
        } else {
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                int c = c0 + t;
                c = c < 0 ? 0 : (c >= w ? w - 1 : c);
                const size_t o = (size_t)c * hp;
                v0[t] = sa[o];
                v1[t] = sb[o];
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------
// prep: u8 image [h][w] -> F_T, G_T [(x+1)*hp + y], sentinel columns x = -1 and x = w.
// grid (ceil(hp/64), w + 2, nimages), block 64.
// ---------------------------------------------------------------------------------------------
struct PrepArgs {
    const uint8_t* I[2];
    float* F[2];
    float* G[2];
};

__global__ __launch_bounds__(64) void k_v2_prep(PrepArgs pa, int w, int h, int hp) {
    const uint8_t* __restrict__ I = pa.I[blockIdx.z];
    float* __restrict__ F = pa.F[blockIdx.z];
    float* __restrict__ G = pa.G[blockIdx.z];
    const int y = blockIdx.x * 64 + threadIdx.x;
    const int x = (int)blockIdx.y - 1;
    if (y >= hp) return;
    float f = 0.0f, g = 0.0f;
    if (x < 0 || x >= w) {
        f = 1e9f; g = 1e9f;
    } else if (y < h) {
        const uint8_t* row = I + (size_t)y * w;
        f = 1.0f * (float)(int)row[x];
        int c1, c2;  // x_derivativeOnGPU costVolume.cu:358-381
        if (x - 1 >= 0 && x + 1 < w) { c1 = row[x + 1]; c2 = row[x - 1]; }
        else if (x + 1 >= w)         { c1 = row[x];     c2 = row[x - 1]; }
        else                         { c1 = row[x + 1]; c2 = row[x];     }
        g = 1.0f * (float)(c2 - c1) / 2;
    }
    const size_t o = (size_t)(x + 1) * hp + y;
    F[o] = f;
    G[o] = g;
}

// ---------------------------------------------------------------------------------------------
// carry prepass.  grid (nbands, nslices, nviews), block 64 (LANE = ROW).
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(64) void k_v2_carry(Launch L) {
    const Args& a = L.v[blockIdx.z];
    const int slice = blockIdx.y;
    const int y = blockIdx.x * 64 + threadIdx.x;
    if (y >= a.h) return;
    Source<MODE> src(a, slice, y);
    float* c0p = a.carry + ((size_t)(slice * 2 + 0) * a.nsegs) * a.hp + y;
    float* c1p = a.carry + ((size_t)(slice * 2 + 1) * a.nsegs) * a.hp + y;
    float acc0 = -0.0f, acc1 = -0.0f;
    int g = 0;
    // segments that start at or left of column 0 begin with the additive identity
    while (g < a.nsegs && seg_c0(a, g) <= 0) {
        c0p[(size_t)g * a.hp] = acc0;
        c1p[(size_t)g * a.hp] = acc1;
        ++g;
    }
    int next = g < a.nsegs ? seg_c0(a, g) : 0x7fffffff;
    for (int cb = 0; cb < a.w; cb += CB) {
        float v0[CB], v1[CB];
        src.template load<CB>(cb, v0, v1);
#pragma unroll
        for (int t = 0; t < CB; ++t) {
            const int c = cb + t;
            if (c < a.w) {
                if (c == next) {   // wave-uniform: carry of segment g = row sum left of column c
                    c0p[(size_t)g * a.hp] = acc0;
                    c1p[(size_t)g * a.hp] = acc1;
                    ++g;
                    next = g < a.nsegs ? seg_c0(a, g) : 0x7fffffff;
                }
                acc0 = v0[t] + acc0;
                acc1 = v1[t] + acc1;
            }
        }
    }
    for (; g < a.nsegs; ++g) {     // segments starting at or beyond column w are never read
        c0p[(size_t)g * a.hp] = acc0;
        c1p[(size_t)g * a.hp] = acc1;
    }
}

// ---------------------------------------------------------------------------------------------
// strip walker.  grid (nstrips, nslices, nviews), block 256.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int ring_row(int y) { return y % RR; }

// per-pixel arithmetic of the stage on the two box means m0, m1
template <int MODE>
__device__ __forceinline__ void stage_out(const Args& a, float m0, float m1, float ga, float gb,
                                          float* __restrict__ dstA, float* __restrict__ dstB,
                                          size_t T, int xo, int yo) {
    if (MODE == GUID) {
        float mm = m0 * m0;          // pixelMultOnGPU(mean, mean) guidedFilter.cu:112
        float var = m1 - mm;         // pixelSousOnGPU :121
        float c = (float)(1.0f / ((double)var + a.eps));   // :350
        dstA[T] = m0;
        dstB[T] = c;
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
        dstA[T] = ak;
        dstB[T] = bk;
    } else {
        float tq = m0 * ga;          // compute_q guidedFilter.cu:363-369
        dstA[T] = tq + m1;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_v2_walk(Launch L) {
    // flat allocation with a small tail pad: the batched phase-B reads of masked-off columns may
    // run up to PB*NSUB + 2R + 1 floats past the last ring row
    __shared__ float ring_s[2 * RR * PITCH + 32];
    float (*ring)[RR][PITCH] = reinterpret_cast<float (*)[RR][PITCH]>(ring_s);
    const Args& a = L.v[blockIdx.z];
    const int k = blockIdx.x, slice = blockIdx.y;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int R = a.R, w = a.w, h = a.h, hp = a.hp, ow = a.ow;
    const int xs = k * ow, cs = xs - (R + 1);
    const int dj = 2 * R + 1;
    const size_t plane = (size_t)w * hp;
    const int nbands = (h + BH - 1) / BH;
    // strips whose every output column has an unclipped window in x take the fast phase B
    const bool x_interior = (cs >= 0) && (xs + ow - 1 + R <= w - 1);
    // phase C ownership: thread -> (integral, column of the tile)
    const int ci = tid / TW, cj = tid - ci * TW;
    float S = -0.0f;
    const float* __restrict__ meanT = a.meanT;
    const float* __restrict__ cinvT = a.cinvT;
    const float* __restrict__ srcF1 = a.F1;
    float* __restrict__ dstA = (MODE == GUID) ? a.dstA : a.dstA + (size_t)slice * plane;
    float* __restrict__ dstB = (MODE == GUID) ? a.dstB : (MODE == S1 ? a.dstB + (size_t)slice * plane : nullptr);

    for (int b = 0; b < nbands; ++b) {
        const int y0 = b * BH;
        const int rows = min(BH, h - y0);
        // ---------------- phase R: LANE = ROW, wave -> sub-strip --------------------------------
        if (lane < rows) {
            const int y = y0 + lane;
            const int g = 4 * k + wave;
            float acc0 = a.carry[((size_t)(slice * 2 + 0) * a.nsegs + g) * hp + y];
            float acc1 = a.carry[((size_t)(slice * 2 + 1) * a.nsegs + g) * hp + y];
            Source<MODE> src(a, slice, y);
            const int rr = ring_row(y);
            const int j0 = wave * SUBW;
            float* r0 = &ring[0][rr][j0];
            float* r1 = &ring[1][rr][j0];
            const int cbeg = cs + j0;
#pragma unroll
            for (int jb = 0; jb < SUBW; jb += RBATCH) {
                float v0[RBATCH], v1[RBATCH];
                src.template load<RBATCH>(cbeg + jb, v0, v1);
#pragma unroll
                for (int t = 0; t < RBATCH; ++t) {
                    const int c = cbeg + jb + t;
                    if (c >= 0 && c < w) {
                        acc0 = v0[t] + acc0;
                        acc1 = v1[t] + acc1;
                        r0[jb + t] = acc0;
                        r1[jb + t] = acc1;
                    }
                }
            }
        }
        __syncthreads();
        // ---------------- phase C: LANE = COLUMN, R -> S in place ------------------------------
        if (ci < 2) {
            int rr = ring_row(y0);
            int r = 0;
            for (; r + 8 <= rows; r += 8) {
                float v[8];
                int ro[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    ro[t] = rr;
                    v[t] = ring[ci][rr][cj];
                    rr = (rr + 1 == RR) ? 0 : rr + 1;
                }
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    S = v[t] + S;
                    ring[ci][ro[t]][cj] = S;
                }
            }
            for (; r < rows; ++r) {
                float v = ring[ci][rr][cj];
                S = v + S;
                ring[ci][rr][cj] = S;
                rr = (rr + 1 == RR) ? 0 : rr + 1;
            }
        }
        __syncthreads();
        // ---------------- phase B: LANE = ROW, box means + stage arithmetic --------------------
        const int ylo = (b == 0) ? 0 : y0 - R;
        const int yhi = (b == nbands - 1) ? h : y0 + BH - R;
        for (int yo = ylo + lane; yo < yhi; yo += 64) {
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
                const float* b1 = &ring[0][rr1][0];
                const float* b0 = &ring[0][rr0][0];
                for (int mb = wave; mb < ow; mb += NSUB * PB) {
                    const size_t Tb = (size_t)(xs + mb) * hp + yo;
                    float ga[PB], gb[PB];
#pragma unroll
                    for (int t = 0; t < PB; ++t) {
                        const bool ok = mb + t * NSUB < ow;
                        const size_t T = ok ? Tb + (size_t)(t * NSUB) * hp : Tb;
                        ga[t] = 0.0f; gb[t] = 0.0f;
                        if (MODE == S1) { ga[t] = meanT[T]; gb[t] = cinvT[T]; }
                        if (MODE == S2) { ga[t] = srcF1[T + hp]; }
                    }
                    const float* p1 = b1 + mb;
                    const float* p0 = b0 + mb;
                    float s11[PB], s10[PB], s01[PB], s00[PB], u11[PB], u10[PB], u01[PB], u00[PB];
#pragma unroll
                    for (int t = 0; t < PB; ++t) {
                        s10[t] = p1[t * NSUB];            s11[t] = p1[t * NSUB + dj];
                        s00[t] = p0[t * NSUB];            s01[t] = p0[t * NSUB + dj];
                        u10[t] = p1[t * NSUB + RR * PITCH];  u11[t] = p1[t * NSUB + dj + RR * PITCH];
                        u00[t] = p0[t * NSUB + RR * PITCH];  u01[t] = p0[t * NSUB + dj + RR * PITCH];
                    }
#pragma unroll
                    for (int t = 0; t < PB; ++t) {
                        if (mb + t * NSUB < ow) {
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
                                m0 = 1.0f * val0 / area;
                                m1 = 1.0f * val1 / area;
                            }
                            stage_out<MODE>(a, m0, m1, ga[t], gb[t], dstA, dstB,
                                            Tb + (size_t)(t * NSUB) * hp, xs + mb + t * NSUB, yo);
                        }
                    }
                }
            } else {
                for (int mb = wave; mb < ow; mb += NSUB * PB) {
                    float ga[PB], gb[PB];
#pragma unroll
                    for (int t = 0; t < PB; ++t) {
                        const int m = mb + t * NSUB;
                        int xo = xs + m;
                        xo = (m < ow && xo < w) ? xo : (w - 1);
                        const size_t T = (size_t)xo * hp + yo;
                        ga[t] = 0.0f; gb[t] = 0.0f;
                        if (MODE == S1) { ga[t] = meanT[T]; gb[t] = cinvT[T]; }
                        if (MODE == S2) { ga[t] = srcF1[T + hp]; }
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
                            stage_out<MODE>(a, m0, m1, ga[t], gb[t], dstA, dstB,
                                            (size_t)xo * hp + yo, xo, yo);
                        }
                    }
                }
            }
        }
        __syncthreads();
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
    const float* __restrict__ q = wa.qT[blockIdx.z] + (size_t)x * hp + y;
    uint64_t* keys = wa.keys[blockIdx.z];
    const size_t id = (size_t)y * w + x;
    uint64_t key = keys[id];
    int z = 0;
    for (; z + 8 <= count; z += 8) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = q[(size_t)(z + t) * plane];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            uint64_t kk = pack_key(v[t], (uint32_t)(slice0 + z + t));
            key = kk < key ? kk : key;
        }
    }
    for (; z < count; ++z) {
        uint64_t kk = pack_key(q[(size_t)z * plane], (uint32_t)(slice0 + z));
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
        if (x < w && y < h) t[i][tx] = qT[(size_t)z * planeT + (size_t)x * hp + y];
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
    L.nsegs = 4 * L.nstrips;
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
    size_t fl = 4 * L9.padded_plane + 2 * L9.plane + L9.carry_slice +
                (size_t)nslices * (3 * L9.plane + L9.carry_slice);
    return fl * sizeof(float) + 24 * 256;
}

// side stream used to overlap the (latency-bound, single-slice) guidance stage with the stage-1
// carry prepass; fork/join with events so the caller's stream semantics are preserved
static thread_local hipStream_t g_side = nullptr;
static thread_local hipEvent_t g_fork = nullptr, g_join = nullptr;

static int ensure_side_stream() {
    if (!g_side) {
        SMX_HIP(hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking));
        SMX_HIP(hipEventCreateWithFlags(&g_fork, hipEventDisableTiming));
        SMX_HIP(hipEventCreateWithFlags(&g_join, hipEventDisableTiming));
    }
    return SMX_OK;
}

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
    hipLaunchKernelGGL(v2::k_v2_walk<MODE>, dim3(a.nstrips, nslices, nviews), dim3(256), 0, st, L);
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
    float *F[2], *G[2], *meanT[2], *cinvT[2], *gcarry[2];
    for (int i = 0; i < nimg; ++i) { F[i] = carve(L.padded_plane); G[i] = carve(L.padded_plane); }
    for (int v = 0; v < nviews; ++v) {
        meanT[v] = carve(L.plane); cinvT[v] = carve(L.plane); gcarry[v] = carve(L.carry_slice);
    }
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
    pa.F[0] = F[0]; pa.F[1] = F[1]; pa.G[0] = G[0]; pa.G[1] = G[1];
    hipLaunchKernelGGL(v2::k_v2_prep, dim3(L.hp / 64, w + 2, 2), dim3(64), 0, st, pa, w, h, L.hp);
    SMX_HIP(hipGetLastError());
    ++nl;

    v2::Launch base_l;
    memset(&base_l, 0, sizeof(base_l));
    for (int v = 0; v < nviews; ++v) {
        v2::Args& a = base_l.v[v];
        a.w = w; a.h = h; a.hp = L.hp; a.R = R; a.ow = L.ow; a.nstrips = L.nstrips; a.nsegs = L.nsegs;
        a.F1 = F[v]; a.G1 = G[v]; a.F2 = F[v ^ 1]; a.G2 = G[v ^ 1];
        a.cc = make_cost_const(p);
        a.eps = p->eps;
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
        v2::Launch s1 = base_l, s2 = base_l;
        for (int v = 0; v < nviews; ++v) {
            s1.v[v].d0 = dmin[v] + s0; s1.v[v].dstA = aT[v]; s1.v[v].dstB = bT[v]; s1.v[v].carry = carry[v];
            s2.v[v].srcA = aT[v]; s2.v[v].srcB = bT[v]; s2.v[v].dstA = qT[v]; s2.v[v].carry = carry[v];
        }
        if ((rc = launch_carry<v2::S1>(s1, cnt, nviews, st))) return rc;
        if (!joined) {
            SMX_HIP(hipStreamWaitEvent(st, g_join, 0));
            joined = true;
        }
        if ((rc = launch_walk<v2::S1>(s1, cnt, nviews, st))) return rc;
        if ((rc = launch_carry<v2::S2>(s2, cnt, nviews, st))) return rc;
        if ((rc = launch_walk<v2::S2>(s2, cnt, nviews, st))) return rc;
        v2::WtaArgs wa;
        for (int v = 0; v < 2; ++v) { wa.qT[v] = qT[v < nviews ? v : 0]; wa.keys[v] = d_keys[v < nviews ? v : 0]; }
        hipLaunchKernelGGL(v2::k_v2_wta, dim3(cdivu(h, 64), w, nviews), dim3(64), 0, st, wa, w, h, L.hp,
                           cnt, s0);
        SMX_HIP(hipGetLastError());
        nl += 5;
        for (int v = 0; v < nviews; ++v) {
            if (d_agg && d_agg[v]) {
                dim3 tg(cdivu(w, 32), cdivu(h, 32), cnt);
                hipLaunchKernelGGL(v2::k_v2_untranspose, tg, dim3(256), 0, st, qT[v],
                                   d_agg[v] + (size_t)(s0 - s_begin) * w * h, w, h, L.hp);
                SMX_HIP(hipGetLastError());
                ++nl;
            }
        }
    }
    if (!joined) SMX_HIP(hipStreamWaitEvent(st, g_join, 0));
    if (launches) *launches = nl;
    return SMX_OK;
}

}  // namespace smx
