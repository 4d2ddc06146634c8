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

constexpr int TW = 112;            // columns of S computed per strip
constexpr int NSUB = 4;            // sub-strips per strip = waves per workgroup
constexpr int SUBW = TW / NSUB;    // 28
constexpr int BH = 64;             // band height = wave width
constexpr int RMAX = 9;            // largest supported box radius
constexpr int RR = BH + 2 * RMAX + 2;  // ring rows (84)
constexpr int PITCH = TW + 1;      // odd pitch: LANE=ROW accesses hit distinct banks
constexpr int RBATCH = 14;         // columns per load batch in phase R (SUBW = 2 batches)
constexpr int PB = 4;              // output columns per load batch in phase B

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

__device__ __forceinline__ int seg_c0(const Args& a, int g) {
    return (g >> 2) * a.ow - (a.R + 1) + (g & 3) * SUBW;
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
// grid (ceil(hp/64), w + 2), block 64.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_v2_prep(const uint8_t* __restrict__ I, float* __restrict__ F,
                                                float* __restrict__ G, int w, int h, int hp) {
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
// carry prepass.  grid (nbands, nslices), block 64 (LANE = ROW).
// ---------------------------------------------------------------------------------------------
constexpr int CB = 16;  // columns per load batch of the carry prepass

template <int MODE>
__global__ __launch_bounds__(64) void k_v2_carry(Args a) {
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
// strip walker.  grid (nstrips, nslices), block 256.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int ring_row(int y) { return y % RR; }

template <int MODE>
__global__ __launch_bounds__(256) void k_v2_walk(Args a) {
    __shared__ float ring[2][RR][PITCH];
    const int k = blockIdx.x, slice = blockIdx.y;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int R = a.R, w = a.w, h = a.h, hp = a.hp;
    const int xs = k * a.ow, cs = xs - (R + 1);
    const size_t plane = (size_t)w * hp;
    const int nbands = (h + BH - 1) / BH;
    // phase C ownership: thread -> (integral, column of the tile)
    const int ci = tid / TW, cj = tid - ci * TW;
    float S = -0.0f;
    const float* __restrict__ meanT = a.meanT;
    const float* __restrict__ cinvT = a.cinvT;
    const float* __restrict__ srcF1 = a.F1;
    float* __restrict__ dstA = a.dstA;
    float* __restrict__ dstB = a.dstB;

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
            for (int mb = wave; mb < a.ow; mb += NSUB * PB) {
                // global operands of PB output columns first (one exposed latency per batch)
                float ga[PB], gb[PB];
#pragma unroll
                for (int t = 0; t < PB; ++t) {
                    const int m = mb + t * NSUB;
                    int xo = xs + m;
                    xo = (m < a.ow && xo < w) ? xo : (w - 1);
                    const size_t T = (size_t)xo * hp + yo;
                    if (MODE == S1) { ga[t] = meanT[T]; gb[t] = cinvT[T]; }
                    if (MODE == S2) { ga[t] = srcF1[(size_t)(xo + 1) * hp + yo]; gb[t] = 0.0f; }
                    if (MODE == GUID) { ga[t] = 0.0f; gb[t] = 0.0f; }
                }
#pragma unroll
                for (int t = 0; t < PB; ++t) {
                    const int m = mb + t * NSUB;
                    const int xo = xs + m;
                    if (m < a.ow && xo < w) {
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
                        const size_t T = (size_t)xo * hp + yo;
                        if (MODE == GUID) {
                            float mm = m0 * m0;          // pixelMultOnGPU(mean, mean) guidedFilter.cu:112
                            float var = m1 - mm;         // pixelSousOnGPU :121
                            float c = (float)(1.0f / ((double)var + a.eps));   // :350
                            dstA[T] = m0;
                            dstB[T] = c;
                            if (a.mean_u8) {             // flToChOnGPU :451-458
                                int ci8 = (int)m0;
                                a.mean_u8[(size_t)yo * w + xo] = (ci8 > 255) ? 255 : (uint8_t)ci8;
                            }
                        } else if (MODE == S1) {
                            float mI = ga[t];
                            float c = gb[t];
                            float mm = mI * m0;          // compute_ak_and_bk guidedFilter.cu:345-354
                            float ak = 1.0f * (m1 - mm) * c;
                            float mb2 = 1.0f * mI * ak;
                            float bk = 1.0f * m0 - mb2;
                            dstA[(size_t)slice * plane + T] = ak;
                            dstB[(size_t)slice * plane + T] = bk;
                        } else {
                            float tq = m0 * ga[t];       // compute_q guidedFilter.cu:363-369
                            dstA[(size_t)slice * plane + T] = tq + m1;
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
// ---------------------------------------------------------------------------------------------
__global__ void k_v2_wta(const float* __restrict__ qT, uint64_t* __restrict__ keys, int w, int h,
                         int hp, int count, int slice0) {
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    const int x = blockIdx.y;
    if (y >= h) return;
    const size_t plane = (size_t)w * hp;
    const float* q = qT + (size_t)x * hp + y;
    const size_t id = (size_t)y * w + x;
    uint64_t key = keys[id];
    for (int z = 0; z < count; ++z) {
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

size_t v2_workspace_bytes(int w, int h, int R, int nslices) {
    if (R < 0 || R > v2::RMAX) R = v2::RMAX;
    V2Layout L = v2_layout(w, h, R);
    size_t fl = 4 * L.padded_plane + 2 * L.plane + L.carry_slice +
                (size_t)nslices * (3 * L.plane + L.carry_slice);
    return fl * sizeof(float) + 16 * 256;
}

template <int MODE>
static int v2_run_stage(const v2::Args& a, int nslices, hipStream_t st) {
    if (nslices <= 0) return SMX_OK;
    const int nbands = (a.h + 63) / 64;
    hipLaunchKernelGGL(v2::k_v2_carry<MODE>, dim3(nbands, nslices), dim3(64), 0, st, a);
    SMX_HIP(hipGetLastError());
    hipLaunchKernelGGL(v2::k_v2_walk<MODE>, dim3(a.nstrips, nslices), dim3(256), 0, st, a);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

// Aggregation + WTA of slices [s_begin, s_end) of one view with the cost built on the fly.
int aggregate_v2(const smx_params* p, const uint8_t* d_guide, const uint8_t* d_other, int w, int h,
                 int dmin, int s_begin, int s_end, uint64_t* d_keys, uint8_t* d_mean_u8, float* d_agg,
                 void* d_ws, size_t ws_bytes, hipStream_t st, int* launches) {
    const int R = p->radius;
    V2Layout L = v2_layout(w, h, R);
    char* base = (char*)align_up((size_t)d_ws, 256);
    size_t avail = ws_bytes - (size_t)(base - (char*)d_ws);
    auto carve = [&](size_t nfloat) {
        float* r = (float*)base;
        size_t b = align_up(nfloat * sizeof(float), 256);
        base += b;
        avail = avail >= b ? avail - b : 0;
        return r;
    };
    const size_t fixed = 4 * align_up(L.padded_plane * 4, 256) + 2 * align_up(L.plane * 4, 256) +
                         align_up(L.carry_slice * 4, 256);
    const size_t per_slice = (3 * L.plane + L.carry_slice) * sizeof(float);
    if (avail < fixed + per_slice + 4 * 256)
        return fail(SMX_E_WS, "aggregate_v2: workspace %zu B too small (need >= %zu B)", ws_bytes,
                    v2_workspace_bytes(w, h, R, 1));
    float* F1 = carve(L.padded_plane);
    float* G1 = carve(L.padded_plane);
    float* F2 = carve(L.padded_plane);
    float* G2 = carve(L.padded_plane);
    float* meanT = carve(L.plane);
    float* cinvT = carve(L.plane);
    float* gcarry = carve(L.carry_slice);
    const int total = s_end - s_begin;
    size_t fit = (avail - 4 * 256) / per_slice;
    int chunk = fit > (size_t)total ? total : (int)fit;
    if (chunk < 1) chunk = 1;
    float* aT = carve((size_t)chunk * L.plane);
    float* bT = carve((size_t)chunk * L.plane);
    float* qT = carve((size_t)chunk * L.plane);
    float* carry = carve((size_t)chunk * L.carry_slice);

    int nl = 0;
    dim3 pgrid(L.hp / 64, w + 2);
    hipLaunchKernelGGL(v2::k_v2_prep, pgrid, dim3(64), 0, st, d_guide, F1, G1, w, h, L.hp);
    SMX_HIP(hipGetLastError());
    hipLaunchKernelGGL(v2::k_v2_prep, pgrid, dim3(64), 0, st, d_other, F2, G2, w, h, L.hp);
    SMX_HIP(hipGetLastError());
    nl += 2;

    v2::Args a;
    memset(&a, 0, sizeof(a));
    a.w = w; a.h = h; a.hp = L.hp; a.R = R; a.ow = L.ow; a.nstrips = L.nstrips; a.nsegs = L.nsegs;
    a.F1 = F1; a.G1 = G1; a.F2 = F2; a.G2 = G2;
    a.cc = make_cost_const(p);
    a.eps = p->eps;
    int rc;
    // guidance statistics
    {
        v2::Args g = a;
        g.dstA = meanT; g.dstB = cinvT; g.mean_u8 = d_mean_u8; g.carry = gcarry;
        if ((rc = v2_run_stage<v2::GUID>(g, 1, st))) return rc;
        nl += 2;
    }
    a.meanT = meanT; a.cinvT = cinvT;
    for (int s0 = s_begin; s0 < s_end; s0 += chunk) {
        const int cnt = (s_end - s0) < chunk ? (s_end - s0) : chunk;
        v2::Args s1 = a;
        s1.d0 = dmin + s0; s1.dstA = aT; s1.dstB = bT; s1.carry = carry;
        if ((rc = v2_run_stage<v2::S1>(s1, cnt, st))) return rc;
        v2::Args s2 = a;
        s2.srcA = aT; s2.srcB = bT; s2.dstA = qT; s2.carry = carry;
        if ((rc = v2_run_stage<v2::S2>(s2, cnt, st))) return rc;
        hipLaunchKernelGGL(v2::k_v2_wta, dim3(cdivu(h, 64), w), dim3(64), 0, st, qT, d_keys, w, h,
                           L.hp, cnt, s0);
        SMX_HIP(hipGetLastError());
        nl += 5;
        if (d_agg) {
            dim3 tg(cdivu(w, 32), cdivu(h, 32), cnt);
            hipLaunchKernelGGL(v2::k_v2_untranspose, tg, dim3(256), 0, st, qT,
                               d_agg + (size_t)(s0 - s_begin) * w * h, w, h, L.hp);
            SMX_HIP(hipGetLastError());
            ++nl;
        }
    }
    if (launches) *launches = nl;
    return SMX_OK;
}

}  // namespace smx
