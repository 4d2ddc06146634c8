// smx_kernels.hip -- gfx950 kernels of the stereo-pair -> disparity-map path (v1 dataflow).
//
// Numerical contract: IEEE f32, every operation rounded on its own in
// the reference's source order.  This file MUST be compiled with -ffp-contract=off; prefix sums
// keep the reference's sequential left->right / top->bottom addition order (integral.cu:82-86,
// 124-128) -- parallelism comes from rows x slices and columns x slices, never from
// re-associating the adds.
//
// Reference citations are relative to the reference's stereo_matching_cuda/ directory.
#include "smx_common.h"

namespace smx {

// =====================================================================================
// rgb -> gray   (rgb_to_grayscale.cu:14-23, sumArraysOnGPU)
// =====================================================================================
__global__ void k_gray(const uint8_t* __restrict__ rgb, int64_t n, int ch,
                       uint8_t* __restrict__ gray, double rw, double gw, double bw) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint8_t* px = rgb + (int64_t)ch * k;
    double val = rw * px[0] + gw * px[1] + bw * px[2];
    gray[k] = (uint8_t)val;
}

// =====================================================================================
// cost volume   (costVolume.cu:358-381 x_derivativeOnGPU, :163-190 costVolumOnGPU2)
// =====================================================================================
__device__ __forceinline__ float xgrad(const uint8_t* __restrict__ row, int x, int w) {
    int c1, c2;
    if (x - 1 >= 0 && x + 1 < w) { c1 = row[x + 1]; c2 = row[x - 1]; }
    else if (x + 1 >= w)         { c1 = row[x];     c2 = row[x - 1]; }
    else                         { c1 = row[x + 1]; c2 = row[x];     }
    return 1.0f * (float)(c2 - c1) / 2;
}

__device__ __forceinline__ float cost_cell(const uint8_t* __restrict__ r1,
                                           const uint8_t* __restrict__ r2, int x, int d, int w1,
                                           int w2, float g1, const CostConst& cc) {
    int xx = x + d;
    float c = cc.border;
    if (xx < w2 && xx >= 0) {
        int di = (int)r1[x] - (int)r2[xx];
        float t1 = 1.0f * (float)(di < 0 ? -di : di);
        float t2 = 1.0f * fabsf(g1 - xgrad(r2, xx, w2));
        float m1 = t1 < cc.th_color ? t1 : cc.th_color;
        float m2 = t2 < cc.th_grad ? t2 : cc.th_grad;
        float a = cc.oma * m1;
        float b = cc.alpha * m2;
        c = a + b;
    }
    return c;
}

constexpr int COST_ZPER = 8;

// grid (ceil(w/256), h, ceil(count/COST_ZPER)); slice z (relative) has d = d0 + z.
__global__ void k_cost(const uint8_t* __restrict__ i1, const uint8_t* __restrict__ i2,
                       float* __restrict__ cost, int w, int h, int d0, int count, CostConst cc) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (x >= w) return;
    const uint8_t* r1 = i1 + (int64_t)y * w;
    const uint8_t* r2 = i2 + (int64_t)y * w;
    const int64_t n = (int64_t)w * h;
    float g1 = xgrad(r1, x, w);
    int z0 = blockIdx.z * COST_ZPER;
    int z1 = min(count, z0 + COST_ZPER);
    for (int z = z0; z < z1; ++z)
        cost[(int64_t)z * n + (int64_t)y * w + x] = cost_cell(r1, r2, x, d0 + z, w, w, g1, cc);
}

// =====================================================================================
// integral image = sequential row prefix then sequential column prefix
// (integral.cu:78-90 rowSum, :121-131 colSum)
// =====================================================================================
constexpr int RS_TX = 32;  // columns per LDS tile; one wave scans 64 rows

// One wave per (plane, band of 64 rows).  The 64 x RS_TX tile is loaded coalesced (128 B row
// segments), transposed through LDS so that lane r owns row r, scanned sequentially with a
// register carry across tiles, and stored coalesced.  MODE 0: out0 = rowscan(in0).
// MODE 1: out0 = rowscan(in0), out1 = rowscan(im * in0)   (guidedFilter.cu:203,209-212)
// MODE 2: out0 = rowscan(in0), out1 = rowscan(in1)
// -0.0f is the exact additive identity, so `acc = v + acc` reproduces out[0] = in[0].
template <int MODE>
__global__ __launch_bounds__(64) void k_rowscan(const float* in0, const float* in1, float* out0,
                                                float* out1, int w, int h, int nplanes) {
    __shared__ float t0[64][RS_TX + 1];
    __shared__ float t1[MODE == 0 ? 1 : 64][RS_TX + 1];
    const int bands = (h + 63) >> 6;
    const int plane = blockIdx.x / bands;
    const int band = blockIdx.x - plane * bands;
    const int y0 = band << 6;
    const int rows = min(64, h - y0);
    const int lane = threadIdx.x;
    const int64_t n = (int64_t)w * h;
    const float* p0 = in0 + (int64_t)plane * n + (int64_t)y0 * w;
    const float* p1 = nullptr;
    if (MODE == 1) p1 = in1 + (int64_t)y0 * w;                       // guidance plane im
    if (MODE == 2) p1 = in1 + (int64_t)plane * n + (int64_t)y0 * w;  // second input stack
    float* q0 = out0 + (int64_t)plane * n + (int64_t)y0 * w;
    float* q1 = (MODE == 0) ? nullptr : out1 + (int64_t)plane * n + (int64_t)y0 * w;
    const int lr = lane >> 5, lc = lane & 31;
    float acc0 = -0.0f, acc1 = -0.0f;
    for (int x0 = 0; x0 < w; x0 += RS_TX) {
        const int cols = min(RS_TX, w - x0);
#pragma unroll 8
        for (int i = 0; i < 32; ++i) {
            int r = 2 * i + lr;
            if (r < rows && lc < cols) {
                int64_t o = (int64_t)r * w + x0 + lc;
                float v = p0[o];
                t0[r][lc] = v;
                if (MODE == 1) t1[r][lc] = p1[o] * v;
                if (MODE == 2) t1[r][lc] = p1[o];
            }
        }
        __syncthreads();
        if (lane < rows) {
            for (int j = 0; j < cols; ++j) {
                acc0 = t0[lane][j] + acc0;
                t0[lane][j] = acc0;
                if (MODE != 0) {
                    acc1 = t1[lane][j] + acc1;
                    t1[lane][j] = acc1;
                }
            }
        }
        __syncthreads();
#pragma unroll 8
        for (int i = 0; i < 32; ++i) {
            int r = 2 * i + lr;
            if (r < rows && lc < cols) {
                int64_t o = (int64_t)r * w + x0 + lc;
                q0[o] = t0[r][lc];
                if (MODE != 0) q1[o] = t1[r][lc];
            }
        }
        __syncthreads();
    }
}

// In-place column prefix: one lane per (plane, column); grid.y selects the buffer.
__global__ void k_colscan(float* buf0, float* buf1, int w, int h, int nplanes) {
    float* buf = blockIdx.y ? buf1 : buf0;
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)nplanes * w) return;
    int64_t plane = gid / w;
    int x = (int)(gid - plane * w);
    float* p = buf + plane * ((int64_t)w * h) + x;
    float acc = -0.0f;
    int y = 0;
    for (; y + 8 <= h; y += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[(int64_t)(y + k) * w];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            acc = v[k] + acc;
            p[(int64_t)(y + k) * w] = acc;
        }
    }
    for (; y < h; ++y) {
        acc = p[(int64_t)y * w] + acc;
        p[(int64_t)y * w] = acc;
    }
}

// =====================================================================================
// box mean from an integral image   (guidedFilter.cu:305-318 computeMeanOnGPU)
// =====================================================================================
struct BoxTaps {
    int64_t i11, i10, i01, i00;  // offsets into a plane; i10/i01/i00 valid iff flag set
    bool hx, hy;                 // xmin >= 0, ymin >= 0
    float area;                  // (float)((xmax-xmin)*(ymax-ymin))
};

__device__ __forceinline__ BoxTaps box_taps(int x, int y, int w, int h, int R) {
    int ymin = max(-1, y - R - 1);
    int ymax = min(h - 1, y + R);
    int xmin = max(-1, x - R - 1);
    int xmax = min(w - 1, x + R);
    BoxTaps t;
    t.hx = xmin >= 0;
    t.hy = ymin >= 0;
    t.i11 = (int64_t)ymax * w + xmax;
    t.i10 = (int64_t)ymax * w + (t.hx ? xmin : 0);
    t.i01 = (int64_t)(t.hy ? ymin : 0) * w + xmax;
    t.i00 = (int64_t)(t.hy ? ymin : 0) * w + (t.hx ? xmin : 0);
    t.area = (float)((xmax - xmin) * (ymax - ymin));
    return t;
}

__device__ __forceinline__ float box_eval(const float* __restrict__ S, const BoxTaps& t) {
    float val = S[t.i11];
    if (t.hx) val -= S[t.i10];
    if (t.hy) val -= S[t.i01];
    if (t.hx && t.hy) val += S[t.i00];
    return 1.0f * val / t.area;
}

// =====================================================================================
// guidance statistics   (guidedFilter.cu:58-123)
// =====================================================================================
__global__ void k_guid_prep(const uint8_t* __restrict__ I, float* __restrict__ im,
                            float* __restrict__ sq, int64_t n) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    int c = (int)I[k];
    float v = 1.0f * (float)c;  // chToFlOnGPU guidedFilter.cu:442-449
    im[k] = v;
    sq[k] = v * v;              // pixelMultOnGPU(d_im, d_im) :111
}

// mean_im = box(S_im); var = box(S_sq) - mean_im*mean_im; cinv = (float)(1.0f/((double)var+EPS))
// (the reciprocal of compute_ak_and_bk, guidedFilter.cu:350, depends on the guidance only).
__global__ void k_guid_finish(const float* __restrict__ S_im, const float* __restrict__ S_sq,
                              float* __restrict__ mean_im, float* __restrict__ cinv,
                              uint8_t* __restrict__ mean_u8, int w, int h, int R, double eps) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (x >= w) return;
    BoxTaps t = box_taps(x, y, w, h, R);
    float m = box_eval(S_im, t);
    float s = box_eval(S_sq, t);
    float m2 = m * m;          // pixelMultOnGPU(mean, mean) :112
    float var = s - m2;        // pixelSousOnGPU :121
    float c = (float)(1.0f / ((double)var + eps));
    int64_t id = (int64_t)y * w + x;
    mean_im[id] = m;
    cinv[id] = c;
    if (mean_u8) {             // flToChOnGPU :451-458
        int ci = (int)m;
        mean_u8[id] = (ci > 255) ? 255 : (uint8_t)ci;
    }
}

// =====================================================================================
// a_k, b_k   (guidedFilter.cu:345-354 compute_ak_and_bk on box means of S_p, S_Ip)
// grid (ceil(w/256), h, nplanes)
// =====================================================================================
__global__ void k_ab(const float* __restrict__ Sp, const float* __restrict__ SIp,
                     const float* __restrict__ mean_im, const float* __restrict__ cinv,
                     float* __restrict__ A, float* __restrict__ B, int w, int h, int R) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (x >= w) return;
    const int64_t n = (int64_t)w * h;
    const int64_t po = (int64_t)blockIdx.z * n;
    BoxTaps t = box_taps(x, y, w, h, R);
    float mp = box_eval(Sp + po, t);
    float mIp = box_eval(SIp + po, t);
    int64_t id = (int64_t)y * w + x;
    float mI = mean_im[id];
    float c = cinv[id];
    float mm = mI * mp;
    float ak = 1.0f * (mIp - mm) * c;
    float mb = 1.0f * mI * ak;
    float bk = 1.0f * mp - mb;
    A[po + id] = ak;
    B[po + id] = bk;
}

// =====================================================================================
// q = box(S_a)*I + box(S_b) and running WTA over the chunk's slices
// (guidedFilter.cu:363-369 compute_q, :403-411 dispSelectOnGPU).  One lane per pixel; the key
// min reproduces `if (best >= q) {...}` with slices ascending: ties go to the larger slice.
// =====================================================================================
__global__ void k_q_wta(const float* __restrict__ Sa, const float* __restrict__ Sb,
                        const float* __restrict__ im, int64_t* __restrict__ keys,
                        float* __restrict__ agg, int w, int h, int count, int slice0, int R) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (x >= w) return;
    const int64_t n = (int64_t)w * h;
    BoxTaps t = box_taps(x, y, w, h, R);
    int64_t id = (int64_t)y * w + x;
    float I = im[id];
    int64_t key = keys[id];
    for (int z = 0; z < count; ++z) {
        const int64_t po = (int64_t)z * n;
        float abar = box_eval(Sa + po, t);
        float bbar = box_eval(Sb + po, t);
        float m = abar * I;
        float q = m + bbar;
        int64_t k = pack_key(q, (uint32_t)(slice0 + z));
        key = k < key ? k : key;
        if (agg) agg[po + id] = q;
    }
    keys[id] = key;
}

__global__ void k_init_keys(int64_t* keys, int64_t n) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) keys[k] = KEY_IDENTITY;
}

__global__ void k_init_wta(float* best, float* dmap, int64_t n) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) {
        best[k] = __builtin_bit_cast(float, 0x7F7F7F7Fu);  // main.cu:112
        dmap[k] = 0.0f;                                     // main.cu:117
    }
}

// dispSelectOnGPU (guidedFilter.cu:403-411) applied once to the winning slice of the key.
__global__ void k_apply_keys(const int64_t* __restrict__ keys, int64_t n, int dmin, float* best,
                             float* dmap) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    int64_t key = keys[k];
    if (key == KEY_IDENTITY) return;
    float q;
    uint32_t s;
    unpack_key(key, &q, &s);
    if (1.0f * best[k] >= 1.0f * q) {
        dmap[k] = (float)(dmin + (int)s);
        best[k] = q;
    }
}

// main.cu:112-141 for both views in one pass: the reference's presets (k_init_wta), the winning slice of
// each key (k_apply_keys) and the copy of the left disparity map that the LR check then overwrites.
// keys / best / dmap hold the left view in [0, n) and the right view in [n, 2n).
__global__ void k_finish_keys(const int64_t* __restrict__ keys, int64_t n, int dminl, int dminr,
                              float* __restrict__ best, float* __restrict__ dmap, float* __restrict__ occlusion) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= 2 * n) return;
    float b = __builtin_bit_cast(float, 0x7F7F7F7Fu);   // main.cu:112
    float d = 0.0f;                                      // main.cu:117
    const int64_t key = keys[k];
    if (key != KEY_IDENTITY) {
        float q;
        uint32_t s;
        unpack_key(key, &q, &s);
        if (1.0f * b >= 1.0f * q) {                      // dispSelectOnGPU guidedFilter.cu:403-411
            d = (float)((k < n ? dminl : dminr) + (int)s);
            b = q;
        }
    }
    best[k] = b;
    dmap[k] = d;
    if (k < n) occlusion[k] = d;                         // main.cu:141
}

// =====================================================================================
// occlusion   (occlusion.cu:3-15 detect_occlusionOnGPU, :134-176 fill_occlusionOnGPU1)
// =====================================================================================
__global__ void k_detect_occlusion(float* dL, const float* __restrict__ dR, int dOcclusion, int w,
                                   int h, int d_lr) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (x >= w) return;
    int64_t id = (int64_t)y * w + x;
    int d = (int)dL[id];
    if (x + d < 0 || x + d >= w || fabsf((float)d + dR[id + d]) > (float)d_lr)
        dL[id] = (float)dOcclusion;
}

// One wave per row, two ballot sweeps: nearest valid value at-or-left, then at-or-right.
// Pure selection (no arithmetic), so it equals the reference's per-pixel searches, including
// its benign in-place race.  Dynamic LDS: 2 w floats.
// With in != disp the filled map is written to disp and `in` stays untouched (main.cu:153 copy folded in):
// the sweeps only ever read values that are valid in the input, and those are never overwritten.
__global__ __launch_bounds__(64) void k_fill_occlusion(const float* in, float* disp, int w, int h, float vMin) {
    extern __shared__ float sbuf[];          // [0, w): nearest valid value at-or-left, [w, 2w): the row itself
    float* sLeft = sbuf;
    float* sVal = sbuf + w;
    const int lane = threadIdx.x;
    const float* row = in + (int64_t)blockIdx.x * w;
    float* out = disp + (int64_t)blockIdx.x * w;
    // the row goes to LDS first, all loads in flight at once: the sweeps below are chains of ballots and
    // must not wait for memory chunk by chunk
    for (int x = lane; x < w; x += 64) sVal[x] = row[x];
    __syncthreads();
    float carry = vMin;
    for (int c0 = 0; c0 < w; c0 += 64) {
        int x = c0 + lane;
        float v = x < w ? sVal[x] : 0.0f;
        bool valid = x < w && v >= vMin;
        unsigned long long mask = __ballot(valid);
        unsigned long long lower = mask & ((2ull << lane) - 1ull);
        int src = lower ? 63 - __clzll((long long)lower) : lane;
        float pick = __shfl(v, src);
        if (x < w) sLeft[x] = lower ? pick : carry;
        int last = mask ? 63 - __clzll((long long)mask) : 0;
        float nv = __shfl(v, last);
        if (mask) carry = nv;
    }
    carry = vMin;
    for (int c0 = ((w - 1) / 64) * 64; c0 >= 0; c0 -= 64) {
        int x = c0 + lane;
        float v = x < w ? sVal[x] : 0.0f;
        bool valid = x < w && v >= vMin;
        unsigned long long mask = __ballot(valid);
        unsigned long long upper = mask & (~0ull << lane);
        int src = upper ? __ffsll((long long)upper) - 1 : lane;
        float pick = __shfl(v, src);
        float right = upper ? pick : carry;
        if (x < w) {
            int dX = (int)v;
            if (!((float)dX >= vMin)) {
                float l = sLeft[x];
                out[x] = l > right ? l : right;
            } else if (in != disp) {
                out[x] = v;
            }
        }
        int first = mask ? __ffsll((long long)mask) - 1 : 0;
        float nv = __shfl(v, first);
        if (mask) carry = nv;
    }
}

// main.cu:112-155 for one image row per workgroup, in one launch: presets + winning slice of each key of both
// views (k_finish_keys), LR check of the left map (k_detect_occlusion; reads the right map only in its own row) and
// filling (k_fill_occlusion: the same two ballot sweeps, by wave 0).  The three steps are pure selections on the
// values of the row, so the results equal those of the three kernels.  All key loads of the row are in flight
// before the first store.  Dynamic LDS: 3 w floats (left map -> occlusion row, right map, nearest-valid-at-left).
constexpr int FP_NT = 256;
constexpr int FP_MAXW = 8192;
__global__ __launch_bounds__(FP_NT) void k_finish_pair_row(const int64_t* __restrict__ keys, int w, int h, int dminl,
                                                            int dminr, int dOcclusion, int d_lr, float vMin,
                                                            float* __restrict__ best, float* __restrict__ dmap,
                                                            float* __restrict__ occlusion, float* __restrict__ filled) {
    extern __shared__ float sbuf[];
    float* sVal = sbuf;                      // left disparities of the row, then the row after the LR check
    float* sR = sbuf + w;                    // right disparities of the row
    float* sLeft = sbuf + 2 * w;
    const int tid = threadIdx.x, y = blockIdx.x;
    const int64_t n = (int64_t)w * h, r0 = (int64_t)y * w;
    auto decode = [&](int64_t key, int dmin, float& b, float& d) {
        b = __builtin_bit_cast(float, 0x7F7F7F7Fu);     // main.cu:112
        d = 0.0f;                                        // main.cu:117
        if (key != KEY_IDENTITY) {
            float q;
            uint32_t s;
            unpack_key(key, &q, &s);
            if (1.0f * b >= 1.0f * q) {                  // dispSelectOnGPU guidedFilter.cu:403-411
                d = (float)(dmin + (int)s);
                b = q;
            }
        }
    };
    for (int x0 = 0; x0 < w; x0 += 4 * FP_NT) {
        int64_t kl[4], kr[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int x = min(x0 + t * FP_NT + tid, w - 1);
            kl[t] = keys[r0 + x];
            kr[t] = keys[n + r0 + x];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int x = x0 + t * FP_NT + tid;
            if (x < w) {
                float b, d;
                decode(kl[t], dminl, b, d);
                best[r0 + x] = b;
                dmap[r0 + x] = d;
                sVal[x] = d;
                decode(kr[t], dminr, b, d);
                best[n + r0 + x] = b;
                dmap[n + r0 + x] = d;
                sR[x] = d;
            }
        }
    }
    __syncthreads();
    // detect_occlusionOnGPU occlusion.cu:3-15
    for (int x = tid; x < w; x += FP_NT) {
        float v = sVal[x];
        const int d = (int)v;
        if (x + d < 0 || x + d >= w || fabsf((float)d + sR[x + d]) > (float)d_lr) v = (float)dOcclusion;
        sVal[x] = v;
        occlusion[r0 + x] = v;
    }
    __syncthreads();
    if (tid >= 64) return;
    // fill_occlusionOnGPU1 occlusion.cu:134-176 (see k_fill_occlusion)
    const int lane = tid;
    float* out = filled + r0;
    float carry = vMin;
    for (int c0 = 0; c0 < w; c0 += 64) {
        int x = c0 + lane;
        float v = x < w ? sVal[x] : 0.0f;
        bool valid = x < w && v >= vMin;
        unsigned long long mask = __ballot(valid);
        unsigned long long lower = mask & ((2ull << lane) - 1ull);
        int src = lower ? 63 - __clzll((long long)lower) : lane;
        float pick = __shfl(v, src);
        if (x < w) sLeft[x] = lower ? pick : carry;
        int last = mask ? 63 - __clzll((long long)mask) : 0;
        float nv = __shfl(v, last);
        if (mask) carry = nv;
    }
    carry = vMin;
    for (int c0 = ((w - 1) / 64) * 64; c0 >= 0; c0 -= 64) {
        int x = c0 + lane;
        float v = x < w ? sVal[x] : 0.0f;
        bool valid = x < w && v >= vMin;
        unsigned long long mask = __ballot(valid);
        unsigned long long upper = mask & (~0ull << lane);
        int src = upper ? __ffsll((long long)upper) - 1 : lane;
        float pick = __shfl(v, src);
        float right = upper ? pick : carry;
        if (x < w) {
            int dX = (int)v;
            if (!((float)dX >= vMin)) {
                float l = sLeft[x];
                out[x] = l > right ? l : right;
            } else {
                out[x] = v;
            }
        }
        int first = mask ? __ffsll((long long)mask) - 1 : 0;
        float nv = __shfl(v, first);
        if (mask) carry = nv;
    }
}

// =====================================================================================
// filter()  (filter.cu:117-207; dead code in the reference: never called from main.cu)
// Direct (2R+1)^2 box filter with zero padding, f32 accumulation in the reference's order (x offset
// outer, y offset inner, filter.cu:57-61), mean truncated to an integer:
//   mean   = (uchar)(int)(sum(I) / (2R+1)^2)                       boxFilterOnGPU      :39-65
//   var    = (float)(int)(sum(I*I) / (2R+1)^2) - (float)(mean*mean) multIm, boxFilterfloatOnGpu, sousIm
// One thread per pixel; the 19 x 19 window comes from the L1/L2-resident u8 image (not a hot path).
// =====================================================================================
__global__ void k_filter(const uint8_t* __restrict__ I, uint8_t* __restrict__ mean,
                         float* __restrict__ var, int w, int h, int R) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w) return;
    float s1 = 0.0f, s2 = 0.0f;
    for (int ix = -R; ix <= R; ++ix) {
        const int xx = x + ix;
        for (int iy = -R; iy <= R; ++iy) {
            const int yy = y + iy;
            float v = 0.0f, v2 = 0.0f;
            if (xx >= 0 && xx < w && yy >= 0 && yy < h) {
                const int c = (int)I[(int64_t)yy * w + xx];
                v = (float)c;
                v2 = (float)(c * c);      // multIm: u8 * u8 in int, then float
            }
            s1 += v;
            s2 += v2;
        }
    }
    const int area = (2 * R + 1) * (2 * R + 1);
    const int m = (int)(s1 / area);
    const int m2 = (int)(s2 / area);
    const uint8_t mu = (uint8_t)m;
    const int64_t id = (int64_t)y * w + x;
    mean[id] = mu;
    const float mm = (float)((int)mu * (int)mu);   // multIm(d_mean, d_mean)
    var[id] = (float)m2 - mm;                     // sousIm
}

}  // namespace smx

// =====================================================================================
// launchers (used by smx_capi.hip)
// =====================================================================================
namespace smx {

static inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

int launch_gray(const smx_params* p, const uint8_t* rgb, int64_t n, int ch, uint8_t* gray,
                hipStream_t st) {
    hipLaunchKernelGGL(k_gray, dim3(cdiv(n, 256)), dim3(256), 0, st, rgb, n, ch, gray, p->r_w,
                       p->g_w, p->b_w);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

int launch_cost(const smx_params* p, const uint8_t* i1, const uint8_t* i2, float* cost, int w,
                int h, int d0, int count, hipStream_t st) {
    if (count <= 0) return SMX_OK;
    CostConst cc = make_cost_const(p);
    dim3 grid(cdiv(w, 256), h, cdiv(count, COST_ZPER));
    hipLaunchKernelGGL(k_cost, grid, dim3(256), 0, st, i1, i2, cost, w, h, d0, count, cc);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

// mode 0: out0 = integral(in0); mode 1: out0 = integral(in0), out1 = integral(im*in0) with
// in1 = im; mode 2: out0 = integral(in0), out1 = integral(in1).
int launch_integral(int mode, const float* in0, const float* in1, float* out0, float* out1, int w,
                    int h, int nplanes, hipStream_t st) {
    if (nplanes <= 0) return SMX_OK;
    const int bands = (h + 63) / 64;
    dim3 grid((unsigned)(bands * (int64_t)nplanes));
    if (mode == 0)
        hipLaunchKernelGGL(k_rowscan<0>, grid, dim3(64), 0, st, in0, in1, out0, out1, w, h, nplanes);
    else if (mode == 1)
        hipLaunchKernelGGL(k_rowscan<1>, grid, dim3(64), 0, st, in0, in1, out0, out1, w, h, nplanes);
    else
        hipLaunchKernelGGL(k_rowscan<2>, grid, dim3(64), 0, st, in0, in1, out0, out1, w, h, nplanes);
    SMX_HIP(hipGetLastError());
    dim3 cgrid(cdiv((int64_t)nplanes * w, 256), mode == 0 ? 1 : 2);
    hipLaunchKernelGGL(k_colscan, cgrid, dim3(256), 0, st, out0, out1, w, h, nplanes);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

int launch_guid_finish(const smx_params* p, const float* S_im, const float* S_sq, float* mean_im,
                       float* cinv, uint8_t* mean_u8, int w, int h, hipStream_t st) {
    dim3 grid(cdiv(w, 256), h);
    hipLaunchKernelGGL(k_guid_finish, grid, dim3(256), 0, st, S_im, S_sq, mean_im, cinv, mean_u8, w,
                       h, p->radius, p->eps);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

int launch_guid_prep(const uint8_t* I, float* im, float* sq, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_guid_prep, dim3(cdiv(n, 256)), dim3(256), 0, st, I, im, sq, n);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

int launch_ab(const smx_params* p, const float* Sp, const float* SIp, const float* mean_im,
              const float* cinv, float* A, float* B, int w, int h, int nplanes, hipStream_t st) {
    if (nplanes <= 0) return SMX_OK;
    dim3 grid(cdiv(w, 256), h, nplanes);
    hipLaunchKernelGGL(k_ab, grid, dim3(256), 0, st, Sp, SIp, mean_im, cinv, A, B, w, h, p->radius);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

int launch_q_wta(const smx_params* p, const float* Sa, const float* Sb, const float* im,
                 int64_t* keys, float* agg, int w, int h, int count, int slice0, hipStream_t st) {
    if (count <= 0) return SMX_OK;
    dim3 grid(cdiv(w, 256), h);
    hipLaunchKernelGGL(k_q_wta, grid, dim3(256), 0, st, Sa, Sb, im, keys, agg, w, h, count, slice0,
                       p->radius);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

int launch_init_keys(int64_t* keys, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_init_keys, dim3(cdiv(n, 256)), dim3(256), 0, st, keys, n);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

int launch_init_wta(float* best, float* dmap, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_init_wta, dim3(cdiv(n, 256)), dim3(256), 0, st, best, dmap, n);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

int launch_apply_keys(const int64_t* keys, int64_t n, int dmin, float* best, float* dmap,
                      hipStream_t st) {
    hipLaunchKernelGGL(k_apply_keys, dim3(cdiv(n, 256)), dim3(256), 0, st, keys, n, dmin, best,
                       dmap);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

int launch_detect_occlusion(const smx_params* p, float* dL, const float* dR, int dOcc, int w, int h,
                            hipStream_t st) {
    dim3 grid(cdiv(w, 256), h);
    hipLaunchKernelGGL(k_detect_occlusion, grid, dim3(256), 0, st, dL, dR, dOcc, w, h, p->d_lr);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

int launch_filter(const smx_params* p, const uint8_t* I, uint8_t* mean, float* var, int w, int h,
                  hipStream_t st) {
    hipLaunchKernelGGL(k_filter, dim3(cdiv(w, 256), h), dim3(256), 0, st, I, mean, var, w, h, p->radius);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

int launch_fill_occlusion(const float* src, float* disp, int w, int h, float vMin, hipStream_t st) {
    size_t lds = (size_t)w * 2 * sizeof(float);
    if (lds > 128 * 1024) return fail(SMX_E_ARG, "fill_occlusion: width %d exceeds LDS row buffer", w);
    static LdsLimitOnce lim;
    if (lds > 64 * 1024) SMX_HIP(lim.ensure((const void*)k_fill_occlusion, 128 * 1024));
    hipLaunchKernelGGL(k_fill_occlusion, dim3(h), dim3(64), lds, st, src, disp, w, h, vMin);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

int launch_finish_keys(const int64_t* keys, int64_t n, int dminl, int dminr, float* best, float* dmap,
                       float* occlusion, hipStream_t st) {
    hipLaunchKernelGGL(k_finish_keys, dim3(cdiv(2 * n, 256)), dim3(256), 0, st, keys, n, dminl, dminr, best, dmap,
                       occlusion);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

// the three launches above as one (rows of up to FP_MAXW pixels; the caller falls back to the three kernels)
bool finish_pair_row_supported(int w) { return w <= FP_MAXW; }
int launch_finish_pair_row(const smx_params* p, const int64_t* keys, int w, int h, int dminl, int dminr, int dOcc,
                           float vMin, float* best, float* dmap, float* occlusion, float* filled, hipStream_t st) {
    const size_t lds = (size_t)w * 3 * sizeof(float);
    static LdsLimitOnce lim;
    if (lds > 64 * 1024) SMX_HIP(lim.ensure((const void*)k_finish_pair_row, (int)((size_t)FP_MAXW * 3 * sizeof(float))));
    hipLaunchKernelGGL(k_finish_pair_row, dim3(h), dim3(FP_NT), lds, st, keys, w, h, dminl, dminr, dOcc, p->d_lr, vMin,
                       best, dmap, occlusion, filled);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

}  // namespace smx
