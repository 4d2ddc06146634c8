// smx_agg_v3.hip -- fused guided-filter aggregation for gfx950: one kernel per slice chunk does
// cost build -> integral (p, I*p) -> box -> a_k, b_k -> integral (a, b) -> box -> q, with a_k, b_k
// never leaving the CU.  Reference: guidedFilter.cu:58-238, costVolume.cu:163-190, integral.cu:78-131.
//
// Work item = (slice-view sv, strip k): a column strip of OW = 64 output columns, walked top -> bottom
// in bands of BH = 32 rows by one 1024-thread workgroup (one per CU, 132 KB of LDS).  Two LDS rings of
// 96 rows x 83 columns of float2 hold the integral images of stage 1 (p, I*p) and stage 2 (a, b): a
// box mean needs the 2R+2 most recent rows, the third band of the ring is the one being produced.
//
// Exactness: every prefix sum keeps the reference's order (sequential left -> right in a row, then
// sequential top -> bottom in a column; integral.cu:82-86, 124-128).  The row scan of a strip starts
// from the running row sums of the strip to its left ("carry", handed over through a small global
// scratch), the column scan keeps its running sum in a register down the strip.  Strips overlap by
// 2R+1 recomputed columns.  Stage 2 lags stage 1 by R rows and R columns; the 2R+1 columns of a, b it
// needs from the strip to the left come through the same scratch ("halo").
//
// Schedule (software pipeline over bands, two workgroup barriers per band):
//   step A(i):  wave 0: row scan stage 1 of band i+1 | wave 1: column scan stage 2 of band i-1 |
//               waves 2-15: box means of stage 1 -> a_k, b_k of band i into ring 2 (+ halo columns)
//   step B(i):  wave 0: row scan stage 2 of band i   | wave 1: column scan stage 1 of band i+1 |
//               waves 2-15: box means of stage 2 -> q of band i-1 to HBM, then cost of band i+2 -> ring 1
// Lane mapping: LANE = ROW in the row scans, LANE = COLUMN everywhere else, so every global access is
// a row-major coalesced run and the aggregated volume comes out in the reference's [z][y][x] layout.
//
// Items are handed out by a ticket counter in strip-major order, so the left neighbour of an item
// always holds an earlier ticket (it is running or done: no deadlock whatever the dispatch order).
// The hand-off is the sc1 form of the guide (write-through stores, drained, one flag per item that
// counts finished bands; sc1 loads behind a relaxed poll + workgroup barrier; no acquire fence).
//
// MODE GUID runs stage 1 alone on (I, I*I) and writes mean_I and 1/(var_I + eps) (guidedFilter.cu:58-123).
// SRC_COST reads p from a materialised [z][y][x] cost volume (the reference's calling convention,
// guidedFilter.cu:198) instead of building it from the two images.
//
// Must be compiled with -ffp-contract=off.
#include <string.h>

#include "smx_common.h"
#include "smx_launch.h"

namespace smx {
namespace v3 {

constexpr int OW = 64;                  // output columns per strip = one wave
constexpr int RMAX = 9;                 // largest supported box radius
constexpr int HWMAX = 2 * RMAX + 1;     // halo / overlap columns
constexpr int TWMAX = OW + HWMAX;       // ring columns in use (83 at R = 9)
constexpr int PITCH = 86;               // float2 per ring row: even (16-B aligned column pairs) and
                                        // PITCH/2 odd (LANE = ROW 16-B accesses hit distinct banks)
constexpr int BH = 32;                  // band height
constexpr int RR = 3 * BH;              // ring rows
constexpr int NT = 1024;
constexpr int NWAVE = NT / 64;
constexpr int W_R = 0;                  // row-scan wave
constexpr int W_C = 1;                  // column-scan wave
constexpr int W_B0 = 2;                 // first box / eval wave
constexpr int NWB = NWAVE - W_B0;       // 14
constexpr int NTB = NWB * 64;           // 896
constexpr int W_POLL = NWAVE - 1;       // wave whose lane 0 polls the left neighbour's flag
static_assert(BH * HWMAX <= NTB, "halo columns of a band are loaded in one pass");
static_assert(RR >= 2 * BH + 2 * RMAX + 2, "ring too small for the two-step pipeline");

enum Mode { GUID = 0, AGG = 1 };
enum Src { SRC_IMG = 0, SRC_COST = 1 };

typedef _Float16 fg_t __attribute__((ext_vector_type(2)));   // (pixel value, x-derivative), exact in fp16
typedef float f2 __attribute__((ext_vector_type(2)));

struct View {
    const fg_t* FG1;      // this view's image plane [h][w+2], sentinel columns at x = -1 and x = w
    const fg_t* FG2;      // the other view's (SRC_IMG)
    const float* cost;    // SRC_COST: [slice][h][w]
    const float* mean;    // AGG in : mean_I [h][w]
    const float* cinv;    // AGG in : 1/(var_I + eps) [h][w]
    float* gmean;         // GUID out
    float* gcinv;         // GUID out
    uint8_t* mean_u8;     // GUID out (optional)
    float* q;             // AGG out: [slice][h][w]
    int d0;               // disparity of local slice 0
};

struct Args {
    View v[2];
    int w, h, R, K, NB, nslices, nsv, nitems;
    f2* carry;            // [parity][stage][sv][h]
    f2* halo;             // [parity][sv][h][HWMAX]
    unsigned* flags;      // [sv][K]   finished-band counters (zeroed before every launch)
    unsigned* ticket;     // work-item counter            (zeroed before every launch)
    unsigned* status;     // != 0: a flag wait timed out (results invalid)
    CostConst cc;
    double eps;
};

// RN(1/d) for d = 0 .. 361 (box areas are (xmax-xmin)*(ymax-ymin) <= 19*19)
struct RcpTable {
    float v[HWMAX * HWMAX + 1];
    constexpr RcpTable() : v() {
        v[0] = 0.0f;
        for (int i = 1; i <= HWMAX * HWMAX; ++i) v[i] = 1.0f / (float)i;
    }
};
__constant__ RcpTable kRcp = RcpTable();

// x / d, correctly rounded, for an integer-valued d in [1, 361] with r = RN(1/d): one residual
// correction step (Markstein).  Bit-identical to IEEE division for |x| >= 2^-100 (exhaustive over
// the significand for every area: tools/check_fastdiv.c); callers route smaller |x| (incl. +-0, whose
// sign the correction would lose) and non-finite x to the true division.
__device__ __forceinline__ float div_small_int(float x, float d, float r) {
    float q = x * r;
    float e = __builtin_fmaf(-q, d, x);
    return __builtin_fmaf(e, r, q);
}
__device__ __forceinline__ bool div_needs_exact(float x) {
    const float ax = fabsf(x);
    return !(ax >= 0x1p-100f && ax < __builtin_inff());   // tiny, zero, inf or NaN
}
__device__ __forceinline__ f2 box_div(f2 val, float area, float rarea) {
    f2 m;
    m.x = div_small_int(val.x, area, rarea);
    m.y = div_small_int(val.y, area, rarea);
    const bool slow = div_needs_exact(val.x) || div_needs_exact(val.y);
    if (__any(slow)) {
        asm volatile("; exact-division slow path");   // keep this a real (rare) wave-uniform branch
        m.x = 1.0f * val.x / area;
        m.y = 1.0f * val.y / area;
    }
    return m;
}

// p = (1-alpha)*min(|I1 - I2|, 7) + alpha*min(|g1 - g2|, 2) and I1*p  (costVolume.cu:187,
// guidedFilter.cu:209).  The halves convert exactly, so the f32 operations equal the reference's; the
// sentinel 60000 of an out-of-range partner saturates both terms = the border constant (:184).
__device__ __forceinline__ f2 cost_pair(fg_t q1, fg_t q2, const CostConst& cc) {
    const float a1 = (float)q1.x, b1 = (float)q1.y, a2 = (float)q2.x, b2 = (float)q2.y;
    float t1 = fabsf(a1 - a2);
    float t2 = fabsf(b1 - b2);
    float m1 = t1 < cc.th_color ? t1 : cc.th_color;
    float m2 = t2 < cc.th_grad ? t2 : cc.th_grad;
    float x = cc.oma * m1;
    float z = cc.alpha * m2;
    f2 r;
    r.x = x + z;
    r.y = a1 * r.x;
    return r;
}

// ---- hand-off accesses: sc1 (bypass this CU's L1, write through the XCD's L2) ----------------
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int AUX_SC1 = 16;
__device__ __forceinline__ rsrc_t mk_rsrc(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0,
                                             (int)(bytes > 0xFFFFFFFFull ? 0xFFFFFFFFull : bytes),
                                             0x00020000);
}
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 ld_sc1(rsrc_t r, unsigned byteoff) {
    u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)byteoff, 0, AUX_SC1);
    return __builtin_bit_cast(f2, v);
}
__device__ __forceinline__ void st_sc1(rsrc_t r, unsigned byteoff, f2 v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), r, (int)byteoff, 0, AUX_SC1);
}
typedef __attribute__((address_space(1))) unsigned gu32;
__device__ __forceinline__ unsigned flag_load(unsigned* p) {
    return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void flag_store(unsigned* p, unsigned v) {
    __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---------------------------------------------------------------------------------------------
// prep: u8 image [h][w] -> (value, x-derivative) half2 plane [h][w+2] with sentinel columns.
// grid (ceil((w+2)/256), h, nimages)
// ---------------------------------------------------------------------------------------------
struct PrepArgs {
    const uint8_t* I[2];
    fg_t* FG[2];
};

__global__ void k_v3_prep(PrepArgs pa, int w, int h) {
    const uint8_t* __restrict__ I = pa.I[blockIdx.z];
    fg_t* __restrict__ FG = pa.FG[blockIdx.z];
    const int xp = blockIdx.x * blockDim.x + threadIdx.x;   // padded column
    const int y = blockIdx.y;
    if (xp >= w + 2) return;
    const int x = xp - 1;
    float f = 60000.0f, g = 60000.0f;
    if (x >= 0 && x < w) {
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
    FG[(size_t)y * (w + 2) + xp] = v;
}

// ---------------------------------------------------------------------------------------------
// the walker
// ---------------------------------------------------------------------------------------------
template <int MODE, int SRC>
__global__ __launch_bounds__(NT) void k_v3_walk(Args A) {
    __shared__ __attribute__((aligned(16))) f2 ring1[RR * PITCH];
    __shared__ __attribute__((aligned(16))) f2 ring2[MODE == AGG ? RR * PITCH : 2];
    __shared__ int s_item;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = A.w, h = A.h, R = A.R, K = A.K, NB = A.NB, nsv = A.nsv;
    const int HW = 2 * R + 1, TW = OW + HW;
    const float invTW = 1.0f / (float)TW, invHW = 1.0f / (float)HW;
    const CostConst cc = A.cc;
    const bool is_b = wave >= W_B0;
    const int wb = wave - W_B0;                 // box-wave index
    const int tb = wb * 64 + lane;              // thread index among the box waves
    const f2 ident = {-0.0f, -0.0f};            // exact additive identity: v + (-0) == v

    for (;;) {
        if (tid == 0) s_item = (int)__hip_atomic_fetch_add((gu32*)A.ticket, 1u, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int item = s_item;
        if (item >= A.nitems) break;
        const int k = item / nsv;
        const int sv = item - k * nsv;
        const int view = MODE == GUID ? sv : sv / A.nslices;
        const int slice = MODE == GUID ? 0 : sv - view * A.nslices;
        const View& V = A.v[view];
        const int xs = k * OW;
        const int cs1 = xs - R - 1;             // image column of ring-1 column 0
        const int cs2 = xs - HW;                // image column of ring-2 column 0
        const int jlo1 = max(0, -cs1), jhi1 = min(TW, w - cs1);   // ring-1 columns inside the image
        const int jlo2 = max(0, -cs2), jhi2 = min(TW, w - cs2);
        const bool pred = k > 0, succ = k + 1 < K;
        const int d = V.d0 + slice;
        unsigned* const myflag = A.flags + (size_t)sv * K + k;
        bool pred_done = !pred;                 // lane 0 of W_POLL: no more polling needed
        // hand-off scratch of this item (in: written by strip k-1, out: read by strip k+1)
        const size_t hrow = (size_t)h;
        const int pin = (k - 1) & 1, pout = k & 1;
        const rsrc_t c1_in = mk_rsrc(A.carry + ((size_t)(pin * 2 + 0) * nsv + sv) * hrow, hrow * 8);
        const rsrc_t c2_in = mk_rsrc(A.carry + ((size_t)(pin * 2 + 1) * nsv + sv) * hrow, hrow * 8);
        const rsrc_t c1_out = mk_rsrc(A.carry + ((size_t)(pout * 2 + 0) * nsv + sv) * hrow, hrow * 8);
        const rsrc_t c2_out = mk_rsrc(A.carry + ((size_t)(pout * 2 + 1) * nsv + sv) * hrow, hrow * 8);
        const rsrc_t h_in = mk_rsrc(A.halo + ((size_t)pin * nsv + sv) * hrow * HWMAX, hrow * HWMAX * 8);
        const rsrc_t h_out = mk_rsrc(A.halo + ((size_t)pout * nsv + sv) * hrow * HWMAX, hrow * HWMAX * 8);
        const fg_t* __restrict__ FG1 = V.FG1;
        const fg_t* __restrict__ FG2 = V.FG2;
        const size_t fgw = (size_t)w + 2;

        // running column sums of the column-scan wave: lane = ring column (second: lane + 64)
        f2 S1a = ident, S1b = ident, S2a = ident, S2b = ident;

        // ---- stage-1 inputs of band b -> ring 1 (box waves, LANE = flattened (row, column)) -----
        constexpr int NE = (BH * TWMAX + NTB - 1) / NTB;   // cells per thread
        auto e1_issue = [&](int b, uint32_t (&ua)[NE], uint32_t (&ub)[NE]) {
            const int y0 = b * BH;
            const int nrows = min(BH, h - y0);
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int t = tb + e * NTB;
                const int r = (int)(((float)t + 0.5f) * invTW);
                const int j = t - r * TW;
                const int c = cs1 + j;
                ua[e] = 0; ub[e] = 0;
                if (r < nrows && c >= 0 && c < w) {
                    const size_t row = (size_t)(y0 + r);
                    ua[e] = __builtin_bit_cast(uint32_t, FG1[row * fgw + c + 1]);
                    if (MODE == AGG && SRC == SRC_IMG) {
                        int xx = c + d;
                        xx = xx < -1 ? -1 : (xx > w ? w : xx);   // sentinel columns
                        ub[e] = __builtin_bit_cast(uint32_t, FG2[row * fgw + xx + 1]);
                    }
                    if (MODE == AGG && SRC == SRC_COST)
                        ub[e] = __builtin_bit_cast(uint32_t, V.cost[((size_t)slice * h + row) * w + c]);
                }
            }
        };
        auto e1_finish = [&](int b, const uint32_t (&ua)[NE], const uint32_t (&ub)[NE]) {
            const int y0 = b * BH;
            const int nrows = min(BH, h - y0);
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int t = tb + e * NTB;
                const int r = (int)(((float)t + 0.5f) * invTW);
                const int j = t - r * TW;
                const int c = cs1 + j;
                if (r < nrows && c >= 0 && c < w) {
                    const fg_t q1 = __builtin_bit_cast(fg_t, ua[e]);
                    f2 v;
                    if (MODE == GUID) {
                        v.x = (float)q1.x;            // chToFlOnGPU guidedFilter.cu:442-449
                        v.y = v.x * v.x;              // pixelMultOnGPU(d_im, d_im) :111
                    } else if (SRC == SRC_IMG) {
                        v = cost_pair(q1, __builtin_bit_cast(fg_t, ub[e]), cc);
                    } else {
                        v.x = __builtin_bit_cast(float, ub[e]);   // copyFromBigToLittleOnGPU :198
                        v.y = (float)q1.x * v.x;                  // pixelMultOnGPU(d_im, d_p) :209
                    }
                    ring1[((y0 + r) % RR) * PITCH + j] = v;
                }
            }
        };

        // ---- row scan of `n` rows starting at image row ylo (LANE = ROW) ----------------------
        auto rowscan = [&](f2* ring, int ylo, int n, int jlo, int jhi, rsrc_t cin, rsrc_t cout) {
            if (lane >= n || jhi <= jlo) return;
            const int y = ylo + lane;
            f2 acc = ident;
            if (pred) acc = ld_sc1(cin, (unsigned)y * 8u);
            f2* row = ring + (y % RR) * PITCH;
            int j = jlo;
            for (; j + 8 <= jhi; j += 8) {
                f2 v[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] = row[j + t];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    acc = v[t] + acc;
                    row[j + t] = acc;
                    if (j + t == OW - 1 && succ) st_sc1(cout, (unsigned)y * 8u, acc);
                }
            }
            for (; j < jhi; ++j) {
                acc = row[j] + acc;
                row[j] = acc;
                if (j == OW - 1 && succ) st_sc1(cout, (unsigned)y * 8u, acc);
            }
        };

        // ---- column scan of rows [ylo, yhi) (LANE = COLUMN, two columns per lane) ---------------
        auto colscan = [&](f2* ring, int ylo, int yhi, f2& Sa, f2& Sb) {
            const bool second = lane + 64 < TW;
            for (int y = ylo; y < yhi; ++y) {
                f2* row = ring + (y % RR) * PITCH;
                f2 va = row[lane];
                f2 vb = second ? row[lane + 64] : ident;
                Sa = va + Sa;
                Sb = vb + Sb;
                row[lane] = Sa;
                if (second) row[lane + 64] = Sb;
            }
        };

        // ---- box mean of output (x, y) from a ring whose column 0 is image column cs -----------
        // (computeBoxFilterOnGPU guidedFilter.cu:305-318: S11 - S10 - S01 + S00 in that order, then
        //  a true division by the clipped window area)
        auto box = [&](const f2* ring, int cs, int x, int y) -> f2 {
            const int ymax = min(h - 1, y + R);
            const int ymin = y - R - 1;
            const bool hy = ymin >= 0;
            const int ych = ymax - (hy ? ymin : -1);
            const int xmax = min(w - 1, x + R);
            const int xmn = x - R - 1;
            const bool hx = xmn >= 0;
            const int xcw = xmax - (hx ? xmn : -1);
            int jmax = xmax - cs, jmin = xmn - cs;
            jmax = min(max(jmax, 0), TW - 1);            // lanes outside the image: stay inside the ring
            jmin = min(max(jmin, 0), TW - 1);
            const f2* r1 = ring + (ymax % RR) * PITCH;
            const f2* r0 = ring + ((hy ? ymin : 0) % RR) * PITCH;
            const f2 s11 = r1[jmax], s10 = r1[jmin], s01 = r0[jmax], s00 = r0[jmin];
            f2 val = s11;
            f2 t = val - s10;
            val = hx ? t : val;
            t = val - s01;
            val = hy ? t : val;
            t = val + s00;
            val = (hx && hy) ? t : val;
            int ai = xcw * ych;
            ai = min(max(ai, 1), HWMAX * HWMAX);
            const float area = (float)ai;
            return box_div(val, area, kRcp.v[ai]);
        };

        // ================= prologue: stage-1 inputs of band 0 ==================================
        if (is_b) {
            uint32_t ua[NE], ub[NE];
            e1_issue(0, ua, ub);
            e1_finish(0, ua, ub);
        }

        // ================= pipelined band loop =================================================
        for (int i = -1; i <= NB; ++i) {
            // left neighbour must have finished steps A(i), B(i): its flag >= i + 2
            if (wave == W_POLL && lane == 0 && !pred_done) {
                const unsigned need = (unsigned)(i + 2);
                const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    const unsigned f = flag_load(myflag - 1);
                    if (f >= need) { pred_done = f >= (unsigned)(NB + 2); break; }
                    __builtin_amdgcn_s_sleep(8);
                    // bounded spin: give up after 2 s (100 MHz counter) or as soon as any workgroup
                    // has given up; the call then reports SMX_E_HIP through smx_dev_agg_status
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull || flag_load(A.status) != 0u) {
                        flag_store(A.status, 1u + (unsigned)item);
                        pred_done = true;
                        break;
                    }
                }
            }
            __syncthreads();
            if (tid == 0 && succ && i >= 0) flag_store(myflag, (unsigned)(i + 1));   // B(i-1) done
            // -------------------------------- step A(i) -----------------------------------------
            if (wave == W_R) {
                const int b = i + 1;                       // row scan, stage 1
                if (b < NB && b * BH < h) rowscan(ring1, b * BH, min(BH, h - b * BH), jlo1, jhi1, c1_in, c1_out);
                drain_vmem();
            } else if (wave == W_C) {
                if (MODE == AGG) {
                    const int b = i - 1;                   // column scan, stage 2
                    if (b >= 0) {
                        if (b == 0) { S2a = ident; S2b = ident; }
                        colscan(ring2, max(0, b * BH - R), min(h, b * BH + BH - R), S2a, S2b);
                    }
                }
            } else {
                const int b = i;                           // box means of stage 1
                const int ylo = max(0, b * BH - R), yhi = min(h, b * BH + BH - R);
                if (b >= 0 && ylo < yhi) {
                    // halo columns of a, b from the strip to the left (issued first, used last)
                    f2 hv = ident;
                    int hy2 = 0, hj = 0;
                    bool hact = false;
                    if (MODE == AGG && pred) {
                        hy2 = (int)(((float)tb + 0.5f) * invHW);
                        hj = tb - hy2 * HW;
                        hact = hy2 < yhi - ylo;
                        hy2 += ylo;
                        if (hact) hv = ld_sc1(h_in, ((unsigned)hy2 * HWMAX + (unsigned)hj) * 8u);
                    }
                    if (xs < w) {
                        const int x = xs + lane;
                        const bool xin = x < w;
                        const int xc = xin ? x : w - 1;
                        for (int y = ylo + wb; y < yhi; y += NWB) {
                            float ga = 0.0f, gb = 0.0f;
                            if (MODE == AGG) {
                                ga = V.mean[(size_t)y * w + xc];
                                gb = V.cinv[(size_t)y * w + xc];
                            }
                            const f2 m = box(ring1, cs1, xc, y);
                            if (MODE == GUID) {
                                float mm = m.x * m.x;          // pixelMultOnGPU(mean, mean) guidedFilter.cu:112
                                float var = m.y - mm;          // pixelSousOnGPU :121
                                float c = (float)(1.0f / ((double)var + A.eps));   // :350
                                if (xin) {
                                    V.gmean[(size_t)y * w + x] = m.x;
                                    V.gcinv[(size_t)y * w + x] = c;
                                    if (V.mean_u8) {           // flToChOnGPU :451-458
                                        int ci8 = (int)m.x;
                                        V.mean_u8[(size_t)y * w + x] = (ci8 > 255) ? 255 : (uint8_t)ci8;
                                    }
                                }
                            } else {
                                float mm = ga * m.x;           // compute_ak_and_bk guidedFilter.cu:345-354
                                float ak = 1.0f * (m.y - mm) * gb;
                                float mb2 = 1.0f * ga * ak;
                                float bk = 1.0f * m.x - mb2;
                                f2 ab = {ak, bk};
                                ring2[(y % RR) * PITCH + HW + lane] = ab;
                                if (succ && lane >= OW - HW)
                                    st_sc1(h_out, ((unsigned)y * HWMAX + (unsigned)(lane - (OW - HW))) * 8u, ab);
                            }
                        }
                    }
                    if (MODE == AGG && hact) ring2[(hy2 % RR) * PITCH + hj] = hv;
                }
                drain_vmem();
            }
            __syncthreads();
            // -------------------------------- step B(i) -----------------------------------------
            if (wave == W_R) {
                if (MODE == AGG) {
                    const int b = i;                       // row scan, stage 2
                    const int ylo = max(0, b * BH - R), yhi = min(h, b * BH + BH - R);
                    if (b >= 0 && ylo < yhi) rowscan(ring2, ylo, yhi - ylo, jlo2, jhi2, c2_in, c2_out);
                    drain_vmem();
                }
            } else if (wave == W_C) {
                const int b = i + 1;                       // column scan, stage 1
                if (b < NB && b * BH < h) {
                    if (b == 0) { S1a = ident; S1b = ident; }
                    colscan(ring1, b * BH, min(h, b * BH + BH), S1a, S1b);
                }
            } else {
                uint32_t ua[NE], ub[NE];
                const int be = i + 2;                      // stage-1 inputs two bands ahead
                const bool ev = be < NB && be * BH < h;
                if (ev) e1_issue(be, ua, ub);
                if (MODE == AGG) {
                    const int b = i - 1;                   // box means of stage 2 -> q
                    const int ylo = max(0, b * BH - 2 * R), yhi = min(h, b * BH + BH - 2 * R);
                    const int x = xs - R + lane;
                    if (b >= 0 && ylo < yhi) {
                        const bool xin = x >= 0 && x < w;
                        const int xc = min(max(x, 0), w - 1);
                        float* __restrict__ qp = V.q + (size_t)slice * h * w;
                        for (int y = ylo + wb; y < yhi; y += NWB) {
                            const float I = (float)FG1[(size_t)y * fgw + xc + 1].x;
                            const f2 m = box(ring2, cs2, xc, y);
                            float tq = m.x * I;                // compute_q guidedFilter.cu:363-369
                            if (xin) __builtin_nontemporal_store(tq + m.y, &qp[(size_t)y * w + x]);
                        }
                    }
                }
                if (ev) e1_finish(be, ua, ub);
            }
        }
        __syncthreads();
        if (tid == 0 && succ) flag_store(myflag, (unsigned)(NB + 2));
    }
}

// ---------------------------------------------------------------------------------------------
// WTA over the chunk's q planes [slice][h][w].  One lane per pixel.  grid (ceil(n/256), nviews)
// (dispSelectOnGPU guidedFilter.cu:403-411 in packed-key form)
// ---------------------------------------------------------------------------------------------
struct WtaArgs {
    const float* q[2];
    uint64_t* keys[2];
};

__global__ __launch_bounds__(256) void k_v3_wta(WtaArgs wa, size_t n, int count, int slice0) {
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= n) return;
    const float* __restrict__ q = wa.q[blockIdx.y] + id;
    uint64_t* keys = wa.keys[blockIdx.y];
    uint64_t key = keys[id];
    int z = 0;
    for (; z + 8 <= count; z += 8) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = __builtin_nontemporal_load(&q[(size_t)(z + t) * n]);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            uint64_t kk = pack_key(v[t], (uint32_t)(slice0 + z + t));
            key = kk < key ? kk : key;
        }
    }
    for (; z < count; ++z) {
        uint64_t kk = pack_key(__builtin_nontemporal_load(&q[(size_t)z * n]), (uint32_t)(slice0 + z));
        key = kk < key ? kk : key;
    }
    keys[id] = key;
}

}  // namespace v3

// =============================================================================================
// host orchestration
// =============================================================================================
static inline unsigned cdivu3(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

struct V3Layout {
    int K, NB;
    size_t fg;        // floats per image plane (half2 = 4 B per pixel)
    size_t plane;     // floats per w*h plane
    size_t sv_carry;  // floats of carry scratch per slice-view
    size_t sv_halo;   // floats of halo scratch per slice-view
};

static V3Layout v3_layout(int w, int h, int R) {
    V3Layout L;
    L.K = (w + R + v3::OW - 1) / v3::OW;
    L.NB = (h + 2 * R + v3::BH - 1) / v3::BH;
    L.fg = (size_t)(w + 2) * h;
    L.plane = (size_t)w * h;
    L.sv_carry = (size_t)2 * 2 * h * 2;              // parity x stage x rows x float2
    L.sv_halo = (size_t)2 * h * v3::HWMAX * 2;       // parity x rows x columns x float2
    return L;
}

bool v3_supported(const smx_params* p) { return p->radius >= 0 && p->radius <= v3::RMAX; }

constexpr size_t V3_CTRL_BYTES = 256;   // ticket, status (zeroed with the flags before every launch)

static size_t v3_flag_bytes(const V3Layout& L, int nsv) {
    return align_up(V3_CTRL_BYTES + (size_t)nsv * L.K * sizeof(unsigned), 256);
}

// bytes for ONE view with `nslices` slices in flight (q planes included)
size_t v3_workspace_bytes(int w, int h, int nslices) {
    V3Layout L = v3_layout(w, h, v3::RMAX);
    size_t b = 0;
    b += 2 * align_up(L.fg * 4, 256);                               // both image planes (single-view calls too)
    b += 2 * align_up(L.plane * 4, 256);                            // mean_I, 1/(var+eps)
    b += align_up((L.sv_carry + L.sv_halo) * 4, 256);               // guidance scratch
    b += v3_flag_bytes(L, 2);                                       // guidance control block (shared)
    b += (size_t)nslices * align_up(L.plane * 4, 256);              // q
    b += align_up((size_t)nslices * (L.sv_carry + L.sv_halo) * 4, 256);
    b += v3_flag_bytes(L, 2 * nslices);                             // control block (shared by both views)
    return b + 16 * 256;
}

template <int MODE, int SRC>
static int launch_walk3(const v3::Args& a, hipStream_t st) {
    int dev = 0, ncu = 256;
    SMX_HIP(hipGetDevice(&dev));
    SMX_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    const int grid = a.nitems < ncu ? a.nitems : ncu;   // persistent: one workgroup per CU
    hipLaunchKernelGGL((v3::k_v3_walk<MODE, SRC>), dim3((unsigned)grid), dim3(v3::NT), 0, st, a);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

// Aggregation + WTA of slices [s_begin, s_end) of `nviews` (1 or 2) views.  View v uses d_guide[v]
// as guidance; its cost slices are d_cost[v] (materialised, slice s at (s - s_begin)*w*h) or, when
// d_cost[v] == NULL, are built on the fly against d_guide[v ^ 1] (nviews == 2) / d_other[0].
// status word of the last fused aggregation that used this workspace (0 = ok)
int v3_read_status(const void* d_ws, unsigned* out) {
    const char* base = (const char*)align_up((size_t)d_ws, 256);
    SMX_HIP(hipMemcpy(out, base, sizeof(unsigned), hipMemcpyDeviceToHost));
    return SMX_OK;
}

int aggregate_v3(const smx_params* p, int nviews, const uint8_t* const* d_guide,
                 const uint8_t* const* d_other, const float* const* d_cost, int w, int h,
                 const int* dmin, int s_begin, int s_end, uint64_t* const* d_keys,
                 uint8_t* const* d_mean_u8, float* const* d_agg, void* d_ws, size_t ws_bytes,
                 hipStream_t st, int* launches) {
    const int R = p->radius;
    const V3Layout L = v3_layout(w, h, R);
    const bool use_cost = d_cost && d_cost[0];
    if (use_cost && nviews == 2 && !d_cost[1])
        return fail(SMX_E_ARG, "aggregate_v3: both views need a cost volume or none");
    char* base = (char*)align_up((size_t)d_ws, 256);
    size_t avail = ws_bytes > (size_t)(base - (char*)d_ws) ? ws_bytes - (size_t)(base - (char*)d_ws) : 0;
    bool oom = false;
    auto carve = [&](size_t bytes) {
        char* r = base;
        size_t b = align_up(bytes, 256);
        if (b > avail) { oom = true; b = avail; }
        base += b;
        avail -= b;
        return (void*)r;
    };
    // first 256 B: status word of the call (smx_dev_agg_status)
    unsigned* status = (unsigned*)carve(256);
    if (oom) return fail(SMX_E_WS, "aggregate_v3: workspace too small");
    SMX_HIP(hipMemsetAsync(status, 0, 256, st));
    // fixed part: image planes, guidance statistics, guidance scratch + control
    v3::fg_t* FG[2];
    float *meanI[2], *cinv[2];
    for (int i = 0; i < 2; ++i) FG[i] = (v3::fg_t*)carve(L.fg * 4);
    for (int v = 0; v < nviews; ++v) { meanI[v] = (float*)carve(L.plane * 4); cinv[v] = (float*)carve(L.plane * 4); }
    v3::f2* gcarry = (v3::f2*)carve((size_t)nviews * L.sv_carry * 4);
    v3::f2* ghalo = (v3::f2*)carve(256);   // unused by the single-stage mode
    char* gctrl = (char*)carve(v3_flag_bytes(L, nviews));
    const int total = s_end - s_begin;
    // per slice-view: q plane (unless the caller's volume is written directly) + scratch + flags
    const bool own_q = !(d_agg && d_agg[0]);
    const size_t per_sv = (own_q ? align_up(L.plane * 4, 256) : 0) + (L.sv_carry + L.sv_halo) * 4 +
                          (size_t)L.K * sizeof(unsigned);
    size_t fit = avail > 8 * 256 + V3_CTRL_BYTES ? (avail - 8 * 256 - V3_CTRL_BYTES) / (per_sv * nviews) : 0;
    if (oom || (fit < 1 && total > 0))
        return fail(SMX_E_WS, "aggregate_v3: workspace %zu B too small (need >= %zu B per view)",
                    ws_bytes, v3_workspace_bytes(w, h, 1));
    int chunk = fit > (size_t)total ? total : (int)fit;
    if (chunk < 1) chunk = 1;
    const int nsv_max = chunk * nviews;
    float* qbuf[2] = {nullptr, nullptr};
    if (own_q)
        for (int v = 0; v < nviews; ++v) qbuf[v] = (float*)carve((size_t)chunk * align_up(L.plane * 4, 256));
    v3::f2* carry = (v3::f2*)carve((size_t)nsv_max * L.sv_carry * 4);
    v3::f2* halo = (v3::f2*)carve((size_t)nsv_max * L.sv_halo * 4);
    char* ctrl = (char*)carve(v3_flag_bytes(L, nsv_max));
    if (oom) return fail(SMX_E_WS, "aggregate_v3: workspace carve overflow");
    // plane stride of q: w*h floats exactly (kernels index planes as slice*w*h), so the 256-B
    // rounding above is only slack
    int nl = 0, rc;

    v3::PrepArgs pa;
    pa.I[0] = d_guide[0];
    pa.I[1] = nviews == 2 ? d_guide[1] : (d_other ? d_other[0] : nullptr);
    pa.FG[0] = FG[0]; pa.FG[1] = FG[1];
    const int nimg = pa.I[1] ? 2 : 1;
    hipLaunchKernelGGL(v3::k_v3_prep, dim3(cdivu3(w + 2, 256), h, nimg), dim3(256), 0, st, pa, w, h);
    SMX_HIP(hipGetLastError());
    ++nl;

    v3::Args a0;
    memset(&a0, 0, sizeof(a0));
    a0.w = w; a0.h = h; a0.R = R; a0.K = L.K; a0.NB = L.NB;
    a0.cc = make_cost_const(p);
    a0.eps = p->eps;
    for (int v = 0; v < nviews; ++v) {
        a0.v[v].FG1 = FG[v]; a0.v[v].FG2 = FG[v ^ 1];
        a0.v[v].mean = meanI[v]; a0.v[v].cinv = cinv[v];
    }
    // ---- guidance statistics (single-stage walker, one slice-view per view) --------------------
    {
        v3::Args g = a0;
        for (int v = 0; v < nviews; ++v) {
            g.v[v].gmean = meanI[v]; g.v[v].gcinv = cinv[v];
            g.v[v].mean_u8 = d_mean_u8 ? d_mean_u8[v] : nullptr;
        }
        g.nslices = 1; g.nsv = nviews; g.nitems = nviews * L.K;
        g.carry = gcarry; g.halo = ghalo;
        g.ticket = (unsigned*)gctrl; g.status = status;
        g.flags = (unsigned*)(gctrl + V3_CTRL_BYTES);
        SMX_HIP(hipMemsetAsync(gctrl, 0, v3_flag_bytes(L, nviews), st));
        if ((rc = launch_walk3<v3::GUID, v3::SRC_IMG>(g, st))) return rc;
        nl += 2;
    }
    for (int s0 = s_begin; s0 < s_end; s0 += chunk) {
        const int cnt = (s_end - s0) < chunk ? (s_end - s0) : chunk;
        v3::Args a = a0;
        v3::WtaArgs wa;
        for (int v = 0; v < 2; ++v) {
            const int vv = v < nviews ? v : 0;
            float* qv = own_q ? qbuf[vv] : d_agg[vv] + (size_t)(s0 - s_begin) * L.plane;
            if (v < nviews) {
                a.v[v].q = qv;
                a.v[v].d0 = dmin[v] + s0;
                a.v[v].cost = use_cost ? d_cost[v] + (size_t)(s0 - s_begin) * L.plane : nullptr;
            }
            wa.q[v] = qv;
            wa.keys[v] = d_keys[vv];
        }
        a.nslices = cnt; a.nsv = cnt * nviews; a.nitems = a.nsv * L.K;
        a.carry = carry; a.halo = halo;
        a.ticket = (unsigned*)ctrl; a.status = status;
        a.flags = (unsigned*)(ctrl + V3_CTRL_BYTES);
        SMX_HIP(hipMemsetAsync(ctrl, 0, v3_flag_bytes(L, a.nsv), st));
        if (use_cost) rc = launch_walk3<v3::AGG, v3::SRC_COST>(a, st);
        else rc = launch_walk3<v3::AGG, v3::SRC_IMG>(a, st);
        if (rc) return rc;
        hipLaunchKernelGGL(v3::k_v3_wta, dim3(cdivu3((int64_t)L.plane, 256), nviews), dim3(256), 0, st, wa,
                           L.plane, cnt, s0);
        SMX_HIP(hipGetLastError());
        nl += 3;
    }
    if (launches) *launches = nl;
    return SMX_OK;
}

}  // namespace smx
