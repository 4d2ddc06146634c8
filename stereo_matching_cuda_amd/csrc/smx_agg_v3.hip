// smx_agg_v3.hip -- fused guided-filter aggregation for gfx950: one kernel per slice chunk does
// cost build -> integral (p, I*p) -> box -> a_k, b_k -> integral (a, b) -> box -> q, with a_k, b_k
// never leaving the CU.  Reference: guidedFilter.cu:58-238, costVolume.cu:163-190, integral.cu:78-131.
//
// Work item = (slice-view sv, strip k): a column strip of OW = 64 output columns, walked top -> bottom
// in bands of BH = 26 rows by one 1024-thread workgroup (one per CU, ~112 KB of LDS).  Two LDS rings of
// 78 rows x 83 columns of float2 hold the integral images of stage 1 (p, I*p) and stage 2 (a, b): a
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
//               wave 2: hand-off I/O | waves 3-15: box means of stage 1 -> a_k, b_k of band i into ring 2
//   step B(i):  wave 0: row scan stage 2 of band i   | wave 1: column scan stage 1 of band i+1 |
//               wave 2: hand-off I/O | waves 3-15: box means of stage 2 -> q of band i-1 to HBM, then the
//               cost of band i+2 -> ring 1
// Lane mapping: in the row scans lanes 0..25 / 32..57 are the rows of the band for the first / second
// component; LANE = COLUMN everywhere else, so every global access is a row-major coalesced run and the
// aggregated volume comes out in the reference's [z][y][x] layout.
//
// Items are handed out by a ticket counter in strip-major order, so the left neighbour of an item
// always holds an earlier ticket (it is running or done: no deadlock whatever the dispatch order).
// The hand-off is the sc1 form of the guide (write-through stores, drained, one flag per item that
// counts finished steps; sc1 loads behind a relaxed poll + workgroup barrier; no acquire fence).
//
// MODE GUID runs stage 1 alone on (I, I*I) and writes mean_I and 1/(var_I + eps) (guidedFilter.cu:58-123);
// MODE FUSED hands those guidance items out first in the same launch as the aggregation items.
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
constexpr int PITCH = 85;               // float2 per ring row: odd, so the LANE = ROW dword accesses of the row
                                        // scan (row stride 2 PITCH dwords) hit distinct banks
#ifndef SMX_V3_NT
#define SMX_V3_NT 1024
#endif
#ifndef SMX_V3_WGPCU
#define SMX_V3_WGPCU 1
#endif
constexpr int NT = SMX_V3_NT;
constexpr int NWAVE = NT / 64;
constexpr int W_R = 0;                  // row-scan wave
constexpr int W_C = 1;                  // column-scan wave (ring columns 0 .. 63)
constexpr int W_IO = 2;                 // hand-off I/O wave (+ column scan of ring columns 64 ..)
constexpr int W_B0 = 3;                 // first box / eval wave
constexpr int NWB = NWAVE - W_B0;       // 13
constexpr int BH = 2 * NWB;             // band height: every box wave owns two rows of a band
constexpr int RR = 3 * BH;              // ring rows
static_assert(RR >= 2 * BH + 2 * RMAX + 2, "ring too small for the two-step pipeline");
static_assert(RR == 3 * BH, "band i starts at ring row (i mod 3) * BH");

enum Mode { GUID = 0, AGG = 1, FUSED = 2 };   // FUSED: guidance items + aggregation items in one launch
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
    int nviews, nguid;    // FUSED: the first nguid = nviews * K items are the guidance items (view, strip)
    f2* hand;             // hand-off records [parity][sv][band] (see REC_F2)
    f2* ghand;            // FUSED: hand-off records of the guidance items [parity][view][band]
    unsigned* flags;      // [sv][K]   finished-band counters (zeroed before every launch)
    unsigned* gflags;     // FUSED: [view][K] the same for the guidance items
    unsigned* gready;     // FUSED: [view][K] bands of mean_I / 1/(var+eps) that are complete and visible
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
typedef __attribute__((address_space(1))) float gf32;
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
// Diagnostic build only (-DSMX_V3_STAMPS, tools/v3_stamps.sh): every wave of one work item records
// the shader clock around its two barriers per band; the product build contains no stamp.
#ifdef SMX_V3_STAMPS
constexpr int STAMP_SLOTS = 4 * 96;
__device__ unsigned long long g_stamps[NWAVE * STAMP_SLOTS];
#define V3_STAMP(n)                                                                          \
    do {                                                                                     \
        if (item == SMX_V3_STAMPS && lane == 0 && (i + 2) * 4 + (n) < STAMP_SLOTS)            \
            g_stamps[wave * STAMP_SLOTS + (i + 2) * 4 + (n)] = __builtin_amdgcn_s_memtime();  \
    } while (0)
#else
#define V3_STAMP(n) ((void)0)
#endif
// Diagnostic build only (-DSMX_V3_ITEMLOG, tools/v3_itemlog.sh): start / end clock and CU of every work item.
#ifdef SMX_V3_ITEMLOG
constexpr int ITEMLOG_MAX = 1 << 16;
__device__ unsigned long long g_itemlog[3 * ITEMLOG_MAX];
#endif
// Workgroup barrier that orders LDS only: __syncthreads() would also wait for every outstanding
// global load and store (vmcnt(0)), which is exactly the latency the cross-step prefetches hide.
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <bool B> struct BoolC { static constexpr bool value = B; };
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 ld16_sc1(rsrc_t r, unsigned byteoff) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byteoff, 0, AUX_SC1));
}
__device__ __forceinline__ void st16_sc1(rsrc_t r, unsigned byteoff, f4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, (int)byteoff, 0, AUX_SC1);
}
__device__ __forceinline__ f2 lo2(f4 v) { f2 r = {v.x, v.y}; return r; }
__device__ __forceinline__ f2 hi2(f4 v) { f2 r = {v.z, v.w}; return r; }

// Hand-off record of one band (per parity and slice-view), written and read in 16-byte units:
//   [0, BH)            stage-1 row carries (float2 per row of the band)
//   [BH, 2 BH)         stage-2 row carries
//   [2 BH, 2 BH + BH*HP) last 2R+1 columns of a, b, HP = 20 float2 per row
constexpr int HP = HWMAX + 1;
constexpr int REC_F2 = 2 * BH + BH * HP;          // float2 per band record
constexpr int NHU = (BH * HP / 2 + 63) / 64;      // 16-byte halo units per lane of the I/O wave
static_assert(BH % 2 == 0 && HP % 2 == 0 && (REC_F2 % 2) == 0, "16-byte hand-off units");
static_assert(BH == 2 * NWB, "every box wave owns two rows of a band");
static_assert(BH % 2 == 0 && BH / 2 <= 16, "the row scan maps half of the rows to 16 lanes of each half wave");

template <int MODE, int SRC>
__global__ __launch_bounds__(NT, (SMX_V3_WGPCU * (NT / 64) + 3) / 4) void k_v3_walk(Args A) {
    __shared__ __attribute__((aligned(16))) f2 ring1[RR * PITCH];
    // ring 1 keeps stage-1 row y at ring row y mod RR; ring 2 keeps a/b row y at (y + R) mod RR, so that the
    // R-lagged bands of stage 2 start at a band slot like those of stage 1 and never wrap inside a band
    __shared__ __attribute__((aligned(16))) f2 ring2[MODE != GUID ? RR * PITCH : 2];
    // hand-off staging (all global hand-off traffic goes through the I/O wave, one step delayed):
    __shared__ __attribute__((aligned(16))) f2 cin[2][BH];    // row carries in : stage -> rows of the band
    __shared__ __attribute__((aligned(16))) f2 cout[2][BH];   // row carries out
    __shared__ __attribute__((aligned(16))) f2 hout[BH * HP]; // last 2R+1 columns of a, b of the band (AGG)
    __shared__ float rcp_s[HWMAX * HWMAX + 1];                // RN(1/area)
    __shared__ int s_item, s_next;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = A.w, h = A.h, R = A.R, K = A.K, NB = A.NB, nsv = A.nsv;
    const int HW = 2 * R + 1, TW = OW + HW;
    const CostConst cc = A.cc;
    const int wb = wave - W_B0;                 // box-wave index
    const f2 ident = {-0.0f, -0.0f};            // exact additive identity: v + (-0) == v
    constexpr unsigned FLAG_DONE = 0x7fffffffu;
    if (tid <= HWMAX * HWMAX) rcp_s[tid] = kRcp.v[tid];
    // the three scan waves are latency chains on the critical path of every step: let them win the
    // issue arbitration against the box waves that share their SIMDs
    if (wave < W_B0) __builtin_amdgcn_s_setprio(3);

    // The ticket of the NEXT item is taken by the I/O wave two band iterations before the end of the current
    // one: no workgroup-wide stall on an atomic at item boundaries.
    if (tid == 0) s_item = (int)__hip_atomic_fetch_add((gu32*)A.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        // LDS-only barriers at the item boundaries too: a full __syncthreads() would make every wave wait for
        // its last q stores (nothing in the next item depends on them; the hand-off data is drained explicitly
        // by the I/O wave before it publishes a flag)
        wg_barrier();
        const int item = s_item;
        if (item >= A.nitems) break;
#ifdef SMX_V3_ITEMLOG
        if (tid == 0 && item < ITEMLOG_MAX) {
            g_itemlog[3 * item] = __builtin_amdgcn_s_memrealtime();
            g_itemlog[3 * item + 2] = blockIdx.x;
        }
#endif
        // FUSED launches hand out the guidance items (view, strip) first: an aggregation item only ever
        // waits for items with smaller tickets (its left neighbour, the guidance item of its strip)
        const bool agg = MODE == AGG || (MODE == FUSED && item >= A.nguid);
        const int nsv_i = agg ? nsv : (MODE == FUSED ? A.nviews : nsv);   // slice-views of this item's kind
        const int item_i = (MODE == FUSED && agg) ? item - A.nguid : item;
        const int k = item_i / nsv_i;
        const int sv = item_i - k * nsv_i;
        const int view = agg ? sv / A.nslices : sv;
        const int slice = agg ? sv - view * A.nslices : 0;
        f2* const hand_i = agg ? A.hand : (MODE == FUSED ? A.ghand : A.hand);
        unsigned* const flags_i = agg ? A.flags : (MODE == FUSED ? A.gflags : A.flags);
        const View& V = A.v[view];
        const int xs = k * OW;
        const int cs1 = xs - R - 1;             // image column of ring-1 column 0
        const int cs2 = xs - HW;                // image column of ring-2 column 0
        const bool pred = k > 0, succ = k + 1 < K;

        // rows of band b in the three lagged row spaces (stage-1 rows, a/b rows, q rows)
        auto rows1 = [&](int b, int& lo, int& hi) { lo = b * BH; hi = min(h, b * BH + BH); if (b < 0 || b >= NB) hi = lo; };
        auto rows2 = [&](int b, int& lo, int& hi) { lo = max(0, b * BH - R); hi = min(h, b * BH + BH - R); if (b < 0) hi = lo; };
        auto rows3 = [&](int b, int& lo, int& hi) { lo = max(0, b * BH - 2 * R); hi = min(h, b * BH + BH - 2 * R); if (b < 0) hi = lo; };

        // column scan of rows [lo, hi) for the ring column `col` of this lane (LANE = COLUMN); runs of
        // consecutive ring rows, so the LDS addresses of a batch are one base + immediates
        auto colscan = [&](f2* ring, int shift, int col, int lo, int hi, f2& S) {
            int rr = (lo + shift) % RR, n = hi - lo;
            while (n > 0) {
                const int run = min(n, RR - rr);
                f2* p = ring + rr * PITCH + col;
                int t0 = 0;
                for (; t0 + 8 <= run; t0 += 8) {      // 8 rows of LDS reads in flight ahead of the adds
                    f2 v[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) v[t] = p[(t0 + t) * PITCH];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        S = v[t] + S;
                        p[(t0 + t) * PITCH] = S;
                    }
                }
                for (; t0 < run; ++t0) {
                    S = p[t0 * PITCH] + S;
                    p[t0 * PITCH] = S;
                }
                n -= run;
                rr = 0;
            }
        };

        // both column groups of a band (ring columns lane and lane + 64).  A full band that does not wrap in
        // the ring (the common case) is straight-line code in batches of 8 rows: the reads of the next batch
        // are issued before the adds and writes of the current one, so the LDS latency is exposed once per
        // column group instead of once per batch (and once per row in the remainder)
        auto colscan_band = [&](f2* ring, int shift, int lo, int hi, f2& S, f2& Sb, bool ccol) {
            const int rr = (lo + shift) % RR;
            if (hi - lo == BH && rr + BH <= RR) {
                constexpr int CB = 8, NCB = (BH + CB - 1) / CB;
                auto group = [&](f2* p, f2& acc) {
                    f2 v[2][CB];
#pragma unroll
                    for (int t = 0; t < CB; ++t) v[0][t] = p[t * PITCH];
#pragma unroll
                    for (int b = 0; b < NCB; ++b) {
                        if (b + 1 < NCB) {
#pragma unroll
                            for (int t = 0; t < CB; ++t)
                                if ((b + 1) * CB + t < BH) v[(b + 1) & 1][t] = p[((b + 1) * CB + t) * PITCH];
                        }
#pragma unroll
                        for (int t = 0; t < CB; ++t)
                            if (b * CB + t < BH) {
                                acc = v[b & 1][t] + acc;
                                p[(b * CB + t) * PITCH] = acc;
                            }
                    }
                };
                f2* const p = ring + rr * PITCH + lane;
                group(p, S);
                if (ccol) group(p + 64, Sb);
                return;
            }
            colscan(ring, shift, lane, lo, hi, S);
            if (ccol) colscan(ring, shift, lane + 64, lo, hi, Sb);
        };

        if (wave == W_IO) {
            // =====================================================================================
            // I/O wave.
            // Global step index t: A(i) = 2i + 4, B(i) = 2i + 5 (i >= -2).
            // Data another strip needs (row carries, halo columns) is produced into LDS at step t by
            // the row-scan / box waves, stored to global (sc1, 16 B per lane) by this wave at the start
            // of step t + 1 and published at step t + 2, after this wave's own s_waitcnt vmcnt(0):
            // flag = t.  Inputs for step t are loaded at step t - 2 (needs the neighbour's flag >= t)
            // and moved to LDS at step t - 1, so neither direction exposes memory latency to the scans.
            // =====================================================================================
            unsigned* const myflag = flags_i + (size_t)sv * K + k;
            unsigned* const gready = MODE == FUSED ? A.gready + (size_t)view * K + k : nullptr;
            bool gdone = false;                  // guidance of this strip complete (FUSED aggregation items)
            bool pred_done = !pred;              // left neighbour finished
            unsigned nx_tk = 0;                  // ticket of the next item
            const size_t recs = (size_t)NB * REC_F2;      // float2 per (parity, slice-view)
            const rsrc_t r_in = mk_rsrc(hand_i + ((size_t)((k - 1) & 1) * nsv_i + sv) * recs, recs * 8);
            const rsrc_t r_out = mk_rsrc(hand_i + ((size_t)(k & 1) * nsv_i + sv) * recs, recs * 8);
            f4 hreg[NHU];
            f4 c1reg = {0, 0, 0, 0}, c2reg = {0, 0, 0, 0};
            int hlo = 0, hhi = 0;        // a/b rows of the halo values in hreg
            // halo unit t = lane + 64 e  ->  row t / (HP/2), column pair t % (HP/2)
            int hu_r[NHU], hu_c[NHU];
#pragma unroll
            for (int e = 0; e < NHU; ++e) {
                const int t = lane + 64 * e;
                hu_r[e] = t / (HP / 2);
                hu_c[e] = (t - hu_r[e] * (HP / 2)) * 2;
            }
            unsigned pseen = 0, gseen = 0;       // last values read from the neighbour's / the guidance flag (both only grow)
            auto wait_pred = [&](int t) {
                const unsigned need = (unsigned)t;
                if (pred_done || pseen >= need) return;      // a neighbour that runs far ahead costs one read per many bands
                const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    const unsigned f = flag_load(myflag - 1);
                    pseen = __builtin_amdgcn_readfirstlane(f);
                    if (f >= need) { pred_done = f == FLAG_DONE; break; }
                    __builtin_amdgcn_s_sleep(4);
                    // bounded spin: give up after 2 s (100 MHz counter) or as soon as any workgroup
                    // has given up; the call then reports SMX_E_HIP through smx_dev_agg_status
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull || flag_load(A.status) != 0u) {
                        flag_store(A.status, 1u + (unsigned)item);
                        pred_done = true;
                        break;
                    }
                }
            };
            for (int i = -2; i <= NB; ++i) {
                int lo, hi;
                // ------------------------------ step A(i), t = 2i + 4 ---------------------------
                // everything this wave issued to global memory was issued at the start of the
                // previous step: the wait is (nearly) free
                drain_vmem();
                if (succ && lane == 0 && i >= -1) flag_store(myflag, (unsigned)(2 * i + 2));
                // guidance item: the box waves stored band i-1 of mean_I / 1/(var+eps) in step A(i-1) (sc1) and
                // drained at the end of step B(i-1), a step later -> publish it
                if (MODE == FUSED && !agg && lane == 0 && i - 1 >= 0 && i - 1 < NB)
                    flag_store(gready, i - 1 == NB - 1 ? FLAG_DONE : (unsigned)i);
                // Next item: its ticket is taken at the END of step A(NB-2), as the last memory instruction of the
                // step, and step B starts with a wait that leaves exactly that one in flight: the latency of the
                // atomic (~1 us under load) passes under two busy steps instead of being waited for by the whole
                // workgroup in a nearly empty drain iteration.  (Built with the atomic optimizer off: it would
                // broadcast -- i.e. wait for -- the result right behind the atomic.)
                if (i == NB - 1) {                         // the ticket is back (drain above): publish it
                    asm volatile("" : "+v"(nx_tk));        // keep the compiler from consuming it (and waiting) earlier
                    if (lane == 0) s_next = (int)nx_tk;
                }
                if (MODE != GUID && agg && pred) {
                    // halo columns + stage-2 carries loaded at B(i-1) -> ring 2 / staging
#pragma unroll
                    for (int e = 0; e < NHU; ++e) {
                        const int r = hu_r[e], c = hu_c[e];
                        if (r < hhi - hlo && c < HW) {
                            int rr = (hlo + R) % RR + r;
                            rr = rr >= RR ? rr - RR : rr;
                            f2* dst = ring2 + rr * PITCH + c;
                            dst[0] = lo2(hreg[e]);
                            if (c + 1 < HW) dst[1] = hi2(hreg[e]);
                        }
                    }
                    if (lane < BH / 2) *(f4*)&cin[1][2 * lane] = c2reg;
                }
                if (MODE != GUID && agg && succ && i - 1 >= 0 && i - 1 < NB) {
                    // stage-2 carries of band i-1 (row scan at B(i-1)) -> global
                    if (lane < BH / 2)
                        st16_sc1(r_out, (unsigned)(((i - 1) * REC_F2 + BH) * 8 + lane * 16), *(const f4*)&cout[1][2 * lane]);
                }
                if (MODE == FUSED && agg && !gdone && i + 1 >= 0 && i + 1 < NB) {
                    // the box waves load mean_I / 1/(var+eps) of the a/b rows of band i+1 in step B(i):
                    // the guidance item of this strip must have stored (and drained) them
                    const unsigned need = (unsigned)(i + 2);
                    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
                    while (gseen < need) {
                        const unsigned f = flag_load(gready);
                        gseen = __builtin_amdgcn_readfirstlane(f);
                        if (f >= need) { gdone = f == FLAG_DONE; break; }
                        __builtin_amdgcn_s_sleep(4);
                        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull || flag_load(A.status) != 0u) {
                            flag_store(A.status, 1u + (unsigned)item);
                            gdone = true;
                            break;
                        }
                    }
                }
                if (pred && i + 2 < NB) {
                    // stage-1 carries for the row scan of band i+2 at A(i+1): load now
                    wait_pred(2 * i + 6);
                    if (lane < BH / 2) c1reg = ld16_sc1(r_in, (unsigned)((i + 2) * REC_F2 * 8 + lane * 16));
                }
                const bool take = i == NB - 2;             // ticket of the next item: last memory instruction of the step
                if (take && lane == 0)
                    nx_tk = __hip_atomic_fetch_add((gu32*)A.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                V3_STAMP(0);
                wg_barrier();
                V3_STAMP(1);
                // ------------------------------ step B(i), t = 2i + 5 ---------------------------
                if (take) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");   // everything but the ticket
                else drain_vmem();
                if (succ && lane == 0 && i >= -1) flag_store(myflag, (unsigned)(2 * i + 3));

                if (pred && lane < BH / 2) *(f4*)&cin[0][2 * lane] = c1reg;
                // stage-1 carries of band i+1 (row scan at A(i)) and halo columns of band i -> global
                if (succ) {
                    if (i + 1 >= 0 && i + 1 < NB && lane < BH / 2)
                        st16_sc1(r_out, (unsigned)((i + 1) * REC_F2 * 8 + lane * 16), *(const f4*)&cout[0][2 * lane]);
                    if (MODE != GUID && agg && i >= 0 && i < NB) {
                        f4 hv[NHU];
#pragma unroll
                        for (int e = 0; e < NHU; ++e) {          // all LDS reads first, then the stores
                            const int t = lane + 64 * e;
                            hv[e] = *(const f4*)&hout[2 * (t < BH * HP / 2 ? t : 0)];
                        }
#pragma unroll
                        for (int e = 0; e < NHU; ++e) {
                            const int t = lane + 64 * e;
                            if (t < BH * HP / 2)
                                st16_sc1(r_out, (unsigned)((i * REC_F2 + 2 * BH) * 8 + t * 16), hv[e]);
                        }
                    }
                }
                // stage-2 carries + halo columns of band i+1 (used at B(i+1) / A(i+1)): load now
                if (MODE != GUID && agg && pred) {
                    rows2(i + 1, lo, hi);
                    hlo = lo; hhi = hi;
                    if (lo < hi && i + 1 < NB) {
                        wait_pred(2 * i + 7);
                        if (lane < BH / 2) c2reg = ld16_sc1(r_in, (unsigned)(((i + 1) * REC_F2 + BH) * 8 + lane * 16));
#pragma unroll
                        for (int e = 0; e < NHU; ++e) {
                            const int t = lane + 64 * e;
                            if (t < BH * HP / 2)
                                hreg[e] = ld16_sc1(r_in, (unsigned)(((i + 1) * REC_F2 + 2 * BH) * 8 + t * 16));
                        }
                    }
                }
                V3_STAMP(2);
                wg_barrier();
                V3_STAMP(3);
            }
            drain_vmem();
            if (succ && lane == 0) flag_store(myflag, FLAG_DONE);
        } else if (wave == W_C) {
            // =====================================================================================
            // column-scan wave: ring columns 0 .. 63, then 64 .. TW-1, of both stages
            // =====================================================================================
            const bool ccol = lane + 64 < TW;             // this lane also scans ring column lane + 64
            f2 S1 = ident, S2 = ident, S1b = ident, S2b = ident;
            for (int i = -2; i <= NB; ++i) {
                int lo, hi;
                if (MODE != GUID && agg) {
                    rows2(i - 1, lo, hi);                  // A(i): stage 2, band i-1
                    if (i - 1 == 0) { S2 = ident; S2b = ident; }
                    colscan_band(ring2, R, lo, hi, S2, S2b, ccol);
                }
                V3_STAMP(0);
                wg_barrier();
                V3_STAMP(1);
                rows1(i + 1, lo, hi);                      // B(i): stage 1, band i+1
                if (i + 1 == 0) { S1 = ident; S1b = ident; }
                colscan_band(ring1, 0, lo, hi, S1, S1b, ccol);
                V3_STAMP(2);
                wg_barrier();
                V3_STAMP(3);
            }
        } else if (wave == W_R) {
            // =====================================================================================
            // row-scan wave: every active lane carries ONE component (p or a / I*p or b) of one row of the band
            // -- the cost of an LDS write grows with
            // the dwords per lane (ds_write_b64 14, ds_write_b128 27 cycles of issue), not with the active
            // lanes, so one dword per lane and column halves the dominant term of the dependent chain.
            // Reads run one batch of 8 columns ahead of the adds.
            // =====================================================================================
            const int jlo1 = max(0, -cs1), jhi1 = min(TW, w - cs1);   // ring-1 columns inside the image
            const int jlo2 = max(0, -cs2), jhi2 = min(TW, w - cs2);
            // lane -> (row of the band, component): each 32-lane group (the unit of LDS banking for dword accesses)
            // takes half of the rows with both components, lanes 0-12 / 16-28 = first / second component.  The
            // 13 first-component lanes of a group then hit 13 distinct even banks and the 13 second-component
            // lanes 13 distinct odd ones (row stride 2 PITCH = 170 dwords = 10 mod 32): conflict-free, where 26
            // rows of one component in a group would be two-way conflicts on the 16 even banks.
            constexpr int HB = BH / 2;
            const int srow = (lane & 15) + HB * (lane >> 5), comp = (lane >> 4) & 1;
            const bool sact = (lane & 15) < HB;
            auto rowscan = [&](f2* ring, int shift, int st, int ylo, int yhi, int jlo, int jhi) {
                if (!sact || srow >= yhi - ylo || jhi <= jlo) return;
                const int y = ylo + srow;
                float acc = pred ? ((const float*)&cin[st][srow])[comp] : -0.0f;
                float* row = (float*)(ring + ((y + shift) % RR) * PITCH) + comp;   // column c of this component: row[2 c]
                if (jlo == 0 && jhi == TWMAX) {
                    // the common case (a strip inside the image at radius 9): 83 columns, fully unrolled so
                    // that every LDS wait is a counted one
                    constexpr int NBATCH = TWMAX / 8;                  // 10 batches of 8 + 3 columns
                    float v[2][8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) v[0][t] = row[2 * t];
#pragma unroll
                    for (int bt = 0; bt < NBATCH; ++bt) {
                        if (bt + 1 < NBATCH) {
#pragma unroll
                            for (int t = 0; t < 8; ++t) v[(bt + 1) & 1][t] = row[2 * (8 * (bt + 1) + t)];
                        }
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            acc = v[bt & 1][t] + acc;
                            row[2 * (8 * bt + t)] = acc;
                            if (8 * bt + t == OW - 1) ((float*)&cout[st][srow])[comp] = acc;
                        }
                    }
#pragma unroll
                    for (int c = 8 * NBATCH; c < TWMAX; ++c) {
                        acc = row[2 * c] + acc;
                        row[2 * c] = acc;
                    }
                    return;
                }
                // general strip: batches of 8 columns, ping-pong; the reads of the next batch are always
                // issued (past the end they fetch bytes nobody uses: LDS reads cannot fault)
                int j = jlo;
                const int nb8 = (jhi - j) >> 3;
                float va[8], vb[8];
                auto rd = [&](float (&v)[8], int c) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) v[t] = row[2 * (c + t)];
                };
                auto run = [&](const float (&v)[8], int c) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        acc = v[t] + acc;
                        row[2 * (c + t)] = acc;
                    }
                };
                if (nb8 > 0) rd(va, j);
                int b = 0;
                for (; b + 2 <= nb8; b += 2) {
                    rd(vb, j + 8);
                    run(va, j);
                    rd(va, j + 16);
                    run(vb, j + 8);
                    j += 16;
                }
                if (b < nb8) {
                    run(va, j);
                    j += 8;
                }
                for (; j < jhi; ++j) {
                    acc = row[2 * j] + acc;
                    row[2 * j] = acc;
                }
                // running row sum left of the next strip's first column
                if (OW - 1 >= jlo && OW - 1 < jhi) ((float*)&cout[st][srow])[comp] = row[2 * (OW - 1)];
            };
            for (int i = -2; i <= NB; ++i) {
                int lo, hi;
                rows1(i + 1, lo, hi);                      // A(i): stage 1, band i+1
                rowscan(ring1, 0, 0, lo, hi, jlo1, jhi1);
                V3_STAMP(0);
                wg_barrier();
                V3_STAMP(1);
                if (MODE != GUID && agg) {                         // B(i): stage 2, band i
                    rows2(i, lo, hi);
                    rowscan(ring2, R, 1, lo, hi, jlo2, jhi2);
                }
                V3_STAMP(2);
                wg_barrier();
                V3_STAMP(3);
            }
        } else {
            // =====================================================================================
            // box / eval waves (LANE = COLUMN); wave wb owns rows 2 wb and 2 wb + 1 of every band
            // =====================================================================================
            const int d = V.d0 + slice;
            const unsigned fgw4 = ((unsigned)w + 2u) * 4u, w4 = (unsigned)w * 4u;
            const unsigned pitch2 = SRC == SRC_IMG ? fgw4 : w4;      // row pitch of the second stage-1 input
            // Every global access of these waves is a buffer instruction: per-lane byte offset fixed for the
            // item + wave-uniform row offset in a scalar register (rows are wave-uniform here), no 64-bit
            // address arithmetic.  A lane whose column lies outside the image carries the offset OOB, which is
            // beyond every plane: its loads return 0 and its stores are dropped by the range check.
            constexpr unsigned OOB = 0x80000000u;
            const size_t plane = (size_t)h * w;
            const rsrc_t r_fg1 = mk_rsrc(V.FG1, (size_t)h * fgw4);
            const rsrc_t r_in2 = !agg ? r_fg1
                               : SRC == SRC_IMG ? mk_rsrc(V.FG2, (size_t)h * fgw4)
                                                : mk_rsrc(V.cost + (size_t)slice * plane, plane * 4);
            const rsrc_t r_ga = mk_rsrc(agg ? V.mean : V.gmean, plane * 4);   // mean_I:       read (agg) / written (guidance)
            const rsrc_t r_gb = mk_rsrc(agg ? V.cinv : V.gcinv, plane * 4);   // 1/(var+eps)
            const rsrc_t r_q = agg ? mk_rsrc(V.q + (size_t)slice * plane, plane * 4)
                                   : mk_rsrc(V.mean_u8, V.mean_u8 ? plane : 0);     // guidance: optional u8 mean image
            constexpr int AUX_G = MODE == FUSED ? AUX_SC1 : 0;   // mean_I / 1/(var+eps) cross workgroups inside a FUSED launch
            constexpr int AUX_NT = 2;

            // per-lane window geometry in x: fixed for the whole item
            struct Geo { int jmax, jmin, xcw; bool hx; };
            unsigned vq, vg, vg8;                     // q / guidance-image column, a_k/b_k column (x4), the latter in bytes x1
            Geo g1, g2;
            {
                auto mkgeo = [&](int x, int cs, unsigned& off4, unsigned& off1) {
                    Geo g;
                    const bool xin = x >= 0 && x < w;
                    const int xc = min(max(x, 0), w - 1);
                    const int xmax = min(w - 1, xc + R), xmn = xc - R - 1;
                    g.hx = xmn >= 0;
                    g.xcw = xmax - (g.hx ? xmn : -1);
                    g.jmax = min(max(xmax - cs, 0), TW - 1);
                    g.jmin = min(max(xmn - cs, 0), TW - 1);
                    off4 = xin ? (unsigned)xc * 4u : OOB;
                    off1 = xin ? (unsigned)xc : OOB;
                    return g;
                };
                unsigned dummy;
                g1 = mkgeo(xs + lane, cs1, vg, vg8);
                g2 = mkgeo(xs - R + lane, cs2, vq, dummy);
            }
            // all 64 windows of the strip unclipped in x: no selects, one area per row
            const bool xint1 = xs - R - 1 >= 0 && xs + OW - 1 + R <= w - 1;
            const bool xint2 = xs - 2 * R - 1 >= 0 && xs + OW - 1 <= w - 1;
            const f2* const p1max = ring1 + g1.jmax;
            const f2* const p1min = ring1 + g1.jmin;
            const f2* const p2max = ring2 + g2.jmax;
            const f2* const p2min = ring2 + g2.jmin;
            // box means of two rows (computeBoxFilterOnGPU guidedFilter.cu:305-318: S11 - S10 - S01 + S00
            // in that order, then a true division by the clipped window area).  Branch-free: all ten
            // LDS reads are issued before the first use; clipped taps are read from a valid dummy
            // address and dropped by a select; the exact-division fix-up is one rare branch at the end.
            auto box2 = [&](const f2* pmax, const f2* pmin, int shift, int xcw, bool hx, bool xint, const int (&yy)[2], f2 (&m)[2]) {
                f2 s11[2], s10[2], s01[2], s00[2], val[2];
                float area[2], ra[2];
                bool hy[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int ymax = min(h - 1, yy[t] + R);
                    const int ymin = yy[t] - R - 1;
                    hy[t] = ymin >= 0;
                    const int ych = ymax - (hy[t] ? ymin : -1);
                    const int o1 = ((ymax + shift) % RR) * PITCH, o0 = (((hy[t] ? ymin : 0) + shift) % RR) * PITCH;
                    s11[t] = pmax[o1]; s10[t] = pmin[o1];
                    s01[t] = pmax[o0]; s00[t] = pmin[o0];
                    const int ai = (xint ? HW : xcw) * ych;
                    area[t] = (float)ai;
                    ra[t] = rcp_s[ai];
                }
                bool slow = false;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f2 v = s11[t];
                    f2 u = v - s10[t];
                    v = (xint || hx) ? u : v;
                    u = v - s01[t];
                    v = hy[t] ? u : v;
                    u = v + s00[t];
                    v = (hy[t] && (xint || hx)) ? u : v;
                    val[t] = v;
                    m[t].x = div_small_int(v.x, area[t], ra[t]);
                    m[t].y = div_small_int(v.y, area[t], ra[t]);
                    slow = slow || div_needs_exact(v.x) || div_needs_exact(v.y);
                }
                if (__any(slow)) {
                    asm volatile("; exact-division slow path");   // keep this a real (rare) wave-uniform branch
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        m[t].x = 1.0f * val[t].x / area[t];
                        m[t].y = 1.0f * val[t].y / area[t];
                    }
                }
            };
            // The same for a band whose rows all exist and are y-interior: the ring rows follow from the band's
            // position in the ring (r1 = ring row of y + R) without any division, the window height is 2R+1.
            // In an x-interior strip (xint) there is no select and one area; in the first / last strip of the
            // image the x-clipping of the general box stays (per-lane width xcw, hx = has a left tap).
            const float area_full = (float)(HW * HW), ra_full = rcp_s[HW * HW];
            // `mid` runs between the issue of the eight LDS reads and their first use: the interior iteration puts
            // its global loads there, so that their issue time hides LDS latency instead of preceding it.
            auto box2_fast = [&](const f2* pmax, const f2* pmin, const int (&r1)[2], bool xint, int xcw, bool hx, f2 (&m)[2], auto mid) {
                f2 s11[2], s10[2], s01[2], s00[2], val[2];
                float area = area_full, ra = ra_full;
                if (!xint) {
                    area = (float)(xcw * HW);
                    ra = rcp_s[xcw * HW];
                }
                const bool left = xint || hx;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    int r0 = r1[t] - HW;
                    r0 = r0 < 0 ? r0 + RR : r0;
                    s11[t] = pmax[r1[t] * PITCH]; s10[t] = pmin[r1[t] * PITCH];
                    s01[t] = pmax[r0 * PITCH];    s00[t] = pmin[r0 * PITCH];
                }
                __builtin_amdgcn_sched_barrier(0);
                mid();
                __builtin_amdgcn_sched_barrier(0);
                bool slow = false;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f2 v = s11[t];
                    f2 u = v - s10[t];
                    v = left ? u : v;
                    v = v - s01[t];
                    u = v + s00[t];
                    v = left ? u : v;
                    val[t] = v;
                    m[t].x = div_small_int(v.x, area, ra);
                    m[t].y = div_small_int(v.y, area, ra);
                    slow = slow || div_needs_exact(v.x) || div_needs_exact(v.y);
                }
                if (__any(slow)) {
                    asm volatile("; exact-division slow path");
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        m[t].x = 1.0f * val[t].x / area;
                        m[t].y = 1.0f * val[t].y / area;
                    }
                }
            };
            // stage-1 input cells of this lane, fixed for the item: rows 2wb, 2wb+1 at ring column `lane`
            // (cells 0, 1) and row 2wb + (lane >> 5) at ring column 64 + (lane & 31) (cell 2).  Columns are
            // clamped into the image instead of predicated: cells outside it are written but never accumulated.
            const int rsel = lane >> 5;      // row of the third cell
            const bool e2_ok = (lane & 31) < HW;
            unsigned in1a, in1b, in2a, in2b;     // byte offsets inside a row: cells 0/1 and cell 2, first / second input
            int ro_a, ro_b;                      // ring offsets (float2) relative to the band's first row
            {
                auto cell = [&](int j, unsigned& o1, unsigned& o2) {
                    const int c = min(max(cs1 + j, 0), w - 1);
                    o1 = (unsigned)(c + 1) * 4u;
                    if (SRC == SRC_IMG) {
                        int xx = c + d;
                        xx = xx < -1 ? -1 : (xx > w ? w : xx);   // sentinel columns
                        o2 = (unsigned)(xx + 1) * 4u;
                    } else {
                        o2 = (unsigned)c * 4u;
                    }
                };
                cell(lane, in1a, in2a);
                cell(64 + (lane & 31), in1b, in2b);
                ro_a = 2 * wb * PITCH + lane;
                ro_b = (2 * wb + rsel) * PITCH + min(64 + (lane & 31), PITCH - 1);
            }
            uint32_t ua[3] = {0, 0, 0}, ub[3] = {0, 0, 0};
            float ga[2] = {0.0f, 0.0f}, gb[2] = {0.0f, 0.0f};
            uint32_t Iraw[2] = {0, 0};     // raw (value, gradient) halves: converted at use, a step after the load

            // The loop is instantiated per item kind (aggregation / guidance) so that the number of memory
            // instructions per step is a compile-time fact in each.
            auto item_loop = [&](auto KIND) __attribute__((always_inline)) {
            constexpr bool AGGK = decltype(KIND)::value;
            int ph1 = 1;                     // i mod 3: position of band i in the three-band rings (i = -2 first)
            for (int i = -2; i <= NB; ++i, ph1 = ph1 == 2 ? 0 : ph1 + 1) {
                int ylo, yhi;
                const int ph2 = ph1 == 0 ? 2 : ph1 - 1;          // (i - 1) mod 3 = (i + 2) mod 3
                if (AGGK && xs < w && (i - 1) * BH >= 3 * R + 1 && (i + 1) * BH <= h) {
                    auto interior = [&](auto XINT) __attribute__((always_inline)) {
                    constexpr bool XI = decltype(XINT)::value;
                    // ---- y-interior iteration of an aggregation item (most iterations of most items): every window
                    // row of both box stages exists and is y-interior, so no row is predicated and the window
                    // height is fixed; in an x-interior strip (XI) no window is clipped in x either.  The rows
                    // loaded for later bands are clamped into the image.  Same arithmetic as the general body
                    // below, straight-line.
                    const int y3 = (i - 1) * BH - 2 * R + 2 * wb;      // q rows
                    const int ye = min((i + 2) * BH + 2 * wb, h - 1);  // stage-1 input rows of band i+2 (clamped)
                    const int ye1 = min((i + 2) * BH + 2 * wb + 1, h - 1);
                    const unsigned dy = (unsigned)(ye1 - ye);          // 1, or 0 on the last image row
                    // step A(i): the loads consumed in step B go out behind the LDS reads of the box
                    auto loadsA = [&]() {
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            Iraw[t] = __builtin_amdgcn_raw_buffer_load_b32(r_fg1, (int)vq, (y3 + t) * (int)fgw4 + 4, 0);
                        ua[0] = __builtin_amdgcn_raw_buffer_load_b32(r_fg1, (int)in1a, ye * (int)fgw4, 0);
                        ua[1] = __builtin_amdgcn_raw_buffer_load_b32(r_fg1, (int)in1a, ye1 * (int)fgw4, 0);
                        ua[2] = __builtin_amdgcn_raw_buffer_load_b32(r_fg1, (int)(in1b + (rsel ? dy * fgw4 : 0u)), ye * (int)fgw4, 0);
                        ub[0] = __builtin_amdgcn_raw_buffer_load_b32(r_in2, (int)in2a, ye * (int)pitch2, 0);
                        ub[1] = __builtin_amdgcn_raw_buffer_load_b32(r_in2, (int)in2a, ye1 * (int)pitch2, 0);
                        ub[2] = __builtin_amdgcn_raw_buffer_load_b32(r_in2, (int)(in2b + (rsel ? dy * pitch2 : 0u)), ye * (int)pitch2, 0);
                    };
                    {
                        f2 m[2];
                        int r1[2];
#pragma unroll
                        for (int t = 0; t < 2; ++t) r1[t] = ph1 * BH + 2 * wb + t;
                        box2_fast(p1max, p1min, r1, XI, g1.xcw, g1.hx, m, loadsA);
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const int ry = r1[t];          // ring 2 keeps a/b row y at ring row (y + R) mod RR
                            float mm = ga[t] * m[t].x;     // compute_ak_and_bk guidedFilter.cu:345-354
                            float ak = 1.0f * (m[t].y - mm) * gb[t];
                            float mb2 = 1.0f * ga[t] * ak;
                            float bk = 1.0f * m[t].x - mb2;
                            f2 ab = {ak, bk};
                            ring2[ry * PITCH + HW + lane] = ab;
                            if (lane >= OW - HW) hout[(2 * wb + t) * HP + lane - (OW - HW)] = ab;
                        }
                    }
                    V3_STAMP(0);
                    wg_barrier();
                    V3_STAMP(1);
                    // step B(i)
                    const int y2 = (i + 1) * BH - R + 2 * wb;          // a/b rows of band i+1 (clamped: unused if missing)
                    // behind the LDS reads of the stage-2 box: the guidance loads for the next step, and the cost of
                    // band i+2 (inputs loaded in step A) -> ring 1
                    auto loadsB = [&]() {
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            ga[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_ga, (int)vg, min(y2 + t, h - 1) * (int)w4, AUX_G));
                            gb[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_gb, (int)vg, min(y2 + t, h - 1) * (int)w4, AUX_G));
                        }
                        if ((i + 2) * BH < h) {                          // band i+2 has rows
                            f2* rb = ring1 + ph2 * BH * PITCH;
#pragma unroll
                            for (int e = 0; e < 3; ++e) {
                                const fg_t q1 = __builtin_bit_cast(fg_t, ua[e]);
                                f2 v;
                                if (SRC == SRC_IMG) {
                                    v = cost_pair(q1, __builtin_bit_cast(fg_t, ub[e]), cc);
                                } else {
                                    v.x = __builtin_bit_cast(float, ub[e]);
                                    v.y = (float)q1.x * v.x;
                                }
                                if (e == 0) rb[ro_a] = v;
                                else if (e == 1) rb[ro_a + PITCH] = v;
                                else if (e2_ok) rb[ro_b] = v;
                            }
                        }
                    };
                    {
                        f2 m[2];
                        int r1[2];
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            r1[t] = ph2 * BH + 2 * wb + t;       // ring-2 row of y + R: (y + 2R) mod RR
                        }
                        box2_fast(p2max, p2min, r1, XI, g2.xcw, g2.hx, m, loadsB);
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const float Iv = (float)__builtin_bit_cast(fg_t, Iraw[t]).x;
                            float tq = m[t].x * Iv;        // compute_q guidedFilter.cu:363-369
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, tq + m[t].y), r_q, (int)vq,
                                                                  (y3 + t) * (int)w4, AUX_NT);
                        }
                    }
                    V3_STAMP(2);
                    wg_barrier();
                    V3_STAMP(3);
                    };
                    if (xint1 && xint2) interior(BoolC<true>{});
                    else interior(BoolC<false>{});
                    continue;
                }
                // ------------------------------ step A(i) -----------------------------------------
                // loads consumed in step B(i): guidance image of the q rows, stage-1 inputs of band i+2
                const int be = i + 2;
                const bool ev = be < NB && be * BH < h;
                int y3lo = 0, y3hi = 0;
                if (AGGK) rows3(i - 1, y3lo, y3hi);
                const int y3 = y3lo + 2 * wb;                    // q rows of this iteration: y3, y3 + 1
                // Every global access of the loop is issued in every iteration, whether its row exists or not (rows
                // clamped into the image for loads, the out-of-range lane offset for stores): with a fixed number
                // of memory instructions per step the waits for the loads of the previous step stay counted ones
                // instead of degrading to "everything, including the q stores just issued".
                if (AGGK) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)    // column + 1 in the padded plane
                        Iraw[t] = __builtin_amdgcn_raw_buffer_load_b32(r_fg1, (int)vq, min(y3 + t, h - 1) * (int)fgw4 + 4, 0);
                }
                {                              // rows clamped into the image (wave-uniform for cells 0, 1)
                    const int ye0 = min(max(be * BH + 2 * wb, 0), h - 1), ye1 = min(max(be * BH + 2 * wb + 1, 0), h - 1);
                    const unsigned dy = (unsigned)(ye1 - ye0);   // 1, or 0 on the last image row
                    ua[0] = __builtin_amdgcn_raw_buffer_load_b32(r_fg1, (int)in1a, ye0 * (int)fgw4, 0);
                    ua[1] = __builtin_amdgcn_raw_buffer_load_b32(r_fg1, (int)in1a, ye1 * (int)fgw4, 0);
                    ua[2] = __builtin_amdgcn_raw_buffer_load_b32(r_fg1, (int)(in1b + (rsel ? dy * fgw4 : 0u)), ye0 * (int)fgw4, 0);
                    if (AGGK) {
                        ub[0] = __builtin_amdgcn_raw_buffer_load_b32(r_in2, (int)in2a, ye0 * (int)pitch2, 0);
                        ub[1] = __builtin_amdgcn_raw_buffer_load_b32(r_in2, (int)in2a, ye1 * (int)pitch2, 0);
                        ub[2] = __builtin_amdgcn_raw_buffer_load_b32(r_in2, (int)(in2b + (rsel ? dy * pitch2 : 0u)), ye0 * (int)pitch2, 0);
                    }
                }
                // box means of stage 1, band i (guidance statistics loaded in step B(i-1))
                rows2(i, ylo, yhi);
                // all BH rows of the band exist and are y-interior
                const bool fast1 = xs < w && i * BH >= HW && (i + 1) * BH <= h;
                if (fast1 || (ylo + 2 * wb < yhi && xs < w)) {
                    // both rows in one straight-line block (the second one clamped onto the first when
                    // it does not exist, its stores predicated): their LDS reads overlap
                    f2 m[2];
                    int yy[2], ry2[2];       // image row, ring-2 row of the a/b values
                    bool ok[2];
                    if (fast1) {
                        int r1[2];
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            ok[t] = true;
                            yy[t] = ylo + 2 * wb + t;
                            r1[t] = ph1 * BH + 2 * wb + t;           // ring row of y + R = i BH + 2 wb + t
                            ry2[t] = r1[t];
                        }
                        box2_fast(p1max, p1min, r1, xint1, g1.xcw, g1.hx, m, [] {});
                    } else {
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            ok[t] = ylo + 2 * wb + t < yhi;
                            yy[t] = ok[t] ? ylo + 2 * wb + t : ylo + 2 * wb;
                            ry2[t] = (yy[t] + R) % RR;
                        }
                        box2(p1max, p1min, 0, g1.xcw, g1.hx, xint1, yy, m);
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        if (!AGGK) {
                            float mm = m[t].x * m[t].x;    // pixelMultOnGPU(mean, mean) guidedFilter.cu:112
                            float var = m[t].y - mm;       // pixelSousOnGPU :121
                            float c = (float)(1.0f / ((double)var + A.eps));   // :350
                            if (ok[t]) {                   // lanes outside the image: dropped by the range check
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m[t].x), r_ga, (int)vg, yy[t] * (int)w4, AUX_G);
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, c), r_gb, (int)vg, yy[t] * (int)w4, AUX_G);
                                int ci8 = (int)m[t].x;     // flToChOnGPU :451-458 (no-op without a u8 plane: 0 records)
                                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)((ci8 > 255) ? 255 : ci8), r_q, (int)vg8, yy[t] * w, 0);
                            }
                        } else {
                            float mm = ga[t] * m[t].x;     // compute_ak_and_bk guidedFilter.cu:345-354
                            float ak = 1.0f * (m[t].y - mm) * gb[t];
                            float mb2 = 1.0f * ga[t] * ak;
                            float bk = 1.0f * m[t].x - mb2;
                            f2 ab = {ak, bk};
                            if (ok[t]) {
                                ring2[ry2[t] * PITCH + HW + lane] = ab;
                                if (lane >= OW - HW) hout[(yy[t] - ylo) * HP + lane - (OW - HW)] = ab;
                            }
                        }
                    }
                }
                V3_STAMP(0);
                wg_barrier();
                V3_STAMP(1);
                // ------------------------------ step B(i) -----------------------------------------
                // loads consumed in step A(i+1): guidance statistics of the a/b rows of band i+1
                if (AGGK) {
                    rows2(i + 1, ylo, yhi);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int y = min(ylo + 2 * wb + t, h - 1);    // clamped: unused if the row does not exist
                        ga[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_ga, (int)vg, y * (int)w4, AUX_G));
                        gb[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_gb, (int)vg, y * (int)w4, AUX_G));
                    }
                    // box means of stage 2 -> q of band i-1
                    const bool fast2 = (i - 1) * BH - 2 * R >= R + 1 && (i - 1) * BH + BH - R <= h;
                    f2 m[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
                    int yy[2] = {y3, y3};
                    bool ok[2] = {false, false};
                    if (fast2 || y3 < y3hi) {
                        if (fast2) {
                            int r1[2];
#pragma unroll
                            for (int t = 0; t < 2; ++t) {
                                ok[t] = true;
                                yy[t] = y3 + t;
                                r1[t] = ph2 * BH + 2 * wb + t;    // ring-2 row of y + R: (y + 2R) mod RR = (i-1) BH + 2 wb + t
                            }
                            box2_fast(p2max, p2min, r1, xint2, g2.xcw, g2.hx, m, [] {});
                        } else {
#pragma unroll
                            for (int t = 0; t < 2; ++t) {
                                ok[t] = y3 + t < y3hi;
                                yy[t] = ok[t] ? y3 + t : y3;
                            }
                            box2(p2max, p2min, R, g2.xcw, g2.hx, xint2, yy, m);
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) {      // always issued: a row that does not exist is dropped by the range check
                        const float Iv = (float)__builtin_bit_cast(fg_t, Iraw[t]).x;
                        float tq = m[t].x * Iv;            // compute_q guidedFilter.cu:363-369
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, tq + m[t].y), r_q, (int)(ok[t] ? vq : OOB),
                                                              min(yy[t], h - 1) * (int)w4, AUX_NT);
                    }
                }
                if (ev) {                                  // stage-1 inputs of band i+2 -> ring 1
                    f2* rb = ring1 + ph2 * BH * PITCH;
#pragma unroll
                    for (int e = 0; e < 3; ++e) {
                        const fg_t q1 = __builtin_bit_cast(fg_t, ua[e]);
                        f2 v;
                        if (!AGGK) {
                            v.x = (float)q1.x;            // chToFlOnGPU guidedFilter.cu:442-449
                            v.y = v.x * v.x;              // pixelMultOnGPU(d_im, d_im) :111
                        } else if (SRC == SRC_IMG) {
                            v = cost_pair(q1, __builtin_bit_cast(fg_t, ub[e]), cc);
                        } else {
                            v.x = __builtin_bit_cast(float, ub[e]);   // copyFromBigToLittleOnGPU :198
                            v.y = (float)q1.x * v.x;                  // pixelMultOnGPU(d_im, d_p) :209
                        }
                        if (e == 0) rb[ro_a] = v;
                        else if (e == 1) rb[ro_a + PITCH] = v;
                        else if (e2_ok) rb[ro_b] = v;
                    }
                }
                // guidance item of a FUSED launch: every storing wave drains its (sc1) stores of step A, a
                // step old by now, before the barrier behind which the I/O wave publishes the band
                if (MODE == FUSED && !AGGK) drain_vmem();
                V3_STAMP(2);
                wg_barrier();
                V3_STAMP(3);
            }
            };
            if (MODE == AGG || (MODE == FUSED && agg)) item_loop(BoolC<true>{});
            else item_loop(BoolC<false>{});
        }
        wg_barrier();
#ifdef SMX_V3_ITEMLOG
        if (tid == 0 && item < ITEMLOG_MAX) g_itemlog[3 * item + 1] = __builtin_amdgcn_s_memrealtime();
#endif
        if (tid == 0) s_item = s_next;
    }
}

// ---------------------------------------------------------------------------------------------
// WTA over the chunk's q planes [slice][h][w].  One lane per pixel.  grid (ceil(n/256), nviews)
// (dispSelectOnGPU guidedFilter.cu:403-411 in packed-key form)
// ---------------------------------------------------------------------------------------------
struct WtaArgs {
    const float* q[2];
    int64_t* keys[2];
};

__global__ __launch_bounds__(256) void k_v3_wta(WtaArgs wa, size_t n, int count, int slice0) {
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= n) return;
    const float* __restrict__ q = wa.q[blockIdx.y] + id;
    int64_t* keys = wa.keys[blockIdx.y];
    int64_t key = keys[id];
    int z = 0;
    for (; z + 8 <= count; z += 8) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = __builtin_nontemporal_load(&q[(size_t)(z + t) * n]);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            int64_t kk = pack_key(v[t], (uint32_t)(slice0 + z + t));
            key = kk < key ? kk : key;
        }
    }
    for (; z < count; ++z) {
        int64_t kk = pack_key(__builtin_nontemporal_load(&q[(size_t)z * n]), (uint32_t)(slice0 + z));
        key = kk < key ? kk : key;
    }
    keys[id] = key;
}

// The same with two pixels per lane (8-byte loads): half the load instructions for the same bytes in
// flight.  Needs an even plane size n (then every plane start is 8-byte aligned).  grid (ceil(n/512), nviews)
__global__ __launch_bounds__(256) void k_v3_wta2(WtaArgs wa, size_t n, int count, int slice0) {
    const size_t id = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (id >= n) return;
    const float* __restrict__ q = wa.q[blockIdx.y] + id;
    int64_t* keys = wa.keys[blockIdx.y];
    int64_t k0 = keys[id], k1 = keys[id + 1];
    int z = 0;
    for (; z + 8 <= count; z += 8) {
        f2 v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = __builtin_nontemporal_load((const f2*)&q[(size_t)(z + t) * n]);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int64_t a = pack_key(v[t].x, (uint32_t)(slice0 + z + t)), b = pack_key(v[t].y, (uint32_t)(slice0 + z + t));
            k0 = a < k0 ? a : k0;
            k1 = b < k1 ? b : k1;
        }
    }
    for (; z < count; ++z) {
        const f2 v = __builtin_nontemporal_load((const f2*)&q[(size_t)z * n]);
        const int64_t a = pack_key(v.x, (uint32_t)(slice0 + z)), b = pack_key(v.y, (uint32_t)(slice0 + z));
        k0 = a < k0 ? a : k0;
        k1 = b < k1 ? b : k1;
    }
    keys[id] = k0;
    keys[id + 1] = k1;
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
    size_t sv_hand;   // floats of hand-off records per slice-view (2 parities x NB bands)
};

static V3Layout v3_layout(int w, int h, int R) {
    V3Layout L;
    L.K = (w + R + v3::OW - 1) / v3::OW;
    L.NB = (h + 2 * R + v3::BH - 1) / v3::BH;
    L.fg = (size_t)(w + 2) * h;
    L.plane = (size_t)w * h;
    L.sv_hand = (size_t)2 * L.NB * v3::REC_F2 * 2;   // parity x bands x record x float2
    return L;
}

void v3_geometry(int* ow, int* bh) { *ow = v3::OW; *bh = v3::BH; }

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
    b += align_up(L.sv_hand * 4, 256);                              // guidance hand-off records
    b += v3_flag_bytes(L, 4);                                       // guidance control block (shared)
    b += (size_t)nslices * align_up(L.plane * 4, 256);              // q
    b += align_up((size_t)nslices * L.sv_hand * 4, 256);
    b += v3_flag_bytes(L, 2 * nslices);                             // control block (shared by both views)
    return b + 16 * 256;
}

template <int MODE, int SRC>
static int launch_walk3(const v3::Args& a, hipStream_t st) {
    int dev = 0, ncu = 256;
    SMX_HIP(hipGetDevice(&dev));
    SMX_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    const int slots = SMX_V3_WGPCU * ncu;
    const int grid = a.nitems < slots ? a.nitems : slots;   // persistent: SMX_V3_WGPCU workgroup(s) per CU
    hipLaunchKernelGGL((v3::k_v3_walk<MODE, SRC>), dim3((unsigned)grid), dim3(v3::NT), 0, st, a);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

// Aggregation + WTA of slices [s_begin, s_end) of `nviews` (1 or 2) views.  View v uses d_guide[v]
// as guidance; its cost slices are d_cost[v] (materialised, slice s at (s - s_begin)*w*h) or, when
// d_cost[v] == NULL, are built on the fly against d_guide[v ^ 1] (nviews == 2) / d_other[0].
#ifdef SMX_V3_ITEMLOG
extern "C" __attribute__((visibility("default"))) int smx_debug_read_itemlog(unsigned long long* out, int n) {
    const int m = 3 * v3::ITEMLOG_MAX;
    SMX_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(v3::g_itemlog), sizeof(unsigned long long) * (n < m ? n : m)));
    return SMX_OK;
}
#endif
#ifdef SMX_V3_STAMPS
extern "C" __attribute__((visibility("default"))) int smx_debug_read_stamps(unsigned long long* out, int n) {
    const int m = v3::NWAVE * v3::STAMP_SLOTS;
    SMX_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(v3::g_stamps), sizeof(unsigned long long) * (n < m ? n : m)));
    return m;
}
#endif

// status word of the last fused aggregation that used this workspace (0 = ok)
int v3_read_status(const void* d_ws, unsigned* out) {
    const char* base = (const char*)align_up((size_t)d_ws, 256);
    SMX_HIP(hipMemcpy(out, base, sizeof(unsigned), hipMemcpyDeviceToHost));
    return SMX_OK;
}

int aggregate_v3(const smx_params* p, int nviews, const uint8_t* const* d_guide,
                 const uint8_t* const* d_other, const float* const* d_cost, int w, int h,
                 const int* dmin, int s_begin, int s_end, int64_t* const* d_keys,
                 uint8_t* const* d_mean_u8, float* const* d_agg, void* d_ws, size_t ws_bytes,
                 hipStream_t st, int* launches) {
    const int R = p->radius;
    const V3Layout L = v3_layout(w, h, R);
    const bool use_cost = d_cost && d_cost[0];
    // the box waves address every plane through 32-bit buffer offsets, with 0x80000000 as "outside the image"
    if ((size_t)h * ((size_t)w + 2) * 4 >= 0x80000000ull)
        return fail(SMX_E_ARG, "aggregate_v3: an image plane of %d x %d exceeds 2 GiB", w, h);
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
    // right behind it the control block of the guidance items (ticket + hand-off flags + ready counters):
    // one memset clears both
    char* gctrl = (char*)carve(v3_flag_bytes(L, 2 * nviews));
    if (oom) return fail(SMX_E_WS, "aggregate_v3: workspace too small");
    SMX_HIP(hipMemsetAsync(status, 0, 256 + v3_flag_bytes(L, 2 * nviews), st));
    // fixed part: image planes, guidance statistics, guidance scratch + control
    v3::fg_t* FG[2];
    float *meanI[2], *cinv[2];
    for (int i = 0; i < 2; ++i) FG[i] = (v3::fg_t*)carve(L.fg * 4);
    for (int v = 0; v < nviews; ++v) { meanI[v] = (float*)carve(L.plane * 4); cinv[v] = (float*)carve(L.plane * 4); }
    v3::f2* ghand = (v3::f2*)carve((size_t)nviews * L.sv_hand * 4);
    const int total = s_end - s_begin;
    // per slice-view: q plane (unless the caller's volume is written directly) + scratch + flags
    const bool own_q = !(d_agg && d_agg[0]);
    const size_t per_sv = (own_q ? align_up(L.plane * 4, 256) : 0) + L.sv_hand * 4 +
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
    v3::f2* hand = (v3::f2*)carve((size_t)nsv_max * L.sv_hand * 4);
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
    // ---- guidance statistics: (view, strip) items of the single-stage walker.  They ride in the first
    //      aggregation launch (FUSED: handed out first, the aggregation items of a strip follow its
    //      guidance item band by band), or run alone when this call has no slices to aggregate.
    v3::Args g = a0;
    for (int v = 0; v < nviews; ++v) {
        g.v[v].gmean = meanI[v]; g.v[v].gcinv = cinv[v];
        g.v[v].mean_u8 = d_mean_u8 ? d_mean_u8[v] : nullptr;
    }
    unsigned* const gflags = (unsigned*)(gctrl + V3_CTRL_BYTES);
    unsigned* const gready = gflags + (size_t)nviews * L.K;
    if (total <= 0) {
        g.nslices = 1; g.nsv = nviews; g.nitems = nviews * L.K; g.nviews = nviews;
        g.hand = ghand;
        g.ticket = (unsigned*)gctrl; g.status = status;
        g.flags = gflags;
        if ((rc = launch_walk3<v3::GUID, v3::SRC_IMG>(g, st))) return rc;
        ++nl;
    }
    bool first = true;
    for (int s0 = s_begin; s0 < s_end; s0 += chunk) {
        const int cnt = (s_end - s0) < chunk ? (s_end - s0) : chunk;
        v3::Args a = first ? g : a0;      // the first launch also carries the guidance outputs
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
        a.nslices = cnt; a.nsv = cnt * nviews; a.nviews = nviews;
        a.nguid = first ? nviews * L.K : 0;
        a.nitems = a.nguid + a.nsv * L.K;
        a.hand = hand; a.ghand = ghand;
        a.ticket = (unsigned*)ctrl; a.status = status;
        a.flags = (unsigned*)(ctrl + V3_CTRL_BYTES);
        a.gflags = gflags; a.gready = gready;
        SMX_HIP(hipMemsetAsync(ctrl, 0, v3_flag_bytes(L, a.nsv), st));
        if (first) rc = use_cost ? launch_walk3<v3::FUSED, v3::SRC_COST>(a, st) : launch_walk3<v3::FUSED, v3::SRC_IMG>(a, st);
        else rc = use_cost ? launch_walk3<v3::AGG, v3::SRC_COST>(a, st) : launch_walk3<v3::AGG, v3::SRC_IMG>(a, st);
        if (rc) return rc;
        first = false;
        if (L.plane % 2 == 0)
            hipLaunchKernelGGL(v3::k_v3_wta2, dim3(cdivu3((int64_t)L.plane, 512), nviews), dim3(256), 0, st, wa,
                               L.plane, cnt, s0);
        else
            hipLaunchKernelGGL(v3::k_v3_wta, dim3(cdivu3((int64_t)L.plane, 256), nviews), dim3(256), 0, st, wa,
                               L.plane, cnt, s0);
        SMX_HIP(hipGetLastError());
        nl += 3;
    }
    if (launches) *launches = nl;
    return SMX_OK;
}

}  // namespace smx
