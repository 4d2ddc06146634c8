// smx_agg_v4.hip -- fused guided-filter aggregation for gfx950, throughput form: cost build ->
// integral (p, I*p) -> box -> a_k, b_k -> integral (a, b) -> box -> q in ONE kernel, with a_k, b_k never
// leaving the CU.  Reference: guidedFilter.cu:171-238, costVolume.cu:163-190, integral.cu:78-131.
//
// Work item = (slice-view sv, strip k): a column strip of OW = 64 output columns, walked top -> bottom in
// bands of BH = 16 rows by one 512-thread workgroup.  THREE workgroups share a CU (49.5 KB of LDS, <= 80 VGPRs
// each): while one of them sits in a latency-bound phase (the sequential scans) the others fill the SIMDs, which
// is what the one-workgroup-per-CU predecessor (smx_agg_v3.hip) could not do.  Two LDS rings of RR = 36 rows x 83
// columns x 2 components hold the integral images of stage 1 (p, I*p) and stage 2 (a, b): a band of box means
// needs BH + 2R + 1 rows.  (BH = 32 with two workgroups per CU also builds: -DSMX_V4_BH=32, 3 % slower.)
//
// Exactness: every prefix sum keeps the reference's order (sequential left -> right in a row, then
// sequential top -> bottom in a column; integral.cu:82-86, 124-128), the box mean its tap order
// (guidedFilter.cu:305-318).  The row scans of a strip start from the running row sums of the strip to its
// left, the column scans keep their running sums in registers down the strip.  Stage 1 recomputes the 2R+1
// columns it shares with its left neighbour; stage 2 lags stage 1 by R rows and R columns and receives the
// 2R+1 finished integral columns it shares with its left neighbour from that neighbour (hand-off record).
//
// Iteration i of an item (four workgroup barriers, each ordering LDS only):
//   W(i)  waves 2..7: (p, I p) of band i (loaded in W(i-1), evaluated in R(i-1)) -> ring 1; all waves: a_k, b_k
//         of band i-1 (computed in X(i-1)) -> ring 2; loads of the stage-1 inputs of band i+1
//   R(i)  wave 0: row scans of stage 1 (band i) and stage 2 (band i-1), lane = (stage, row, component), four
//         columns per LDS instruction | waves 2..7: evaluate band i+1's costs, the left neighbour's stage-2 halo
//         columns (record i, loaded in X(i-1)) -> ring 2, drain of the record stores of X(i-1)
//   C(i)  waves 0..2: column scan of stage 1, band i | waves 3, 4: column scan of stage 2, band i-1
//         (lane = one dword of a ring row, running sums in registers down the strip); one lane publishes record i-1
//   X(i)  all waves, LANE = COLUMN, 2 rows each: box means of stage 1 -> a_k, b_k of band i (lagged by R,
//         registers); box means of stage 2 -> q rows of band i-1 (lagged by 2R) -> HBM; record i -> global;
//         load of the neighbour's record i+1; guidance loads for X(i+1)
// Every global access is a row-major run issued as a buffer instruction (per-lane byte offset kept in a VGPR for the
// whole item + scalar row offset).
//
// Items are handed out by a ticket counter in strip-major order, so the left neighbour of an item always
// holds an earlier ticket (it is running or done: no deadlock whatever the dispatch order).  The hand-off is
// the sc1 form of the guide (write-through 16-byte stores, every storing wave drained before the workgroup
// barrier behind which ONE lane publishes a band counter; sc1 loads behind a relaxed poll of that counter
// and a workgroup barrier; no acquire fence).
//
// Must be compiled with -ffp-contract=off.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "smx_agg_dev.h"
#include "smx_agg_v5.h"
#include "smx_launch.h"

namespace smx {
namespace v4 {
using namespace aggdev;

constexpr int OW = 64;                  // output columns per strip = one wave
constexpr int RMAX = 9;                 // largest supported box radius
constexpr int HWMAX = 2 * RMAX + 1;     // halo / overlap columns
constexpr int TWMAX = OW + HWMAX;       // ring columns in use (83 at R = 9)
// A ring row is component-planar: ROWF floats = the first components (p / a) of its columns at [0, OFF1), the
// second ones (I p / b) at [OFF1, OFF1 + NCOLP).  The row scan then moves four columns of one component per LDS
// instruction (16-byte aligned: ROWF, OFF1 are multiples of 4), the LANE = COLUMN phases read a cell's pair with
// one ds_read2_b32.  Row stride 172 dwords = 12 mod 32: the 16-byte stores of 8 consecutive rows hit distinct banks.
constexpr int NCOLP = 84;               // columns per component plane of a row (TWMAX rounded up to quads)
constexpr int OFF1 = 88;
constexpr int ROWF = OFF1 + NCOLP;      // 172
static_assert(NCOLP >= TWMAX && NCOLP % 4 == 0 && OFF1 % 4 == 0 && OFF1 >= NCOLP, "planar ring row");
constexpr int NT = 512;
constexpr int NWAVE = NT / 64;
// Band height: 32 rows (rings of 52 rows, 72 KB of LDS, two workgroups per CU) or 16 rows (rings of 36 rows,
// 50 KB, three workgroups per CU, fewer idle rows at the bottom of a strip)
#ifndef SMX_V4_BH
#define SMX_V4_BH 16
#endif
constexpr int BH = SMX_V4_BH;           // band height
constexpr int RPW = BH / NWAVE;         // rows of a band per wave in the LANE = COLUMN phases
constexpr int RR = BH + 2 * RMAX + 2;   // ring rows: a band of box means needs BH + 2R + 1 rows
constexpr int WG_PER_CU = BH == 32 ? 2 : 3;
static_assert(BH == 32 || BH == 16, "band height");
static_assert(RR % 4 == 0 && RR % RPW == 0 && BH % RPW == 0 && RPW % 2 == 0,
              "groups of four (column scan) and of RPW (box, cost) consecutive ring rows never wrap");

enum Src { SRC_IMG = 0, SRC_COST = 1 };


struct View {
    const fg_t* FG1;      // this view's image plane [h][w + 2 PADX], sentinel columns on either side
    const fg_t* FG2;      // the other view's (SRC_IMG)
    const float* cost;    // SRC_COST: [slice][h][w]
    const f2* guid;       // (mean_I, 1/(var_I + eps)) [h][w]
    float* q;             // out: [slice][h][w]
    int d0;               // disparity of local slice 0
};

struct Args {
    View v[2];
    int w, h, R, K, NI, nslices, nsv, nitems;
    f2* hand;             // hand-off records [parity][sv][iteration] (see REC_F2)
    unsigned* flags;      // [sv][K]  published-record counters (zeroed before every launch)
    unsigned* ticket;     // work-item counter              (zeroed before every launch)
    unsigned* status;     // != 0: a flag wait timed out (results invalid)
    const unsigned* only_if;   // != NULL: the launch does nothing unless this word is nonzero (the queued fall-back behind the comb walker)
    CostConst cc;
};

// ---------------------------------------------------------------------------------------------
// prep: u8 image [h][w] -> (value, x-derivative) half2 plane [h][w + 2 PADX] with PADX sentinel columns on
// either side (x_derivativeOnGPU costVolume.cu:358-381).  Done by the workgroups of k_v4_guid_rows for their image rows
// (round 5: a 5 us launch of its own before).
// ---------------------------------------------------------------------------------------------
struct PrepArgs {
    const uint8_t* I[2];
    fg_t* FG[2];
    unsigned* zero[2];      // two word ranges this launch clears (status word, control block of the first chunk):
    unsigned nzero[2];      // saves two fill launches per call
};

// ---------------------------------------------------------------------------------------------
// guidance statistics (guidedFilter.cu:58-123): integral images of I and I*I (integral.cu:78-131, the
// reference's sequential row prefix then sequential column prefix), box means, variance, 1/(var + eps).
// Three small latency-bound kernels; the planes live in the workspace.
// ---------------------------------------------------------------------------------------------
struct GuidArgs {
    const fg_t* FG[2];      // image planes of k_v4_prep
    float* S[2][2];         // [view][plane]: integral of I, integral of I*I (scratch)
    f2* G[2];               // out: (mean_I, 1/(var_I + eps))
    uint8_t* mean_u8[2];    // out, optional: mean_I as u8 (flToChOnGPU guidedFilter.cu:451-458)
    // k_v4_guid_rows also does k_v4_prep's work (one launch less): the image planes of `nimg` images, the cleared words
    PrepArgs prep;
    int nimg, nviews;
};

// Row prefix sums (rowSum integral.cu:78-90).  One workgroup per `rows` image rows of one view, whole rows in
// LDS: (1) all loads, (2) lanes (row, plane) of wave 0 run the sequential prefix 64 columns at a time through
// registers, (3) all stores.  No wait for a load ever has a store in front of it (a wave's accesses complete in
// order: a scan that loads and stores tile by tile waits for the acknowledgement of its stores in every tile).
// Dynamic LDS: 2 planes x rows x wpad floats, wpad = roundup(w, 128) + 4.
constexpr int GR_NT = 256;
constexpr int GR_MAXROWS = 8;
__host__ __device__ inline int gr_wpad(int w) { return ((w + 127) & ~127) + 4; }
#ifndef SMX_GR_WHATIF
#define SMX_GR_WHATIF 0     // diagnostic builds (WRONG results): 1 no scan, 2 no S stores, 4 no image planes, 8 no pixel loads
#endif
__global__ __launch_bounds__(GR_NT) void k_v4_guid_rows(GuidArgs ga, int w, int h, int rows) {
    extern __shared__ __attribute__((aligned(16))) float gr_lds[];
    const int tid = threadIdx.x, view = blockIdx.y, y0 = blockIdx.x * rows;
    const int wp = w + 2 * PADX;
    // ---- k_v4_prep's part I: the words to clear
    {
        const unsigned gid = (blockIdx.y * gridDim.x + blockIdx.x) * GR_NT + tid, gsz = gridDim.y * gridDim.x * GR_NT;
#pragma unroll
        for (int r = 0; r < 2; ++r)
            for (unsigned i = gid; i < ga.prep.nzero[r]; i += gsz) ga.prep.zero[r][i] = 0u;
    }
    const uint8_t* __restrict__ Iu = ga.prep.I[view];
    float* __restrict__ S0 = ga.S[view][0];
    float* __restrict__ S1 = ga.S[view][1];
    const int wpad = gr_wpad(w), wr = wpad - 4;               // wr: a multiple of 128
    float* P0 = gr_lds;
    float* P1 = gr_lds + rows * wpad;
    // (1) the rows of this workgroup, 16 loads per thread in flight
    const int total = rows * wr;
    for (int e0 = 0; e0 < total; e0 += 16 * GR_NT) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int e = min(e0 + k * GR_NT + tid, total - 1);
            const int r = e / wr, x = e - r * wr;
            if (SMX_GR_WHATIF & 8) v[k] = (float)(e & 255); else
            v[k] = 1.0f * (float)(int)Iu[(size_t)min(y0 + r, h - 1) * w + min(x, w - 1)];     // chToFlOnGPU guidedFilter.cu:442-449
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int e = e0 + k * GR_NT + tid;
            if (e < total) {
                const int r = e / wr, x = e - r * wr;
                P0[r * wpad + x] = v[k];
                P1[r * wpad + x] = v[k] * v[k];               // pixelMultOnGPU(d_im, d_im) :111
            }
        }
    }
    __syncthreads();
    // ---- k_v4_prep's part II: the padded (value, x-derivative) rows out of the pixel values in LDS (x_derivativeOnGPU
    // costVolume.cu:358-381: (I[x-1] - I[x+1]) / 2, one-sided at the image edges; integers <= 255: exact whichever way formed).
    // (From global memory, a cell per loop trip, this part cost 5 us: its byte loads were waited for trip by trip.)
    if (view < ga.nimg && !(SMX_GR_WHATIF & 4)) {
        fg_t* __restrict__ FGo = ga.prep.FG[view];
        const int nr = min(rows, h - y0);
        for (int e = tid; e < nr * wp; e += GR_NT) {
            const int r = e / wp, xp = e - r * wp, x = xp - PADX;
            float f = 60000.0f, g = 60000.0f;
            if (x >= 0 && x < w) {
                const float* pr = P0 + r * wpad;
                f = pr[x];
                const float c1 = pr[x + 1 < w ? x + 1 : x], c2 = pr[x - 1 >= 0 ? x - 1 : x];     // (w >= 2)
                g = 1.0f * (c2 - c1) / 2;
            }
            fg_t v;
            v.x = (_Float16)f;
            v.y = (_Float16)g;
            FGo[(size_t)(y0 + r) * wp + xp] = v;
        }
    }
    if (view >= ga.nviews) return;          // (an image that is only the other view's partner: no statistics)
    __syncthreads();                        // (the scan below overwrites the pixel values in place)
    // (2) columns behind the image hold copies of the last pixel; their sums are never stored
    if (tid < 2 * rows && !(SMX_GR_WHATIF & 1)) {
        float* row = gr_lds + tid * wpad;                     // tid = plane * rows + row
        float acc = -0.0f;                                    // exact additive identity
        f4 c[16], n[16];
        auto rd = [&](f4 (&t)[16], int x0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) t[k] = *(const f4*)(row + x0 + 4 * k);
        };
        auto scan_wr = [&](f4 (&t)[16], int x0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                acc = t[k].x + acc; t[k].x = acc;
                acc = t[k].y + acc; t[k].y = acc;
                acc = t[k].z + acc; t[k].z = acc;
                acc = t[k].w + acc; t[k].w = acc;
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) *(f4*)(row + x0 + 4 * k) = t[k];
        };
        rd(c, 0);
        for (int x0 = 0; x0 < wr; x0 += 128) {
            rd(n, x0 + 64);
            scan_wr(c, x0);
            rd(c, min(x0 + 128, wr - 64));                    // (behind the last pair: a harmless re-read)
            scan_wr(n, x0 + 64);
        }
    }
    __syncthreads();
    for (int r = 0; r < rows; ++r) {
        if (y0 + r >= h || (SMX_GR_WHATIF & 2)) break;
        float* d0 = S0 + (size_t)(y0 + r) * w;
        float* d1 = S1 + (size_t)(y0 + r) * w;
        for (int x = tid; x < w; x += GR_NT) {
            d0[x] = P0[r * wpad + x];
            d1[x] = P1[r * wpad + x];
        }
    }
}

// Column prefix sums in place (colSum integral.cu:121-131): one latency chain of h dependent adds per column.  A workgroup
// owns 64 columns of one plane; its GC_NW waves take the segments of GC_SEG rows round robin: every wave has the loads of
// its segment in flight from the start (a wave can track 63 vector-memory operations, and a load takes ~1 us here: one wave per
// column, round 4, spent 16 us on KITTI shape waiting twelve times for 32 loads queued behind the stores of the batch before),
// and the chain itself -- the reference's top -> bottom order, one running sum per lane -- is handed from wave to wave
// through LDS (s_turn: whose segment it is; the waves of a workgroup are co-resident, so the spin cannot starve).
// grid (ceil(w/64), 2 planes, nviews), block 64 GC_NW
constexpr int GC_NW = 16, GC_SEG = 32;
__global__ __launch_bounds__(64 * GC_NW) void k_v4_guid_cols(GuidArgs ga, int w, int h) {
    __shared__ float s_acc[64];
    __shared__ int s_turn;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane;
    const bool on = x < w;
    float* __restrict__ p = ga.S[blockIdx.z][blockIdx.y] + min(x, w - 1);
    if (threadIdx.x == 0) s_turn = 0;
    __syncthreads();
    const int nseg = (h + GC_SEG - 1) / GC_SEG;
    float v[GC_SEG];
    for (int seg = wv; seg < nseg; seg += GC_NW) {
        const int y0 = seg * GC_SEG;
#pragma unroll
        for (int t = 0; t < GC_SEG; ++t) v[t] = p[(size_t)min(y0 + t, h - 1) * w];
        while (__hip_atomic_load(&s_turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != seg) __builtin_amdgcn_s_sleep(1);
        float acc = seg == 0 ? -0.0f : s_acc[lane];               // -0: the exact additive identity
#pragma unroll
        for (int t = 0; t < GC_SEG; ++t) { acc = v[t] + acc; v[t] = acc; }
        s_acc[lane] = acc;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&s_turn, seg + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        // (the next wave is on the chain already while this one stores)
        if (on) {
#pragma unroll
            for (int t = 0; t < GC_SEG; ++t)
                if (y0 + t < h) p[(size_t)(y0 + t) * w] = v[t];
        }
    }
}

// mean_I = box(S_I); var = box(S_II) - mean_I*mean_I; 1/(var + eps) in double as the reference's
// compute_ak_and_bk (guidedFilter.cu:350).  grid (ceil(w/256), h, nviews)
__global__ void k_v4_guid_finish(GuidArgs ga, int w, int h, int R, double eps) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, view = blockIdx.z;
    if (x >= w) return;
    const f2 g = guid_point(ga.S[view][0], ga.S[view][1], x, y, w, h, R, eps);
    const size_t id = (size_t)y * w + x;
    ga.G[view][id] = g;
    if (ga.mean_u8[view]) ga.mean_u8[view][id] = mean_to_u8(g.x);
}

// Hand-off record of one iteration (per parity and slice-view), written and read in 16-byte units:
//   [0, BH)              stage-1 row carries of band i        (float2 per row)
//   [BH, 2 BH)           stage-2 row carries of band i-1
//   [2 BH, 2 BH + BH*HP) last 2R+1 columns of the stage-2 integral of band i-1, HP = 20 float2 per row
constexpr int HP = HWMAX + 1;
constexpr int REC_F2 = 2 * BH + BH * HP;          // float2 per record
constexpr int REC_U = REC_F2 / 2;                 // 16-byte units per record: one per thread
static_assert(REC_U <= NT && HP % 2 == 0 && BH % 2 == 0, "one 16-byte hand-off unit per thread");

// stage-1 input quads: four ring columns of one row per lane (one 16-byte load per image), held by waves 2..7
constexpr int NQ = (TWMAX + 3) / 4;               // quads per ring row (21; the last one ends in the pitch padding)
constexpr int QW0 = 2;                            // first wave that holds quads
constexpr int QTH = NT - 64 * QW0;                // threads that hold quads
constexpr int NQR = BH / 16;                      // quads per quad-holding thread: rows r, r + 16, ..
static_assert(NQ * 4 <= NCOLP && NQ <= 4 * (NWAVE - QW0), "one wave-instruction covers 16 rows x 4 quads");

// Diagnostic build only (-DSMX_V4_STAMPS=<item>): every wave of one work item records the shader clock
// at its phase boundaries; the product build contains no stamp.
#ifdef SMX_V4_STAMPS
constexpr int STAMP_W = 12;             // stamps per iteration
constexpr int STAMP_SLOTS = STAMP_W * 40;
__device__ unsigned long long g_stamps[NWAVE * STAMP_SLOTS];
#define V4_STAMP(n)                                                                          \
    do {                                                                                     \
        if (item == SMX_V4_STAMPS && lane == 0 && i * STAMP_W + (n) < STAMP_SLOTS)            \
            g_stamps[wave * STAMP_SLOTS + i * STAMP_W + (n)] = __builtin_amdgcn_s_memtime();  \
    } while (0)
#elif defined(SMX_V4_MARK)
#define V4_STAMP(n) asm volatile("; V4_MARK " #n)      // (to find the phases in the ISA listing)
#else
#define V4_STAMP(n) ((void)0)
#endif
// Diagnostic build only (-DSMX_V4_DUMP=<item> -DSMX_V4_DUMP_IT=<iteration> -DSMX_V4_DUMP_PH=<0..3: behind the
// barrier that ends W / R / C / X>): both LDS rings of one work item at one point -> global memory
#ifdef SMX_V4_DUMP
__device__ float g_dump[2 * RR * ROWF];
#define V4_DUMP(ph)                                                                               \
    do {                                                                                          \
        if (item == SMX_V4_DUMP && i == SMX_V4_DUMP_IT && (ph) == SMX_V4_DUMP_PH) {                \
            for (int e_ = tid; e_ < RR * ROWF; e_ += NT) {                                         \
                g_dump[e_] = ring1[e_];                                                            \
                g_dump[RR * ROWF + e_] = ring2[e_];                                                \
            }                                                                                     \
            wg_barrier();                                                                         \
        }                                                                                         \
    } while (0)
#else
#define V4_DUMP(ph) ((void)0)
#endif
#ifdef SMX_V4_ITEMLOG
constexpr int ITEMLOG_MAX = 1 << 16;
__device__ unsigned long long g_itemlog[3 * ITEMLOG_MAX];
#endif

// Hand-in hysteresis (experiment, default off): an item that has caught up with its left neighbour meets the slow
// hand-in -- poll, barrier, exposed cross-XCD load -- in every iteration; with SLACK > 0 it waits once until the
// neighbour is SLACK records ahead.  Measured on KITTI (1242x375, D=192): SLACK 0 / 2 / 3 / 5 / 8 -> 1.266 / 1.274 /
// 1.274 / 1.279 / 1.280 ms per pair: the slow hand-in is not what a caught-up item loses time on.
#ifndef SMX_V4_SLACK
#define SMX_V4_SLACK 0
#endif
constexpr unsigned SLACK = SMX_V4_SLACK;
// Diagnostic build only (-DSMX_V4_WHATIF=<bits>): leaves parts of the work out (WRONG results) to see what
// the kernel time is sensitive to.  1: no q stores / guidance loads in X; 2: box taps not read from LDS;
// 4: no cost evaluation; 8: no column scans; 16: no row scans; 32: no stage-1 cost loads; 64: no box at all;
// 128: no exact-division check; 256: no division; 512: no stage-1 box; 1024: no stage-2 box; 2048: no hand-off
// between strips; 4096: the row-scan wave never waits for the neighbour's record
#ifndef SMX_V4_WHATIF
#define SMX_V4_WHATIF 0
#endif
constexpr int WHATIF = SMX_V4_WHATIF;

// one ds_read_b64 the compiler cannot pair into a ds_read2_b64 (which takes four times the LDS cycles of two
// ds_read_b64 for the same bytes); the caller waits with lds_wait16() before the first use
#define LDS_RD64(dst, addr, imm) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(imm))
// a cell's (first, second) component of a component-planar ring row as ONE register pair: the compiler would pair
// the loads of a row by tap instead (both taps of a component per instruction) and then shuffle eight registers
#define LDS_RD2(dst, addr, o0, o1) \
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(addr), "i"(o0), "i"(o1))

// RT: compile-time box radius (RMAX) or -1 = the radius of the call (A.R <= RMAX)
// FAST: the non-bit-exact mode: the row prefix sums are wave-parallel DPP scans over the
// columns of a row instead of the reference's sequential left -> right chain, i.e. the additions are
// re-associated.  Reported separately, never the product default.
template <int SRC, int RT, bool FAST>
__global__ __launch_bounds__(NT, 2 * WG_PER_CU) void k_v4_walk(Args A) {
    __shared__ __attribute__((aligned(16))) float ring1[RR * ROWF];   // stage-1 row y at ring row y mod RR
    __shared__ __attribute__((aligned(16))) float ring2[RR * ROWF];   // a/b row y at ring row (y + R) mod RR
    __shared__ __attribute__((aligned(16))) f2 cout[2][BH];         // row carries out
    __shared__ __attribute__((aligned(16))) f2 cin_fast[FAST ? 2 : 1][FAST ? BH : 1];   // FAST: row carries in
    __shared__ float rcp_s[HWMAX * HWMAX + 1];                      // RN(1/area)
    __shared__ int s_item, s_next;
    __shared__ unsigned s_seen;                                     // last value read from the left neighbour's flag

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = A.w, h = A.h, K = A.K, NI = A.NI, nsv = A.nsv;
    const int R = RT >= 0 ? RT : A.R;
    const int HW = 2 * R + 1, TW = OW + HW;
    const CostConst cc = A.cc;
    const f2 ident = {-0.0f, -0.0f};            // exact additive identity: v + (-0) == v
    if (tid <= HWMAX * HWMAX) rcp_s[tid] = kRcp.v[tid];

    // Hand-off units.  In: the row-scan lanes (waves 0, 1) load their own carry dword; the threads of waves
    // QW0.. load one 16-byte halo unit each (two columns of one row).  Out: the same threads store the halo
    // units, the next BH of them the row carries of both stages (two rows per unit).
    const int qt = tid - 64 * QW0;              // index among the threads of waves QW0..
    constexpr int NHALO_U = BH * HP / 2;        // halo units per record
    static_assert(NHALO_U + BH <= QTH, "one hand-off unit per thread of waves QW0..");
    const bool hu_halo = qt >= 0 && qt < NHALO_U;
    const bool hu_carry = qt >= NHALO_U && qt < NHALO_U + BH;
    // halo unit: row of the band, first of the two columns; byte offset of the unit inside a record
    auto hu_rc = [&](int& r, int& c) {
        const int u = max(opaque(qt), 0);
        r = u / (HP / 2);
        c = (u - r * (HP / 2)) * 2;
    };
    auto hu_off = [&]() {
        const int u = opaque(qt);
        return u < NHALO_U ? (unsigned)(2 * BH * 8 + u * 16) : (unsigned)((u - NHALO_U) * 16);
    };
    // row-scan lanes: each 32-lane group (the unit of LDS banking for dword accesses) takes half of the rows
    // with both components, lanes 0-15 / 16-31 = first / second component: 16 distinct even + 16 distinct
    // odd banks (row stride 170 dwords = 10 mod 32)
    // BH = 32: wave 0 scans stage 1, wave 1 stage 2; BH = 16: wave 0 scans both, lanes 32.. = stage 2
    auto srow_of = [&]() { const int l = opaque(lane); return BH == 32 ? (l & 15) + 16 * (l >> 5) : (l & 15); };
    auto scomp_of = [&]() { return (opaque(lane) >> 4) & 1; };
    auto sstage_of = [&]() { return BH == 32 ? wave : (opaque(lane) >> 5); };
    constexpr int NRSW = BH == 32 ? 2 : 1;      // row-scan waves
    // this thread's two stage-1 input quads (waves QW0..): row of the band, first ring column
    // One wave-instruction covers 16 rows x 4 quads: the LDS writes of a 16-lane group then go to 16 different
    // rows (row stride 170 dwords = 10 mod 32: conflict-free), the global loads to 64-byte runs of 16 rows.
    // Wave QW0 + g takes the quads 4g .. 4g+3 of a row; its two rounds the rows 0..15 / 16..31 of the band.
    const bool q_on = wave >= QW0 && 4 * (wave - QW0) + (lane >> 4) < NQ;
    auto quad_rc = [&](int e, int& r, int& c) {
        const int l = opaque(lane);
        r = 16 * e + (l & 15);
        c = (4 * (wave - QW0) + (l >> 4)) * 4;
    };

    if (A.only_if && flag_load(const_cast<unsigned*>(A.only_if)) == 0u) return;     // (uniform: every thread reads the same word)
    if (tid == 0) s_item = (int)__hip_atomic_fetch_add((gu32*)A.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        wg_barrier();
        const int item = s_item;
        if (item >= A.nitems) break;
#ifdef SMX_V4_ITEMLOG
        if (tid == 0 && item < ITEMLOG_MAX) {
            g_itemlog[3 * item] = __builtin_amdgcn_s_memrealtime();
            g_itemlog[3 * item + 2] = blockIdx.x;
        }
#endif
        const int k = item / nsv;
        const int sv = item - k * nsv;
        const int view = sv / A.nslices;
        const int slice = sv - view * A.nslices;
        const View& V = A.v[view];
        const int xs = k * OW;
        const int cs1 = xs - R - 1;             // image column of ring-1 column 0
        const int cs2 = xs - HW;                // image column of ring-2 column 0
        const bool pred = k > 0 && !(WHATIF & 2048), succ = k + 1 < K && !(WHATIF & 2048);   // (2048: no hand-off between strips)
        const int d = V.d0 + slice;
        unsigned* const myflag = A.flags + (size_t)sv * K + k;
        const size_t recs = (size_t)NI * REC_F2;      // float2 per (parity, slice-view)
        const rsrc_t r_in = mk_rsrc(A.hand + ((size_t)((k - 1) & 1) * nsv + sv) * recs, recs * 8);
        const rsrc_t r_out = mk_rsrc(A.hand + ((size_t)(k & 1) * nsv + sv) * recs, recs * 8);

        // Every global access of the LANE = COLUMN phases is a buffer instruction: per-lane byte offset fixed
        // for the item + wave-uniform row offset in a scalar register, no 64-bit address arithmetic.  A lane
        // whose column lies outside the image carries the offset OOB, which is beyond every plane: its loads
        // return 0 and its stores are dropped by the range check.
        const unsigned fgw4 = ((unsigned)w + 2u * PADX) * 4u, w4 = (unsigned)w * 4u;
        const size_t plane = (size_t)h * w;
        const rsrc_t r_fg1 = mk_rsrc(V.FG1, (size_t)h * fgw4);
        const rsrc_t r_in2 = SRC == SRC_IMG ? mk_rsrc(V.FG2, (size_t)h * fgw4)
                                            : mk_rsrc(V.cost + (size_t)slice * plane, plane * 4);
        const rsrc_t r_g = mk_rsrc(V.guid, plane * 8);
        const rsrc_t r_q = mk_rsrc(V.q + (size_t)slice * plane, plane * 4);

        // per-lane window geometry in x (needed by the clipped box only) and the byte offset (x4) of a lane's
        // column in a row
        struct Geo { int jmax, jmin, xcw; bool hx; };
        auto mkgeo = [&](int x, int cs) {
            Geo g;
            const int xc = min(max(x, 0), w - 1);
            const int xmax = min(w - 1, xc + R), xmn = xc - R - 1;
            g.hx = xmn >= 0;
            g.xcw = xmax - (g.hx ? xmn : -1);
            g.jmax = min(max(xmax - cs, 0), TW - 1);
            g.jmin = min(max(xmn - cs, 0), TW - 1);
            return g;
        };
        auto coloff = [&](int x) { return x >= 0 && x < w ? (unsigned)x * 4u : OOB; };
        // Per-lane constants of the item that every iteration needs stay in VGPRs (the kernel has 80 to spend at three
        // workgroups per CU): re-deriving them costs more vector instructions than anything else they could buy.
        const unsigned vg_keep = coloff(xs + lane), vq_keep = coloff(xs - R + lane);
        auto vg_of = [&]() { return vg_keep; };                          // a_k, b_k column
        auto vq_of = [&]() { return vq_keep; };                          // q column
        // the hand-off unit of this thread: byte offset inside a record; ring-2 float offset of its (row, column)
        // relative to the first row of the band; which of its two columns lie inside the halo
        const unsigned rec_voff = wave < NRSW ? (unsigned)(((sstage_of() * BH + srow_of()) * 8) & ~15)
                                              : ((hu_halo || hu_carry) ? hu_off() : 0u);
        int hu_r0, hu_c0;
        hu_rc(hu_r0, hu_c0);
        const int hu_lds = hu_r0 * ROWF + hu_c0;
        const bool hu_in = hu_c0 < HW, hu_two = hu_c0 + 1 < HW;
        // ring float offset of row (rbase + r) for a kept offset o = r * ROWF + c (c + 64 < ROWF)
        auto ring_off = [&](int rbase, int o) {
            const int v = rbase * ROWF + o;
            return v >= RR * ROWF ? v - RR * ROWF : v;
        };
        // the stage-1 quad of this thread: ring-1 offset, byte offsets in the two input planes (without the band term)
        int q_r0, q_c0;
        quad_rc(0, q_r0, q_c0);
        const int q_lds = q_r0 * ROWF + q_c0;
        const unsigned q_off1 = (unsigned)(min(max(cs1 + q_c0, -PADX), w) + PADX) * 4u + (unsigned)q_r0 * fgw4;
        const unsigned q_off2 = (unsigned)(min(max(cs1 + q_c0 + d, -PADX), w) + PADX) * 4u + (unsigned)q_r0 * fgw4;
        // all 64 windows of the strip unclipped in x: no selects, one area
        const bool xint1 = xs - R - 1 >= 0 && xs + OW - 1 + R <= w - 1;
        const bool xint2 = xs - 2 * R - 1 >= 0 && xs + OW - 1 <= w - 1;
        const float area_full = (float)(HW * HW), ra_full = rcp_s[HW * HW];
        // iterations whose box stage is interior (every window of the band unclipped): [f_lo, f_lo + f_n)
        //   stage 1, band i:   BH i - 2R - 1 >= 0 and BH i + BH <= h
        //   stage 2, band i-1: BH (i-1) - 3R - 1 >= 0 and BH i - R <= h
        const int f1_lo = (2 * R + 1 + BH - 1) / BH, f2_lo = 1 + (3 * R + 1 + BH - 1) / BH;
        const unsigned f1_n = xint1 ? (unsigned)max(h / BH - f1_lo, 0) : 0u;
        const unsigned f2_n = xint2 ? (unsigned)max((h + R) / BH - f2_lo + 1, 0) : 0u;

        const int jlo1 = max(0, -cs1), jhi1 = min(TW, w - cs1);   // ring-1 columns inside the image
        const int jlo2 = HW, jhi2 = min(TW, w - cs2);             // new ring-2 columns inside the image

        // stage-1 inputs of the next band: raw quads (four (value, gradient) cells of this view / of the other view,
        // or four raw costs), loaded in W(i); evaluated to (p, I p) under the row scans of R(i); written in W(i+1)
        u4 qa[NQR], qb[NQR];
        f4 qresx[NQR], qresy[NQR];          // (p, I p) of the quad, planar like the ring rows they are written to
        f2 gab[RPW];                         // (mean_I, 1/(var+eps)) of the a/b rows of the next X phase
        uint32_t Iraw[RPW];                  // raw (value, gradient) halves of the q rows of the next X phase
        f2 abreg[RPW];                       // a_k, b_k of this wave's rows, written to ring 2 in the next W phase
        f4 hreg = {0, 0, 0, 0};              // this thread's unit of the left neighbour's next record: a halo unit (waves
                                             // QW0..) or the unit with this row-scan lane's carry
        bool have_pref = false;
        unsigned seen = 0;
#pragma unroll
        for (int t = 0; t < RPW; ++t) { abreg[t] = ident; }

        // loads of the stage-1 inputs of band ib (rows clamped into the image: every load is issued)
        auto issue_cost = [&](int ib) {
            if (!q_on) return;
            if (SRC == SRC_IMG && NQR == 1) {
                // kept per-lane offsets + the band's row offset as the scalar operand; the rows of the last band are
                // clamped per lane.  (One load site: loads in two branches would be waited for where they merge.)
                unsigned vo1 = q_off1, vo2 = q_off2;
                int soff = BH * ib * (int)fgw4;
                if (BH * ib + BH > h) {
                    const unsigned dy = (unsigned)(BH * ib + q_r0 - min(BH * ib + q_r0, h - 1)) * fgw4;
                    vo1 -= dy;
                    vo2 -= dy;
                }
                qa[0] = __builtin_amdgcn_raw_buffer_load_b128(r_fg1, (int)vo1, soff, 0);
                qb[0] = __builtin_amdgcn_raw_buffer_load_b128(r_in2, (int)vo2, soff, 0);
                return;
            }
#pragma unroll
            for (int e = 0; e < NQR; ++e) {
                int qr, qc;
                quad_rc(e, qr, qc);
                // a quad wholly outside the image is moved onto the sentinel columns (its cells are written but
                // never accumulated)
                const unsigned y = (unsigned)min(BH * ib + qr, h - 1);
                const int c1 = min(max(cs1 + qc, -PADX), w);
                qa[e] = __builtin_amdgcn_raw_buffer_load_b128(r_fg1, (int)((unsigned)(c1 + PADX) * 4u + y * fgw4), 0, 0);
                if (SRC == SRC_IMG) {
                    const int c2 = min(max(cs1 + qc + d, -PADX), w);
                    qb[e] = __builtin_amdgcn_raw_buffer_load_b128(r_in2, (int)((unsigned)(c2 + PADX) * 4u + y * fgw4), 0, 0);
                } else {
                    unsigned c4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = min(max(cs1 + qc + j, 0), w - 1);
                        c4[j] = ldu(r_in2, (unsigned)c * 4u + y * w4, 0);
                    }
                    qb[e] = (u4){c4[0], c4[1], c4[2], c4[3]};
                }
            }
        };
        // raw -> (p, I p)
        auto eval_cost = [&]() {
            if (!q_on) return;
#pragma unroll
            for (int e = 0; e < NQR; ++e) {
                const unsigned ra[4] = {qa[e].x, qa[e].y, qa[e].z, qa[e].w};
                const unsigned rb4[4] = {qb[e].x, qb[e].y, qb[e].z, qb[e].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const fg_t q1 = __builtin_bit_cast(fg_t, ra[j]);
                    f2 v;
                    if (SRC == SRC_IMG) {
                        v = cost_pair(q1, __builtin_bit_cast(fg_t, rb4[j]), cc);
                    } else {
                        v.x = __builtin_bit_cast(float, rb4[j]);   // copyFromBigToLittleOnGPU guidedFilter.cu:198
                        v.y = (float)q1.x * v.x;                   // pixelMultOnGPU(d_im, d_p) :209
                    }
                    qresx[e][j] = v.x;
                    qresy[e][j] = v.y;
                }
            }
        };
        // loads for the X phase of iteration ib: guidance of its a/b rows, guidance image of its q rows
        auto issue_guid = [&](int ib) {
            const unsigned vg = vg_of(), vq = vq_of();
            const unsigned vg8 = vg == OOB ? OOB : 2u * vg;          // this lane's a/b column in the float2 guidance plane
            const unsigned vqI = vq == OOB ? OOB : vq + 4u * PADX;   // this lane's q column in the padded image plane
            const int ya0 = BH * ib - R + RPW * wave;
#pragma unroll
            for (int t = 0; t < RPW; ++t) {
                const int y = min(max(ya0 + t, 0), h - 1);
                const u2 g = __builtin_amdgcn_raw_buffer_load_b64(r_g, (int)vg8, y * (int)(2u * w4), 0);
                gab[t] = __builtin_bit_cast(f2, g);
            }
            const int yq0 = BH * (ib - 1) - 2 * R + RPW * wave;
#pragma unroll
            for (int t = 0; t < RPW; ++t) {
                const int y = min(max(yq0 + t, 0), h - 1);
                Iraw[t] = ldu(r_fg1, vqI, y * (int)fgw4);
            }
        };

        // loads of this thread's part of the left neighbour's record `rec` (sc1: the hand-off form of the guide)
        // One 16-byte load per thread, without a branch (a load under a condition makes the compiler merge old and
        // new register values right behind it, i.e. wait for it on the spot): the halo unit of a thread of waves
        // QW0.., the unit that holds its carry for a row-scan lane (any unit for the others: never used).
        auto fetch_rec = [&](int rec) {
            hreg = ld16_sc1(r_in, (unsigned)(rec * REC_F2 * 8) + rec_voff);
        };
        // bounded wait for the left neighbour's flag >= need (thread 0 only); result -> s_seen
        auto spin_pred = [&](unsigned need) {
            const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                const unsigned f = flag_load(myflag - 1);
                if (f >= need) { s_seen = f; break; }
                __builtin_amdgcn_s_sleep(4);
                // give up after 2 s (100 MHz counter) or as soon as any workgroup has given up; the call then
                // reports SMX_E_HIP through smx_dev_agg_status
                if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull || flag_load(A.status) != 0u) {
                    flag_store(A.status, 1u + (unsigned)item);
                    s_seen = FLAG_DONE;
                    break;
                }
            }
        };

        // ---- row scans of iteration i: this lane = one component of one row of one stage --------------------
        // stage 1: band i, ring-1 columns [jlo1, jhi1); stage 2: a/b band i-1, the new ring-2 columns [jlo2, jhi2)
        auto rowscans = [&](int i, int rb, int rbp) {
            const int st = sstage_of(), srow = srow_of(), scomp = scomp_of();
            const int y = st == 0 ? BH * i + srow : BH * (i - 1) - R + srow;
            const int jlo = st == 0 ? jlo1 : jlo2, jhi = st == 0 ? jhi1 : jhi2;
            const bool act = y >= 0 && y < h && jhi > jlo && (st == 0 || i >= 1);
            int rr = (st == 0 ? rb : rbp) + srow;
            rr = rr >= RR ? rr - RR : rr;
            // the unit holds the carries (p, I p) / (a, b) of rows 2k, 2k+1
            const float c01 = scomp ? hreg.y : hreg.x, c23 = scomp ? hreg.w : hreg.z;
            float acc = (pred && !(WHATIF & 4096)) ? ((srow & 1) ? c23 : c01) : -0.0f;   // (4096: wave 0 never waits for the record)
#ifdef SMX_V4_STAMPS
            asm volatile("" : "+v"(acc));
            V4_STAMP(11);
#endif
            float* row = (st == 0 ? ring1 : ring2) + rr * ROWF + scomp * OFF1;   // column c of this component: row[c]
            float* const co = (float*)&cout[st][srow] + scomp;
            // NG groups of four columns from group G0, unrolled; the reads run two groups ahead of the adds.
            // hook(g) runs behind group g.
            auto run4 = [&](auto G0c, auto NGc, auto hook) {
                constexpr int G0 = decltype(G0c)::value, NG = decltype(NGc)::value;
                f4* const r4 = (f4*)row;
                f4 v[3];
                f4 xo[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
                v[0] = r4[G0];
                if (NG > 1) v[1] = r4[G0 + 1];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (g + 2 < NG) v[(g + 2) % 3] = r4[G0 + g + 2];
                    const f4 in = v[g % 3];
                    f4& x = xo[g & 1];
                    acc = in.x + acc; x.x = acc;
                    acc = in.y + acc; x.y = acc;
                    acc = in.z + acc; x.z = acc;
                    hook(G0 + g, acc);
                    acc = in.w + acc; x.w = acc;
                    r4[G0 + g] = x;
                    // The sums of two consecutive groups live in different registers: the adds of a group then do not
                    // have to wait until the 16-byte store of the previous one has read its four source registers.
                    asm volatile("" :: "v"(xo[(g & 1) ^ 1]));
                }
            };
            if (RT == RMAX && jlo1 == 0 && jhi1 == TWMAX && jhi2 == TWMAX) {
                // the common case (a strip inside the image at radius 9): stage 1 scans ring columns 0 .. 82 (and
                // the unused column 83), stage 2 the columns 19 .. 82; the carry for the next strip is the running
                // sum behind ring column 63 (stage 1) / behind column 82 (stage 2)
                static_assert(HWMAX == 19 && OW == 64 && TWMAX == 83, "group boundaries below");
                if (act && st == 0) run4(std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{}, [](int, float) {});
                if (act && st == 1) { acc = row[HWMAX] + acc; row[HWMAX] = acc; }    // (columns 16 .. 18 are the neighbour's)
                if (act) {
                    run4(std::integral_constant<int, 5>{}, std::integral_constant<int, 11>{}, [](int, float) {});
                    if (st == 0) *co = acc;                                           // behind column 63
                    run4(std::integral_constant<int, 16>{}, std::integral_constant<int, 5>{},
                         [&](int g, float a) { if (g == 20 && st == 1) *co = a; });   // behind column 82
                }
                return;
            }
            if (!act) return;
            // general strip: batches of 8 columns, ping-pong; the reads of the next batch are always issued
            // (past the end they fetch bytes nobody uses: LDS reads cannot fault)
            int j = jlo;
            const int nb8 = (jhi - j) >> 3;
            float va[8], vb[8];
            auto rd = [&](float (&v)[8], int c) {
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] = row[c + t];
            };
            auto runb = [&](const float (&v)[8], int c) {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    acc = v[t] + acc;
                    row[c + t] = acc;
                }
            };
            if (nb8 > 0) rd(va, j);
            int b = 0;
            for (; b + 2 <= nb8; b += 2) {
                rd(vb, j + 8);
                runb(va, j);
                rd(va, j + 16);
                runb(vb, j + 8);
                j += 16;
            }
            if (b < nb8) {
                runb(va, j);
                j += 8;
            }
            for (; j < jhi; ++j) {
                acc = row[j] + acc;
                row[j] = acc;
            }
            // running row sum left of the next strip's first column (stage 1: ring column OW-1; stage 2: the last one)
            if (st == 0) { if (OW - 1 >= jlo && OW - 1 < jhi) *co = row[OW - 1]; }
            else *co = acc;
        };

        // ---- FAST mode: wave-parallel row scans.  The 4 BH (stage, row, component) scans of an iteration are
        // dealt to all eight waves; one scan = an inclusive DPP prefix sum over ring columns jlo .. 63 (LANE =
        // COLUMN), then over the columns 64 .. jhi-1 with the first part's total as carry.
        auto wave_scan = [&](float x) {
            // Hillis-Steele inside the 16-lane rows, then across them (row_bcast15 / row_bcast31)
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xF, 0xF, false));
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x112, 0xF, 0xF, false));
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x114, 0xF, 0xF, false));
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x118, 0xF, 0xF, false));
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x142, 0xA, 0xF, false));
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x143, 0xC, 0xF, false));
            return x;
        };
        auto rowscans_fast = [&](int i, int rb, int rbp) {
            // every wave takes 4 BH / NWAVE scans, four at a time (independent chains interleave); a scan that
            // has nothing to do is predicated off, not branched around
            constexpr int NU = 4 * BH / NWAVE, UB = 4;
            static_assert(NU % UB == 0, "scans per wave");
#pragma unroll
            for (int ub = 0; ub < NU; ub += UB) {
                float* row[UB];
                float* co[UB];
                float carry[UB], s0[UB], s1[UB], tot[UB];
                int jlo[UB], jhi[UB], st[UB];
                bool on[UB];
#pragma unroll
                for (int k = 0; k < UB; ++k) {
                    const int u = wave + NWAVE * (ub + k);
                    st[k] = u / (2 * BH);
                    const int srow = (u >> 1) % BH, scomp = u & 1;
                    const int y = st[k] == 0 ? BH * i + srow : BH * (i - 1) - R + srow;
                    jlo[k] = st[k] == 0 ? jlo1 : jlo2;
                    jhi[k] = st[k] == 0 ? jhi1 : jhi2;
                    on[k] = y >= 0 && y < h && jhi[k] > jlo[k] && (st[k] == 0 || i >= 1);       // wave-uniform
                    int rr = (st[k] == 0 ? rb : rbp) + srow;
                    rr = rr >= RR ? rr - RR : rr;
                    row[k] = (st[k] == 0 ? ring1 : ring2) + rr * ROWF + scomp * OFF1;
                    co[k] = (float*)&cout[st[k]][srow] + scomp;
                    carry[k] = pred ? ((const float*)&cin_fast[st[k]][srow])[scomp] : 0.0f;
                    s0[k] = on[k] && lane >= jlo[k] && lane < jhi[k] ? row[k][lane] : 0.0f;
                    s1[k] = on[k] && 64 + lane < jhi[k] ? row[k][64 + lane] : 0.0f;
                }
#pragma unroll
                for (int k = 0; k < UB; ++k) s0[k] = wave_scan(s0[k]) + carry[k];
#pragma unroll
                for (int k = 0; k < UB; ++k) tot[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s0[k]), 63));
#pragma unroll
                for (int k = 0; k < UB; ++k) s1[k] = wave_scan(s1[k]) + tot[k];
#pragma unroll
                for (int k = 0; k < UB; ++k) {
                    if (on[k] && lane >= jlo[k] && lane < jhi[k]) row[k][lane] = s0[k];
                    if (on[k] && 64 + lane < jhi[k]) row[k][64 + lane] = s1[k];
                    // the carry for the next strip: behind ring column OW-1 (stage 1) / behind the last column (stage 2)
                    const int last = st[k] == 0 ? OW - 1 : jhi[k] - 1;
                    if (on[k] && last >= jlo[k] && last < jhi[k]) {
                        if (last < 64) { if (lane == last) *co[k] = s0[k]; }
                        else if (lane == last - 64) *co[k] = s1[k];
                    }
                }
            }
        };

        // ---- column scan of one band for the dword `idx` (column, component) of this lane -------------------
        // rows yband + t at ring rows (rbase + t) mod RR; groups of four rows never wrap (RR % 4 == 0)
        auto colscan = [&](float* ring, int idx, int rbase, int yband, float& S) {
            float* const pc = ring + idx;
            constexpr int P2 = ROWF;
            if (yband >= 0 && yband + BH <= h) {
                // full band: up to 16 rows of reads in flight ahead of the dependent chain of adds and its writes
                auto rowp = [&](int t) {
                    int rr = rbase + (t & ~3);
                    rr = rr >= RR ? rr - RR : rr;
                    return pc + rr * P2 + (t & 3) * P2;
                };
                constexpr int NPF = BH < 16 ? BH : 16;
                float v[NPF];
#pragma unroll
                for (int t = 0; t < NPF; ++t) v[t] = *rowp(t);
#pragma unroll
                for (int t = 0; t < BH; ++t) {
                    S = v[t % NPF] + S;
                    *rowp(t) = S;
                    if (t + NPF < BH) v[t % NPF] = *rowp(t + NPF);
                }
                return;
            }
            for (int t = 0; t < BH; ++t) {
                const int y = yband + t;
                if (y < 0 || y >= h) continue;
                int rr = rbase + t;
                rr = rr >= RR ? rr - RR : rr;
                S = pc[rr * P2] + S;
                pc[rr * P2] = S;
            }
        };

        // ---- box means of this wave's four rows (computeBoxFilterOnGPU guidedFilter.cu:305-318: S11 - S10 -
        // S01 + S00 in that order, then a true division by the window area) ---------------------------------
        // fast form: every window of the band is unclipped in x and y.  rbase = ring row of the band's first
        // bottom tap row (the top tap row is 2R+1 ring rows above it).
        // a cell's (first, second) component: one ds_read2_b32
        auto cell = [&](const float* p) { return (f2){p[0], p[OFF1]}; };
        // obf = ring float offset of the wave's first bottom tap row (its group of RPW rows never wraps)
        auto box2_fast = [&](const float* ring, int obf, int half, f2* m) {
            const int ob0 = obf + 2 * half * ROWF;
            f2 s11[2], s10[2], s01[2], s00[2], val[2];
            if (RT == RMAX && !(WHATIF & 2)) {
                static_assert(HWMAX + OFF1 < 256, "ds_read2_b32 offsets");
                // top tap rows: HW ring rows above, each wrapped on its own
                int ot0 = ob0 - HW * ROWF;
                ot0 = ot0 < 0 ? ot0 + RR * ROWF : ot0;
                int ot1 = ot0 + ROWF;
                ot1 = ot1 >= RR * ROWF ? ot1 - RR * ROWF : ot1;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const unsigned ab = lds_off(ring + ob0 + t * ROWF + lane), at = lds_off(ring + (t ? ot1 : ot0) + lane);
                    LDS_RD2(s11[t], ab, HWMAX, HWMAX + OFF1);
                    LDS_RD2(s10[t], ab, 0, OFF1);
                    LDS_RD2(s01[t], at, HWMAX, HWMAX + OFF1);
                    LDS_RD2(s00[t], at, 0, OFF1);
                }
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(s11[0]), "+v"(s10[0]), "+v"(s01[0]), "+v"(s00[0]), "+v"(s11[1]), "+v"(s10[1]), "+v"(s01[1]),
                               "+v"(s00[1]));
            } else {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    int ot = ob0 + (t - HW) * ROWF;
                    ot = ot < 0 ? ot + RR * ROWF : ot;
                    const float* pb = ring + ob0 + t * ROWF + lane;
                    const float* pt = ring + ot + lane;
                    if (WHATIF & 2) {
                        s11[t] = s10[t] = s01[t] = s00[t] = (f2){(float)(ob0 + lane), 2.0f + ot};
                    } else {
                        s11[t] = cell(pb + HW); s10[t] = cell(pb);
                        s01[t] = cell(pt + HW); s00[t] = cell(pt);
                    }
                }
            }
            // (the two rows' dependent chains interleaved: a packed instruction that reads the result of the one in
            // front of it costs a wait state)
            f2 v0 = s11[0] - s10[0], v1 = s11[1] - s10[1];
            v0 = v0 - s01[0]; v1 = v1 - s01[1];
            v0 = v0 + s00[0]; v1 = v1 + s00[1];
            val[0] = v0; val[1] = v1;
            if (WHATIF & 256) {
                m[0] = v0; m[1] = v1;
            } else {
                const f2 d2 = {area_full, area_full}, r2 = {ra_full, ra_full};
                f2 q0 = v0 * r2, q1 = v1 * r2;
                f2 e0 = __builtin_elementwise_fma(-q0, d2, v0), e1 = __builtin_elementwise_fma(-q1, d2, v1);
                m[0] = __builtin_elementwise_fma(e0, r2, q0);
                m[1] = __builtin_elementwise_fma(e1, r2, q1);
            }
            // smallest / largest magnitude of the four sums: v_min3 + v_min, v_max3 + v_max
            const float amin = fminf(fminf(fminf(fabsf(val[0].x), fabsf(val[0].y)), fabsf(val[1].x)), fabsf(val[1].y));
            const float amax = fmaxf(fmaxf(fmaxf(fabsf(val[0].x), fabsf(val[0].y)), fabsf(val[1].x)), fabsf(val[1].y));
            // tiny, zero, infinite window sums (a NaN sum gives a NaN mean on either path)
            if (!(WHATIF & 128) && __any(!(amin >= 0x1p-100f) || !(amax < __builtin_inff()))) {
                asm volatile("; exact-division slow path");   // keep this a real (rare) wave-uniform branch
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    m[t].x = 1.0f * val[t].x / area_full;
                    m[t].y = 1.0f * val[t].y / area_full;
                }
            }
        };
        auto box4_fast = [&](const float* ring, int obf, f2 (&m)[RPW]) {
#pragma unroll
            for (int hf = 0; hf < RPW / 2; ++hf) box2_fast(ring, obf, hf, &m[2 * hf]);
        };
        // general form: windows clipped at the image borders; rows y0 + t that do not exist are skipped
        // (ok[t] = false).  shift = 0 (ring 1) or R (ring 2).
        auto box4_gen = [&](const float* ring, int shift, const Geo& g, bool xint, int y0, f2 (&m)[RPW], bool (&ok)[RPW]) {
            const float* const pmax = ring + g.jmax;
            const float* const pmin = ring + g.jmin;
            const bool left = xint || g.hx;
#pragma unroll
            for (int t = 0; t < RPW; ++t) {
                const int y = y0 + t;
                ok[t] = y >= 0 && y < h;
                m[t] = ident;
                if (!ok[t]) continue;                       // wave-uniform
                const int ymax = min(h - 1, y + R);
                const int ymin = y - R - 1;
                const bool hy = ymin >= 0;
                const int ych = ymax - (hy ? ymin : -1);
                const int o1 = ((ymax + shift) % RR) * ROWF, o0 = (((hy ? ymin : 0) + shift) % RR) * ROWF;
                const f2 s11 = cell(pmax + o1), s10 = cell(pmin + o1), s01 = cell(pmax + o0), s00 = cell(pmin + o0);
                const int ai = (xint ? HW : g.xcw) * ych;
                const float area = (float)ai, ra = rcp_s[ai];
                f2 v = s11;
                f2 u = v - s10;
                v = left ? u : v;
                u = v - s01;
                v = hy ? u : v;
                u = v + s00;
                v = (hy && left) ? u : v;
                m[t].x = div_small_int(v.x, area, ra);
                m[t].y = div_small_int(v.y, area, ra);
                if (__any(div_needs_exact(v.x) || div_needs_exact(v.y))) {
                    asm volatile("; exact-division slow path");
                    m[t].x = 1.0f * v.x / area;
                    m[t].y = 1.0f * v.y / area;
                }
            }
        };

        // q rows of X(iq)
        float qout[RPW];
        bool qok[RPW];
        // (interior: every row exists and every lane's column is its own -- no predicate, no clamp)
        auto store_q = [&](int iq, bool interior) {
            if (WHATIF & 1) return;
            const unsigned vq = vq_of();
            const int yq0 = BH * (iq - 1) - 2 * R + RPW * wave;
            unsigned vo[RPW];
            int so[RPW];
            if (interior) {
#pragma unroll
                for (int t = 0; t < RPW; ++t) { vo[t] = vq; so[t] = (yq0 + t) * (int)w4; }
            } else {
#pragma unroll
                for (int t = 0; t < RPW; ++t) { vo[t] = qok[t] ? vq : OOB; so[t] = min(max(yq0 + t, 0), h - 1) * (int)w4; }
            }
#pragma unroll
            for (int t = 0; t < RPW; ++t)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, qout[t]), r_q, (int)vo[t], so[t], AUX_NT);
        };

        // ===================================== the band loop ==============================================
        // prologue: stage-1 inputs of band 0 (evaluated here; band i+1 is loaded in W(i) and evaluated under the
        // row scans of R(i)), the loads of X(0), the left neighbour's record 0
        issue_cost(0);
        issue_guid(0);
        if (pred) {
            if (tid == 0) spin_pred(1u + SLACK);
            wg_barrier();
            seen = s_seen;
        }
        fetch_rec(0);
        have_pref = pred;
        eval_cost();
        float Sc = -0.0f;                   // running column sum of this lane's dword (waves 0..4)
        int rb = 0, rbp = 0;                // ring row of the first row of band i / band i-1 (both rings)
        // ring float offset of this wave's group of RPW rows of band i / band i-1 (a group never wraps)
        int ob1 = RPW * wave * ROWF, ob2 = ob1;
        for (int i = 0; i < NI; ++i) {
            // ------------------------------------ W(i) --------------------------------------------------
            V4_STAMP(0);
            {
                if (q_on) {
#pragma unroll
                    for (int e = 0; e < NQR; ++e) {
                        float* dst;
                        if (NQR == 1) {
                            dst = ring1 + ring_off(rb, q_lds);
                        } else {
                            int qr, qc;
                            quad_rc(e, qr, qc);
                            int rw = rb + qr;
                            rw = rw >= RR ? rw - RR : rw;
                            dst = ring1 + rw * ROWF + qc;
                        }
                        *(f4*)dst = qresx[e];
                        *(f4*)(dst + OFF1) = qresy[e];
                    }
                }
                if (i >= 1) {
#pragma unroll
                    for (int t = 0; t < RPW; ++t) {
                        float* dst = ring2 + ob2 + t * ROWF + HW + lane;
                        dst[0] = abreg[t].x;
                        dst[OFF1] = abreg[t].y;
                    }
                }
                if (pred && !have_pref) {
                    // the neighbour had not published record i when this item looked: wait for it now
                    if (tid == 0) spin_pred((unsigned)i + 1u + SLACK);
                    wg_barrier();
                    seen = s_seen;
                    fetch_rec(i);
                }
                have_pref = false;
                if (FAST && wave < NRSW && pred) {
                    // the carries of this iteration: from the row-scan lanes' units to where every wave finds them
                    const int st = sstage_of(), srow = srow_of(), scomp = scomp_of();
                    const float c01 = scomp ? hreg.y : hreg.x, c23 = scomp ? hreg.w : hreg.z;
                    ((float*)&cin_fast[st][srow])[scomp] = (srow & 1) ? c23 : c01;
                }
            }
            if (!(WHATIF & 32)) issue_cost(i + 1);      // lands under the row scans (rows clamped: harmless behind the last iteration)
            V4_STAMP(1);
            wg_barrier();
            V4_STAMP(2);
            V4_DUMP(0);
            // ------------------------------------ R(i) --------------------------------------------------
            // the scans are dependent chains on the critical path of the iteration: let them win the issue
            // arbitration against the waves (of this and the other workgroup) that share their SIMDs
            if (!FAST && wave < NRSW) {
                __builtin_amdgcn_s_setprio(3);
                if (!(WHATIF & 16)) rowscans(i, rb, rbp);
                __builtin_amdgcn_s_setprio(0);
            } else {
                if (wave == NWAVE - 1 && lane == 0) {
                    // an otherwise idle lane looks at the left neighbour's flag for the prefetch of record i+1
                    if (pred && seen != FLAG_DONE && seen < (unsigned)i + 2u) s_seen = flag_load(myflag - 1);
                    // ticket of the next item, one iteration before the end
                    if (i == NI - 1)
                        s_next = (int)__hip_atomic_fetch_add((gu32*)A.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (!(WHATIF & 4)) eval_cost();
                // the left neighbour's last 2R+1 columns of the stage-2 integral of band i-1 -> ring 2 (nobody
                // touches these columns before X(i))
                if (pred && hu_halo && i >= 1 && hu_in) {
                    // unit = (first, second) component of column hu_c, then of column hu_c + 1
                    float* dst = ring2 + ring_off(rbp, hu_lds);
                    if (hu_two) {
                        *(f2*)dst = (f2){hreg.x, hreg.z};
                        *(f2*)(dst + OFF1) = (f2){hreg.y, hreg.w};
                    } else {
                        dst[0] = hreg.x;
                        dst[OFF1] = hreg.y;
                    }
                }
                // every storing wave drains its global accesses here: the record stored in X(i-1) is complete in
                // memory before the barrier behind which one lane publishes it
                drain_vmem();
            }
            if (FAST) rowscans_fast(i, rb, rbp);
            V4_STAMP(3);
            wg_barrier();
            V4_STAMP(4);
            V4_DUMP(1);
            // ------------------------------------ C(i) --------------------------------------------------
            if (succ && tid == NT - 1 && i >= 1) flag_store(myflag, (unsigned)i);
            // one dword (column, component) of a ring row per lane, lane-linear along the row
            if (wave < 3) {
                __builtin_amdgcn_s_setprio(3);
                const int cidx1 = 64 * wave + opaque(lane);                      // stage 1: every dword of a row
                if (cidx1 < ROWF && !(WHATIF & 8)) colscan(ring1, cidx1, rb, BH * i, Sc);
                __builtin_amdgcn_s_setprio(0);
            } else if (wave < 5) {
                __builtin_amdgcn_s_setprio(3);
                const int cidx2 = OFF1 * (wave - 3) + HW + opaque(lane);         // stage 2: the new columns of either plane
                if (i >= 1 && !(WHATIF & 8)) colscan(ring2, cidx2, rbp, BH * (i - 1) - R, Sc);
                __builtin_amdgcn_s_setprio(0);
            }
            V4_STAMP(5);
            wg_barrier();
            V4_STAMP(6);
            V4_DUMP(2);
            // ------------------------------------ X(i) --------------------------------------------------
            seen = s_seen;
            V4_STAMP(8);
#ifdef SMX_V4_EXTRA_SALU
            {   // sensitivity experiment: SMX_V4_EXTRA_SALU dependent scalar adds per wave and iteration
                int dummy = i;
#pragma unroll
                for (int z = 0; z < SMX_V4_EXTRA_SALU; ++z) asm volatile("s_add_u32 %0, %0, 1" : "+s"(dummy));
                asm volatile("" :: "s"(dummy));
            }
#endif
#ifdef SMX_V4_EXTRA_VALU
            {
                int dummy = lane;
#pragma unroll
                for (int z = 0; z < SMX_V4_EXTRA_VALU; ++z) asm volatile("v_add_u32 %0, %0, 1" : "+v"(dummy));
                asm volatile("" :: "v"(dummy));
            }
#endif
            // Order of the phase: everything that consumes a value loaded in the previous iteration comes before
            // the first global access of this one.  (The compiler cannot count loads across the loop edge: the
            // first such use behind a new access waits for ALL outstanding accesses, the new one included.)
            // The left neighbour's record i+1 is needed first thing in the next iteration and comes from another
            // XCD's writes (1 - 2 us): its load goes out at the start of the phase, right behind a "use" of the
            // values loaded in the previous iteration (they arrived long ago: that wait is free).
#pragma unroll
            for (int t = 0; t < RPW; ++t) asm volatile("" : "+v"(gab[t]), "+v"(Iraw[t]));
            // The loads are issued in every iteration of every item: a load under a condition makes the compiler
            // merge old and new register values right behind it, i.e. wait for it on the spot.  What they return
            // counts only if the record had been published (have_pref).
            fetch_rec(min(i + 1, NI - 1));
            have_pref = pred && (seen == FLAG_DONE || seen >= (unsigned)i + 2u);
            {
                // box means of stage 1 -> a_k, b_k rows [BH i - R, BH i + BH - R)   (compute_ak_and_bk guidedFilter.cu:345-354)
                const int ya0 = BH * i - R + RPW * wave;
                f2 m[RPW];
                bool ok[RPW];
                if (WHATIF & (64 | 512)) {
#pragma unroll
                    for (int t = 0; t < RPW; ++t) m[t] = (f2){1.0f + lane, 2.0f};
                } else if ((unsigned)(i - f1_lo) < f1_n) {
                    box4_fast(ring1, ob1, m);
                } else {
                    box4_gen(ring1, 0, mkgeo(xs + opaque(lane), cs1), xint1, ya0, m, ok);
                }
#pragma unroll
                for (int t = 0; t < RPW; ++t) {
                    float mm = gab[t].x * m[t].x;
                    float ak = 1.0f * (m[t].y - mm) * gab[t].y;
                    float mb2 = 1.0f * gab[t].x * ak;
                    float bk = 1.0f * m[t].x - mb2;
                    abreg[t] = (f2){ak, bk};
                }
            }
            V4_STAMP(9);
            const int yq0 = BH * (i - 1) - 2 * R + RPW * wave;
#pragma unroll
            for (int t = 0; t < RPW; ++t) { qout[t] = 0.0f; qok[t] = false; }
            const bool fast2 = (unsigned)(i - f2_lo) < f2_n;
            if (i >= 1) {
                // box means of stage 2 -> q rows [BH (i-1) - 2R, BH i - 2R)
                f2 m[RPW];
                if (WHATIF & (64 | 1024)) {
#pragma unroll
                    for (int t = 0; t < RPW; ++t) m[t] = (f2){1.0f + lane, 2.0f};
                } else if (fast2) {
                    box4_fast(ring2, ob2, m);
#pragma unroll
                    for (int t = 0; t < RPW; ++t) qok[t] = true;
                } else {
                    box4_gen(ring2, R, mkgeo(xs - R + opaque(lane), cs2), xint2, yq0, m, qok);
                }
#pragma unroll
                for (int t = 0; t < RPW; ++t) {
                    const float Iv = (float)__builtin_bit_cast(fg_t, Iraw[t]).x;
                    float tq = m[t].x * Iv;            // compute_q guidedFilter.cu:363-369
                    qout[t] = tq + m[t].y;
                }
            }
            V4_STAMP(10);
            __builtin_amdgcn_sched_barrier(0);
            // ---- the global accesses of the phase: q rows, record i, the loads of iteration i+1 ----
            if (i >= 1) store_q(i, fast2 && !(WHATIF & (64 | 1024)));
            if (succ && (hu_halo || hu_carry)) {
                // record i: row carries of this iteration's row scans, last 2R+1 columns of the stage-2 integral
                f4 hov;
                if (hu_carry) {
                    hov = *(const f4*)(&cout[0][0] + 2 * (qt - NHALO_U));   // cout[0][0 .. BH), cout[1][0 .. BH) are contiguous
                } else {
                    const float* p = ring2 + ring_off(rbp, hu_lds + OW);
                    const f2 a = *(const f2*)p, b = *(const f2*)(p + OFF1);
                    hov = (f4){a.x, b.x, a.y, b.y};
                }
                st16_sc1(r_out, (unsigned)((i * REC_F2) * 8) + rec_voff, hov);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(WHATIF & 1)) issue_guid(i + 1);      // (rows are clamped into the image: harmless behind the last iteration)
            V4_STAMP(7);
            wg_barrier();
            rbp = rb;
            rb += BH;
            rb = rb >= RR ? rb - RR : rb;
            ob2 = ob1;
            ob1 += BH * ROWF;
            ob1 = ob1 >= RR * ROWF ? ob1 - RR * ROWF : ob1;
        }
        // the last record and the last q rows: drained, then published
        drain_vmem();
        wg_barrier();
        if (tid == 0) {
            if (succ) flag_store(myflag, FLAG_DONE);
            s_item = s_next;
        }
#ifdef SMX_V4_ITEMLOG
        if (tid == 0 && item < ITEMLOG_MAX) g_itemlog[3 * item + 1] = __builtin_amdgcn_s_memrealtime();
#endif
    }
}

// ---------------------------------------------------------------------------------------------
// WTA over the chunk's q planes [slice][h][w] (dispSelectOnGPU guidedFilter.cu:403-411 in packed-key
// form).  One lane per pixel; grid (ceil(n/256), nviews)
// ---------------------------------------------------------------------------------------------
struct WtaArgs {
    const float* q[2];
    int64_t* keys[2];
    const unsigned* gate;     // != NULL: the pass runs only if (*gate != 0) == gate_nonzero (which walker's q planes are valid)
    int gate_nonzero;
    int fresh;                // != 0: the keys hold nothing yet: start from the identity instead of loading them
};
__device__ __forceinline__ bool wta_gate_closed(const unsigned* gate, int gate_nonzero) {
    return gate && (int)(flag_load(const_cast<unsigned*>(gate)) != 0u) != gate_nonzero;
}

__global__ __launch_bounds__(256) void k_v4_wta(WtaArgs wa, size_t n, int count, int slice0) {
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= n || wta_gate_closed(wa.gate, wa.gate_nonzero)) return;
    const float* __restrict__ q = wa.q[blockIdx.y] + id;
    int64_t* keys = wa.keys[blockIdx.y];
    const int64_t key = wa.fresh ? KEY_IDENTITY : keys[id];
    WtaRun r;                           // (smx_common.h: the winner of this call's slices, packed once)
    int z = 0;
    for (; z + 8 <= count; z += 8) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = __builtin_nontemporal_load(&q[(size_t)(z + t) * n]);
#pragma unroll
        for (int t = 0; t < 8; ++t) r.step(v[t], (uint32_t)(slice0 + z + t));
    }
    for (; z < count; ++z) r.step(__builtin_nontemporal_load(&q[(size_t)z * n]), (uint32_t)(slice0 + z));
    const int64_t kk = r.key();
    keys[id] = kk < key ? kk : key;
}

// The same with two pixels per lane (8-byte loads).  Needs an even plane size n and 8-byte aligned planes.
// grid (ceil(n/512), nviews)
__global__ __launch_bounds__(256) void k_v4_wta2(WtaArgs wa, size_t n, int count, int slice0) {
    const size_t id = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (id >= n || wta_gate_closed(wa.gate, wa.gate_nonzero)) return;
    const float* __restrict__ q = wa.q[blockIdx.y] + id;
    int64_t* keys = wa.keys[blockIdx.y];
    const int64_t k0 = wa.fresh ? KEY_IDENTITY : keys[id], k1 = wa.fresh ? KEY_IDENTITY : keys[id + 1];
    WtaRun r0, r1;
    int z = 0;
    constexpr int U = 8;               // loads in flight per lane (4 / 16 / 24 measure the same)
    for (; z + U <= count; z += U) {
        f2 v[U];
#pragma unroll
        for (int t = 0; t < U; ++t) v[t] = __builtin_nontemporal_load((const f2*)&q[(size_t)(z + t) * n]);
#pragma unroll
        for (int t = 0; t < U; ++t) {
            r0.step(v[t].x, (uint32_t)(slice0 + z + t));
            r1.step(v[t].y, (uint32_t)(slice0 + z + t));
        }
    }
    for (; z < count; ++z) {
        const f2 v = __builtin_nontemporal_load((const f2*)&q[(size_t)z * n]);
        r0.step(v.x, (uint32_t)(slice0 + z));
        r1.step(v.y, (uint32_t)(slice0 + z));
    }
    const int64_t a = r0.key(), c = r1.key();
    keys[id] = a < k0 ? a : k0;
    keys[id + 1] = c < k1 ? c : k1;
}

}  // namespace v4

// =============================================================================================
// host orchestration
// =============================================================================================
static inline unsigned cdivu4(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

struct V4Layout {
    int K, NI;
    size_t fg;        // floats per image plane (half2 = 4 B per pixel)
    size_t plane;     // floats per w*h plane
    size_t sv_hand;   // floats of hand-off records per slice-view (2 parities x NI records)
};

static V4Layout v4_layout(int w, int h, int R) {
    V4Layout L;
    L.K = (w + R + v4::OW - 1) / v4::OW;
    L.NI = (h - 1 + 2 * R) / v4::BH + 2;     // the q rows of iteration i end at BH i - 2R
    L.fg = (size_t)(w + 2 * v4::PADX) * h;
    L.plane = (size_t)w * h;
    L.sv_hand = (size_t)2 * L.NI * v4::REC_F2 * 2;   // parity x records x float2
    return L;
}

void v4_geometry(int* ow, int* bh) { *ow = v4::OW; *bh = v4::BH; }

bool v4_supported(const smx_params* p) { return p->radius >= 0 && p->radius <= v4::RMAX; }

constexpr size_t V4_CTRL_BYTES = 256;   // ticket (zeroed with the flags before every launch)

static size_t v4_flag_bytes_k(int K, int nsv) { return align_up(V4_CTRL_BYTES + (size_t)nsv * K * sizeof(unsigned), 256); }
static size_t v4_flag_bytes(const V4Layout& L, int nsv) { return v4_flag_bytes_k(L.K, nsv); }

// bytes for ONE view with `nslices` slices in flight (q planes included)
size_t v4_workspace_bytes(int w, int h, int nslices) {
    V4Layout L = v4_layout(w, h, v4::RMAX);
    size_t b = 256;
    b += 2 * align_up(L.fg * 4, 256);                               // both image planes (single-view calls too)
    b += align_up(L.plane * 8, 256);                                // (mean_I, 1/(var+eps))
    b += 2 * align_up(L.plane * 4, 256);                            // guidance scratch: integrals of I, I*I
    const size_t qp5 = v5::q_plane_floats(w, h);                     // comb-ordered q plane of the comb walker
    b += (size_t)nslices * align_up((L.plane > qp5 ? L.plane : qp5) * 4, 256);   // q
    b += align_up((size_t)v5::strips(w) * v5::bands(h) * v5::CLP * 100, 256) + 512;   // comb-ordered guidance planes (80 + 20 B per lane and band)
    const size_t hand5 = v5::sv_hand_floats(h);                    // the comb walker's records (smx_agg_v5.hip)
    b += align_up((size_t)nslices * (L.sv_hand > hand5 ? L.sv_hand : hand5) * 4, 256);
    b += v4_flag_bytes(L, 2 * nslices);                             // control block (shared by both views; K of either walker <= L.K)
    return b + 16 * 256;
}

template <int SRC, bool FAST>
static int launch_walk4(const v4::Args& a, hipStream_t st) {
    int dev = 0, ncu = 256;
    SMX_HIP(hipGetDevice(&dev));
    SMX_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    int per_cu = v4::WG_PER_CU;                          // persistent: WG_PER_CU workgroups per CU
    static const int env_per_cu = env_int_once("SMX_V4_WG_PER_CU", 0);   // experiments: fewer workgroups per CU
    if (env_per_cu >= 1 && env_per_cu <= v4::WG_PER_CU) per_cu = env_per_cu;
    const int slots = per_cu * ncu;
    const int grid = a.nitems < slots ? a.nitems : slots;
    if (a.R == v4::RMAX)
        hipLaunchKernelGGL((v4::k_v4_walk<SRC, v4::RMAX, FAST>), dim3((unsigned)grid), dim3(v4::NT), 0, st, a);
    else
        hipLaunchKernelGGL((v4::k_v4_walk<SRC, -1, FAST>), dim3((unsigned)grid), dim3(v4::NT), 0, st, a);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

#ifdef SMX_V4_ITEMLOG
extern "C" __attribute__((visibility("default"))) int smx_debug_read_itemlog(unsigned long long* out, int n) {
    const int m = 3 * v4::ITEMLOG_MAX;
    SMX_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(v4::g_itemlog), sizeof(unsigned long long) * (n < m ? n : m)));
    return SMX_OK;
}
#endif
#ifdef SMX_V4_DUMP
extern "C" __attribute__((visibility("default"))) int smx_debug_read_dump(float* out, int n) {
    const int m = 2 * v4::RR * v4::ROWF;
    SMX_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(v4::g_dump), sizeof(float) * (n < m ? n : m)));
    return m;
}
#endif
#ifdef SMX_V4_STAMPS
extern "C" __attribute__((visibility("default"))) int smx_debug_read_stamps(unsigned long long* out, int n) {
    const int m = v4::NWAVE * v4::STAMP_SLOTS;
    SMX_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(v4::g_stamps), sizeof(unsigned long long) * (n < m ? n : m)));
    return m;
}
#endif

// status words of the last fused aggregation that used this workspace: [0] != 0: a hand-off wait timed out;
// [1] != 0: the comb walker met cost values outside its exactness argument and the queued ring walker redid the chunk
int v4_read_status(const void* d_ws, unsigned* out, int nwords) {
    const char* base = (const char*)align_up((size_t)d_ws, 256);
    SMX_HIP(hipMemcpy(out, base, sizeof(unsigned) * (size_t)nwords, hipMemcpyDeviceToHost));
    return SMX_OK;
}

// Bytes of the region the comb walker addresses through its one 32-bit-offset descriptor: both image planes, the
// guidance planes of the views and their comb-ordered copies (the carving order of aggregate_v4)
size_t v5_fix_bytes(int w, int h, int nviews) {
    const V4Layout L = v4_layout(w, h, v4::RMAX);
    const size_t permb = (size_t)v5::strips(w) * v5::bands(h) * v5::CLP;
    return 2 * align_up(L.fg * 4, 256) + (size_t)nviews * (align_up(L.plane * 8, 256) + align_up(permb * 5 * 16, 256) + align_up(permb * 20, 256));
}

// Aggregation + WTA of slices [s_begin, s_end) of `nviews` (1 or 2) views.  View v uses d_guide[v]
// as guidance; its cost slices are d_cost[v] (materialised, slice s at (s - s_begin)*w*h) or, when
// d_cost[v] == NULL, are built on the fly against d_guide[v ^ 1] (nviews == 2) / d_other[0].
int aggregate_v4(const smx_params* p, int nviews, const uint8_t* const* d_guide,
                 const uint8_t* const* d_other, const float* const* d_cost, int w, int h,
                 const int* dmin, int s_begin, int s_end, int64_t* const* d_keys,
                 uint8_t* const* d_mean_u8, float* const* d_agg, void* d_ws, size_t ws_bytes,
                 hipStream_t st, const AggOpts& opt, AggInfo* info) {
    const int R = p->radius;
    const bool fast = opt.fast;
    V4Layout L = v4_layout(w, h, R);
    const bool use_cost = d_cost && d_cost[0];
    // The comb walker (smx_agg_v5.hip) serves the hot case: radius 9, costs built from the images, exact mode.
    // opt.walker: 0 = choose, 4 = the ring walker of this file.  Both share this orchestration: image planes, guidance
    // statistics, chunking, WTA pass; only the strip / band geometry and the records differ.
    // The comb walker addresses both image planes, the guidance planes and their comb-ordered copies through ONE buffer
    // descriptor with 32-bit offsets, 0x80000000 marking "outside": the whole region must stay below 2 GiB
    // (v5_fix_bytes: the same terms the carving below uses).
    const bool v5_fits = v5_fix_bytes(w, h, nviews) < 0x80000000ull;
    // Materialised cost volumes (the reference's calling convention, guidedFilter.cu:198-200) run on the comb walker too
    // (round 5): its cost wave loads the costs and CHECKS them -- +0 or a normal number in [2^-60, 2^60] is what its exactness
    // argument covers.  A violation cannot come back to the host of an asynchronous call, so the ring walker, which takes
    // any input, is queued behind it with a device-side gate (`only_if`): it does nothing unless the comb walker raised the
    // second status word, and the two WTA passes are gated the other way round.  Cost: two empty launches and a memset.
    const bool use_v5 = opt.walker != 4 && v5_fits &&
                        (use_cost ? v5_supported_cost(p) && (size_t)w * h >= 4 : v5_supported(p));
    const bool fallback4 = use_v5 && use_cost;
    if (opt.walker == 5 && !use_v5)
        return fail(SMX_E_ARG, "aggregate_v4: the comb walker does not apply (radius 9, eps >= 1, default-like cost parameters where "
                               "the costs are built from the images, planes of %d x %d within its 2 GiB descriptor: %s)", w, h,
                    v5_fits ? "yes" : "no");
    if (info) { *info = AggInfo(); info->walker_used = use_v5 ? 5 : 4; }
    const V4Layout L4 = L;      // the ring walker's geometry (the queued fall-back uses it)
    if (use_v5) {
        L.K = v5::strips(w);
        L.NI = v5::bands(h);
        L.sv_hand = v5::sv_hand_floats(h);
    }
    // records and flags are shared by the two walkers of a call with a queued fall-back: the larger of each
    const size_t sv_hand_c = fallback4 && L4.sv_hand > L.sv_hand ? L4.sv_hand : L.sv_hand;
    const int K_c = fallback4 && L4.K > L.K ? L4.K : L.K;
    // every plane is addressed through 32-bit buffer offsets, with 0x80000000 as "outside the image"
    if ((size_t)h * ((size_t)w + 2 * v4::PADX) * 8 >= 0x80000000ull)
        return fail(SMX_E_ARG, "aggregate_v4: an image plane of %d x %d exceeds 2 GiB", w, h);
    if (use_cost && nviews == 2 && !d_cost[1])
        return fail(SMX_E_ARG, "aggregate_v4: both views need a cost volume or none");
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
    if (oom) return fail(SMX_E_WS, "aggregate_v4: workspace too small");
    // fixed part: image planes, guidance statistics, guidance scratch
    v4::fg_t* FG[2];
    float* gs[4];
    v4::f2* gpair[2];
    for (int i = 0; i < 2; ++i) FG[i] = (v4::fg_t*)carve(L.fg * 4);
    for (int v = 0; v < nviews; ++v) gpair[v] = (v4::f2*)carve(L.plane * 8);
    // comb-ordered guidance planes of the comb walker (smx_agg_v5.h): [K][h][CLP] per view
    v4::f2* g1p[2] = {nullptr, nullptr};
    unsigned* i2p[2] = {nullptr, nullptr};
    // (band-major, smx_agg_v5.h: 5 NI row pairs of 16 B per lane; per band 16 B + 4 B per lane)
    const size_t permb = (size_t)v5::strips(w) * v5::bands(h) * v5::CLP;
    if (use_v5)
        for (int v = 0; v < nviews; ++v) {
            g1p[v] = (v4::f2*)carve(permb * 5 * 16);
            i2p[v] = (unsigned*)carve(permb * 20);
        }
    const char* const fix_end = base;       // image planes + guidance planes: the comb walker addresses them through one descriptor
    if (use_v5 && (size_t)(fix_end - (const char*)FG[0]) >= 0x80000000ull)
        return fail(SMX_E_ARG, "aggregate_v4: comb walker planes exceed the 2 GiB descriptor (v5_fix_bytes out of step with the carving)");
    for (int i = 0; i < 2 * nviews; ++i) gs[i] = (float*)carve(L.plane * 4);
    const int total = s_end - s_begin;
    // per slice-view: q plane (unless the caller's volume is written directly) + records + flags
    const bool own_q = !(d_agg && d_agg[0]);
    // (the comb walker's own q planes are comb-ordered: K * OWS >= w columns per row)
    const size_t qplane = use_v5 && own_q ? v5::q_plane_floats(w, h) : L.plane;
    const size_t per_sv = (own_q ? align_up(qplane * 4, 256) : 0) + sv_hand_c * 4 +
                          (size_t)K_c * sizeof(unsigned);
    size_t fit = avail > 8 * 256 + V4_CTRL_BYTES ? (avail - 8 * 256 - V4_CTRL_BYTES) / (per_sv * nviews) : 0;
    if (oom || (fit < 1 && total > 0))
        return fail(SMX_E_WS, "aggregate_v4: workspace %zu B too small (need >= %zu B per view)",
                    ws_bytes, v4_workspace_bytes(w, h, 1));
    int chunk = fit > (size_t)total ? total : (int)fit;
    if (opt.max_chunk > 0 && chunk > opt.max_chunk) chunk = opt.max_chunk;
    if (chunk < 1) chunk = 1;
    if (info) info->chunk = chunk;
    const int nsv_max = chunk * nviews;
    float* qbuf[2] = {nullptr, nullptr};
    if (own_q)
        for (int v = 0; v < nviews; ++v) qbuf[v] = (float*)carve((size_t)chunk * align_up(qplane * 4, 256));
    v4::f2* hand = (v4::f2*)carve((size_t)nsv_max * sv_hand_c * 4);
    char* ctrl = (char*)carve(v4_flag_bytes_k(K_c, nsv_max));
    if (oom) return fail(SMX_E_WS, "aggregate_v4: workspace carve overflow");
    int nl = 0, rc;

    v4::PrepArgs pa;
    pa.I[0] = d_guide[0];
    pa.I[1] = nviews == 2 ? d_guide[1] : (d_other ? d_other[0] : nullptr);
    pa.FG[0] = FG[0]; pa.FG[1] = FG[1];
    const int nimg = pa.I[1] ? 2 : 1;
    pa.zero[0] = status; pa.nzero[0] = 64;
    pa.zero[1] = (unsigned*)ctrl; pa.nzero[1] = (unsigned)(v4_flag_bytes_k(K_c, nsv_max) / 4);
    // (no launch of its own: k_v4_guid_rows below does this kernel's work for its image rows)

    // ---- guidance statistics (guidedFilter.cu:58-123): (mean_I, 1/(var_I + eps)), optional u8 mean image
    {
        v4::GuidArgs ga;
        memset(&ga, 0, sizeof(ga));
        for (int v = 0; v < nviews; ++v) {
            ga.FG[v] = FG[v];
            ga.S[v][0] = gs[2 * v]; ga.S[v][1] = gs[2 * v + 1];
            ga.G[v] = gpair[v];
            ga.mean_u8[v] = d_mean_u8 ? d_mean_u8[v] : nullptr;
        }
        ga.prep = pa; ga.nimg = nimg; ga.nviews = nviews;
        {
            // rows per workgroup: few enough that the launch fills the chip, and whole rows fit the LDS
            const int wpad = v4::gr_wpad(w);
            int rows = (int)((size_t)(152 * 1024) / ((size_t)8 * wpad));
            const int fillrows = h * nviews / 256;
            rows = rows > fillrows ? fillrows : rows;
            rows = rows > v4::GR_MAXROWS ? v4::GR_MAXROWS : rows;
            rows = rows < 1 && (size_t)8 * wpad <= (size_t)(152 * 1024) ? 1 : rows;
            if (rows < 1) return fail(SMX_E_ARG, "aggregate_v4: image too wide for the guidance row scan");
            static LdsLimitOnce lim;     // (once per device: the sharded driver runs several devices from one process)
            SMX_HIP(lim.ensure((const void*)v4::k_v4_guid_rows, 160 * 1024));
            hipLaunchKernelGGL(v4::k_v4_guid_rows, dim3(cdivu4(h, rows), nimg > nviews ? nimg : nviews), dim3(v4::GR_NT), (size_t)8 * rows * wpad, st,
                               ga, w, h, rows);
        }
        hipLaunchKernelGGL(v4::k_v4_guid_cols, dim3(cdivu4(w, 64), 2, nviews), dim3(64 * v4::GC_NW), 0, st, ga, w, h);
        if (use_v5) {
            // (the comb walker's planes: the statistics are evaluated where the comb-ordered copy is written, and G / the u8
            // mean leave from there too -- one launch and one round trip of G less than finish + permute)
            const float* S0[2] = {ga.S[0][0], ga.S[1][0]};
            const float* S1[2] = {ga.S[0][1], ga.S[1][1]};
            if ((rc = v5_perm_launch(nviews, S0, S1, gpair, ga.mean_u8, FG, g1p, i2p, w, h, p->eps, st))) return rc;
        } else {
            hipLaunchKernelGGL(v4::k_v4_guid_finish, dim3(cdivu4(w, 256), h, nviews), dim3(256), 0, st, ga, w, h, R, p->eps);
        }
        SMX_HIP(hipGetLastError());
        nl += 3;
    }
    stage_mark(ST_GUIDANCE, st);

    v4::Args a0;
    memset(&a0, 0, sizeof(a0));
    a0.w = w; a0.h = h; a0.R = R; a0.K = L.K; a0.NI = L.NI;
    a0.cc = make_cost_const(p);
    for (int v = 0; v < nviews; ++v) {
        a0.v[v].FG1 = FG[v]; a0.v[v].FG2 = FG[v ^ 1];
        a0.v[v].guid = gpair[v];
    }
    for (int s0 = s_begin; s0 < s_end; s0 += chunk) {
        const int cnt = (s_end - s0) < chunk ? (s_end - s0) : chunk;
        v4::Args a = a0;
        v4::WtaArgs wa;
        for (int v = 0; v < 2; ++v) {
            const int vv = v < nviews ? v : 0;
            float* qv = own_q ? qbuf[vv] : d_agg[vv] + (size_t)(s0 - s_begin) * L.plane;   // (own planes: `qplane` floats apart)
            if (v < nviews) {
                a.v[v].q = qv;
                a.v[v].d0 = dmin[v] + s0;
                a.v[v].cost = use_cost ? d_cost[v] + (size_t)(s0 - s_begin) * L.plane : nullptr;
            }
            wa.q[v] = qv;
            wa.keys[v] = d_keys[vv];
        }
        a.nslices = cnt; a.nsv = cnt * nviews;
        a.nitems = a.nsv * L.K;
        a.hand = hand;
        a.ticket = (unsigned*)ctrl; a.status = status;
        a.flags = (unsigned*)(ctrl + V4_CTRL_BYTES);
        if (s0 != s_begin) SMX_HIP(hipMemsetAsync(ctrl, 0, v4_flag_bytes_k(K_c, a.nsv), st));   // (first chunk: cleared by k_v4_guid_rows)
        if (use_v5) {
            v5::Args b;
            memset(&b, 0, sizeof(b));
            b.fix = (const char*)FG[0];
            b.fix_bytes = (size_t)(fix_end - (const char*)FG[0]);
            for (int v = 0; v < 2; ++v) {
                const int vv = v < nviews ? v : 0;
                b.o_fg[v] = (unsigned)((const char*)FG[v] - b.fix);
                b.o_g1p[v] = (unsigned)((const char*)g1p[vv] - b.fix);
                b.o_i2p[v] = (unsigned)((const char*)i2p[vv] - b.fix);
                b.q[v] = a.v[vv].q;
                b.d0[v] = a.v[vv].d0;
            }
            b.w = w; b.h = h; b.K = L.K; b.NI = L.NI;
            static const int env_no_overlap = env_int_once("SMX_V5_NO_OVERLAP", 0);     // (A/B runs: a period of the whole item)
            b.P = env_no_overlap ? ((v5::bands(h) + 3) & ~1) : v5::period(h, L.K);
            b.nslices = a.nslices; b.nsv = a.nsv; b.nitems = a.nitems;
            b.hand = (float*)hand; b.flags = a.flags; b.ticket = a.ticket; b.status = a.status;
            b.src_cost = use_cost ? 1 : 0;
            b.cost_plane = L.plane;
            b.bad = status + 1;
            for (int v = 0; v < 2; ++v) b.cost[v] = a.v[v < nviews ? v : 0].cost;
            b.cc = a.cc;
            {
                const _Float16 hc = (_Float16)a.cc.th_color, hg = (_Float16)a.cc.th_grad;
                unsigned short uc, ug;
                memcpy(&uc, &hc, 2); memcpy(&ug, &hg, 2);
                b.th2 = (unsigned)uc | ((unsigned)ug << 16);
            }
            b.fast = fast ? 1 : 0;
            {
                // Role priorities (smx_agg_v5.hip PRIO_*).  Measured in rounds 4 and 5 (tools/prio_ab*.sh, profiles/r05_prio_*):
                // they are worth 2-7 % on every launch whose aggregated planes stay within a few GB -- KITTI geometry up to
                // 1 500 slices (5.7 GB of q), Motorcycle / 4K geometry with few slices, every aspect ratio at KITTI's volume --
                // and cost 0.4-4.8 % on 4K (34 GB of q per launch); Motorcycle (15 GB) came out at -3.8 %, +3.0 % and -1.7 % on
                // three boxes.  The counters of the losing case (profiles/r05_prio_pmc_motorcycle.txt): identical instruction
                // counts, vector-memory operations 32 % longer in flight, 30 % more cycles in s_waitcnt.  What in a large q
                // footprint does that is not established (address translation of 512 workgroups streaming into planes 27-33 MB
                // apart is the suspect); the rule is therefore stated in the variable the effect follows: the q bytes of the
                // launch.  SMX_V5_PRIO=0/1 (read once per process) overrides it for A/B runs.
                constexpr double PRIO_MAX_Q_BYTES = 6e9;
                const double q_bytes = (double)a.nsv * (double)qplane * 4.0;
                b.prio = q_bytes < PRIO_MAX_Q_BYTES ? 1 : 0;
                static const int env_prio = env_int_once("SMX_V5_PRIO", -1);     // (A/B runs)
                if (env_prio >= 0) b.prio = env_prio != 0;
            }
            b.qperm = own_q ? 1 : 0;
            b.q_plane = qplane;
            rc = v5_launch(b, st);
            if (!rc && fallback4) {
                // the queued ring walker (does nothing unless status[1] was raised): its own geometry, fresh tickets and flags
                SMX_HIP(hipMemsetAsync(ctrl, 0, v4_flag_bytes_k(K_c, a.nsv), st));
                v4::Args a4 = a;
                a4.K = L4.K; a4.NI = L4.NI;
                a4.nitems = a4.nsv * L4.K;
                a4.only_if = status + 1;
                rc = fast ? launch_walk4<v4::SRC_COST, true>(a4, st) : launch_walk4<v4::SRC_COST, false>(a4, st);
                nl += 2;
            }
        } else if (fast) rc = use_cost ? launch_walk4<v4::SRC_COST, true>(a, st) : launch_walk4<v4::SRC_IMG, true>(a, st);
        else rc = use_cost ? launch_walk4<v4::SRC_COST, false>(a, st) : launch_walk4<v4::SRC_IMG, false>(a, st);
        if (rc) return rc;
        if (info) ++info->walker_launches;
        stage_mark(ST_WALK, st);
        bool al8 = L.plane % 2 == 0;
        for (int v = 0; v < nviews; ++v) al8 = al8 && ((uintptr_t)wa.q[v] & 7) == 0;
        wa.gate = nullptr; wa.gate_nonzero = 0;
        // (opt.keys_fresh: the caller's keys hold nothing yet -- the first WTA pass of the call starts from the identity instead
        // of loading them, which saves the smx_dev_init_keys launch in front of the call; with the gated pair of passes of a
        // queued fall-back exactly one of the two runs, so both may take the flag)
        const bool fresh = opt.keys_fresh && s0 == s_begin;
        wa.fresh = fresh ? 1 : 0;
        if (use_v5 && own_q) {
            if ((rc = v5_wta_launch(nviews, wa.q, wa.keys, w, h, cnt, s0, fallback4 ? status + 1 : nullptr, fresh, st))) return rc;
            if (fallback4) {
                // ... and the WTA over the ring walker's planes ([slice][h][w] at the start of the same buffers), if it ran
                wa.gate = status + 1; wa.gate_nonzero = 1;
                if (al8) hipLaunchKernelGGL(v4::k_v4_wta2, dim3(cdivu4((int64_t)L.plane, 512), nviews), dim3(256), 0, st, wa, L.plane, cnt, s0);
                else hipLaunchKernelGGL(v4::k_v4_wta, dim3(cdivu4((int64_t)L.plane, 256), nviews), dim3(256), 0, st, wa, L.plane, cnt, s0);
                SMX_HIP(hipGetLastError());
                ++nl;
            }
        } else if (al8)
            hipLaunchKernelGGL(v4::k_v4_wta2, dim3(cdivu4((int64_t)L.plane, 512), nviews), dim3(256), 0, st, wa,
                               L.plane, cnt, s0);
        else
            hipLaunchKernelGGL(v4::k_v4_wta, dim3(cdivu4((int64_t)L.plane, 256), nviews), dim3(256), 0, st, wa,
                               L.plane, cnt, s0);
        SMX_HIP(hipGetLastError());
        stage_mark(ST_WTA, st);
        nl += s0 != s_begin ? 3 : 2;
    }
    if (info) info->launches = nl;
    return SMX_OK;
}

}  // namespace smx
