// smx_agg_v5.hip -- fused guided-filter aggregation for gfx950, comb form (radius 9, costs built on the fly):
// cost build -> integral (p, I*p) -> box -> a_k, b_k -> integral (a, b) -> box -> q in ONE kernel.
// Reference: guidedFilter.cu:171-238, costVolume.cu:163-190, integral.cu:78-131.
//
// What changed against smx_agg_v4.hip: the integral images no longer live in LDS rings.  A work item is a strip of
// OWS = 19 (L - 1) output columns of one slice-view, walked top -> bottom in bands of BH = 10 rows by one workgroup (two
// per CU).  Its 19 L integral-image columns are dealt to COMB lanes: lane i of a comb with residue r owns column
// base + 19 i + r, so
//   * the left box tap (2R+1 = 19 columns to the left) is always lane i-1 of the same comb: `wave_shr:1` (`row_shr:1` for
//     combs of 16) fused into the subtraction / addition, no LDS access, no bank conflict;
//   * the column prefix sum S[y][c] = S[y-1][c] + R[y][c] is a register of the lane, and the 19 rows of history the
//     top taps need are a 20-slot register ring of the lane (static slots: the band loop is unrolled over two
//     bands = one turn of the ring);
//   * only the ROW prefix sums go through LDS: a band tile of 10 rows x 19 L columns x 2 components per stage, scanned in
//     place by one wave (lane = (stage, row, component), four columns per LDS instruction, the reference's left -> right
//     order) and read once by the comb lanes.
//
// Comb length L = 9 (smx_agg_v5.h), the PIPELINED form: 512 threads = 3 comb waves of stage 1 (p, I p -> a_k,
// b_k), 3 comb waves of stage 2 (a, b -> q), one ROW-SCAN wave and one COST wave; seven combs of nine lanes per comb
// wave, 152 output columns per strip, 128 VGPRs, 79 KB of LDS: two workgroups per CU.  One workgroup barrier (ordering
// LDS only) per band; in slot sl of an item
//   cost wave      evaluates the stage-1 inputs of band sl+2 (packed-half arithmetic, exact: cost_trunc_h2) -> tile 1[(sl+2)%3]
//   scan wave      row prefix of stage 1 on band sl+1 (tile 1[(sl+1)%3]) and of stage 2 on the a/b band sl-1 (tile 2[(sl-1)%2])
//   stage-1 combs  S1 += R1 of band sl, box, division -> a_k, b_k of rows [10 sl - 9, 10 sl + 1) -> tile 2[sl%2];
//                  then the left neighbour's record of pass sl+1 (halo columns of stage 2, row carries of stage 1) -> LDS
//   stage-2 combs  the a/b band sl-2 out of tile 2[sl%2] into registers (first thing: the stage-1 waves wait for that
//                  through an LDS counter before their first store), hand-off record of pass sl-1 -> global (sc1);
//                  S2 += R2, box, division, q rows [10 sl - 38, 10 sl - 28) -> HBM
// (The earlier three-barrier forms with comb lengths 12 / 16 -- W / R / X phases, the row scans on wave 0 -- are history:
// tools/variants/smx_agg_v5_r04_L9_L12_L16.hip, not built into the library.)
// Hand-off, tickets and the bounded flag waits are those of smx_agg_v4.hip (strip-major tickets: the left neighbour of
// an item always holds an earlier ticket); every wave takes the flag value it acts on from an LDS word written in the
// slot before (s_peek), so that all eight waves agree on whether the slot has the extra barrier of a wait.
// Round 5: a workgroup's items are PIPELINED -- the front roles (cost wave, stage-1 scan lanes, stage 1) start the next
// ticket every P = period(h) slots while stage 2 finishes the last three slots of the item before (see the slot loop).
//
// Exactness (all checked bit for bit by the CPU model tools/v5_model.cpp against the oracle): virtual rows and
// columns outside the image hold -0, the exact additive identity, so the clamped window corners of
// computeBoxFilterOnGPU (guidedFilter.cu:305-318) fall out of the sums themselves (S[y >= h] = S[h-1],
// S[c >= w] = S[w-1]); a tap that does not exist (no row / column in front of the image) reads +0 from the
// zero-filled DPP source or the ring's initial state, and v - (+0), v + (+0) are exact because no sum of this
// path can be -0: every p is >= +0 (costVolume.cu:187) and neither a_k nor b_k can be -0 (x - y == -0 only for
// x == -0).  That argument needs costs built from the images, so this kernel serves SRC_IMG only; materialised
// cost volumes and other radii stay on smx_agg_v4.hip.
//
// Must be compiled with -ffp-contract=off.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "smx_agg_dev.h"
#include "smx_agg_v5.h"

namespace smx {
namespace v5 {
using namespace aggdev;

constexpr int R = 9, HW = 2 * R + 1;
constexpr int SWU = HW * L;             // integral-image columns per strip (171)
constexpr int SW = (SWU + 3) / 4 * 4;   // tile columns: whole quads (the columns behind SWU are never used)
static_assert(OWS == HW * (L - 1) && SWU == OWS + HW && CPW * L <= 64 && CPW * NS1 >= HW, "strip geometry");
constexpr int RD = 20;                  // ring slots (>= 2R+2; a multiple of BH: static slots)
static_assert(RD % BH == 0 && RD >= HW + 1, "ring");
static_assert(L == 9, "the shipped form: seven combs of nine lanes per comb wave");
constexpr int NX = 2;                               // a row-scan wave and a cost wave beside the comb waves
constexpr int NWAVE = 2 * NS1 + NX, NT = 64 * NWAVE;   // 512 threads
constexpr int WPE = 4;                              // waves per SIMD the register budget is set for (128 VGPRs)
constexpr int NT1 = 3, NT2 = 2;                     // tile buffers per stage
enum Role { ROLE_S1 = 0, ROLE_S2 = 1, ROLE_SCAN = 2, ROLE_COST = 3 };
// A tile row is component-planar: first components (p / a) at [0, 304), second ones (I p / b) at [P1, P1 + 304):
// the row scan moves four columns of one component per LDS instruction, a comb lane reads its cell's pair with
// one ds_read2st64_b32 (offset1 = P1 / 64).
constexpr int P1 = (SW + 63) / 64 * 64;                 // 320 / 256
constexpr int RS = (P1 + SW + 16 + 63) / 64 * 64 + 4;   // row stride in floats, 4 mod 64: the scan lanes of different rows spread over the banks (644 / 516)
constexpr int TILE_F = BH * RS;
static_assert(P1 % 64 == 0 && P1 >= SW && RS >= P1 + SW && RS % 4 == 0, "tile row");

// Hand-off record of one band, 16-byte units:
//   [0, 100)    unit t*10 + jp: stage-2 row prefix of columns 285 + 2 jp, 286 + 2 jp of a/b row t of band i-1 as
//               (c0[j], c1[j], c0[j+1], c1[j+1])  (column 304 is padding)
//   [100, 105)  unit 100 + tp: stage-1 running row sums behind tile column 284 of rows 2 tp, 2 tp + 1 of band i
constexpr int NHU = BH * 10, NCU = BH / 2;
static_assert(REC_U == NHU + NCU, "record layout");

// Diagnostic build only (-DSMX_V5_STAMPS=<item>): every wave of one work item records the shader clock at the
// start and end of its work in the W, R and X phases (the gaps are barrier waits); the product build has no stamp.
#ifdef SMX_V5_STAMPS
constexpr int STAMP_W = 12;     // 0..5: slot phases; 6..10: after each of the five row pairs (comb waves) / quarters of the scan / cost batches
constexpr int STAMP_SLOTS = STAMP_W * 48;
// (the stamps are those of ONE workgroup -- SMX_V5_STAMPS mod 512 -- over 48 global slots from STAMP_G0: its items overlap)
#ifndef SMX_V5_STAMP_G0
#define SMX_V5_STAMP_G0 30
#endif
constexpr int STAMP_G0 = SMX_V5_STAMP_G0;
__device__ unsigned long long g_stamps[10 * STAMP_SLOTS];
#define V5_STAMP(n)                                                                          \
    do {                                                                                     \
        if ((int)blockIdx.x == SMX_V5_STAMPS % 512 && lane == 0 && i >= STAMP_G0 && (i - STAMP_G0) * STAMP_W + (n) < STAMP_SLOTS) \
            g_stamps[wave * STAMP_SLOTS + (i - STAMP_G0) * STAMP_W + (n)] = __builtin_amdgcn_s_memtime();  \
    } while (0)
#else
#define V5_STAMP(n) ((void)0)
#endif
// Diagnostic build only (-DSMX_V5_WHATIF=<bits>): leaves parts of the work out (WRONG results) to see what the
// kernel time is sensitive to.  1: no row scans; 2: no cost evaluation; 4: no stage-1 comb rows; 8: no stage-2 comb
// rows; 16: no q stores; 32: no guidance loads; 64: no hand-off (every strip like strip 0); 128: no input loads;
// 256: row scans without their LDS writes; 512: row scans without the adds; 1024: row scans at normal priority;
// 2048: no record / flag stores (every strip like the last); 4096: no hand-in (every strip like the first); 8192: the stage-1
// waves do not wait for the stage-2 copy-out
#ifndef SMX_V5_WHATIF
#define SMX_V5_WHATIF 0
#endif
constexpr int WHATIF = SMX_V5_WHATIF;
// Diagnostic build only (-DSMX_V5_MARK): comments in the ISA around the interior-path regions tools/isa_budget.py counts
#ifdef SMX_V5_MARK
#define V5_MARK(name) asm volatile("; MARK " name)
#else
#define V5_MARK(name) ((void)0)
#endif
#ifndef SMX_V5_WMAP
#define SMX_V5_WMAP 0
#endif
constexpr int WMAP = SMX_V5_WMAP;
// Wave priorities of the roles while they work (the row-scan wave runs at 3, the stage-2 waves' copy-out too): a slot ends
// with its slowest wave, and the stage-2 comb waves have a quarter of a slot to spare -- measured: cost 2 / stage 1 1 /
// stage 2 0 is 4 % faster than all 0 on KITTI shape and 1-2 % on shapes up to 4 Mpix x 128 disparities, but 3-4 % SLOWER on
// Motorcycle and 4K, where no priority at all is best (Args::prio: the host switches the whole set by the size of the launch)
#ifndef SMX_V5_PRIO_COST
#define SMX_V5_PRIO_COST 2
#endif
#ifndef SMX_V5_PRIO_S1
#define SMX_V5_PRIO_S1 1
#endif
#ifndef SMX_V5_PRIO_S2HEAD
#define SMX_V5_PRIO_S2HEAD 3
#endif
#ifndef SMX_V5_PRIO_SCAN
#define SMX_V5_PRIO_SCAN 3
#endif
#ifndef SMX_V5_S2_KEEP
#define SMX_V5_S2_KEEP 12
#endif
#ifndef SMX_V5_TOUCH
#define SMX_V5_TOUCH 0      // (A/B: load-to-LDS touches of the next band's cost lines: slower, 0.84 against 0.78 ms per KITTI pair)
#endif
constexpr int S2_KEEP = SMX_V5_S2_KEEP;     // vector-memory operations a stage-2 wave issues behind its record store in an interior slot (ten q rows + two loads; five less with row-pair stores)
constexpr int PRIO_COST = SMX_V5_PRIO_COST, PRIO_S1 = SMX_V5_PRIO_S1, PRIO_S2HEAD = SMX_V5_PRIO_S2HEAD, PRIO_SCAN = SMX_V5_PRIO_SCAN;

#if (SMX_V5_WHATIF & 16384)
__device__ uint64_t g_masksink[1 << 16];
#endif
#ifdef SMX_V5_DUMP
// Diagnostic build only: tile 1 (current buffer), tile 2 and the comb registers of one item behind the barrier that
// ends phase SMX_V5_DUMP_PH (0 W, 1 R, 2 X) of iteration SMX_V5_DUMP_IT
__device__ float g_dump[2 * TILE_F + NT * 48];
#endif

// ---- DPP left taps fused into the arithmetic: lane i reads lane i-1 (`wave_shr:1`: the combs of nine lanes are not whole DPP
// rows, so there is no zero fill in front of a comb).  (The compiler keeps a separate v_mov_b32_dpp per tap; the fused forms
// halve the box.)  A VGPR written by the VALU instruction in front may not be read by DPP for two wait states and the
// compiler's hazard recogniser does not look into inline assembly: box_bottom, whose sources are the column sums of
// this very row, brings its own s_nop 1; box_top reads ring slots written 19 rows ago.
__device__ __forceinline__ f2 box_bottom(f2 s) {            // s - s[lane-1], both components
    f2 d;
    asm("s_nop 1\n\tv_subrev_f32_dpp %0, %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_subrev_f32_dpp %1, %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(d.x), "=&v"(d.y) : "v"(s.x), "v"(s.y));
    return d;
}
__device__ __forceinline__ f2 box_top(f2 u, f2 t) {         // u + t[lane-1], both components
    f2 d;
    asm("v_add_f32_dpp %0, %2, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_add_f32_dpp %1, %3, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(d.x), "=&v"(d.y) : "v"(t.x), "v"(t.y), "v"(u.x), "v"(u.y));
    return d;
}

// x / area for both components of a cell with ca = (RN(1/area), area) in ONE register pair: div_small_int2 with the
// operands broadcast by op_sel (low half = 1/area, high half = area) instead of two ready-made pairs
__device__ __forceinline__ f2 div_ca(f2 x, f2 ca) {
    f2 q, e, m;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(q) : "v"(x), "v"(ca));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(e) : "v"(q), "v"(ca), "v"(x));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(m) : "v"(e), "v"(ca), "v"(q));
    return m;
}

__device__ __forceinline__ f2 div_ca_q(f2 x, f2 ca) { f2 q; asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(q) : "v"(x), "v"(ca)); return q; }
__device__ __forceinline__ f2 div_ca_e(f2 q, f2 x, f2 ca) {
    f2 e;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(e) : "v"(q), "v"(ca), "v"(x));
    return e;
}
__device__ __forceinline__ f2 div_ca_m(f2 e, f2 q, f2 ca) { f2 m; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(m) : "v"(e), "v"(ca), "v"(q)); return m; }

// ---- matching cost of a cell in packed halves (cost wave of the pipelined form).  q = (pixel value, x-derivative) as two
// halves: pixel values are the integers 0 .. 255 and derivatives multiples of 0.5 in [-127.5, 127.5] (k_v4_prep), so the
// difference of two cells is EXACT in fp16 and min(|d|, threshold) -- thresholds exact in fp16: v5_supported -- is the very
// number the reference's f32 arithmetic gets (costVolume.cu:187).  Against the sentinel 60000 of a partner outside the image
// the difference rounds but stays finite and far above either threshold.  v_fma_mix_f32 with a +0 addend is the
// correctly rounded product of a half and a float (no product here is negative), so the two weighted terms, their sum
// and I * p round exactly where the reference's do.
__device__ __forceinline__ unsigned cost_trunc_h2(unsigned q1, unsigned q2, unsigned th2) {
    unsigned d;
    asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(q1), "v"(q2));
    d &= 0x7fff7fffu;
    asm("v_pk_min_f16 %0, %1, %2" : "=v"(d) : "v"(d), "s"(th2));
    return d;
}
__device__ __forceinline__ float mix_mul_lo(unsigned h2, float s) {   // RN((float)low half * s)
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "s"(s));
    return r;
}
__device__ __forceinline__ float mix_mul_hi(unsigned h2, float s) {   // RN((float)high half * s)
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "s"(s));
    return r;
}
__device__ __forceinline__ float mix_mul_lo_v(unsigned h2, float v) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(v));
    return r;
}

// a cell's (first, second) component of a tile row (the compiler forms ds_read2st64_b32 / ds_write2st64_b32:
// the planes are 5 x 64 dwords apart)
__device__ __forceinline__ f2 tile_rd(const float* p) { return (f2){p[0], p[P1]}; }
__device__ __forceinline__ void tile_wr(float* p, f2 v) { p[0] = v.x; p[P1] = v.y; }

// FASTK = 1: the FAST mode (smx_set_agg_path(4)): window means by multiplication with the rounded reciprocal of the area
// instead of the correctly rounded division, no exactness vote -- the same sums in the same order, at most an ulp or two away
// per mean; NOT bit-exact, never the default
// QP = 1: q goes to the library's own comb-ordered scratch (Args::qperm; the product default), 0: into the caller's [slice][h][w]
// volume -- a template parameter because a run-time choice between the two store forms inside the row pairs cost 1.4 % of a pair.
template <int FASTK, int QP>
__global__ __launch_bounds__(NT, WPE) void k_v5_walk(Args A) {
    constexpr bool QPERM = QP != 0;
    __shared__ __attribute__((aligned(16))) float tile1[NT1][TILE_F];
    __shared__ __attribute__((aligned(16))) float tile2s[NT2][TILE_F];
    __shared__ float cin1s[2][BH][2];                               // stage-1 row carries of the band the next scan pass takes (by pass parity)
    __shared__ float rcp_s[RCP_N];                                  // RN(1/area)
    __shared__ int s_queue[4];                                      // tickets of the workgroup's items, by item number & 3 (written by the cost wave a few slots ahead)
    __shared__ unsigned s_peekn;                                    // the NEXT item's predecessor flag as peeked at in the last slot of the front item
    __shared__ unsigned s_seen;                                     // last value read from the left neighbour's flag
    __shared__ unsigned s_peek[2];                                  // the flag as peeked at during slot sl -> [(sl + 1) & 1], read by every wave at the top of slot sl + 1
    __shared__ unsigned touch_sink[64];                             // where the cost wave's prefetch touches of a cost volume land (never read)
    __shared__ unsigned s_x1;                                       // stage-2 waves that have taken their rows out of tile 2 (counts up through an item)

    const int lane = threadIdx.x & 63;
    const int hwave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // role of a hardware wave (waves w and w + 4 of a workgroup share a SIMD)
    const int wave = WMAP == 1 ? (int)((0x75436210u >> (4 * hwave)) & 7u) : WMAP == 2 ? (int)((0x54317620u >> (4 * hwave)) & 7u)
                     : WMAP == 3 ? (int)((0x53764210u >> (4 * hwave)) & 7u) : WMAP == 4 ? (int)((0x57436210u >> (4 * hwave)) & 7u) : hwave;
    const int tid = 64 * wave + lane;
    const int w = A.w, h = A.h, K = A.K, NI = A.NI, nsv = A.nsv;
    const CostConst cc = A.cc;
    const f2 NZ2 = {-0.0f, -0.0f};
    if (tid < RCP_N) rcp_s[tid] = kRcp.v[tid];

    // ---- comb geometry of this thread: stage, DPP row = residue, position in the comb, tile column ----------
    const bool st2w = wave >= NS1 && wave < 2 * NS1;
    // (per-lane values that only one phase of an iteration needs are re-derived from the thread index where they
    // are used -- a handful of integer instructions per band -- instead of living in VGPRs through the comb rows)
    // comb of this lane = residue rho (>= 19: the lane idles), position il in the comb
    auto comb_il = [&]() { return opaque(lane) % L; };
    auto comb_rho = [&]() {
        const int l = opaque(lane), c = l / L;
        return c < CPW ? CPW * (wave - (st2w ? NS1 : 0)) + c : HW;
    };
    auto comb_jt = [&]() {
        const int rho = comb_rho(), il = comb_il();
        return rho < HW ? HW * il + rho : 0;                        // (idle lanes run along on column 0)
    };
    // stage-1 inputs (cost evaluation): done by the stage-1 waves that do not scan (waves 1 .. NS1-1), in the shadow of
    // the row scans; NRQ rounds of one quad (four tile columns of one row) per thread, the last round almost full
    constexpr int NQROW = SW / 4;                                   // quads per tile row (76 / 57)
    constexpr int NCT = 64;                                         // cost threads: the cost wave
    constexpr int NRB = 4;                                          // rounds whose loads are in flight together
#ifndef SMX_V5_NRBC
#define SMX_V5_NRBC 4
#endif
    constexpr int NRBC = SMX_V5_NRBC;                               // ... with materialised cost volumes (their quads come from HBM)
    static_assert(2 * NRBC >= (BH * (SW / 4) + 63) / 64, "two batches");
    constexpr int NCT2 = 0;
    constexpr int NQT = BH * NQROW;                                 // quads per band
    constexpr int NRQ = (NQT - NCT2 + NCT - 1) / NCT;               // rounds

    // ---- items pipelined across the workgroup's tickets (round 5) ---------------------------------------------------------
    // An item needs the cost wave in its local slots -2 .. s1_last - 2, stage 1 in -2 .. s1_last, stage 2 in 1 .. q_last: every
    // role idles for several of an item's NI + 2 slots.  The workgroup therefore starts a new item every P = A.P slots
    // (max(s1_last + 3, q_last): NI - 1 or NI): the FRONT roles (cost wave, stage-1 scan lanes, stage 1) take the next ticket
    // while stage 2 (and, in the first overlap slot, the stage-2 scan lanes) finish the last q_last - P + 3 slots of the BACK
    // item.  One global slot counter g per wave; the front item's local slot is slf = g - P n - 2.  Tile buffers are indexed by
    // GLOBAL band (local band + P n = `boff`), so the front item's band b and the back item's band b + P share a buffer exactly
    // as two bands of one item do, and the same protocols (slot barrier, x1 counter) order their uses.
    // (Short or wide-and-short images get a period of the whole item, NI + 2: smx_agg_v5.h period() and DESIGN.md 4.1 -- the
    // argument why no cycle of flag waits exists; tools/v5_protocol_sim.py is this protocol as a CPU model.)
    if (tid == 0) {
        s_queue[0] = (int)__hip_atomic_fetch_add((gu32*)A.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_x1 = 0u; s_peek[0] = 0u; s_peek[1] = 0u; s_peekn = 0u;
    }
    const int P = A.P, q_last = (h + 37) / 10;                     // period in slots; last local slot with q rows inside the image
    // Buffer descriptors: ONE over the fixed part of the workspace (both image planes, the guidance plane: the
    // plane is chosen by a scalar offset), one over the hand-off records, one (per item) over the slice's q plane.
    const unsigned fgw4 = ((unsigned)w + 2u * PADX) * 4u, w4 = (unsigned)w * 4u;
    const rsrc_t r_fix = mk_rsrc(A.fix, A.fix_bytes);
    const unsigned recb = (unsigned)(NI + 2) * REC_U * 16u;  // bytes per (parity, slice-view)  (smx_agg_v5.h records())
    const rsrc_t r_hand = mk_rsrc(A.hand, (size_t)2 * nsv * recb);
    {
        auto item_body = [&](auto ROLEc) {
        constexpr int ROLE = decltype(ROLEc)::value;
        constexpr bool ST2 = ROLE == ROLE_S2, COMB = ROLE <= ROLE_S2;
        // ---- the item this role works on (set by `take`) ----------------------------------------------------------------
        int item = 0, k = 0, sv = 0, view = 0, slice = 0, d = 0;
        int base1 = 0;          // image column of tile-1 column 0
        int base2 = 0;          // image column of tile-2 column 0 (= a/b column of the same comb lane)
        bool pred = false, succ = false;
        bool xedge = false;     // the strip has columns outside the image (virtual: -0)
        unsigned* myflag = A.flags;
        int o_fg1 = 0, o_fg2 = 0, o_g1p = 0, o_i2p = 0, o_i2b = 0, o_in = 0, o_out = 0;
        rsrc_t r_q = r_fix;
#if (SMX_V5_WHATIF & 16384)
        // (what-if 16384: the cost of a pairwise pre-reduction of q in stage 2 -- odd slices load the q rows of the slice before,
        // select, store the minimum and a ballot; results wrong by construction)
        [[maybe_unused]] float qo[(ST2 && (WHATIF & 16384)) ? BH : 1];
        [[maybe_unused]] rsrc_t r_qp = r_fix;
        [[maybe_unused]] bool pairB = false;
#endif
        // q rows: comb-ordered scratch (row y of this strip at (k h + y) * OWS) or the caller's [h][w]
        int q_pitch = 0, q_row0 = 0;
        auto decode = [&](int it) {
            item = it;
            k = it / nsv;
            sv = it - k * nsv;
            view = sv / A.nslices;
            slice = sv - view * A.nslices;
            base1 = OWS * k - 1;
            base2 = OWS * k - R - 1;
            pred = k > 0 && !(WHATIF & (64 | 4096)); succ = k + 1 < K && !(WHATIF & (64 | 2048));
            d = A.d0[view] + slice;
            myflag = A.flags + (size_t)sv * K + k;
            o_fg1 = (int)A.o_fg[view]; o_fg2 = (int)A.o_fg[view ^ 1];
            // this strip's rows of the comb-ordered guidance planes
            // (band-major: 5 NI row pairs of 16 B per lane for stage 1; per band 16 B + 4 B per lane for stage 2 -- smx_agg_v5.h)
            o_g1p = (int)(A.o_g1p[view] + (unsigned)k * (unsigned)(5 * NI) * (CLP * 16u));
            o_i2p = (int)(A.o_i2p[view] + (unsigned)k * (unsigned)NI * (CLP * 16u));
            o_i2b = (int)(A.o_i2p[view] + (unsigned)K * (unsigned)NI * (CLP * 16u) + (unsigned)k * (unsigned)NI * (CLP * 4u));
            o_in = (int)((((unsigned)(k - 1) & 1u) * (unsigned)nsv + (unsigned)sv) * recb);
            o_out = (int)((((unsigned)k & 1u) * (unsigned)nsv + (unsigned)sv) * recb);
            r_q = mk_rsrc(A.q[view] + (size_t)slice * A.q_plane, A.q_plane * 4);
#if (SMX_V5_WHATIF & 16384)
            pairB = (slice & 1) != 0;
            r_qp = mk_rsrc(A.q[view] + (size_t)(slice > 0 ? slice - 1 : 0) * A.q_plane, A.q_plane * 4);
#endif
            // (qperm: q_pitch is the pitch of a row PAIR, the rows of a pair lie side by side -- smx_agg_v5.h)
            q_pitch = QPERM ? OWS * 8 : (int)w4; q_row0 = QPERM ? k * ((h + 1) / 2) * (OWS * 8) : 0;
            xedge = base1 < 0 || base1 + SW > w;
        };
        // the tile-2 buffer the comb rows of stage 1 write / the next scan pass takes for stage 2 and the hand-in fills,
        // and the stage-1 carries of that pass (they alternate from band to band)
        float* tile2 = tile2s[0];
        float (*cin1)[2] = cin1s[0];
        // ---- per-lane constants of the item that every comb row needs ------------------------------------------
        const int jt = comb_jt();                                   // tile column
        int xw = 1;                                                 // window width of this lane's output column
        float rcp_i = 1.0f;                                         // 1 / (19 xw): interior rows
        unsigned vo = OOB;                                          // stage 2: byte offset of the q column in a q row (stage 1: unused)
        unsigned vg = 0;                                            // byte offset of this lane in a row of the comb-ordered guidance plane
        bool col_ok = false;
        f2 ca_i = {1.0f, 1.0f};                                     // interior rows: (1/area, area)
        uint64_t okmask = 0;                                        // lanes with an output
        const bool il0 = comb_il() == 0;
        // stage 1: tile-2 column this lane's a_k, b_k go to -- its own tile column, or a padding column of the row when
        // that column is the left neighbour's halo (lane 0 of a comb of a strip with a neighbour) or the lane idles
        int jw = 0;
        static_assert(SW + 16 <= P1 && P1 + SW + 16 <= RS, "padding columns behind both planes of a tile row");
        auto lane_consts = [&]() {
            const int il = comb_il();
            const int xo = (ST2 ? base2 - R : base2) + jt;          // a/b column (stage 1) / q column (stage 2)
            col_ok = comb_rho() < HW && xo >= 0 && xo < w && (!ST2 || il >= 1);
            xw = col_ok ? min(w - 1, xo + R) - max(-1, xo - R - 1) : 1;
            rcp_i = rcp_s[HW * xw];
            {
                const int cl = 64 * (wave - (st2w ? NS1 : 0)) + lane;          // slot in a row of the comb-ordered planes
                vg = (unsigned)cl * (ST2 ? 4u : 16u);
                // comb-ordered scratch: 8 rho + (i - 1): the eight outputs of a comb are 32 contiguous bytes.  The LAST strip
                // is stored in column order instead (19 (i - 1) + rho = the local column), so that what lies outside the image is
                // one contiguous tail of its rows that the WTA pass never reads.
                vo = !col_ok ? OOB : (QPERM ? (unsigned)(k + 1 < K ? (L - 1) * comb_rho() + il - 1 : HW * (il - 1) + comb_rho()) * 8u : (unsigned)xo * 4u);
            }
            ca_i = (f2){rcp_i, (float)(HW * xw)};
            okmask = __builtin_amdgcn_ballot_w64(col_ok);
            jw = (comb_rho() < HW && !(pred && comb_il() == 0)) ? jt : SW + (lane & 15);
        };

        // ---- register state ------------------------------------------------------------------------------------
        f2 ring[COMB ? RD : 1];              // ring[y mod RD] = S[y] of this lane's column; the slot of row y-1 is the running sum
        auto ring_reset = [&]() {
#pragma unroll
            for (int s = 0; s < (COMB ? RD : 1); ++s) ring[s] = (f2){0.0f, 0.0f};
            if constexpr (COMB) ring[ST2 ? 10 : RD - 1] = NZ2;   // the slot in front of the first row (stage 1: row 0; stage 2: row -9)
        };
        ring_reset();
        // (arrays of the other role shrink to one element: the two roles are separate instantiations, so that no
        // register carries state of the other role around the band loop)
        // guidance of the band's output rows: loaded at the end of R (a barrier and the X1 phase ahead of the rows that
        // use it), consumed row by row in X2
        // stage 1: (mean_I, 1/(var+eps)) of two a/b rows per entry -- a ring of TWO row pairs: the pair P = 5 sl + j of the band grid
        // lives in gr[P & 1] = gr[(sl + j) & 1] and is loaded while pair P - 2 is computed, i.e. 2 x ~450 cycles ahead of its use
        // (round 4 loaded the five pairs of a band in a burst at the end of the slot before -- 20 registers through the rows,
        // and the burst was waited for in front of the slot barrier)
        f4 gr[ROLE == ROLE_S1 ? 2 : 1];
        for (int e = 0; e < (ROLE == ROLE_S1 ? 2 : 1); ++e) gr[e] = (f4){0, 0, 0, 0};
        unsigned gI[ST2 ? BH / 2 : 1];       // stage 2: guidance image values of two q rows each (fp16 pairs)
        f2 r2[ST2 ? BH : 1];                 // stage 2: the band's (R2 a, R2 b) rows, taken out of tile 2 in X(i), used in R(i+1)
#pragma unroll
        for (int t = 0; t < (ST2 ? BH : 1); ++t) r2[t] = NZ2;
        f4 hreg = {0, 0, 0, 0};              // (stage-1 role, threads 0 .. REC_U-1) this thread's unit of the left neighbour's next record
        bool have_pref = false;
        unsigned seen = 0;

        // stage-1 input units of this thread (waves 1..9): round A one quad, round B one pair; tile offsets and the
        // byte offsets in the two image planes without the band term
        // tile (row, column) of this thread's quad of round r; re-derived from the thread index where it is used (a few
        // integer instructions) instead of living in registers through the comb rows
        auto cost_unit = [&](int r, int& row, int& col, bool& on) {
            const int u = NCT2 + r * NCT + opaque(lane);
            on = u < NQT;
            const int uc = min(u, NQT - 1);
            row = uc / NQROW;
            col = (uc - row * NQROW) * 4;
        };
        // Loads of the stage-1 inputs of band ib (rows clamped into the image: every load is issued) and their evaluation
        // raw -> (p, I p) -> tile 1 buffer `dst`; cells outside the image are -0.  One phase: no register carries the raw
        // values on.  The loads of NRB rounds are in flight together.
        // Pipelined form: the cost wave evaluates the whole band, NRQ quads per lane.  Per quad, fixed for the item: byte
        // offsets of its four cells in the two image planes (row term included; the band term is the scalar offset of the
        // load), byte offset in a tile, the row term alone and which of the four columns lie in the image (edge items).
        constexpr int CWN = ROLE == ROLE_COST ? NRQ : 1;
        unsigned cw_a1[CWN], cw_a2[CWN], cw_t[CWN], cw_rowb[CWN], cw_m[CWN];
        for (int r = 0; r < CWN; ++r) { cw_a1[r] = 0u; cw_a2[r] = 0u; cw_t[r] = 0u; cw_rowb[r] = 0u; cw_m[r] = 0u; }
        auto cost_consts = [&]() {
          if constexpr (ROLE == ROLE_COST) {
#pragma unroll
            for (int r = 0; r < NRQ; ++r) {
                int row, col; bool on;
                cost_unit(r, row, col, on);
                const unsigned rowb = (unsigned)row * fgw4;
                cw_rowb[r] = (unsigned)opaque((int)rowb);       // (opaque: computed here, once per item, not where the band loop uses them)
                cw_a1[r] = (unsigned)opaque((int)((unsigned)(min(max(base1 + col, -PADX), w) + PADX) * 4u + rowb));
                // the partner cell in the other view's plane -- or, for a materialised cost volume (A.src_cost), the byte offset
                // of the quad in a slice plane of the volume, band term excluded (can be -4: column -1 of row 0 in strip 0)
                cw_a2[r] = A.src_cost ? (unsigned)opaque((row * w + base1 + col) * 4)
                                      : (unsigned)opaque((int)((unsigned)(min(max(base1 + col + d, -PADX), w) + PADX) * 4u + rowb));
                cw_t[r] = (unsigned)opaque((int)(on ? (unsigned)(row * RS + col) * 4u : (unsigned)SW * 4u));      // (lanes without a quad: a padding column)
                unsigned m = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) m |= (base1 + col + j >= 0 && base1 + col + j < w) ? 1u << j : 0u;
                cw_m[r] = (unsigned)opaque((int)(m | ((unsigned)row << 8)));     // (bits 8..: the tile row of the quad)
            }
          }
        };
        // ---- materialised cost volumes (the reference's calling convention, guidedFilter.cu:198-200: slice s of view v at
        // cost[v] + s * w * h): the cost wave LOADS the four costs of a quad instead of evaluating them; I p as before.
        // One descriptor per item over the slice plane.  A quad starts at column OWS k - 1 + 4 j: unaligned 16-byte loads (4-byte
        // aligned).  Edge items clamp the quad into the plane and shift the lanes of the vector back (cost_lin).
        // What the exactness argument at the top of this file needs from the values -- +0 or a normal number in [2^-60, 2^60],
        // so that no window sum is negative, -0, tiny or non-finite -- is CHECKED here on every cost inside the image: the running
        // unsigned maximum of the bit patterns and minimum of (pattern - 1) per lane, compared once per item; a violation raises
        // the second status word, and the host has queued the ring walker behind this kernel to redo the chunk if it is set.
        rsrc_t r_c = r_fix;
        unsigned c_max = 0u, c_min1 = 0xffffffffu;
        const unsigned plane4 = (unsigned)(A.cost_plane * 4);
        // (the cost wave's verdict on the item it leaves; then the descriptor of the new item's slice)
        auto cost_verdict = [&]() {
            if constexpr (ROLE == ROLE_COST) {
                // a cost of the volume outside {+0} U [2^-60, 2^60] (negative, -0, denormal, tiny, huge, infinite, NaN): this
                // kernel's results for the chunk do not count (smx_agg_v4.hip has queued the ring walker behind it)
                if (A.src_cost && __builtin_amdgcn_ballot_w64(c_max > 0x5d800000u || c_min1 < 0x21800000u - 1u) != 0 && lane == 0)
                    flag_store(A.bad, 1u);
                c_max = 0u; c_min1 = 0xffffffffu;
            }
        };
        auto cost_rsrc = [&]() {
            if constexpr (ROLE == ROLE_COST) {
                if (A.src_cost) r_c = mk_rsrc(A.cost[view] + (size_t)slice * A.cost_plane, A.cost_plane * 4);
            }
        };
        // edge items: byte offset of quad r of band ib clamped into the slice plane, and by how many elements it was moved
        auto cost_lin = [&](int ib, int r, int& linc4, int& sh) {
            const int row = (int)(cw_m[r] >> 8);
            const int e = max(0, BH * ib + row - (h - 1));                  // rows below the image read the last image row
            const int lin4 = (int)cw_a2[r] + (BH * ib - e) * (int)w4;
            linc4 = min(max(lin4, 0), (int)plane4 - 16);
            sh = (lin4 - linc4) >> 2;
        };
        // (loading a band ahead of its slot -- 56 registers through the slot -- spills and is slower: measured)
        u4 cw_ra[CWN], cw_rb[CWN];
        auto cw_issue = [&](int ib, auto R0c, auto EDGEc, auto SRCc) {
            constexpr int NB = decltype(SRCc)::value ? NRBC : NRB;   // (cost volumes: larger batches, see eval_band_p)
            constexpr int R0 = decltype(R0c)::value, R1 = R0 + NB < CWN ? R0 + NB : CWN;
            constexpr bool EDGE = decltype(EDGEc)::value, SRCC = decltype(SRCc)::value;
            const unsigned bandb = (unsigned)(BH * ib) * fgw4, ymaxb = (unsigned)(h - 1) * fgw4;
            if (BH * ib + BH <= h) {
#pragma unroll
                for (int r = R0; r < R1; ++r) {
                    cw_ra[r] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(r_fix, (int)cw_a1[r], o_fg1 + (int)bandb, 0));
                    if constexpr (!SRCC)
                        cw_rb[r] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(r_fix, (int)cw_a2[r], o_fg2 + (int)bandb, 0));
                }
            } else {
                // rows behind the image read the last image row (every load is issued; their cells become -0 in cw_finish)
#pragma unroll
                for (int r = R0; r < R1; ++r) {
                    const int yadj = min((int)bandb, (int)ymaxb - (int)cw_rowb[r]);
                    cw_ra[r] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(r_fix, (int)cw_a1[r] + yadj, o_fg1, 0));
                    if constexpr (!SRCC)
                        cw_rb[r] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(r_fix, (int)cw_a2[r] + yadj, o_fg2, 0));
                }
            }
            if constexpr (SRCC) {
#pragma unroll
                for (int r = R0; r < R1; ++r) {
                    if constexpr (EDGE) {
                        int linc4, sh;
                        cost_lin(ib, r, linc4, sh);
                        cw_rb[r] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(r_c, linc4, 0, 0));
                    } else {
                        cw_rb[r] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(r_c, (int)cw_a2[r], BH * ib * (int)w4, 0));
                    }
                }
            }
        };
        // materialised cost volumes: the cost quads of a band come from HBM (the image planes from L2), and they are consumed in the
        // slot they are issued in -- their latency sat on the cost wave's path twice per slot (0.78 ms per KITTI pair against 0.63
        // from the images).  Holding a band of them in registers across the slot spills (28 more live registers: 0.96 ms).
        // Instead the cost wave TOUCHES the next band's lines a slot ahead -- one dword per 128 bytes of its ten row segments,
        // 70 lanes of two load-to-LDS instructions, no register, nothing waits for them -- so that the real loads hit in L2.
        auto cw_touch_p = [&](int ib) {
            if constexpr (ROLE == ROLE_COST) {
                constexpr int SEGS = (SW * 4 + 124 + 127) / 128;            // 128-byte lines a row segment of SW costs can touch (7)
                static_assert(BH * SEGS <= 128, "two touch instructions per band");
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int l = min(opaque(lane) + 64 * half, BH * SEGS - 1);
                    const int row = l / SEGS, seg = l - row * SEGS;
                    const int y = min(BH * ib + row, h - 1);
                    const int off = min(max((y * w + base1) * 4 + seg * 128, 0), (int)plane4 - 4);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r_c, (__attribute__((address_space(3))) void*)touch_sink, 4, off, 0, 0, 0);
                }
            }
        };
        auto cw_finish = [&](int ib, float* dst, auto R0c, auto EDGEc, auto SRCc) {
            constexpr bool EDGE = decltype(EDGEc)::value, SRCC = decltype(SRCc)::value;
            constexpr int NB = SRCC ? NRBC : NRB;
            constexpr int R0 = decltype(R0c)::value, R1 = R0 + NB < CWN ? R0 + NB : CWN;
            const unsigned bandb = (unsigned)(BH * ib) * fgw4, ymaxb = (unsigned)(h - 1) * fgw4;
            char* const dstb = (char*)dst;
#pragma unroll
            for (int r = R0; r < R1; ++r) {
                const unsigned q1[4] = {cw_ra[r].x, cw_ra[r].y, cw_ra[r].z, cw_ra[r].w};
                unsigned q2[4] = {cw_rb[r].x, cw_rb[r].y, cw_rb[r].z, cw_rb[r].w};
                f4 pp, ip;
                if constexpr (!SRCC) {
                    f4 t1, t2;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned m = cost_trunc_h2(q1[j], q2[j], A.th2);
                        t1[j] = mix_mul_lo(m, cc.oma);
                        t2[j] = mix_mul_hi(m, cc.alpha);
                    }
                    pp = t1 + t2;
                } else {
                    if constexpr (EDGE) {
                        // the quad was loaded `sh` elements away from where it belongs (clamped into the plane): element j of the
                        // quad is element j + sh of the vector; positions that fall off are outside the image (masked below)
                        int linc4, sh;
                        cost_lin(ib, r, linc4, sh);
                        if (sh == -1) { q2[3] = q2[2]; q2[2] = q2[1]; q2[1] = q2[0]; }
                        else if (sh == 1) { q2[0] = q2[1]; q2[1] = q2[2]; q2[2] = q2[3]; }
                        else if (sh == 2) { q2[0] = q2[2]; q2[1] = q2[3]; }
                        else if (sh == 3) { q2[0] = q2[3]; }
                        const bool rowok = bandb + cw_rowb[r] <= ymaxb;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (!(rowok && (((cw_m[r] & 0xffu) >> j) & 1u) != 0u)) q2[j] = 0u;   // (outside the image: not checked, replaced by -0 below)
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        c_max = max(c_max, q2[j]);
                        c_min1 = min(c_min1, q2[j] - 1u);
                        pp[j] = __builtin_bit_cast(float, q2[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) ip[j] = mix_mul_lo_v(q1[j], pp[j]);     // I p (guidedFilter.cu:200 pixelMultOnGPU)
                if constexpr (EDGE) {
                    const bool rowok = bandb + cw_rowb[r] <= ymaxb;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (!(rowok && (((cw_m[r] & 0xffu) >> j) & 1u) != 0u)) { pp[j] = -0.0f; ip[j] = -0.0f; }
                }
                *(f4*)(dstb + cw_t[r]) = pp;
                *(f4*)(dstb + cw_t[r] + P1 * 4) = ip;
            }
        };
        auto eval_band_p = [&](int ib, float* dst) {
            static_assert(2 * NRB >= CWN, "two batches");
            auto run = [&](auto EDGEc, auto SRCc) {
                constexpr int NB = decltype(SRCc)::value ? NRBC : NRB;
                cw_issue(ib, std::integral_constant<int, 0>{}, EDGEc, SRCc);
                cw_finish(ib, dst, std::integral_constant<int, 0>{}, EDGEc, SRCc);
                if constexpr (NB < CWN) {
                    cw_issue(ib, std::integral_constant<int, NB>{}, EDGEc, SRCc);
                    cw_finish(ib, dst, std::integral_constant<int, NB>{}, EDGEc, SRCc);
                }
            };
            // (versions of the whole band: the interior one has no trace of the edge handling, the cost-volume one none of the
            // cost evaluation)
            const bool edge = xedge || BH * ib + BH > h;
            if (A.src_cost) {
                if (SMX_V5_TOUCH) cw_touch_p(ib + 1);
                if (edge) run(std::true_type{}, std::true_type{}); else run(std::false_type{}, std::true_type{});
            } else if (edge) run(std::true_type{}, std::false_type{});
            else { V5_MARK("cost begin"); run(std::false_type{}, std::false_type{}); V5_MARK("cost end"); }
        };
        // guidance of the output rows of iteration ib: FEW, WIDE loads -- a vector-memory instruction costs its wave and the CU's
        // address path the same whatever its width, and ten 8-byte loads per stage-1 wave and band were 700-1000 cycles of its
        // slot.  Stage 1: five 16-byte loads of two rows each (the planes hold row pairs on the band grid: pair P = rows
        // 2 P - 9, 2 P - 8).  Stage 2: the five fp16 row pairs of the band as one 16-byte and one 4-byte load.
        static_assert(BH == 10, "five row pairs per band");
        // stage 1: row pair P of the band grid -> its ring entry.  Unconditional (a load under a condition is waited for where the
        // branches merge); the lane's offset in a row of the plane is re-derived from the lane index -- two instructions -- instead
        // of living in a register through the rows.  (bit_cast of the whole vector: taking .y/.z/.w of the builtin's result
        // through a u4 copy let the compiler narrow the load to ONE dword -- wrong results, found by bisection)
        auto g1_load = [&](int P, auto Sc) {
            if constexpr (ROLE == ROLE_S1 && !(WHATIF & 32)) {
                const int Pc = __builtin_amdgcn_readfirstlane(min(max(P, 0), 5 * NI - 1));
                const unsigned vgl = (unsigned)(64 * wave + opaque(lane)) * 16u;
                gr[decltype(Sc)::value] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r_fix, (int)vgl, o_g1p + Pc * (CLP * 16), 0));
            }
        };
        auto issue_guid = [&](int ib, int yq0) {
            if (WHATIF & 32) return;
            const int ibc = __builtin_amdgcn_readfirstlane(min(max(ib, 0), NI - 1));
            if constexpr (ST2) {
                // (yq0 = 10 (ib - 1) - 18 at every call site: the band grid of the planes)
#ifdef SMX_V5_GI_B32     // (A/B: five 4-byte loads straight into their registers instead of a 16-byte load that needs an aligned quad)
#pragma unroll
                for (int c = 0; c < 4; ++c) gI[c] = ldu(r_fix, vg * 4u, o_i2p + ibc * (CLP * 16) + 4 * c);
#else
                const u4 a = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(r_fix, (int)(vg * 4u), o_i2p + ibc * (CLP * 16), 0));
                gI[0] = a.x; gI[1] = a.y; gI[2] = a.z; gI[3] = a.w;
#endif
                gI[4] = ldu(r_fix, vg, o_i2b + ibc * (CLP * 4));
            }
        };

        // hand-off unit of this thread (waves 5..): index hq < REC_U; halo units: row, first of its two columns
        auto hu_idx = [&]() { return ST2 ? opaque(tid) - 64 * NS1 : opaque(tid); };
        auto fetch_rec = [&](int rec) {
            const int hq = hu_idx();
            hreg = ld16_sc1(r_hand, (unsigned)(o_in + rec * REC_U * 16) + (hq >= 0 && hq < REC_U ? (unsigned)hq * 16u : 0u));
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

        // ---- row scans of iteration i (wave 0): lanes 0..19 stage 1, lanes 32..51 stage 2; lane = (row, component)
        // stage 1: band i1 in tile t1p; stage 2: the a/b band i2 (rows 10 i2 - 9 ..) in tile t2p
        auto rowscans = [&](int i1, int i2, float* t1p, float* t2p, float (*cinp)[2], bool pred1, bool pred2) {
            const int sc_l = lane & 31, sc_st = lane >> 5, sc_row = sc_l % BH, sc_comp = sc_l / BH;
            const bool sc_on = sc_l < 2 * BH;
            const int y = sc_st == 0 ? BH * i1 + sc_row : BH * i2 - R + sc_row;
            const bool act = sc_on && y >= 0 && y < h && (sc_st == 0 ? i1 >= 0 : i2 >= 0);
            if (!act) return;
            float* const row = (sc_st == 0 ? t1p : t2p) + sc_row * RS + sc_comp * P1;
            // stage 1 starts from the left neighbour's running row sum (or -0); stage 2 of a strip with a left
            // neighbour leaves the 19 halo columns alone and starts behind them from the halo's last column
            // (pred1 / pred2: the item of the stage-1 / stage-2 lanes has a left neighbour -- two items in an overlap slot)
            const bool keep = sc_st == 1 && pred2;
            float acc = (sc_st == 0 && pred1) ? cinp[sc_row][sc_comp] : -0.0f;
            f4* const r4 = (f4*)row;
            constexpr int NG = SW / 4;
#ifndef SMX_V5_SCAN_PF
#define SMX_V5_SCAN_PF 4        // (round 5 A/B on KITTI shape: 4 -> 0.816 ms per pair, 8 -> 0.823, 12 -> 0.819)
#endif
            constexpr int PF = SMX_V5_SCAN_PF;          // groups of reads in flight ahead of the dependent adds
            f4 v[PF];
#pragma unroll
            for (int g = 0; g < PF; ++g) v[g] = r4[g];
            f4 xo[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
            // fully unrolled: static registers for the reads in flight; the sums of two consecutive groups live in
            // different registers, so that the adds of a group do not wait until the 16-byte store of the previous one has
            // read its four source registers.  Groups 0 .. 4 (columns 0 .. 19) can leave the halo columns 0 .. 18 untouched.
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const f4 in = v[g % PF];
                if (g + PF < NG) v[g % PF] = r4[g + PF];
                f4& x = xo[g & 1];
                if ((WHATIF & 512) && g >= 5) { x = in; } else {
                acc = in.x + acc; x.x = acc;
                acc = in.y + acc; x.y = acc;
                acc = in.z + acc; x.z = acc;
                if (g == 4 && keep) acc = in.z;            // column 18 = the halo's last: the carry
                acc = in.w + acc; x.w = acc;
                }
                if (g < 5 && keep) { x.x = in.x; x.y = in.y; x.z = in.z; if (g < 4) x.w = in.w; }
#ifdef SMX_V5_SCAN_W64
                if (!((WHATIF & 256) && g >= 5)) { ((f2*)row)[2 * g] = (f2){x.x, x.y}; ((f2*)row)[2 * g + 1] = (f2){x.z, x.w}; }
#else
                if (!((WHATIF & 256) && g >= 5)) r4[g] = x; else asm volatile("" :: "v"(x));
#endif
                asm volatile("" :: "v"(xo[(g & 1) ^ 1]));
            }
        };

        // ---- comb rows ---------------------------------------------------------------------------------------------
        // (box sum S11 - S10 - S01 + S00, computeBoxFilterOnGPU guidedFilter.cu:305-318, in that order: box_bottom of the ring
        // slot of the row, minus the slot 19 rows up, box_top with that slot -- the left taps are lane i-1 of the comb)
        // window area of output row y for this lane in a band whose windows are clipped in y: xw x (window rows inside the image);
        // RN(1 / area) comes from the table in LDS -- looked up a row PAIR ahead of its use (rcp_pair), so that the LDS latency
        // is not on the path of the division
        // (rows outside the image get SOME valid table index -- their outputs are dropped; few scalar instructions matter here: a
        // border pair that re-derived both its own and the next pair's areas ran 300 cycles longer than an interior one)
        auto area_ix = [&](int y) {
            const unsigned yh = min((unsigned)(min(y + R, h - 1) - max(y - R - 1, -1)), (unsigned)HW);
            return (int)__umul24((unsigned)xw, max(yh, 1u));
        };
        auto rcp_pair = [&](int y) {
            const int a0 = area_ix(y), a1 = area_ix(y + 1);
            return (f4){rcp_s[a0], (float)a0, rcp_s[a1], (float)a1};
        };
        f4 rcb = {1.0f, 1.0f, 1.0f, 1.0f};   // border bands: (1/area, area) of rows a and b of the NEXT row pair

        // stage-1 role: tile 2 of a band's parity is free once every stage-2 wave has taken the a/b band two bands back out
        // of it (they do that first thing in the slot and count up s_x1); waited for right in front of the first a/b store
        unsigned x1_need = 0;
        auto wait_x1 = [&]() {
            if (WHATIF & 8192) return;
            for (unsigned spins = 0; __hip_atomic_load(&s_x1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < x1_need; ++spins) {
                __builtin_amdgcn_s_sleep(1);
                if (spins > (1u << 24)) { flag_store(A.status, 1u + (unsigned)item); break; }   // (cannot happen: every wave reaches its increment)
            }
            asm volatile("" ::: "memory");
        };
        // BORDER: a band with windows clipped in y (the first and last two of an image): per-row window areas from the table,
        // a/b rows outside the image become -0; it carries the x-edge handling along (EDGE), so that there are three
        // instantiations, not four.  (Round 4 ran border bands row by row, `row1`: 1 500-2 000 cycles more per slot, ten of an
        // item's 43 slots -- 8 % of the kernel.)
        auto rows1_pair = [&](auto N0c, auto EDGEc, auto BORDERc, auto WAITc, int i, const float* t1, f2& rv) {
            f2 caa = ca_i, cab = ca_i;
            f4 rcn = {1.0f, 1.0f, 1.0f, 1.0f};
            constexpr int N0 = decltype(N0c)::value, T0 = N0 % BH, GS = (N0 / BH + T0 / 2) & 1;
            constexpr bool BORDER = decltype(BORDERc)::value;
            constexpr bool EDGE = decltype(EDGEc)::value || BORDER;   // strip 0 or one with columns outside the image
            constexpr int SLa = N0, SLb = N0 + 1, SLPa = (N0 + RD - 1) % RD, S01a = (N0 + 1) % RD, S01b = (N0 + 2) % RD;
            const f2 rvb = tile_rd(t1 + (T0 + 1) * RS + jt);
            const f2 rvn = tile_rd(t1 + (T0 + 2 < BH ? T0 + 2 : T0 + 1) * RS + jt);
            const int ya = BH * i - R + T0;            // a/b rows ya, ya + 1
            if constexpr (decltype(BORDERc)::value) {
                caa = (f2){rcb.x, rcb.y};
                cab = (f2){rcb.z, rcb.w};
                rcn = rcp_pair(ya + 2);
            }
            const f2 old_a = ring[S01a];               // (slot S01a == SLb: the top taps of row a are what row b overwrites)
            ring[SLa] = rv + ring[SLPa];               // colSum integral.cu:124-128
            f2 ua = box_bottom(ring[SLa]);
            const f2 sb = rvb + ring[SLa];
            ua = ua - old_a;
            f2 ub = box_bottom(sb);
            ua = box_top(ua, old_a);
            ring[SLb] = sb;
            ub = ub - ring[S01b];
            if constexpr (EDGE) {
                // (no zero fill in front of a comb that is not a whole DPP row: the first lane of a comb of strip 0 -- its a/b
                // columns 0 .. 8 are outputs -- takes the box without left taps)
                const f2 u0a = ring[SLa] - old_a;
                if (il0 && k == 0) ua = u0a;
            }
            const f2 qa = div_ca_q(ua, caa);
            ub = box_top(ub, ring[S01b]);
            if constexpr (EDGE) {
                const f2 u0b = sb - ring[S01b];
                if (il0 && k == 0) ub = u0b;
            }
            const f2 qb = div_ca_q(ub, cab);
            f2 ma = qa, mb = qb;                       // (mean_p, mean_Ip)
            if constexpr (!FASTK) ma = div_ca_m(div_ca_e(qa, ua, caa), qa, caa);
            const f2 ga = {gr[GS].x, gr[GS].y}, gb = {gr[GS].z, gr[GS].w};
            const float mma = ga.x * ma.x;             // compute_ak_and_bk guidedFilter.cu:345-354
            if constexpr (!FASTK) mb = div_ca_m(div_ca_e(qb, ub, cab), qb, cab);
            const float ta = ma.y - mma;
            const float mmb = gb.x * mb.x;
            const float aka = 1.0f * ta * ga.y;
            const float tb = mb.y - mmb;
            const float mb2a = 1.0f * ga.x * aka;
            const float akb = 1.0f * tb * gb.y;
            const float bka = 1.0f * ma.x - mb2a;
            const float mb2b = 1.0f * gb.x * akb;
            f2 aba = {aka, bka};
            if (EDGE && !col_ok) aba = NZ2;            // a/b columns outside the image: -0
            if (BORDER && !(ya >= 0 && ya < h)) aba = NZ2;          // ... and rows
            if constexpr (decltype(WAITc)::value) wait_x1();
            tile_wr(tile2 + T0 * RS + jw, aba);
            const float bkb = 1.0f * mb.x - mb2b;
            f2 abb = {akb, bkb};
            if (EDGE && !col_ok) abb = NZ2;
            if (BORDER && !(ya + 1 >= 0 && ya + 1 < h)) abb = NZ2;
            tile_wr(tile2 + (T0 + 1) * RS + jw, abb);
            rv = rvn;
            // the pair that takes this ring entry next (the band's last pair leaves that to the end of the slot, behind the
            // hand-in, whose wait for the record -- the compiler makes it a wait for every load in flight -- then finds only
            // the load of the pair before, issued ~500 cycles earlier)
            if constexpr (T0 != BH - 2) g1_load(5 * i + T0 / 2 + 2, std::integral_constant<int, GS>{});
            if constexpr (BORDER) rcb = rcn;
            __builtin_amdgcn_sched_barrier(0);
        };
        // the same for stage 2; one vote on tiny window sums for both rows.  BORDER: per-row window areas, only the q rows
        // inside the image are stored (and vote)
        auto rows2_pair = [&](auto N0c, auto BORDERc, int i) {
            constexpr int N0 = decltype(N0c)::value, T0 = N0 % BH, PAR = N0 / BH;
            constexpr bool BORDER = decltype(BORDERc)::value;
            const int yq = BH * (i - 2) - 2 * R + T0;
            const bool va = !BORDER || (yq >= 0 && yq < h), vb = !BORDER || (yq + 1 >= 0 && yq + 1 < h);
            f2 caa = ca_i, cab = ca_i;
            f4 rcn = {1.0f, 1.0f, 1.0f, 1.0f};
            if constexpr (BORDER) {
                caa = (f2){rcb.x, rcb.y};
                cab = (f2){rcb.z, rcb.w};
                rcn = rcp_pair(yq + 2);
            }
            constexpr int SLa = (BH * PAR + T0 + 11) % RD, SLb = (SLa + 1) % RD, SLPa = (SLa + RD - 1) % RD, S01a = SLb, S01b = (SLb + 1) % RD;
            const f2 old_a = ring[S01a];
            ring[SLa] = r2[T0] + ring[SLPa];
            f2 ua = box_bottom(ring[SLa]);
            const f2 sb = r2[T0 + 1] + ring[SLa];
            ua = ua - old_a;
            f2 ub = box_bottom(sb);
            ua = box_top(ua, old_a);
            ring[SLb] = sb;
            ub = ub - ring[S01b];
            const f2 qa = div_ca_q(ua, caa);
            ub = box_top(ub, ring[S01b]);
            const f2 qb = div_ca_q(ub, cab);
            f2 ma = qa, mb = qb;
            // tiny (or zero, or non-finite) window sums of a, b take the true division (wave-uniform, rare); lanes without an
            // output do not vote
            bool tiny = false;
            if constexpr (!FASTK) {
                ma = div_ca_m(div_ca_e(qa, ua, caa), qa, caa);
                if constexpr (!BORDER) {
                    float amin;
                    asm("v_min3_f32 %0, |%1|, |%2|, |%3|" : "=v"(amin) : "v"(ua.x), "v"(ua.y), "v"(ub.x));
                    mb = div_ca_m(div_ca_e(qb, ub, cab), qb, cab);
                    asm("v_min_f32 %0, %1, |%2|" : "=v"(amin) : "v"(amin), "v"(ub.y));
                    tiny = !(amin >= 0x1p-100f);
                } else {
                    // (a row outside the image has no output and does not vote: its window sums are zero)
                    float m1, m2;
                    asm("v_min_f32 %0, |%1|, |%2|" : "=v"(m1) : "v"(ua.x), "v"(ua.y));
                    mb = div_ca_m(div_ca_e(qb, ub, cab), qb, cab);
                    asm("v_min_f32 %0, |%1|, |%2|" : "=v"(m2) : "v"(ub.x), "v"(ub.y));
                    tiny = (va && !(m1 >= 0x1p-100f)) || (vb && !(m2 >= 0x1p-100f));
                }
            }
            if (!FASTK && (__builtin_amdgcn_ballot_w64(tiny) & okmask) != 0) {
                asm volatile("; exact-division slow path");
                ma.x = 1.0f * ua.x / caa.y; ma.y = 1.0f * ua.y / caa.y;
                mb.x = 1.0f * ub.x / cab.y; mb.y = 1.0f * ub.y / cab.y;
            }
            const fg_t ip = __builtin_bit_cast(fg_t, gI[T0 / 2]);
            const float tqa = ma.x * (float)ip.x;      // compute_q guidedFilter.cu:363-369
            const float tqb = mb.x * (float)ip.y;
            float qva = tqa + ma.y;
            float qvb = tqb + mb.y;
#if (SMX_V5_WHATIF & 16384)
            if constexpr (ST2) {
                if (pairB) {
                    const float oa = qo[T0], ob = qo[T0 + 1];
                    const bool sa = qva == qva && !(oa < qva), sb = qvb == qvb && !(ob < qvb);
                    const uint64_t ba = __builtin_amdgcn_ballot_w64(sa), bb = __builtin_amdgcn_ballot_w64(sb);
                    qva = sa ? qva : oa;
                    qvb = sb ? qvb : ob;
                    if (lane == 0) {
                        g_masksink[((blockIdx.x * 8 + wave) * 64 + (yq & 31)) & 0xffff] = ba;
                        g_masksink[((blockIdx.x * 8 + wave) * 64 + 32 + (yq & 31)) & 0xffff] = bb;
                    }
                }
            }
#endif
            if (!(WHATIF & 16)) {
                if constexpr (QPERM) {
                    // own scratch: the two rows of the pair as ONE 8-byte store (yq is even; a row behind an image of odd height
                    // lands in the padding row of the last pair)
                    if (va) __builtin_amdgcn_raw_buffer_store_b64((u2){__builtin_bit_cast(unsigned, qva), __builtin_bit_cast(unsigned, qvb)}, r_q, (int)vo, q_row0 + (yq >> 1) * q_pitch, AUX_NT);
                } else {
                    if (va) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, qva), r_q, (int)vo, q_row0 + yq * q_pitch, AUX_NT);
                    if (vb) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, qvb), r_q, (int)vo, q_row0 + (yq + 1) * q_pitch, AUX_NT);
                }
            }
            if constexpr (BORDER) rcb = rcn;
            __builtin_amdgcn_sched_barrier(0);
        };

        // hand-in of record `rec` (stage-1 role, strips with a left neighbour): stage-2 halo columns -> tile 2 (scanned
        // around in R(rec)), stage-1 row carries -> LDS
        auto hand_in = [&](bool halo, const f4 hreg) {
            const int hq = hu_idx();
            if (hq >= 0 && hq < NHU) {
                if (halo) {
                    const int t = hq / 10, j = (hq - 10 * t) * 2;
                    float* dst = tile2 + t * RS + j;
                    dst[0] = hreg.x;
                    dst[P1] = hreg.y;
                    if (j + 1 < HW) { dst[1] = hreg.z; dst[P1 + 1] = hreg.w; }
                }
            } else if (hq >= NHU && hq < REC_U) {
                const int tp = hq - NHU;
                cin1[2 * tp][0] = hreg.x; cin1[2 * tp][1] = hreg.y;
                cin1[2 * tp + 1][0] = hreg.z; cin1[2 * tp + 1][1] = hreg.w;
            }
        };

        // ===================================== the slot loop: the row scans run beside the comb rows =======================
        // Scan pass s (s = -1 .. ) = stage-1 band s+1 and the a/b band s-1, by the scan wave during slot s; slot s also has the
        // comb rows of stage 1 on band s (scanned in pass s-1) and of stage 2 on the a/b band s-2 (scanned in pass s-1, taken out
        // of its tile at the start of the slot), and the cost wave's band s+2.  ONE workgroup barrier per slot.
        // Stage-1 band b lives in tile1[gb mod 3], the a/b band b in tile2s[gb mod 2], the carries of pass s in cin1s[gs mod 2]
        // with gb = b + (global band of the item's band 0).  The hand-off record of pass s -- {carries of band s+1, halo of the
        // a/b band s-1} -- has index s+1.
        // The buffers by GLOBAL band, relative to the band gb of the current global slot (gb = local slot + the item's band 0; it
        // advances by one per slot across item boundaries, whichever item a role is on): G1[j] = tile 1 of band gb + j,
        // G2[j] = tile 2 of band gb - j, GC[j] = the carries of pass gb + j.  Rotated at the end of every slot: no modulo
        // arithmetic in the loop (the first pipelined build spent 60 % more scalar instructions than the per-item loop).
        float* G1[3] = {tile1[1], tile1[2], tile1[0]};                 // gb = -2 at the first slot: (gb + 6) % 3 = 1
        float* G2[2] = {tile2s[0], tile2s[1]};                         // (gb + 6) & 1 = 0; band gb - 1: 1
        float (*GC[2])[2] = {cin1s[0], cin1s[1]};
        int g = 0;                  // global slot of this workgroup
        int nf = 0;                 // items the front roles have started
        int slf = -2;               // local slot of the front item (-2 .. P-3)
        bool ending = false;        // no ticket left: the front roles idle while stage 2 finishes the last item
        bool f_pred = false;        // the front item has a left neighbour (every wave tracks it: the flag wait has a barrier in it)
        bool own = false;           // this role has an item
        // stage 2 moves to the front item in front slot 1: the back item's last q rows left in front slot q_last - P <= 0, and
        // its own first slot on an item is 1 (a fixed slot keeps the item's state invariant in the main slot loop below)
        constexpr int sw = 1;
        int pend_item = 0; bool pend = false;                   // (stage 2) the front item, taken over at slot sw
        int sl2 = 0;                                            // (stage 2) local slot of its own item
        bool b_pred = false, b_own = false;                     // (scan wave) the back item: its a/b band P-3 is scanned in front slot -2
        int nxt = 0;                                            // (cost wave, lane 63) the ticket after the front item's

        // ---- stage 1: one slot of its item (sl = -2: the first record and the first guidance pairs; -1: a record; >= 0: rows)
        auto slot_s1 = [&](auto PARc, int sl) {
            constexpr int PAR = decltype(PARc)::value;
            [[maybe_unused]] const int i = g;           // (V5_STAMP)
            if constexpr (ROLE == ROLE_S1) {
                // the record of pass sl+1 is needed at the end of this slot: its load goes out now (unconditionally);
                // what it returns counts only if the record had been published
                fetch_rec(min(sl + 2, NI - 1));
                if (sl == -2) {
                    g1_load(0, std::integral_constant<int, 0>{});
                    g1_load(1, std::integral_constant<int, 1>{});
                }
                if (sl >= 0) {
                    const float* const t1 = G1[0];
                    f2 rv = tile_rd(t1 + jt);
                    tile2 = G2[0];
                    V5_STAMP(3);
                    if (sl == 0 && succ && tid < NCU) {
                        // record 0 (the stage-1 row carries of band 0; no halo yet) leaves from HERE: stage 2 is still on the
                        // item before this one.  It is drained at the end of this slot and published by stage 2 in the next.
                        const float* p = t1 + 2 * tid * RS + OWS - 1;
                        st16_sc1(r_hand, (unsigned)o_out + (unsigned)(NHU + tid) * 16u, (f4){p[0], p[P1], p[RS], p[RS + P1]});
                    }
                    if (A.prio) __builtin_amdgcn_s_setprio(PRIO_S1);
                    x1_need = (unsigned)NS1 * (unsigned)(g + 1);
                    const bool border = BH * sl - R < R + 1 || BH * sl - R + BH - 1 > h - 1 - R;
                    if (WHATIF & 4) wait_x1();                  // (the rows wait in front of their first store)
                    V5_STAMP(4);
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(WHATIF & 4)) {
#define V5_P1(TT) rows1_pair(std::integral_constant<int, BH * PAR + TT>{}, std::false_type{}, std::false_type{}, std::integral_constant<bool, TT == 0>{}, sl, t1, rv);
#define V5_P1E(TT) rows1_pair(std::integral_constant<int, BH * PAR + TT>{}, std::true_type{}, std::false_type{}, std::integral_constant<bool, TT == 0>{}, sl, t1, rv);
#define V5_P1B(TT) rows1_pair(std::integral_constant<int, BH * PAR + TT>{}, std::true_type{}, std::true_type{}, std::integral_constant<bool, TT == 0>{}, sl, t1, rv);
                    if (border) { rcb = rcp_pair(BH * sl - R); V5_P1B(0) V5_STAMP(6); V5_P1B(2) V5_STAMP(7); V5_P1B(4) V5_STAMP(8); V5_P1B(6) V5_STAMP(9); V5_P1B(8) V5_STAMP(10); }
                    else if (xedge || k == 0) { V5_P1E(0) V5_P1E(2) V5_P1E(4) V5_P1E(6) V5_P1E(8) }
                    else { V5_MARK("s1rows begin"); V5_P1(0) V5_STAMP(6); V5_P1(2) V5_STAMP(7); V5_P1(4) V5_STAMP(8); V5_P1(6) V5_STAMP(9); V5_P1(8) V5_STAMP(10); V5_MARK("s1rows end"); }
                    }
#undef V5_P1
#undef V5_P1E
#undef V5_P1B
                    __builtin_amdgcn_s_setprio(0);
                    V5_STAMP(11);
                }
            }
        };
        // ---- stage 2: one slot (1 <= sl <= q_last) of its item: the a/b band sl-2 out of its tile, record sl out, q rows
        auto slot_s2 = [&](auto PARc, int sl) {
            constexpr int PAR = decltype(PARc)::value;
            [[maybe_unused]] const int i = g;           // (V5_STAMP)
            if constexpr (ST2) {
                if (succ && tid == 64 * 2 * NS1 - 1) flag_store(myflag, (unsigned)sl);     // records 0 .. sl-1 are complete
                // the a/b band sl-2 (scanned in pass sl-1) out of its tile; the record of pass sl-1 (index sl): halo of that
                // band, carries of stage-1 band sl
                if (A.prio) __builtin_amdgcn_s_setprio(PRIO_S2HEAD);      // (the stage-1 waves are waiting for this copy-out)
                V5_MARK("s2head begin");
                const float* const t2 = G2[0];
#pragma unroll
                for (int t = 0; t < BH; ++t) r2[t] = tile_rd(t2 + t * RS + jt);
                const int hq = hu_idx();
                const bool rec_out = succ && hq >= 0 && hq < REC_U;
                f4 hov = {0.0f, 0.0f, 0.0f, 0.0f};
                if (rec_out) {
                    if (hq < NHU) {
                        const int t = hq / 10, j = (hq - 10 * t) * 2;
                        const float* p = t2 + t * RS + OWS + j;
                        hov = (f4){p[0], p[P1], p[1], p[P1 + 1]};
                    } else {
                        const float* p = G1[0] + 2 * (hq - NHU) * RS + OWS - 1;
                        hov = (f4){p[0], p[P1], p[RS], p[RS + P1]};
                    }
                }
                // everything this wave needs from tile 2 is in registers: tell the stage-1 waves at once (they are waiting to
                // overwrite the tile); the record goes to memory behind that
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r2[0]), "+v"(r2[1]), "+v"(r2[2]), "+v"(r2[3]), "+v"(r2[4]), "+v"(r2[5]),
                             "+v"(r2[6]), "+v"(r2[7]), "+v"(r2[8]), "+v"(r2[9]), "+v"(hov) :: "memory");
                if (lane == 0) __hip_atomic_fetch_add(&s_x1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (rec_out) st16_sc1(r_hand, (unsigned)(o_out + sl * REC_U * 16) + (unsigned)hq * 16u, hov);
                // a/b rows behind the image are -0 (stage 1 writes them as long as it is on this item; in the last slots of an
                // item it has moved on, and the tile holds an older band)
                if (BH * (sl - 2) - R + BH - 1 >= h) {          // (only in the last slots of an item)
#pragma unroll
                    for (int t = 0; t < BH; ++t)
                        if (BH * (sl - 2) - R + t >= h) r2[t] = NZ2;
                }
                V5_MARK("s2head end");
                __builtin_amdgcn_s_setprio(0);
                V5_STAMP(3);
                V5_STAMP(4);
                [[maybe_unused]] bool s2_interior = false;
                if (sl >= 2) {
                    const int yq0 = BH * (sl - 2) - 2 * R;
                    const bool border = yq0 < R + 1 || yq0 + BH - 1 > h - 1 - R;
                    s2_interior = !border && !(WHATIF & (8 | 16 | 32)) && sl != q_last;
#if (SMX_V5_WHATIF & 16384)
                    if constexpr (ST2) {
                        if (pairB) {
#pragma unroll
                            for (int t = 0; t < BH; ++t)
                            {
                                const int yc = min(max(yq0 + t, 0), h - 1);
                                qo[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_qp, (int)vo, q_row0 + (QPERM ? (yc >> 1) * q_pitch + (yc & 1) * 4 : yc * q_pitch), AUX_NT));
                            }
                        }
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(WHATIF & 8)) {
#define V5_P2(TT) rows2_pair(std::integral_constant<int, BH * PAR + TT>{}, std::false_type{}, sl);
#define V5_P2B(TT) rows2_pair(std::integral_constant<int, BH * PAR + TT>{}, std::true_type{}, sl);
                    if (border) { rcb = rcp_pair(yq0); V5_P2B(0) V5_STAMP(6); V5_P2B(2) V5_STAMP(7); V5_P2B(4) V5_STAMP(8); V5_P2B(6) V5_STAMP(9); V5_P2B(8) V5_STAMP(10); }
                    else { V5_MARK("s2rows begin"); V5_P2(0) V5_STAMP(6); V5_P2(2) V5_STAMP(7); V5_P2(4) V5_STAMP(8); V5_P2(6) V5_STAMP(9); V5_P2(8) V5_STAMP(10); V5_MARK("s2rows end"); }
                    }
#undef V5_P2
#undef V5_P2B
                }
                issue_guid(sl, BH * (sl - 1) - 2 * R);
                // the record stored above is complete in memory before the barrier behind which it is published
                // (vector-memory operations of a wave complete in issue order: what was issued behind the record store -- the
                // q stores of an interior band, ten rows or five row pairs, and the two guidance loads -- may stay in flight;
                // -DSMX_V5_S2_KEEP=0: the full drain of round 4; the item's last slot drains everything: FLAG_DONE follows)
#ifdef SMX_V5_GI_B32
                constexpr int KEEP = S2_KEEP > 0 ? (QPERM ? S2_KEEP - BH / 2 : S2_KEEP) + 3 : 0;
#else
                constexpr int KEEP = S2_KEEP > 0 ? (QPERM ? S2_KEEP - BH / 2 : S2_KEEP) : 0;
#endif
                if (KEEP > 0 && s2_interior) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KEEP) : "memory"); else
                drain_vmem();
            }
        };

        // One global slot.  The period P is EVEN (smx_agg_v5.h period()), so the local slot of every role's item has the parity of the
        // global slot: the loop below alternates the two parities statically (the register rings and the guidance ring use
        // static slots per parity; a run-time dispatch would put loads under a condition -- they are then waited for where the
        // branches merge: measured, +13 %).  Returns true when the workgroup is done.
        auto iter = [&](auto PARc) -> bool {
            wg_barrier();
            [[maybe_unused]] const int i = g;           // (V5_STAMP: by global slot)
            V5_STAMP(0); V5_STAMP(1); V5_STAMP(2);
            if constexpr (ROLE >= ROLE_SCAN) { V5_STAMP(3); V5_STAMP(4); }
            if (slf == -2) {
                // ---- a new front item (or none) -----------------------------------------------------------------------------
                const int it = s_queue[nf & 3];
                ending = it >= A.nitems;
                if (ending && nf == 0) return true;     // (a workgroup without any ticket)
                f_pred = !ending && it / nsv > 0 && !(WHATIF & (64 | 4096));
                // every wave must come to the same `seen` (the wait below has a barrier in it): what the cost wave peeked at in
                // the slot before, for THIS item
                seen = nf > 0 ? s_peekn : 0u;
                if constexpr (ROLE == ROLE_COST) {
                    if (own) cost_verdict();
                    own = !ending;
                    if (own) { decode(it); cost_consts(); cost_rsrc(); }
                } else if constexpr (ROLE == ROLE_S1) {
                    own = !ending;
                    if (own) {
                        decode(it); lane_consts(); ring_reset();
                        gr[0] = (f4){0, 0, 0, 0}; gr[1] = (f4){0, 0, 0, 0};
                    }
                } else if constexpr (ROLE == ROLE_SCAN) {
                    b_pred = pred; b_own = own;
                    own = !ending;
                    if (own) decode(it);
                } else {
                    pend_item = it; pend = !ending;
                }
            } else {
                // (the peek of the previous slot, written before the barrier above; the cost wave writes it every slot)
                seen = max(seen, s_peek[g & 1]);
            }
            if constexpr (ST2) {
                // the slot behind an item's last one: the item is complete (its last record and q rows were drained at the end of
                // that slot, the barrier above lies in between) -- published NOW, before this slot's work and its flag wait: an
                // item's completion must not hang on the progress of the workgroup's next item any longer than its last rows do
                if (own && sl2 == q_last + 1 && succ && tid == 64 * 2 * NS1 - 1) flag_store(myflag, FLAG_DONE);
                if (slf == sw) {
                    own = pend;
                    sl2 = sw;
                    if (own) {
                        decode(pend_item); lane_consts(); ring_reset();
#pragma unroll
                        for (int t = 0; t < BH; ++t) r2[t] = NZ2;
                    }
                }
            }
            // ---- the slot's work ----------------------------------------------------------------------------------------
            if constexpr (ROLE == ROLE_SCAN) {
                if (!(WHATIF & 1024) && A.prio) __builtin_amdgcn_s_setprio(PRIO_SCAN);
                V5_MARK("scan begin");
                if (!(WHATIF & 1)) {
                    const int i1 = own && slf + 1 < NI ? slf + 1 : -1;
                    // (front slot -2: the back item's last a/b band with rows inside the image can still be waiting, P == s1_last + 3;
                    // it is the band in front of this global slot's like any other)
                    const int i2 = slf == -2 ? (b_own ? P - 3 : -1) : (own ? slf - 1 : -1);
                    rowscans(i1, i2, G1[1], G2[1], GC[0], pred, slf == -2 ? b_pred : pred);
                }
                V5_MARK("scan end");
                __builtin_amdgcn_s_setprio(0);
            } else if constexpr (ROLE == ROLE_COST) {
                if (lane == 63) {
                    // an otherwise idle lane looks at the left neighbour's flag for the prefetch of the next record (written every
                    // slot: a stale entry could belong to the item before)
                    unsigned pk = 0u;
                    if (own && pred && seen != FLAG_DONE && seen < (unsigned)(slf + 4)) pk = flag_load(myflag - 1);
                    s_peek[(g + 1) & 1] = pk;
                    // the ticket after this item's, a few slots before the front roles need it -- and never in the item's first
                    // slot when items overlap (P >= 6 then): behind the flag wait of slot -1 the left neighbour's workgroup has
                    // taken ITS next ticket already, and the deadlock-freedom argument needs that order
                    if (own && slf == P - 6) {
                        nxt = (int)__hip_atomic_fetch_add((gu32*)A.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        s_queue[(nf + 1) & 3] = nxt;
                    }
                    // ... and, in the last slot of this item, the flag of THAT item's left neighbour for its first slot
                    if (own && slf == P - 3) {
                        unsigned pn = 0u;
                        if (nxt < A.nitems) {
                            const int kn = nxt / nsv, svn = nxt - kn * nsv;
                            if (kn > 0 && !(WHATIF & (64 | 4096))) pn = flag_load(A.flags + (size_t)svn * K + kn - 1);
                        }
                        s_peekn = pn;
                    }
                }
                if (A.prio) __builtin_amdgcn_s_setprio(PRIO_COST);
                if (own && !(WHATIF & 2) && slf + 2 < NI) eval_band_p(slf + 2, G1[2]);
                __builtin_amdgcn_s_setprio(0);
            } else if constexpr (ROLE == ROLE_S1) {
                if (own) slot_s1(PARc, slf);
            } else {
                const int sl = sl2;                                 // local slot of stage 2's own item
                if (own && sl >= 1 && sl <= q_last) {
                    slot_s2(PARc, sl);
                } else if (lane == 0) {
                    // (no copy-out in this slot: the stage-1 waves count one per global slot all the same)
                    __hip_atomic_fetch_add(&s_x1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            // ---- hand-in of the FRONT item's record of pass slf+1 (index slf+2), prefetched at the top of the slot if it had been
            // published.  (two copies of the hand-in on purpose: the prefetched record is OLDER than the guidance loads of this slot,
            // so its wait leaves those in flight; a record fetched again behind the flag wait is the newest load, and one merged
            // copy would wait for everything)
            have_pref = f_pred && (seen == FLAG_DONE || seen >= (unsigned)(slf + 3));
            if (f_pred && !have_pref && slf + 2 < NI) {
                if (tid == 0) spin_pred((unsigned)(slf + 3));
                wg_barrier();
                seen = s_seen;
                if constexpr (ROLE == ROLE_S1) {
                    const int hq = hu_idx();
                    const f4 late = ld16_sc1(r_hand, (unsigned)(o_in + min(slf + 2, NI - 1) * REC_U * 16) + (hq >= 0 && hq < REC_U ? (unsigned)hq * 16u : 0u));
                    cin1 = GC[1]; tile2 = G2[0];
                    hand_in(slf >= 0, late);
                }
            } else if constexpr (ROLE == ROLE_S1) {
                // (the halo is that of the a/b band slf, which the comb rows above have just written into the same tile)
                V5_MARK("s1handin begin");
                if (own && pred && slf + 2 < NI) { cin1 = GC[1]; tile2 = G2[0]; hand_in(slf >= 0, hreg); }
                V5_MARK("s1handin end");
            }
            if constexpr (ROLE == ROLE_S1) {
                // (pair 4 of this slot used ring entry (parity + 4) & 1 = parity.  Unconditional: in the two slots in front of the
                // rows it re-loads the pair the entry holds already -- pair 0 in slot -2, pair 1 in slot -1 -- and with no item it
                // is harmless)
                g1_load(slf >= 0 ? 5 * slf + 6 : slf + 2, PARc);
                if (own && slf == 0 && succ) drain_vmem();      // record 0 (see slot_s1) is in memory before the next barrier
            }
            V5_STAMP(5);
            if (ending && slf == 1) return true;
            { float* const t = G1[0]; G1[0] = G1[1]; G1[1] = G1[2]; G1[2] = t; }
            { float* const t = G2[0]; G2[0] = G2[1]; G2[1] = t; }
            { float (*const t)[2] = GC[0]; GC[0] = GC[1]; GC[1] = t; }
            ++g;
            ++sl2;
            return false;
        };
        // Per front item: four slots in which items change hands (-2: the front roles; 1: stage 2), then the main loop over
        // slots 2 .. P-3, in which every role's item state is loop-invariant (the compiler hoists what it derives from it: a
        // single loop over all slots cost 60 % more scalar instructions).  P is even: P - 4 slots in the main loop.
        for (;;) {
            bool done = false;
            for (slf = -2; slf < 2 && !done; slf += 2) {
                done = iter(std::integral_constant<int, 0>{});
                if (!done) { ++slf; done = iter(std::integral_constant<int, 1>{}); --slf; }
            }
            if (done) break;
            for (slf = 2; slf <= P - 3; slf += 2) {
                (void)iter(std::integral_constant<int, 0>{});
                ++slf;
                (void)iter(std::integral_constant<int, 1>{});
                --slf;
            }
            ++nf;
        }
        if constexpr (ROLE == ROLE_COST) { if (own) cost_verdict(); }
        };
        if (wave < NS1) item_body(std::integral_constant<int, ROLE_S1>{});
        else if (wave < 2 * NS1) item_body(std::integral_constant<int, ROLE_S2>{});
        else if (wave == 2 * NS1) item_body(std::integral_constant<int, ROLE_SCAN>{});
        else item_body(std::integral_constant<int, ROLE_COST>{});
    }
}

// ---------------------------------------------------------------------------------------------------------------
// guidance statistics (guidedFilter.cu:58-123, from the integral images) + their comb-ordered copies: grid (K * NI * 5, nviews), block CLP
// ---------------------------------------------------------------------------------------------------------------
struct PermArgs {
    const float* S0[2];     // integral images of I and I*I (k_v4_guid_rows / k_v4_guid_cols)
    const float* S1[2];
    f2* G[2];               // out: (mean_I, 1/(var_I + eps)) [h][w] (the ring walker's layout; the caller's mean image comes from it)
    uint8_t* mean_u8[2];    // out, optional
    double eps;
    const fg_t* FG[2];
    f2* g1p[2];
    unsigned* i2p[2];
};
__global__ __launch_bounds__(CLP) void k_v5_perm(PermArgs pa, int w, int h, int K, int NI) {
    const int cl = threadIdx.x, cw = cl >> 6, ln = cl & 63;
    const int rho = ln / L < CPW ? CPW * cw + ln / L : HW, il = ln % L;
    const int kb = blockIdx.x / (BH / 2), j = blockIdx.x - kb * (BH / 2);   // one row pair of a band per workgroup
    const int k = kb / NI, ib = kb - k * NI, v = blockIdx.y;
    const int x1 = OWS * k - R - 1 + HW * il + rho;     // a/b column of stage-1 comb lane cl
    const int xq = x1 - R;                              // q column of stage-2 comb lane cl
    const bool on1 = rho < HW && x1 >= 0 && x1 < w, on2 = rho < HW && xq >= 0 && xq < w;
    // stage 1: the a/b rows 10 ib - 9 + 2 j, + 1 of band ib (rows clamped into the image; the a_k, b_k of rows outside it are
    // replaced by -0 whatever their guidance)
    {
        const int ya = min(max(BH * ib - R + 2 * j, 0), h - 1), yb = min(max(BH * ib - R + 2 * j + 1, 0), h - 1);
        f4 g = {0.0f, 0.0f, 0.0f, 0.0f};
        if (on1) {
            // (a pixel is evaluated by every strip whose a/b columns hold it and once more per clamped row: 1.13 evaluations
            // per pixel on KITTI shape, all with the same result -- whichever store lands last, G holds that value)
            const f2 a = guid_point(pa.S0[v], pa.S1[v], x1, ya, w, h, R, pa.eps);
            const f2 b = yb == ya ? a : guid_point(pa.S0[v], pa.S1[v], x1, yb, w, h, R, pa.eps);
            g = (f4){a.x, a.y, b.x, b.y};
            pa.G[v][(size_t)ya * w + x1] = a;
            pa.G[v][(size_t)yb * w + x1] = b;
            if (pa.mean_u8[v]) {
                pa.mean_u8[v][(size_t)ya * w + x1] = mean_to_u8(a.x);
                pa.mean_u8[v][(size_t)yb * w + x1] = mean_to_u8(b.x);
            }
        }
        ((f4*)pa.g1p[v])[((size_t)k * (5 * NI) + 5 * ib + j) * CLP + cl] = g;
    }
    // stage 2: the q rows 10 (ib - 1) - 18 + 2 j, + 1 of band ib: image values as an fp16 pair (row pairs clamped like the
    // loads they replace; a row behind the image holds 0); pairs 0 .. 3 of a band form a u32x4, pair 4 lies behind them
    {
        const int yp = min(max((BH * (ib - 1) - 2 * R) / 2 + j, 0), (h - 1) / 2), y = 2 * yp;
        fg_t p2 = {(_Float16)0.0f, (_Float16)0.0f};
        if (on2) {
            p2.x = pa.FG[v][(size_t)y * (w + 2 * PADX) + PADX + xq].x;
            if (y + 1 < h) p2.y = pa.FG[v][(size_t)(y + 1) * (w + 2 * PADX) + PADX + xq].x;
        }
        const unsigned pr = __builtin_bit_cast(unsigned, p2);
        unsigned* const i2a = pa.i2p[v];
        unsigned* const i2b = pa.i2p[v] + (size_t)K * NI * CLP * 4;
        const size_t e = ((size_t)k * NI + ib) * CLP + cl;
        if (j < 4) i2a[4 * e + j] = pr; else i2b[e] = pr;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// WTA over the chunk's comb-ordered q planes [slice][K][ceil(h/2)][OWS][2] (dispSelectOnGPU guidedFilter.cu:403-411):
// four elements per lane (two columns x the two rows of a pair), coalesced nt loads, 8 in flight; the keys stay [h][w] --
// the lane finds its pixels once per call.  grid (ceil(plane / 1024), nviews)
// ---------------------------------------------------------------------------------------------------------------
struct Wta5Args {
    const float* q[2];
    int64_t* keys[2];
    const unsigned* skip_if;      // != NULL: the pass does nothing if this word is nonzero (the comb walker's planes do not count)
    int fresh;                    // != 0: the keys hold nothing yet (no smx_dev_init_keys ran): start from the identity, do not load them
};
// Four elements per lane, 16-byte loads: a row pair of a strip is 2 OWS = 304 floats, so a plane is a whole number of quads
// and every plane starts 64-byte aligned (q_plane_floats is a multiple of 304; the scratch is carved 256-byte aligned)
static_assert((2 * OWS) % 4 == 0, "quads (two columns x the two rows of a pair) do not straddle pair rows");
__global__ __launch_bounds__(256) void k_v5_wta(Wta5Args wa, int w, int h, int K, int count, int slice0) {
    constexpr int EPL = 4;
    const int hp = (h + 1) / 2;                                 // row pairs: [K][hp][OWS][2] (smx_agg_v5.h)
    const size_t np = (size_t)K * hp * (2 * OWS);
#ifdef SMX_WTA_REV       // (A/B: the planes back to front -- the strips the walker wrote last first)
    const size_t e0 = ((size_t)(gridDim.x - 1 - blockIdx.x) * 256 + threadIdx.x) * EPL;
#else
    const size_t e0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * EPL;
#endif
    if (e0 >= np || (wa.skip_if && flag_load(const_cast<unsigned*>(wa.skip_if)) != 0u)) return;
    const float* __restrict__ q = wa.q[blockIdx.y] + e0;
    int64_t* const keys = wa.keys[blockIdx.y];
    // the pixels of this lane's elements (once per call)
    int64_t* kp[EPL];
    int64_t key[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        const size_t e = e0 + j;
        const int prow = (int)(e / (2 * OWS)), rem = (int)(e - (size_t)prow * (2 * OWS));   // prow = k hp + yp
        const int p = rem >> 1, k = prow / hp, y = 2 * (prow - k * hp) + (rem & 1);
        const int rho = p / (L - 1), i1 = p - (L - 1) * rho;
        const int x = OWS * k + (k + 1 < K ? HW * i1 + rho : p);        // (the last strip is in column order)
        kp[j] = e < np && x < w && y < h ? keys + (size_t)y * w + x : nullptr;
        key[j] = kp[j] && !wa.fresh ? *kp[j] : KEY_IDENTITY;
    }
    // nothing of this lane lies in the image (the tail of a row of the last strip): no load at all
    bool any = false;
#pragma unroll
    for (int j = 0; j < EPL; ++j) any = any || kp[j];
    if (!any) return;
    typedef float fv __attribute__((ext_vector_type(EPL)));
    WtaRun run[EPL];                    // (smx_common.h: the winner of this call's slices in the float domain, packed once)
    auto step = [&](const fv v, unsigned slice) {
#pragma unroll
        for (int j = 0; j < EPL; ++j) run[j].step(v[j], slice);
    };
    int z = 0;
    constexpr int U = 8;
    for (; z + U <= count; z += U) {
        fv v[U];
#pragma unroll
#ifdef SMX_WTA_LD_PLAIN
        for (int t = 0; t < U; ++t) v[t] = *(const fv*)&q[(size_t)(z + t) * np];
#else
        for (int t = 0; t < U; ++t) v[t] = __builtin_nontemporal_load((const fv*)&q[(size_t)(z + t) * np]);
#endif
#pragma unroll
        for (int t = 0; t < U; ++t) step(v[t], (unsigned)(slice0 + z + t));
    }
    for (; z < count; ++z) step(__builtin_nontemporal_load((const fv*)&q[(size_t)z * np]), (unsigned)(slice0 + z));
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        const int64_t kk = run[j].key();
        key[j] = kk < key[j] ? kk : key[j];
    }
#pragma unroll
    for (int j = 0; j < EPL; ++j)
        if (kp[j]) *kp[j] = key[j];
}

}  // namespace v5

int v5_perm_launch(int nviews, const float* const* S0, const float* const* S1, aggdev::f2* const* G, uint8_t* const* mean_u8,
                   const aggdev::fg_t* const* FG, aggdev::f2* const* g1p, unsigned* const* i2p, int w, int h, double eps,
                   hipStream_t st) {
    v5::PermArgs pa;
    memset(&pa, 0, sizeof(pa));
    for (int v = 0; v < nviews; ++v) {
        pa.S0[v] = S0[v]; pa.S1[v] = S1[v]; pa.G[v] = G[v]; pa.mean_u8[v] = mean_u8[v];
        pa.FG[v] = FG[v]; pa.g1p[v] = g1p[v]; pa.i2p[v] = i2p[v];
    }
    pa.eps = eps;
    const int K = v5::strips(w), NI = v5::bands(h);
    hipLaunchKernelGGL(v5::k_v5_perm, dim3((unsigned)(K * NI * (v5::BH / 2)), (unsigned)nviews), dim3(v5::CLP), 0, st, pa, w, h, K, NI);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

int v5_wta_launch(int nviews, const float* const* q, int64_t* const* keys, int w, int h, int count, int slice0,
                  const unsigned* skip_if, bool fresh, hipStream_t st) {
    v5::Wta5Args wa;
    wa.skip_if = skip_if;
    wa.fresh = fresh ? 1 : 0;
    for (int v = 0; v < 2; ++v) { wa.q[v] = q[v < nviews ? v : 0]; wa.keys[v] = keys[v < nviews ? v : 0]; }
    const int K = v5::strips(w);
    const size_t np = v5::q_plane_floats(w, h);
    for (int v = 0; v < nviews; ++v)
        if (((uintptr_t)wa.q[v] & 15) != 0) return fail(SMX_E_ARG, "v5_wta_launch: q scratch of view %d is not 16-byte aligned", v);
    hipLaunchKernelGGL(v5::k_v5_wta, dim3((unsigned)((np / 4 + 255) / 256), (unsigned)nviews), dim3(256), 0, st, wa, w, h, K,
                       count, slice0);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

void v5_geometry(int* ow, int* bh) { *ow = v5::OWS; *bh = v5::BH; }
void v5_slots(int h, int K, int* bands, int* q_last, int* period) { *bands = v5::bands(h); *q_last = (h + 37) / 10; *period = v5::period(h, K); }

bool v5_supported(const smx_params* p) {
    if (p->radius != v5::R) return false;
    if (!(p->eps >= 1.0) || !(p->eps < 1e30)) return false;
    const CostConst c = make_cost_const(p);
    if (!(c.alpha >= 0.0f && c.alpha <= 1.0f && c.th_color >= 0.0f && c.th_color < 1e6f && c.th_grad >= 0.0f && c.th_grad < 1e6f))
        return false;
    // smallest nonzero truncated terms: |dI| >= 1 (integers), |dg| >= 0.5 (halves), or the thresholds themselves
    const float m1 = c.th_color < 1.0f ? c.th_color : 1.0f, m2 = c.th_grad < 0.5f ? c.th_grad : 0.5f;
    const float t1 = c.oma * m1, t2 = c.alpha * m2;
    auto ok = [](float t) { return t == 0.0f || t >= 0x1p-60f; };
    if (!(ok(t1) && ok(t2))) return false;
    // the pipelined form truncates in packed halves: thresholds exact (and finite) in fp16
    if (!((float)(_Float16)c.th_color == c.th_color && (float)(_Float16)c.th_grad == c.th_grad && c.th_color < 60000.0f && c.th_grad < 60000.0f))
        return false;
    return true;
}

// materialised cost volumes: the cost parameters do not matter (the kernel checks the VALUES it loads), radius and eps do
bool v5_supported_cost(const smx_params* p) { return p->radius == v5::R && p->eps >= 1.0 && p->eps < 1e30; }

int v5_launch(const v5::Args& a, hipStream_t st) {
    int dev = 0, ncu = 256;
    SMX_HIP(hipGetDevice(&dev));
    SMX_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    int per_cu = v5::WG_PER_CU;                          // persistent: WG_PER_CU workgroups per CU
    static const int env_per_cu = env_int_once("SMX_V5_WG_PER_CU", 0);   // experiments: fewer workgroups per CU
    if (env_per_cu >= 1 && env_per_cu <= v5::WG_PER_CU) per_cu = env_per_cu;
    const int slots = per_cu * ncu;
    const int grid = a.nitems < slots ? a.nitems : slots;
    if (a.fast && a.qperm) hipLaunchKernelGGL((v5::k_v5_walk<1, 1>), dim3((unsigned)grid), dim3(v5::NT), 0, st, a);
    else if (a.fast) hipLaunchKernelGGL((v5::k_v5_walk<1, 0>), dim3((unsigned)grid), dim3(v5::NT), 0, st, a);
    else if (a.qperm) hipLaunchKernelGGL((v5::k_v5_walk<0, 1>), dim3((unsigned)grid), dim3(v5::NT), 0, st, a);
    else hipLaunchKernelGGL((v5::k_v5_walk<0, 0>), dim3((unsigned)grid), dim3(v5::NT), 0, st, a);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

// Residency of the walker as the runtime sees it (dev tool: tools/v5_quick.py --occupancy)
extern "C" __attribute__((visibility("default"))) int smx_debug_v5_occupancy(int* blocks_per_cu, int* vgprs, int* lds_bytes) {
    int nb = 0;
    SMX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)v5::k_v5_walk<0, 1>, v5::NT, 0));
    hipFuncAttributes fa;
    SMX_HIP(hipFuncGetAttributes(&fa, (const void*)v5::k_v5_walk<0, 1>));
    if (blocks_per_cu) *blocks_per_cu = nb;
    if (vgprs) *vgprs = fa.numRegs;
    if (lds_bytes) *lds_bytes = (int)fa.sharedSizeBytes;
    return SMX_OK;
}

#ifdef SMX_V5_STAMPS
extern "C" __attribute__((visibility("default"))) int smx_debug_read_stamps5(unsigned long long* out, int n) {
    const int m = 10 * v5::STAMP_SLOTS;
    SMX_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(v5::g_stamps), sizeof(unsigned long long) * (n < m ? n : m)));
    return m;
}
#endif
#ifdef SMX_V5_DUMP
extern "C" __attribute__((visibility("default"))) int smx_debug_read_dump5(float* out, int n) {
    const int m = 2 * v5::TILE_F + v5::NT * 48;
    SMX_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(v5::g_dump), sizeof(float) * (n < m ? n : m)));
    return m;
}
#endif

}  // namespace smx
