// smx_agg_v5.h -- interface of the comb-form aggregation kernel (smx_agg_v5.hip) towards the host orchestration
// of smx_agg_v4.hip: argument block, strip / band / record geometry, eligibility.
#pragma once
#include "smx_agg_dev.h"

namespace smx {
namespace v5 {

// Comb length 9: seven combs per wave, three comb waves per stage + a row-scan wave + a cost wave (512 threads, two
// workgroups per CU at 128 VGPRs); the row scans run beside the comb rows.  (Round 4 also built comb lengths 12 and 16 as
// three-barrier forms -- slower; tools/variants/ keeps that file.)
constexpr int L = 9;                    // lanes of a comb
constexpr int CPW = 64 / L;             // combs per wave
constexpr int NS1 = (19 + CPW - 1) / CPW;   // comb waves per stage
constexpr int OWS = 19 * (L - 1);       // output columns per strip (19 combs x (L - 1) outputs)
constexpr int BH = 10;                  // band height
constexpr int REC_U = 105;              // 16-byte units per hand-off record
constexpr int WG_PER_CU = 2;
constexpr int CLP = 64 * NS1;           // comb lane slots per stage and strip

struct Args {
    // the fixed part of the workspace: both image planes [h][w + 2 PADX] of k_v4_prep and the guidance planes
    // (mean_I, 1/(var_I + eps)) [h][w] of both views, addressed through ONE buffer descriptor
    const char* fix;
    size_t fix_bytes;
    unsigned o_fg[2];     // byte offset of view v's image plane (the other view's is o_fg[v ^ 1])
    // comb-ordered copies (k_v5_perm): what a comb lane cl = 16 rho + il of strip k needs in row y sits at
    // [k][y][cl], so that a wave's guidance load is one contiguous run instead of 16 clusters of 4 columns
    unsigned o_g1p[2];    // float4 [K][5 NI][CLP]: (mean_I, 1/(var_I + eps)) of the a/b rows 2 P - 9, 2 P - 8 (row pair P on the band grid) at the a/b column OWS k - 10 + 19 il + rho
    unsigned o_i2p[2];    // u32x4 [K][NI][CLP] + u32 [K][NI][CLP] behind it: image values (fp16 pairs) of the ten q rows of band ib at the q column OWS k - 19 + 19 il + rho
    // out, per view: qperm != 0: comb-ordered scratch [slice][K][ceil(h/2)][OWS][2]: the rows 2 yp, 2 yp + 1 of column
    // OWS k + 19 (il-1) + rho side by side at [(L-1) rho + il - 1] (a wave stores ONE contiguous run of 8-byte units per row
    // pair -- round 5: half the store instructions of a row at a time; read back by k_v5_wta); else the caller's [slice][h][w]
    // materialised cost volumes (src_cost != 0): slice s of view v at cost[v] + s * cost_plane, [h][w] each (the reference's
    // calling convention, guidedFilter.cu:198); else the costs are built from the image planes
    const float* cost[2];
    size_t cost_plane;    // floats per slice of a cost volume (w * h)
    int src_cost;
    unsigned* bad;        // src_cost: raised when a cost value falls outside what the exactness argument covers
    float* q[2];
    int qperm;
    size_t q_plane;       // floats per slice of q
    int d0[2];            // disparity of local slice 0 per view
    int w, h, K, NI, nslices, nsv, nitems;
    int P;                // slots between the starts of two items of a workgroup (period(h): items are pipelined, smx_agg_v5.hip)
    float* hand;          // hand-off records [parity][sv][iteration][REC_U x 4 floats]
    unsigned* flags;      // [sv][K]  published-record counters (zeroed before every launch)
    unsigned* ticket;     // work-item counter              (zeroed before every launch)
    unsigned* status;     // != 0: a flag wait timed out (results invalid)
    CostConst cc;
    unsigned th2;        // (th_color, th_grad) as packed halves (exact: v5_supported)
    int prio;            // 1: role priorities for the cost and stage-1 waves (smx_agg_v5.hip PRIO_*): small and medium launches
    int fast;            // 1: FAST mode (reciprocal multiplication instead of the exact division; not bit-exact)
};

inline int strips(int w) { return (w + OWS - 1) / OWS; }
inline size_t q_plane_floats(int w, int h) { return (size_t)strips(w) * ((h + 1) / 2) * 2 * OWS; }   // comb-ordered q scratch per slice: row pairs
inline int bands(int h) { return (h + 2 * 9 + BH - 1) / BH + 2; }     // the q rows of iteration i end at 10 i - 28
inline int records(int h) { return bands(h) + 2; }                   // hand-off records per (strip boundary, slice-view)
// An item occupies its roles for fewer slots than it has: the cost wave for local slots -2 .. s1_last - 2, stage 1 for
// -2 .. s1_last (s1_last = the slot of the last a/b row inside the image), stage 2 for 1 .. q_last (the last q row).  A
// workgroup starts its next item after period(h) slots: the longest of those ranges (bands(h) - 1 or bands(h)), at least 4,
// rounded up to an even number.
inline int period(int h, int K) {
    const int s1_last = (h + 8) / 10, q_last = (h + 37) / 10;
    int p = s1_last + 3 > q_last ? s1_last + 3 : q_last;
    // NO DEADLOCK needs p >= 2 K + 2 (and >= 6); the argument is spelled out in DESIGN.md 4.1.  While a workgroup runs slots
    // -2 .. 0 of its next item x' it may wait for flags 1 .. 3 of that item's left neighbour, and until it is through them
    // the LAST records of the item x it is finishing stay unpublished: late(x) <- early(pred(x')), a dependency on a LARGER
    // ticket.  Two things keep the wait-for graph acyclic all the same: (1) a wait on such a late record happens in slots
    // >= p - 4 of the waiter, and from a workgroup in slots -2 / -1 a chain of waits through left neighbours in the same row
    // (2 slots of lag per strip, at most K - 2 of them) reaches slot 2 K - 5 < p - 4 at most, so between two late waits a
    // cycle passes a workgroup that has not even STARTED the awaited item; (2) tickets are taken in slot p - 6 >= 0, behind a
    // flag wait, so a left neighbour's workgroup takes its next ticket before the right neighbour's does.  With both, the
    // ticket of the item a workgroup has just started falls strictly from one late wait of a cycle to the next: there is no cycle.  (h = 9, p = 4: two workgroups
    // waited for each other's last records; caught by the bounded wait,
    // tests/test_gpu_parity.py::test_items_pipelined_across_a_workgroups_tickets.)  Short or wide-and-short images get a
    // period of the whole item: nothing overlaps, an item's completion is published before its workgroup waits for anything
    // of the next one, and every wait points to a smaller ticket.
    if (p < 6 || p < 2 * K + 2) p = bands(h) + 2;
    return p + (p & 1);     // even: the local slot of an item then has the parity of the workgroup's global slot (static ring slots)
}
inline size_t sv_hand_floats(int h) { return (size_t)2 * records(h) * REC_U * 4; }   // parity x records

}  // namespace v5

// radius 9, and parameters for which no window sum of the matching cost can be tiny without being zero (the
// kernel then needs no exact-division check in stage 1) nor any sum non-finite: every nonzero truncated cost
// term >= 2^-60, eps >= 1
bool v5_supported(const smx_params* p);
bool v5_supported_cost(const smx_params* p);     // the same for materialised cost volumes (values checked in the kernel)
int v5_launch(const v5::Args& a, hipStream_t st);
// comb-ordered guidance planes of one call: G (mean_I, 1/(var+eps)) [h][w] and the image planes FG -> g1p, i2p
int v5_perm_launch(int nviews, const float* const* S0, const float* const* S1, aggdev::f2* const* G, uint8_t* const* mean_u8,
                   const aggdev::fg_t* const* FG, aggdev::f2* const* g1p, unsigned* const* i2p, int w, int h, double eps,
                   hipStream_t st);
// packed-key WTA over `count` comb-ordered q planes (slice slice0 ..) of `nviews` views -> keys [h][w]
// (skip_if != NULL: a device word; the pass does nothing when it is nonzero)
int v5_wta_launch(int nviews, const float* const* q, int64_t* const* keys, int w, int h, int count, int slice0,
                  const unsigned* skip_if, bool fresh, hipStream_t st);

}  // namespace smx
