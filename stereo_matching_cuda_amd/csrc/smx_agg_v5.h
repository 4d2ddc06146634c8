// smx_agg_v5.h -- interface of the comb-form aggregation kernel (smx_agg_v5.hip) towards the host orchestration
// of smx_agg_v4.hip: argument block, strip / band / record geometry, eligibility.
#pragma once
#include "smx_agg_dev.h"

namespace smx {
namespace v5 {

constexpr int OWS = 285;                // output columns per strip (19 combs x 15 outputs)
constexpr int BH = 10;                  // band height
constexpr int REC_U = 105;              // 16-byte units per hand-off record
constexpr int WG_PER_CU = 2;

struct Args {
    // the fixed part of the workspace: both image planes [h][w + 2 PADX] of k_v4_prep and the guidance planes
    // (mean_I, 1/(var_I + eps)) [h][w] of both views, addressed through ONE buffer descriptor
    const char* fix;
    size_t fix_bytes;
    unsigned o_fg[2];     // byte offset of view v's image plane (the other view's is o_fg[v ^ 1])
    unsigned o_guid[2];   // byte offset of view v's guidance plane
    float* q[2];          // out: [slice][h][w] per view
    int d0[2];            // disparity of local slice 0 per view
    int w, h, K, NI, nslices, nsv, nitems;
    float* hand;          // hand-off records [parity][sv][iteration][REC_U x 4 floats]
    unsigned* flags;      // [sv][K]  published-record counters (zeroed before every launch)
    unsigned* ticket;     // work-item counter              (zeroed before every launch)
    unsigned* status;     // != 0: a flag wait timed out (results invalid)
    CostConst cc;
};

inline int strips(int w) { return (w + OWS - 1) / OWS; }
inline int bands(int h) { return (h + 2 * 9 + BH - 1) / BH + 1; }     // the q rows of iteration i end at 10 i - 18
inline size_t sv_hand_floats(int h) { return (size_t)2 * bands(h) * REC_U * 4; }   // parity x records

}  // namespace v5

// radius 9, and parameters for which no window sum of the matching cost can be tiny without being zero (the
// kernel then needs no exact-division check in stage 1) nor any sum non-finite: every nonzero truncated cost
// term >= 2^-60, eps >= 1
bool v5_supported(const smx_params* p);
int v5_launch(const v5::Args& a, hipStream_t st);

}  // namespace smx
