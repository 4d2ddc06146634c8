/*
 * smx_rccl.h -- the multi-GPU exchange step of the stereo path behind a C-ABI (libsmx_rccl.so, built
 * from stereo_matching_cuda_amd/csrc/smx_rccl.cpp; links librccl, libsmx_hip).
 *
 * The reference is single-GPU (main.cu:44-48 hard-wires device 0); this has no counterpart there.
 * The disparity slices of both cost volumes are sharded over the GPUs of one node; the only coupling
 * between slices is the running winner-take-all (dispSelectOnGPU, guidedFilter.cu:403-411), which in
 * packed-key form (winner_take_all.cuh, smx.h: smx_pack_key) is a per-pixel signed-int64 MIN:
 * ONE ncclAllReduce(ncclInt64, ncclMin) over xGMI reassembles the map.
 */
#ifndef SMX_RCCL_H
#define SMX_RCCL_H

#include "smx.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

/* In-place MIN all-reduce of n packed WTA keys on `stream` of the current device.
 * nccl_comm: an ncclComm_t (as void*) whose rank lives on the current device. */
int smx_wta_allreduce(int64_t* d_keys, int64_t n, void* nccl_comm, void* stream);

/* main.cu:65-155 for one pair with the slices [0, size_d) of both volumes sharded over the first
 * `ngpu` devices of this node, driven from ONE host thread (ncclCommInitAll; grouped all-reduce).
 * Device g aggregates slices [g*size_d/ngpu, (g+1)*size_d/ngpu); images are replicated, guidance
 * statistics recomputed per device (deterministic, bit-equal).  Decode + LR check + fill run on
 * device 0.  Host pointers in/out like smx_stereo_pair; cost_* / agg_* outputs must be NULL. */
int smx_stereo_pair_sharded(const smx_params* p, const uint8_t* gray_l, const uint8_t* gray_r, int w,
                            int h, int size_d, int dminl, int dminr, int ngpu, const smx_pair_out* out);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* SMX_RCCL_H */
