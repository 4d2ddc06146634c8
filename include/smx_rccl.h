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

/* The same as a MIN reduce to rank `root` only: the LR check and the filling run on one rank
 * (SURVEY 8e), so the other ranks need not receive the reassembled maps. */
int smx_wta_reduce(int64_t* d_keys, int64_t n, int root, void* nccl_comm, void* stream);

/* Persistent sharded context: main.cu:65-155 for pairs of one shape with the slices [0, size_d) of both
 * volumes sharded over the first `ngpu` devices of this node, driven from ONE host thread.  Created once:
 * the RCCL communicator (ncclCommInitAll), two streams and three events per device, image / key / mean
 * buffers and the aggregation workspace of every device, the decode buffers of device 0.  A pair is then
 * only: upload, aggregate (device g: slices [g*size_d/ngpu, (g+1)*size_d/ngpu); images replicated, guidance
 * statistics recomputed per device: deterministic, bit-equal), ONE exchange step (grouped ncclReduce to
 * device 0 of the 2*n packed keys), decode + LR check + fill on device 0, download.
 * flags: SMX_SHARDED_OVERLAP_VIEWS = one aggregation launch per view, the exchange of the left keys runs on
 *        a second stream under the aggregation of the right volume (pays when the exchange is a visible
 *        part of a pair; costs a second, smaller launch per device);
 *        SMX_SHARDED_ALLREDUCE = all-reduce instead of reduce (every rank ends with the reassembled keys).
 * Host pointers in/out like smx_stereo_pair; cost_* / agg_* outputs must be NULL. */
#define SMX_SHARDED_OVERLAP_VIEWS 1
#define SMX_SHARDED_ALLREDUCE 2
typedef struct smx_sharded_ctx smx_sharded_ctx;
int smx_sharded_create(const smx_params* p, int w, int h, int size_d, int ngpu, int flags, smx_sharded_ctx** ctx);
int smx_sharded_run(smx_sharded_ctx* ctx, const uint8_t* gray_l, const uint8_t* gray_r, int dminl, int dminr,
                    const smx_pair_out* out);
int smx_sharded_destroy(smx_sharded_ctx* ctx);

/* One pair on a temporary context (smx_sharded_create + smx_sharded_run + smx_sharded_destroy): pays the
 * communicator set-up and every allocation per call.  Kept for the reference-style one-shot driver. */
int smx_stereo_pair_sharded(const smx_params* p, const uint8_t* gray_l, const uint8_t* gray_r, int w,
                            int h, int size_d, int dminl, int dminr, int ngpu, const smx_pair_out* out);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* SMX_RCCL_H */
