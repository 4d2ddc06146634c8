// winner_take_all.cuh -- the reference header of this name is entirely commented out
// (winner_take_all.cuh:2-95); its live winner-take-all is dispSelectOnGPU inside the guided filter
// (guidedFilter.cu:403-411).  Here the header carries the packed-key form of that rule, which is
// what makes the running argmin mergeable across disparity shards / GPUs:
//   key = sord(cost) << 32 | (0xFFFFFFFF - slice) as a signed 64-bit integer,
//   min key == `if (best >= q) { dmap = label; best = q; }` with ascending slices
// and the one exchange step of the multi-GPU path (libsmx_rccl.so, include/smx_rccl.h):
//   smx_wta_allreduce(d_keys, n, comm, stream) = ncclAllReduce(ncclInt64, ncclMin) over xGMI.
#pragma once
#include "SystemIncludes.h"

inline long long wta_pack(float cost, unsigned slice) { return smx_pack_key(cost, slice); }
inline void wta_unpack(long long key, float* cost, unsigned* slice) {
    uint32_t s = 0;
    smx_unpack_key(key, cost, &s);
    if (slice) *slice = s;
}
