"""Disparity-slice sharding across ranks (one process per GPU) and the one exchange step of the
path: a per-pixel MIN all-reduce of packed (cost, slice) keys (RCCL over xGMI with backend "nccl",
gloo on CPU).  No reference counterpart (the reference is single-GPU); SURVEY.md 8e.

Rank g of G owns the contiguous slice range [g*D//G, (g+1)*D//G) of BOTH volumes; the two gray
images are replicated and the guidance statistics recomputed per rank (deterministic, bit-equal).
The signed min of the packed keys is exactly the sequential `best >= q` rule of dispSelectOnGPU
(guidedFilter.cu:403-411): smallest cost, and among equal costs the largest slice.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(size_d, rank, world):
    """Contiguous, balanced, possibly empty slice range of `rank`."""
    return (rank * size_d) // world, ((rank + 1) * size_d) // world


def allreduce_min_keys_(keys_i64, group=None):
    """In-place MIN all-reduce of packed int64 keys (the kernels emit them in signed order, so this
    is the collective and nothing else).  Device tensors go through RCCL (backend "nccl"); with a
    gloo group (CPU rehearsal of the N>1 path, several ranks sharing one GPU) they are staged
    through host memory."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if keys_i64.is_cuda and dist.get_backend(group) == "gloo":
            host = keys_i64.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.MIN, group=group)
            keys_i64.copy_(host)
        else:
            dist.all_reduce(keys_i64, op=dist.ReduceOp.MIN, group=group)
    return keys_i64


def merge_keys_host(key_arrays):
    """Reference merge for tests: elementwise signed min over a list of numpy int64 arrays."""
    out = np.asarray(key_arrays[0], dtype=np.int64).copy()
    for k in key_arrays[1:]:
        np.minimum(out, np.asarray(k, dtype=np.int64), out=out)
    return out


class ShardedPair:
    """D-sharded stereo pair: local aggregation of this rank's slices, ONE all-reduce of both
    views' keys, then decode + LR check + filling (replicated on every rank: n-sized, microseconds)."""

    def __init__(self, w, h, size_d, rank=0, world=1, group=None, **kw):
        from .device import PairPipeline
        self.rank, self.world, self.group = rank, world, group
        s0, s1 = shard_range(size_d, rank, world)
        self.pipe = PairPipeline(w, h, size_d, s_begin=s0, s_end=s1, **kw)

    def run(self, gray_l, gray_r):
        self.pipe.aggregate(gray_l, gray_r)
        if self.world > 1:
            allreduce_min_keys_(self.pipe.keys, self.group)
        self.pipe.finish()
        return self.pipe
