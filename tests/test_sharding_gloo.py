"""The N>1 path on CPU: world_size-2 (and 3) gloo process groups run the product's slice sharding
(shard_range) and its one exchange step (allreduce_min_keys_: MIN all-reduce of packed u64 keys held
as int64 bit patterns).  Per-shard aggregation is done by the oracle here (no GPU in this container);
the merged result must equal the unsharded oracle result bit for bit.  On the GPU box the same merge
is exercised with the HIP kernels (tests/test_gpu_parity.py::test_virtual_shards_*)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stereo_matching_cuda_amd.sharded import allreduce_min_keys_, merge_keys_host, shard_range


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, Il, Ir, D, dmin, out_dir):
    import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        h, w = Il.shape
        s0, s1 = shard_range(D, rank, world)
        keys = np.full((2, h, w), np.int64(orc.KEY_IDENTITY))
        for view, (I, J, dm) in enumerate(((Il, Ir, dmin), (Ir, Il, 0))):
            if s1 > s0:
                cost = orc.cost_volume(I, J, D, dm)
                best, dmap, _, _ = orc.guided_filter(I, cost, dm, s_begin=s0, s_end=s1)
                keys[view] = orc.pack_keys(best, (dmap - dm).astype(np.int64))
        t = torch.from_numpy(keys.copy())
        allreduce_min_keys_(t)
        np.save(os.path.join(out_dir, f"keys{rank}.npy"), t.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_merge_equals_unsharded(tmp_path, golden, orc, world):
    Il = orc.gray(golden["tsukuba0"])[100:164, 120:216].copy()
    Ir = orc.gray(golden["tsukuba1"])[100:164, 120:216].copy()
    D, dmin = 8, -7
    mp.spawn(_worker, args=(world, _free_port(), Il, Ir, D, dmin, str(tmp_path)), nprocs=world,
             join=True)
    want = orc.stereo_pair(Il, Ir, D, dminl=dmin, dminr=0)
    merged = [np.load(tmp_path / f"keys{r}.npy") for r in range(world)]
    for m in merged[1:]:
        assert np.array_equal(m, merged[0])  # all-reduce: every rank holds the same keys
    for view, (dm, side) in enumerate(((dmin, "l"), (0, "r"))):
        best, sl = orc.unpack_keys(merged[0][view])
        assert np.array_equal(best.view(np.uint32), want["best" + side].view(np.uint32))
        assert np.array_equal((sl + dm).astype(np.float32), want["dmap" + side])


def test_shard_ranges_partition_the_slices():
    for D in (1, 7, 16, 192, 280, 512):
        for G in (1, 2, 3, 4, 8, 9, 600):
            r = [shard_range(D, g, G) for g in range(G)]
            assert r[0][0] == 0 and r[-1][1] == D
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_signed_key_order_and_tie_break(orc):
    rng = np.random.default_rng(2)
    cost = np.concatenate([rng.normal(size=64), [0.0, -0.0, 1.5, 1.5, 1.5, -2.0, 3.3961514e38]]).astype(np.float32)
    sl = rng.integers(0, 512, size=cost.size)
    k = orc.pack_keys(cost, sl)
    assert k.dtype == np.int64
    # the signed order of the keys is (cost ascending, slice descending)
    assert np.array_equal(np.argsort(k, kind="stable"),
                          np.lexsort((-sl, np.where(cost == 0, np.float32(0), cost))))
    # equal cost: the larger slice has the smaller key (dispSelect `>=`: later slice wins)
    a = orc.pack_keys(np.float32([1.5, 1.5]), [3, 9])
    assert a[1] < a[0]
    # smaller cost always wins regardless of slice
    b = orc.pack_keys(np.float32([1.25, 1.5]), [0, 511])
    assert b[0] < b[1]
    # identity element
    m = merge_keys_host([k, np.full_like(k, np.int64(orc.KEY_IDENTITY))])
    assert np.array_equal(m, k)


def test_allreduce_is_identity_without_a_group():
    t = torch.tensor([5, -3, 0], dtype=torch.int64)
    u = t.clone()
    allreduce_min_keys_(u)
    assert torch.equal(t, u)
