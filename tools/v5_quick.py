"""Quick GPU check of the comb walker (smx_agg_v5.hip, path 5) against the oracle on a few shapes, and an A/B
timing of the two fused walkers on the KITTI shape (dev tool; run on the GPU box)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import oracle
import stereo_matching_cuda_amd as smx
from stereo_matching_cuda_amd import synth
from stereo_matching_cuda_amd.device import PairPipeline

KEYS = ("meanl", "meanr", "aggl", "aggr", "dmapl", "dmapr", "bestl", "bestr", "occlusion", "filled")


def run(w, h, D, seed, path=5, want_agg=True):
    Il, Ir = synth.gen_pair(w, h, D, seed)
    ref = oracle.stereo_pair(Il, Ir, D, want_agg=True)
    pipe = PairPipeline(w, h, D, want_agg=want_agg)
    dl, dr = torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda()
    smx.check(smx.lib().smx_set_agg_path(path))
    try:
        pipe.run(dl, dr)
        got = pipe.results()
    finally:
        smx.lib().smx_set_agg_path(0)
    ok = True
    for k in KEYS:
        if k not in got:
            continue
        a, b = np.asarray(got[k]), np.asarray(ref[k]).reshape(np.asarray(got[k]).shape)
        bad = (a.view(np.uint32) != b.view(np.uint32)) if a.dtype == np.float32 else (a != b)
        nb = int(bad.sum())
        if nb:
            ok = False
            idx = np.argwhere(bad)
            print(f"  {k}: {nb} mismatches of {bad.size}; first {idx[:4].tolist()} "
                  f"rows {idx[:,-2].min()}..{idx[:,-2].max()} cols {idx[:,-1].min()}..{idx[:,-1].max()}")
            if a.dtype == np.float32:
                i0 = tuple(idx[0])
                print("    got", a[i0[:-1]][i0[-1]:i0[-1] + 6], "\n    ref", b[i0[:-1]][i0[-1]:i0[-1] + 6],
                      "max abs diff", float(np.nanmax(np.abs(a - b))))
            if k.startswith("agg"):
                cols = np.unique(idx[:, -1]); rows = np.unique(idx[:, -2]); sl = np.unique(idx[:, 0])
                print("    slices", sl[:8], "rows", rows[:12], "cols", cols[:24], "ncols", cols.size, "nrows", rows.size)
    print(f"{w}x{h} D={D} path {path} agg={want_agg}: {'OK' if ok else 'MISMATCH'}", flush=True)
    return ok


def timeit(path, w, h, D, n=20):
    Il, Ir = synth.gen_pair(w, h, D, 20150101)
    pipe = PairPipeline(w, h, D)
    dl, dr = torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda()
    smx.check(smx.lib().smx_set_agg_path(path))
    try:
        for _ in range(3):
            pipe.run(dl, dr)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for i in range(n):
            pipe.run(dl, dr)
            ev[i + 1].record()
        torch.cuda.synchronize()
        pipe.check_status()
    finally:
        smx.lib().smx_set_agg_path(0)
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    print(f"path {path} {w}x{h}x{D}: ms/pair median {ts[n // 2]:.3f} min {ts[0]:.3f} max {ts[-1]:.3f}", flush=True)


def occupancy():
    import ctypes as C
    from stereo_matching_cuda_amd import _lib
    L = C.CDLL(_lib.SO_PATH)
    nb, vg, lds = C.c_int(), C.c_int(), C.c_int()
    rc = L.smx_debug_v5_occupancy(C.byref(nb), C.byref(vg), C.byref(lds))
    print(f"k_v5_walk: rc {rc} blocks per CU {nb.value} VGPRs {vg.value} LDS {lds.value} B", flush=True)


if __name__ == "__main__":
    if "--occupancy" in sys.argv:
        torch.zeros(1).cuda()
        occupancy()
    shapes = [(70, 40, 3, 1), (152, 30, 2, 2), (153, 21, 2, 6), (384, 288, 16, 4), (600, 95, 4, 3), (1242, 64, 3, 5)]
    allok = True
    if "--time-only" in sys.argv:
        shapes = []
    for s in shapes:
        allok &= run(*s)
        allok &= run(*s, want_agg=False)      # the comb-ordered q scratch + its WTA pass
        if not allok and "--all" not in sys.argv:
            break
    if allok and "--kitti" in sys.argv:
        w, h, D = synth.SHAPES["kitti"]
        timeit(3, w, h, D)
        timeit(5, w, h, D)
    sys.exit(0 if allok else 1)
