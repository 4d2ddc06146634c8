"""Quick GPU check of the fused aggregation against the oracle on a few shapes (dev tool)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import oracle
import stereo_matching_cuda_amd as smx
from stereo_matching_cuda_amd import synth
from stereo_matching_cuda_amd.device import PairPipeline

def run(w, h, D, seed, want_agg=True):
    Il, Ir = synth.gen_pair(w, h, D, seed)
    ref = oracle.stereo_pair(Il, Ir, D, want_agg=want_agg)
    pipe = PairPipeline(w, h, D, want_agg=want_agg)
    dl, dr = torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda()
    pipe.run(dl, dr)
    got = pipe.results()
    ok = True
    for k in ("meanl", "meanr", "dmapl", "dmapr", "bestl", "bestr", "occlusion", "filled", "aggl", "aggr"):
        if k not in got or k not in ref:
            continue
        a, b = np.asarray(got[k]), np.asarray(ref[k]).reshape(np.asarray(got[k]).shape)
        bad = (a.view(np.uint32) != b.view(np.uint32)) if a.dtype == np.float32 else (a != b)
        nb = int(bad.sum())
        if nb:
            ok = False
            idx = np.argwhere(bad)
            print(f"  {k}: {nb} mismatches of {bad.size}; first {idx[:4].tolist()} "
                  f"rows {idx[:,-2].min()}..{idx[:,-2].max()} cols {idx[:,-1].min()}..{idx[:,-1].max()}")
            if a.dtype == np.float32 and "-v" in sys.argv:
                i0 = tuple(idx[0])
                print("    got", a[i0[:-1]][i0[-1]:i0[-1] + 6], "\n    ref", b[i0[:-1]][i0[-1]:i0[-1] + 6],
                      "max abs diff", float(np.nanmax(np.abs(a - b))))
    print(f"{w}x{h} D={D}: {'OK' if ok else 'MISMATCH'}", flush=True)
    return ok

if __name__ == "__main__":
    shapes = [(70, 40, 3, 1), (64, 32, 2, 2), (150, 100, 4, 3), (384, 288, 16, 4), (500, 375, 8, 5)]
    allok = True
    for s in shapes:
        allok &= run(*s)
    if "--kitti" in sys.argv:
        w, h, D = synth.SHAPES["kitti"]
        Il, Ir = synth.gen_pair(w, h, D, 20150101)
        pipe = PairPipeline(w, h, D)
        dl, dr = torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda()
        for _ in range(3):
            pipe.run(dl, dr)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            pipe.run(dl, dr)
        torch.cuda.synchronize()
        print("kitti ms/pair", (time.perf_counter() - t0) / 10 * 1e3)
        pipe.check_status()
    sys.exit(0 if allok else 1)
