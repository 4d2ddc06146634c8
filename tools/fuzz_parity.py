"""Randomised parity sweep of the device pipeline against the oracle (dev tool; the committed tests cover
fixed shapes and 12 seeds -- this one walks radius, shape, D, chunking and the cost-volume calling convention)."""
import sys, itertools
import numpy as np
import torch
sys.path.insert(0, ".")
import oracle
import stereo_matching_cuda_amd as smx
from stereo_matching_cuda_amd import synth, _lib
from stereo_matching_cuda_amd.device import PairPipeline

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
bad = 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
BIG = len(sys.argv) > 3          # third argument: larger shapes (more strips, more bands)
for it in range(N):
    R = int(rng.integers(0, 10))
    w = int(rng.choice([int(rng.integers(2, 40)), int(rng.integers(40, 140)), int(rng.integers(140, 330))]))
    h = int(rng.choice([int(rng.integers(1, 12)), int(rng.integers(12, 60)), int(rng.integers(60, 140))]))
    if BIG:
        w, h = int(rng.integers(330, 900)), int(rng.integers(140, 420))
    D = int(rng.integers(1, 7))
    sif = int(rng.integers(1, D + 1))
    p = _lib.default_params()
    p.radius = R
    Il, Ir = synth.gen_pair(w, h, D, int(rng.integers(1, 1 << 30)))
    ref = oracle.stereo_pair(Il, Ir, D, want_agg=True, params=oracle.Params.from_buffer_copy(bytes(p)))
    dl, dr = torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda()
    ok = True
    # both q layouts: the caller's [z][y][x] volume, and the product's default (own scratch -- comb-ordered where the comb
    # walker runs -- with its own WTA pass)
    for want_agg in (True, False):
        pipe = PairPipeline(w, h, D, want_agg=want_agg, params=p, slices_in_flight=sif)
        pipe.run(dl, dr)
        got = pipe.results()
        for k in ("meanl", "meanr", "dmapl", "dmapr", "bestl", "bestr", "occlusion", "filled", "aggl", "aggr"):
            if k not in got or k not in ref:
                continue
            a, b = np.asarray(got[k]), np.asarray(ref[k]).reshape(np.asarray(got[k]).shape)
            neq = (a.view(np.uint32) != b.view(np.uint32)) if a.dtype == np.float32 else (a != b)
            if a.dtype == np.float32:
                neq &= ~(np.isnan(a) & np.isnan(b))
            if neq.any():
                ok = False
                print(f"  MISMATCH {k} (want_agg={want_agg}): {int(neq.sum())} of {neq.size}")
    print(f"R={R} {w}x{h} D={D} sif={sif}: {'OK' if ok else 'MISMATCH'}", flush=True)
    bad += not ok
print("bad", bad)
sys.exit(1 if bad else 0)
