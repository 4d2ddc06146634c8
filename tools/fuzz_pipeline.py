"""Randomised parity + liveness sweep of the comb walker with MORE work items than workgroup slots (dev tool): a workgroup
then takes several tickets and pipelines them (smx_agg_v5.hip, period()); shapes are drawn so that the strip count K, the
band count and the period fall on both sides of the rules in smx_agg_v5.h period().  Each shape runs `REPS` times (the
failures this hunts are a matter of timing) through the image source and, every other shape, the cost-volume source.
usage: python tools/fuzz_pipeline.py [seed] [shapes] [reps]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import oracle
import stereo_matching_cuda_amd as smx
from stereo_matching_cuda_amd import synth
from stereo_matching_cuda_amd.device import PairPipeline

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 11)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
KEYS = ("dmapl", "dmapr", "bestl", "bestr", "occlusion", "filled")
lib = smx.lib()
bad = 0
for it in range(N):
    K = int(rng.integers(1, 10))
    w = int(rng.integers(152 * (K - 1) + 1, 152 * K + 1))
    h = int(rng.choice([int(rng.integers(1, 30)), int(rng.integers(30, 110)), int(rng.integers(110, 260))]))
    D = int(rng.integers(max(2, 300 // K), max(3, 1100 // K)))          # 2 K D between ~600 and ~2200 items
    D = min(D, 230)
    Il, Ir = synth.gen_pair(w, h, D, int(rng.integers(1, 1 << 30)))
    want = oracle.stereo_pair(Il, Ir, D, want_cost=True)
    dl, dr = torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda()
    ok = True
    lib.smx_set_agg_path(5)
    try:
        for rep in range(REPS):
            src = "cost" if (it & 1) and rep == REPS - 1 else "images"
            pipe = PairPipeline(w, h, D)
            if src == "cost":
                cl, cr = torch.from_numpy(want["costl"]).cuda(), torch.from_numpy(want["costr"]).cuda()
                pipe.aggregate(dl, dr, cl, cr)
                pipe.finish()
            else:
                pipe.run(dl, dr)
            assert lib.smx_last_agg_path() == 5, lib.smx_last_agg_path()
            got = pipe.results()
            for k in KEYS:
                a, b = np.asarray(got[k]), np.asarray(want[k]).reshape(np.asarray(got[k]).shape)
                if (a != b).any():
                    ok = False
                    print(f"  MISMATCH {k} run {rep} ({src}): {int((a != b).sum())} of {a.size}")
    except Exception as e:                                               # a timed-out hand-off raises SmxError
        ok = False
        print(f"  ERROR {type(e).__name__}: {e}")
    finally:
        lib.smx_set_agg_path(0)
    print(f"K={K} {w}x{h} D={D} items={2 * K * D}: {'OK' if ok else 'FAILED'}", flush=True)
    bad += not ok
print("bad", bad)
sys.exit(1 if bad else 0)
