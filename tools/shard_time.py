"""Per-rank time of a disparity-sharded pair, measured on ONE GPU: the pipeline of a rank that owns D/N of the D
slices (everything but the exchange of the keys).  Input to the projected scaling table of DESIGN.md -- nothing here
runs on several GPUs.   usage: python tools/shard_time.py [workload] [N ...]"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import stereo_matching_cuda_amd as smx
from stereo_matching_cuda_amd import synth
from stereo_matching_cuda_amd.device import PairPipeline
args = sys.argv[1:]
wl = args.pop(0) if args and not args[0].isdigit() else 'kitti'
w,h,D = synth.SHAPES[wl]
Il,Ir = synth.gen_pair(w,h,D,synth.SEEDS.get(wl, 1))
dl,dr = torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda()
reps = 100 if wl == 'kitti' else 5
for N in [int(a) for a in args] or (1,2,4,8):
    pipe = PairPipeline(w,h,D,s_begin=0,s_end=D//N)
    for _ in range(10): pipe.run(dl,dr)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(reps): pipe.run(dl,dr)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/reps*1e3
    pipe.check_status()
    print(wl,'N',N,'slices',D//N,'ms/step (no all-reduce)',round(dt,3), 'keys MB', round(2*w*h*8/1e6,2), flush=True)
    del pipe
    torch.cuda.empty_cache()
