import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import stereo_matching_cuda_amd as smx
from stereo_matching_cuda_amd import synth
from stereo_matching_cuda_amd.device import PairPipeline
w,h,D = synth.SHAPES['kitti']
Il,Ir = synth.gen_pair(w,h,D,synth.SEEDS['kitti'])
dl,dr = torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda()
for N in [int(a) for a in sys.argv[1:]] or (1,2,4,8):
    pipe = PairPipeline(w,h,D,s_begin=0,s_end=D//N)
    for _ in range(3): pipe.run(dl,dr)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(20): pipe.run(dl,dr)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/20*1e3
    print('N',N,'slices',D//N,'ms/step (no all-reduce)',round(dt,3))
