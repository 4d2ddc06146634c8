"""ms per KITTI-shape pair of the device-resident pipeline without any event in the loop (dev tool, GPU box):
python tools/pair_time.py [path] [repeats]   -- SMX_LIB_PATH + SMX_ALLOW_LIB_OVERRIDE=1 select a variant build"""
import sys, time
import torch
sys.path.insert(0, ".")
import stereo_matching_cuda_amd as smx
from stereo_matching_cuda_amd import synth
from stereo_matching_cuda_amd.device import PairPipeline
path = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w, h, D = synth.SHAPES["kitti"]
Il, Ir = synth.gen_pair(w, h, D, 20150101)
pipe = PairPipeline(w, h, D)
dl, dr = torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda()
smx.check(smx.lib().smx_set_agg_path(path))
def step():
    pipe.init_keys(); pipe.aggregate_pair(dl, dr); pipe.finish()
for _ in range(300): step()
torch.cuda.synchronize()
out = []
for _ in range(reps):
    t0 = time.perf_counter()
    for _ in range(300): step()
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / 300 * 1e3)
pipe.check_status()
print("path", path, "ms/pair", " ".join(f"{v:.4f}" for v in out), flush=True)
