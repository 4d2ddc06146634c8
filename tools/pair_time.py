"""ms per KITTI-shape pair of the device-resident pipeline without any event in the loop (dev tool, GPU box):
python tools/pair_time.py [path] [repeats] [workload]   -- SMX_LIB_PATH + SMX_ALLOW_LIB_OVERRIDE=1 select a variant build"""
import sys, time
import torch
sys.path.insert(0, ".")
import stereo_matching_cuda_amd as smx
from stereo_matching_cuda_amd import synth
from stereo_matching_cuda_amd.device import PairPipeline
path = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
wl = sys.argv[3] if len(sys.argv) > 3 else "kitti"
import os
if "," in wl:                                   # "w,h,D": an ad-hoc shape
    w, h, D = (int(v) for v in wl.split(","))
else:
    w, h, D = synth.SHAPES[wl]
Il, Ir = synth.gen_pair(w, h, D, synth.SEEDS.get(wl, 1))
N = max(10, int(300 * (1242 * 375 * 192) / (float(w) * h * D)))
sif = int(os.environ["SMX_SIF"]) if "SMX_SIF" in os.environ else None       # slices per walker launch (A/B runs)
pipe = PairPipeline(w, h, D, slices_in_flight=sif)
dl, dr = torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda()
smx.check(smx.lib().smx_set_agg_path(path))
cost = pipe.cost_volumes(dl, dr) if os.environ.get("SMX_SRC", "") == "cost" else None     # materialised cost volumes (A/B runs)
def step():
    if cost is None: pipe.aggregate(dl, dr)
    else: pipe.aggregate(dl, dr, cost[0], cost[1])
    pipe.finish()
for _ in range(N): step()
torch.cuda.synchronize()
out = []
for _ in range(reps):
    t0 = time.perf_counter()
    for _ in range(N): step()
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / N * 1e3)
pipe.check_status()
print(wl, "path", path, "chunk", pipe.last_chunk(), "ms/pair", " ".join(f"{v:.4f}" for v in out), flush=True)
