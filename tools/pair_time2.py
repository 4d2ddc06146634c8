"""ms per KITTI-shape pair with TWO pairs in flight: two device-resident pipelines (own workspace, keys, results) on two
streams, steps alternate between them -- the small kernels of one pair (key preset, guidance, WTA pass, finish) and the ramp-up
of its walker fill the CUs the other pair's persistent walker leaves idle in its tail (dev tool, GPU box).
python tools/pair_time2.py [workload] [repeats]"""
import sys, time
import torch
sys.path.insert(0, ".")
import stereo_matching_cuda_amd as smx
from stereo_matching_cuda_amd import synth
from stereo_matching_cuda_amd.device import PairPipeline
wl = sys.argv[1] if len(sys.argv) > 1 else "kitti"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
NF = int(sys.argv[3]) if len(sys.argv) > 3 else 2
w, h, D = synth.SHAPES[wl]
Il, Ir = synth.gen_pair(w, h, D, synth.SEEDS.get(wl, 1))
N = max(10, int(300 * (1242 * 375 * 192) / (float(w) * h * D)))
pipes = [PairPipeline(w, h, D) for _ in range(NF)]
streams = [torch.cuda.Stream() for _ in range(NF)]
dl, dr = torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda()
torch.cuda.synchronize()
def step(i):
    p = pipes[i % NF]
    with torch.cuda.stream(streams[i % NF]):
        p.init_keys(); p.aggregate_pair(dl, dr); p.finish()
for i in range(N): step(i)
torch.cuda.synchronize()
out = []
for _ in range(reps):
    t0 = time.perf_counter()
    for i in range(N): step(i)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / N * 1e3)
for p in pipes: p.check_status()
print(wl, "pairs in flight", NF, "ms/pair", " ".join(f"{v:.4f}" for v in out), flush=True)
