"""Print the per-wave barrier timeline of one work item (diagnostic build, tools/v3_stamps.sh)."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
os.environ["SMX_LIB_PATH"] = os.path.join("stereo_matching_cuda_amd", "_build_stamps", "libsmx_hip.so")
os.environ["SMX_ALLOW_LIB_OVERRIDE"] = "1"
import stereo_matching_cuda_amd as smx
from stereo_matching_cuda_amd import synth
from stereo_matching_cuda_amd.device import PairPipeline
w, h, D = synth.SHAPES["kitti"]
Il, Ir = synth.gen_pair(w, h, D, 20150101)
pipe = PairPipeline(w, h, D)
dl, dr = torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda()
for _ in range(3):
    pipe.run(dl, dr)
torch.cuda.synchronize()
L = C.CDLL(os.environ["SMX_LIB_PATH"])
NW, NS = 16, 4 * 96
buf = np.zeros(NW * NS, np.uint64)
L.smx_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), buf.size)
st = buf.reshape(NW, NS).astype(np.int64)
t0 = st[st > 0].min()
nit = 0
while nit * 4 < NS and st[:, nit * 4].max() > 0:
    nit += 1
print("iterations", nit)
names = {0: "R", 1: "C"}
for it in range(nit):
    row = st[:, it * 4:(it + 1) * 4] - t0
    # arrive at barrier A (slot0), leave (slot1), arrive at barrier B (slot2), leave (slot3)
    print(f"it {it-2:3d} | leaveB_prev->arriveA (work A) / wait A / work B / wait B per wave (cycles)")
    for wv in range(NW):
        prev = st[wv, it * 4 - 1] - t0 if it > 0 else row[wv, 0]
        print(f"   w{wv:2d}{names.get(wv,'B')}: workA {row[wv,0]-prev:6d} waitA {row[wv,1]-row[wv,0]:6d} "
              f"workB {row[wv,2]-row[wv,1]:6d} waitB {row[wv,3]-row[wv,2]:6d}")
