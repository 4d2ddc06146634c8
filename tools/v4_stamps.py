"""Per-wave phase timeline of one work item of k_v4_walk (diagnostic build:
tools/exp_build.sh stamps -DSMX_V4_STAMPS=<item>).  Slots per iteration: start / end of the work of the
W, R, C and X phases (the gaps are barrier waits)."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
os.environ["SMX_LIB_PATH"] = os.path.join("stereo_matching_cuda_amd", "_build_exp", sys.argv[1] if len(sys.argv) > 1 else "stamps", "libsmx_hip.so")
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
NW, SW = 8, 12
NS = SW * 40
buf = np.zeros(NW * NS, np.uint64)
L.smx_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), buf.size)
st = buf.reshape(NW, NS).astype(np.int64)
nit = 0
while nit * SW < NS and st[:, nit * SW].max() > 0:
    nit += 1
print("iterations", nit, "item cycles", st[:, :nit * SW].max() - st[st > 0].min())
for it in range(nit):
    r = st[:, it * SW:(it + 1) * SW]
    nxt = st[:, (it + 1) * SW] if it + 1 < nit else r[:, 7]
    print(f"it {it:2d}: iteration {int(nxt.max() - r[:, 0].min()):6d} cycles")
    for wv in range(NW):
        a = r[wv]
        print(f"   w{wv}: W {a[1]-a[0]:5d} (+wait {a[2]-a[1]:5d})  R {a[3]-a[2]:5d} (+{a[4]-a[3]:5d})  "
              f"C {a[5]-a[4]:5d} (+{a[6]-a[5]:5d})  X {a[7]-a[6]:5d} (+{int(nxt[wv]-a[7]):5d})"
              f"   X: out {a[8]-a[6]:5d} box1 {a[9]-a[8]:5d} box2 {a[10]-a[9]:5d} next {a[7]-a[10]:5d}  R: carry {a[11]-a[2] if a[11] else 0:5d}")
# summary: mean over the interior iterations of the phase times (work + barrier wait, max over waves = the phase)
if nit > 8:
    acc = np.zeros((4,))
    accw = np.zeros((4, NW))
    cnt = 0
    for it in range(3, nit - 3):
        r = st[:, it * SW:(it + 1) * SW]
        nxt = st[:, (it + 1) * SW]
        b = [r[:, 0].min(), r[:, 2].min(), r[:, 4].min(), r[:, 6].min(), nxt.min()]
        acc += np.diff(b)
        accw += np.stack([r[:, 1] - r[:, 0], r[:, 3] - r[:, 2], r[:, 5] - r[:, 4], r[:, 7] - r[:, 6]])
        cnt += 1
    acc /= cnt; accw /= cnt
    print("SUMMARY phase length W R C X (cycles):", " ".join(f"{v:7.0f}" for v in acc), " iteration", f"{acc.sum():7.0f}")
    for ph, nm in enumerate("WRCX"):
        print(f"SUMMARY work of phase {nm} per wave:", " ".join(f"{v:6.0f}" for v in accw[ph]))
