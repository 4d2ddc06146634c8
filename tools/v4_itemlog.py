"""Per-item timeline of one k_v4_walk launch (diagnostic build: tools/exp_build.sh itemlog -DSMX_V4_ITEMLOG):
how long items of each strip take, when they start, how much of the launch is fill / drain."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
os.environ["SMX_LIB_PATH"] = os.path.join(os.getcwd(), "stereo_matching_cuda_amd", "_build_exp", "itemlog", "libsmx_hip.so")
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
N = 1 << 16
buf = np.zeros(3 * N, np.uint64)
L.smx_debug_read_itemlog(buf.ctypes.data_as(C.c_void_p), buf.size)
lg = buf.reshape(N, 3).astype(np.int64)
n = int((lg[:, 0] > 0).sum())
lg = lg[:n]
t0 = lg[:, 0].min()
st, en, wg = (lg[:, 0] - t0) / 100.0, (lg[:, 1] - t0) / 100.0, lg[:, 2]   # 100 MHz counter -> us
dur = en - st
K = (w + 9 + 63) // 64
nsv = 2 * D
nwg = int(wg.max()) + 1
span = en.max()
print(f"items {n}  workgroups {nwg}  kernel span {span:.1f} us  busy fraction of the workgroup slots {dur.sum() / (nwg * span):.3f}")
for k in range(K):
    d = dur[k * nsv:(k + 1) * nsv]
    s = st[k * nsv:(k + 1) * nsv]
    e = en[k * nsv:(k + 1) * nsv]
    print(f"strip {k:2d}: item time mean {d.mean():6.1f} min {d.min():6.1f} max {d.max():6.1f} us | starts {s.min():7.1f} .. {s.max():7.1f} | ends .. {e.max():7.1f}")
last = np.array([en[wg == g].max() for g in range(nwg)])
first = np.array([st[wg == g].min() for g in range(nwg)])
print(f"first item start per workgroup: max {first.max():.1f} us;  last item end per workgroup: min {last.min():.1f} mean {last.mean():.1f} max {last.max():.1f} us")
print(f"drain: idle workgroup-time behind the last item = {(span - last).sum() / (nwg * span):.3f} of the launch")
# running workgroups over time
ts = np.linspace(0, span, 21)
act = [(int(((st <= t) & (en > t)).sum())) for t in ts]
print("items in flight at 5% steps:", act)
