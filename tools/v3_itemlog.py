"""Per-item timeline of one fused aggregation launch (diagnostic build: tools/exp_build.sh itemlog -DSMX_V3_ITEMLOG)."""
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
nguid = 2 * K
print(f"items {n}  kernel span {en.max():.1f} us  sum of item time per WG: mean {dur.sum() / 256:.1f} us")
print(f"guidance items: mean {dur[:nguid].mean():.1f} us")
for k in range(K):
    d = dur[nguid + k * nsv: nguid + (k + 1) * nsv]
    s = st[nguid + k * nsv: nguid + (k + 1) * nsv]
    print(f"strip {k:2d}: dur mean {d.mean():6.1f} min {d.min():6.1f} max {d.max():6.1f} us | starts {s.min():7.1f} .. {s.max():7.1f}")
# idle gaps per WG
gaps = []
for g in range(256):
    idx = np.where(wg == g)[0]
    o = np.argsort(st[idx])
    s, e = st[idx][o], en[idx][o]
    gaps.append((s[1:] - e[:-1]).sum() if len(s) > 1 else 0.0)
    if g < 3:
        print(f"wg {g}: {len(idx)} items, first start {s[0]:.1f}, last end {e[-1]:.1f}, busy {dur[idx].sum():.1f}, gaps {gaps[-1]:.1f}")
print(f"per-WG gaps between items: mean {np.mean(gaps):.1f} us; last end: min {min(en[wg == g].max() for g in range(256)):.1f} max {en.max():.1f}")
