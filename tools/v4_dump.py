"""Compare the LDS rings of one work item of k_v4_walk with the oracle (diagnostic build:
tools/exp_build.sh dump -DSMX_V4_DUMP=<item> -DSMX_V4_DUMP_IT=<i> -DSMX_V4_DUMP_PH=<0..3>).
usage: v4_dump.py w h D item it ph"""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
os.environ["SMX_LIB_PATH"] = os.path.join("stereo_matching_cuda_amd", "_build_exp", "dump", "libsmx_hip.so")
os.environ["SMX_ALLOW_LIB_OVERRIDE"] = "1"
import oracle
import stereo_matching_cuda_amd as smx
from stereo_matching_cuda_amd import synth
from stereo_matching_cuda_amd.device import PairPipeline
w, h, D, item, it, ph = [int(a) for a in sys.argv[1:7]]
BH = int(os.environ.get('SMX_V4_BH', '16'))
R, RR, ROWF, OFF1, OW = 9, BH + 20, 172, 88, 64
Il, Ir = synth.gen_pair(w, h, D, 1)
pipe = PairPipeline(w, h, D)
pipe.run(torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda())
torch.cuda.synchronize()
L = C.CDLL(os.environ["SMX_LIB_PATH"])
buf = np.zeros(2 * RR * ROWF, np.float32)
L.smx_debug_read_dump(buf.ctypes.data_as(C.c_void_p), buf.size)
rings = buf.reshape(2, RR, ROWF)
K = (w + R + OW - 1) // OW
nsv = 2 * D
k, sv = divmod(item, nsv)
view, sl = divmod(sv, D)
I1, I2 = (Il, Ir) if view == 0 else (Ir, Il)
dmin = -(D - 1) if view == 0 else 0
cost = oracle.cost_volume(I1, I2, D, dmin)[sl]
p = cost.astype(np.float32)
Ip = (I1.astype(np.float32) * p).astype(np.float32)
xs = k * OW
cs1 = xs - R - 1
y0 = BH * it
print(f"item {item}: view {view} slice {sl} strip {k}; iteration {it} phase {ph}; ring-1 band rows {y0}..{y0+BH-1}")
def cmp(name, exp, ring, rows, y_of, cols, x_of):
    bad = 0
    for r in rows:
        y = y_of(r)
        if y < 0 or y >= h: continue
        for j in cols:
            x = x_of(j)
            if x < 0 or x >= w: continue
            rr_ = rings[ring, (y + (R if ring == 1 else 0)) % RR]
            got = np.array([rr_[j], rr_[OFF1 + j]], np.float32)
            e = np.array([exp[0][y, x], exp[1][y, x]], np.float32)
            if got.view(np.uint32).tolist() != e.view(np.uint32).tolist():
                if bad < 6: print(f"  {name} mismatch y={y} x={x} (ring col {j}): got {got} want {e}")
                bad += 1
    print(f"{name}: {bad} mismatches")
band = range(BH)
if ph == 0:
    cmp("cost (p, Ip)", (p, Ip), 0, band, lambda r: y0 + r, range(OW + 2 * R + 1), lambda j: cs1 + j)
else:
    # row prefix (ph 1) / integral (ph 2, 3) over the whole image: the oracle's order
    def rowpre(a):
        o = np.zeros_like(a)
        acc = np.zeros(a.shape[0], np.float32)
        for x in range(a.shape[1]):
            acc = (acc + a[:, x]).astype(np.float32) if x else a[:, 0].copy()
            o[:, x] = acc
        return o
    Rp, RIp = rowpre(p), rowpre(Ip)
    if ph == 1:
        cmp("row prefix", (Rp, RIp), 0, band, lambda r: y0 + r, range(OW + 2 * R + 1), lambda j: cs1 + j)
    else:
        Sp, SIp = oracle.integral(p), oracle.integral(Ip)
        cmp("integral", (Sp, SIp), 0, band, lambda r: y0 + r, range(OW + 2 * R + 1), lambda j: cs1 + j)
