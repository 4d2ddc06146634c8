"""Per-wave phase timeline of one WORKGROUP of k_v5_walk over 48 of its global slots (diagnostic build:
tools/exp_build.sh stamps5 -DSMX_V5_STAMPS=<workgroup> [-DSMX_V5_STAMP_G0=<first global slot>]; since the items of a workgroup are
pipelined a slot no longer belongs to one item).  Slots per iteration: start / end of the work of the W, R and X phases (the
gaps are barrier waits).  The stamps themselves cost more than in round 4 (every wave of every workgroup evaluates their
condition): use them for the shape of a slot, not for its length."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
os.environ["SMX_LIB_PATH"] = os.path.join("stereo_matching_cuda_amd", "_build_exp", sys.argv[1] if len(sys.argv) > 1 else "stamps5", "libsmx_hip.so")
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
NS = SW * 48
buf = np.zeros(NW * NS, np.uint64)
L.smx_debug_read_stamps5(buf.ctypes.data_as(C.c_void_p), buf.size)
st = buf.reshape(NW, NS).astype(np.int64)
nit = 0
while nit * SW < NS and st[:, nit * SW].max() > 0:
    nit += 1
print("iterations", nit, "item cycles", st[:, :nit * SW].max() - st[st > 0].min())
for it in range(nit):
    r = st[:, it * SW:(it + 1) * SW]
    nxt = st[:, (it + 1) * SW] if it + 1 < nit else r[:, 5]
    if it < 6 or it > nit - 4 or "-v" in sys.argv:
        print(f"it {it:2d}: iteration {int(nxt.max() - r[:, 0].min()):6d} cycles")
        for wv in range(NW):
            a = r[wv]
            fine = ("  pairs end at " + " ".join(f"{a[6 + j] - a[4]:5d}" for j in range(5)) + f"  stamp11 {a[11] - a[4]:5d}") if wv < 6 and a[6] > 0 else ""
            print(f"   w{wv}: W {a[1]-a[0]:5d} (+wait {a[2]-a[1]:5d})  R {a[3]-a[2]:5d} (+{a[4]-a[3]:5d})  X {a[5]-a[4]:5d} (+{int(nxt[wv]-a[5]):5d}){fine}")
if nit > 8:
    acc = np.zeros((3,))
    accw = np.zeros((3, NW))
    cnt = 0
    for it in range(4, nit - 4):
        r = st[:, it * SW:(it + 1) * SW]
        nxt = st[:, (it + 1) * SW]
        b = [r[:, 0].min(), r[:, 2].min(), r[:, 4].min(), nxt.min()]
        acc += np.diff(b)
        accw += np.stack([r[:, 1] - r[:, 0], r[:, 3] - r[:, 2], r[:, 5] - r[:, 4]])
        cnt += 1
    acc /= cnt; accw /= cnt
    print("SUMMARY phase length W R X (cycles):", " ".join(f"{v:7.0f}" for v in acc), " iteration", f"{acc.sum():7.0f}")
    for ph, nm in enumerate("WRX"):
        print(f"SUMMARY work of phase {nm} per wave:", " ".join(f"{v:6.0f}" for v in accw[ph]))

if nit > 8:
    # fine stamps of the comb waves: end of each of the five row pairs, relative to the start of the rows (stamp 4)
    fine = np.zeros((NW, 6)); cnt = 0
    for it in range(6, nit - 6):
        r = st[:, it * SW:(it + 1) * SW]
        if (r[:6, 6:11] > 0).all():
            fine[:6, :5] += r[:6, 6:11] - r[:6, 4:5]
            fine[:6, 5] += r[:6, 5] - r[:6, 10]
            cnt += 1
    if cnt:
        fine /= cnt
        for wv in range(6):
            print(f"SUMMARY comb wave {wv}: end of row pair 1..5 after the start of the rows:", " ".join(f"{v:6.0f}" for v in fine[wv, :5]), f"  rest of the slot {fine[wv, 5]:6.0f}")

if nit > 8:
    print("stage-1 wave 0, slots 8..19: rows end -> guidance issued (stamp 11) -> end of the slot function (stamp 5) -> next slot starts; cycles")
    for it in range(8, min(20, nit - 1)):
        r = st[0, it * SW:(it + 1) * SW]; nx = st[0, (it + 1) * SW]
        print(f"  slot {it:2d}: guid {r[11]-r[10]:5d}  tail {r[5]-r[11]:5d}  barrier {nx-r[5]:5d}   (slot {nx - r[0]:5d})")
