"""ms per pair through the HOST-pointer entry (smx_create / smx_ctx_stereo_pair / smx_destroy): uploads of both
gray images, the whole path, downloads of the eight result planes (no cost / aggregated volumes), on a persistent
context -- the PCIe-inclusive rate that DESIGN.md quotes next to bench.py's device-resident one.
usage: python tools/pcie_pair.py [workload] [pairs]"""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
import stereo_matching_cuda_amd as smx
from stereo_matching_cuda_amd import synth
from stereo_matching_cuda_amd._lib import PairOut
wl = sys.argv[1] if len(sys.argv) > 1 else "kitti"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
w, h, D = synth.SHAPES[wl]
Il, Ir = synth.gen_pair(w, h, D, synth.SEEDS.get(wl, 1))
L = smx.lib()
params = smx.default_params()
ctx = C.c_void_p()
smx.check(L.smx_create(C.byref(params), w, h, D, C.byref(ctx)))
n = w * h
bufs = {k: np.empty(n, np.float32) for k in ("best_l", "best_r", "dmap_l", "dmap_r", "occlusion", "filled")}
bufs.update({k: np.empty(n, np.uint8) for k in ("mean_l", "mean_r")})
out = PairOut()
for k, a in bufs.items():
    setattr(out, k, a.ctypes.data)
run = lambda: smx.check(L.smx_ctx_stereo_pair(ctx, Il.ctypes.data, Ir.ctypes.data, -(D - 1), 0, C.byref(out)))
for _ in range(3):
    run()
t0 = time.perf_counter()
for _ in range(K):
    run()
dt = (time.perf_counter() - t0) / K
# pipelined entry: pair k+1 goes up and pair k-1 comes down under the aggregation of pair k; results are read through the
# staged pointers (pinned staging of the context)
staged = PairOut()
def run_async(K):
    smx.check(L.smx_ctx_stereo_pair_async(ctx, Il.ctypes.data, Ir.ctypes.data, -(D - 1), 0))
    for _ in range(K - 1):
        smx.check(L.smx_ctx_stereo_pair_async(ctx, Il.ctypes.data, Ir.ctypes.data, -(D - 1), 0))
        smx.check(L.smx_ctx_wait(ctx, C.byref(staged), None))
    smx.check(L.smx_ctx_wait(ctx, C.byref(staged), None))
run_async(5)
t0 = time.perf_counter()
run_async(K)
dta = (time.perf_counter() - t0) / K
got = np.ctypeslib.as_array(C.cast(staged.filled, C.POINTER(C.c_float)), (n,))
assert np.array_equal(got.view(np.uint32), bufs["filled"].view(np.uint32)), "pipelined != synchronous"
smx.check(L.smx_destroy(ctx))
print(f"{wl} {w}x{h} D={D}: {dta * 1e3:.3f} ms per pair through smx_ctx_stereo_pair_async / smx_ctx_wait (pinned staging, two pairs "
      f"in flight, results read in the staging), {w * h / dta / 1e6:.1f} MPix/s")
print(f"{wl} {w}x{h} D={D}: {dt * 1e3:.3f} ms per pair through host pointers (pageable numpy buffers, synchronous call), "
      f"{w * h / dt / 1e6:.1f} MPix/s; bytes over PCIe per pair: {2 * n + 6 * 4 * n + 2 * n}")
