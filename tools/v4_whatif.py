"""Kernel time of k_v4_walk for the what-if variants under _build_exp/wi<bits> (tools/exp_build.sh wi<b>
-DSMX_V4_WHATIF=<b>): which part of the work is the kernel time sensitive to?  (Results of these builds are
wrong by construction; only the timing means something.)  usage: v4_whatif.py <variant dir names...>"""
import os, subprocess, sys
code = r'''
import sys, time, torch
sys.path.insert(0, ".")
import stereo_matching_cuda_amd as smx
from stereo_matching_cuda_amd import synth
from stereo_matching_cuda_amd.device import PairPipeline
w, h, D = synth.SHAPES["kitti"]
Il, Ir = synth.gen_pair(w, h, D, 20150101)
dl, dr = torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda()
pipe = PairPipeline(w, h, D)
for _ in range(3): pipe.run(dl, dr)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): pipe.run(dl, dr)
torch.cuda.synchronize(); print("%.3f" % ((time.perf_counter() - t0) / 30 * 1e3))
'''
for v in sys.argv[1:] or ["."]:
    env = dict(os.environ)
    if v != ".":
        env["SMX_LIB_PATH"] = os.path.join("stereo_matching_cuda_amd", "_build_exp", v, "libsmx_hip.so")
        env["SMX_ALLOW_LIB_OVERRIDE"] = "1"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    print(f"{v:10s} kitti ms/pair {r.stdout.strip() or r.stderr[-300:]}", flush=True)
