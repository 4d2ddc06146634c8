#!/bin/bash
# dev tool (GPU box): A/B timing (KITTI shape, tools/v5_quick.py --kitti) of every variant under _build_exp/ whose name
# starts with x5 (what-if builds give wrong results by construction: only the timing lines count)
cd $GRAFT_REPO_ROOT
for d in stereo_matching_cuda_amd/_build_exp/x5*/; do
  n=$(basename $d)
  SMX_ALLOW_LIB_OVERRIDE=1 SMX_LIB_PATH=$PWD/$d/libsmx_hip.so timeout -k 10 120 python tools/v5_quick.py --time-only --kitti > gpurun_out/exp_$n.log 2>&1
  echo "$n: $(grep -c ' OK' gpurun_out/exp_$n.log) ok; $(grep 'path 5 1242' gpurun_out/exp_$n.log)"
done
# residency check: one workgroup per CU instead of two
SMX_V5_WG_PER_CU=1 timeout -k 10 120 python tools/v5_quick.py --time-only --kitti 2>&1 | grep "path 5" | sed 's/^/wg_per_cu=1: /'
