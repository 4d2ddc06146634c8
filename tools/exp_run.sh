#!/bin/bash
# dev tool (GPU box): time every variant under _build_exp/ with tools/v3_quick.py --kitti
cd $GRAFT_REPO_ROOT
for d in stereo_matching_cuda_amd/_build_exp/*/; do
  n=$(basename $d)
  SMX_ALLOW_LIB_OVERRIDE=1 SMX_LIB_PATH=$PWD/$d/libsmx_hip.so timeout -k 10 120 python tools/v3_quick.py --kitti > gpurun_out/exp_$n.log 2>&1
  echo "$n: $(grep -c OK gpurun_out/exp_$n.log) ok, $(grep kitti gpurun_out/exp_$n.log)"
done
