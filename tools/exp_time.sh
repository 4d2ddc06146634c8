#!/bin/bash
# dev tool (GPU box): event-free KITTI timing (tools/pair_time.py) of the product build and of every variant _build_exp/x5*
cd $GRAFT_REPO_ROOT
echo "product: $(timeout -k 10 120 python tools/pair_time.py 0 2 ${1:-kitti} 2>&1 | grep path)"
for d in stereo_matching_cuda_amd/_build_exp/x5*/; do
  n=$(basename $d)
  echo "$n: $(SMX_ALLOW_LIB_OVERRIDE=1 SMX_LIB_PATH=$PWD/$d/libsmx_hip.so timeout -k 10 120 python tools/pair_time.py 0 2 ${1:-kitti} 2>&1 | grep path)"
done
