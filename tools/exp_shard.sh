#!/bin/bash
# dev tool (GPU box): shard timings (tools/shard_time.py) for every variant under _build_exp/
cd $GRAFT_REPO_ROOT
for d in stereo_matching_cuda_amd/_build_exp/*/; do
  n=$(basename $d); echo "== $n"
  SMX_ALLOW_LIB_OVERRIDE=1 SMX_LIB_PATH=$PWD/$d/libsmx_hip.so timeout -k 10 150 python tools/shard_time.py 1 4 8 2>&1 | grep "^N"
done
