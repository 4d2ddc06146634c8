#!/bin/bash
# A/B on one box: the workgroup's items pipelined (period(h, K)) against a period of the whole item (SMX_V5_NO_OVERLAP=1),
# alternating (the 4K times of one box drift by several per cent within minutes).
# usage: tools/overlap_ab.sh <tag> [workload ...]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R; shift
for wl in ${@:-kitti motorcycle 4k}; do
  for rep in 1 2 3 4; do
    for nov in 0 1; do
      echo -n "no_overlap=$nov " >> $O/overlap_ab.txt
      SMX_V5_NO_OVERLAP=$nov timeout -k 10 200 python3 tools/pair_time.py 0 3 $wl 2>&1 | grep -v amdgpu.ids >> $O/overlap_ab.txt || exit 1
    done
  done
done
cat $O/overlap_ab.txt
