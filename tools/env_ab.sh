#!/bin/bash
# A/B of an environment knob on one box, alternating: tools/env_ab.sh <tag> <VAR> "<values>" [workload ...]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R; VAR=$2; VALS=$3; shift 3
for wl in ${@:-kitti}; do
  for rep in 1 2 3; do
    for v in $VALS; do
      echo -n "$VAR=$v " >> $O/env_ab.txt
      env $VAR=$v timeout -k 10 200 python3 tools/pair_time.py 0 3 $wl 2>&1 | grep -v amdgpu.ids >> $O/env_ab.txt || exit 1
    done
  done
done
cat $O/env_ab.txt
