#!/bin/bash
# dev tool (GPU box): role priorities of the comb walker on / off (SMX_V5_PRIO) on the three bench shapes, event-free timing
cd $GRAFT_REPO_ROOT
for wl in kitti motorcycle 4k; do
  for p in 1 0; do
    echo "prio=$p $(SMX_V5_PRIO=$p timeout -k 10 300 python tools/pair_time.py 0 2 $wl 2>&1 | grep path)"
  done
done
