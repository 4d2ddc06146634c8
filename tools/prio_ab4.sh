#!/bin/bash
# dev tool (GPU box): role priorities on / off on KITTI geometry with more and more slices per launch: the hand-off records of
# a launch are 2 x (slice-views) x (bands + 4) x 1.68 KB x (K-1)/K... = ~0.3 MB per slice -- 57 MB at D = 192
cd $GRAFT_REPO_ROOT
for D in 192 400 700 1000 1500; do
  for p in 1 0; do
    echo "D=$D prio=$p $(SMX_V5_PRIO=$p timeout -k 10 300 python tools/pair_time.py 0 2 1242,375,$D 2>&1 | grep path)"
  done
done
