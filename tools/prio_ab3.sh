#!/bin/bash
# dev tool (GPU box): role priorities on / off on Motorcycle shape against the slices per walker launch
cd $GRAFT_REPO_ROOT
for sif in 280 140 70 35 18; do
  for p in 1 0; do
    echo "sif=$sif prio=$p $(SMX_SIF=$sif SMX_V5_PRIO=$p timeout -k 10 300 python tools/pair_time.py 0 2 motorcycle 2>&1 | grep path)"
  done
done
