#!/bin/bash
# dev tool (GPU box): what decides whether the role priorities pay -- the volume of the launch or the height of the image
# (the per-strip guidance + image working set against the 4 MB L2 of an XCD)?  Same cells, different aspect.
cd $GRAFT_REPO_ROOT
for wl in 1242,375,192 1242,2000,36 2964,375,80 2964,2000,15 1242,1000,72 1242,1500,48 608,2000,72; do
  for p in 1 0; do
    echo "prio=$p $(SMX_V5_PRIO=$p timeout -k 10 300 python tools/pair_time.py 0 2 $wl 2>&1 | grep path)"
  done
done
