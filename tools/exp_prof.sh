#!/bin/bash
# dev tool (GPU box): rocprofv3 kernel stats of bench.py for every variant under _build_exp/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in $R/stereo_matching_cuda_amd/_build_exp/*/; do
  n=$(basename $d)
  SMX_ALLOW_LIB_OVERRIDE=1 SMX_LIB_PATH=$d/libsmx_hip.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/xp_$n -- python3 $R/bench.py --no-cpu-baseline --steps 10 > $R/gpurun_out/xp_$n.log 2>&1
  echo "== $n"; cat $R/gpurun_out/xp_$n/*/*kernel_stats.csv | cut -d, -f1-4 | head -4
done
