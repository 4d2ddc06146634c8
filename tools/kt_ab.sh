#!/bin/bash
# Per-kernel A/B on one box: rocprofv3 kernel stats of bench.py for the product build and a variant build.
# usage: tools/kt_ab.sh <tag> <variant dir under _build_exp> <workload> [steps]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; V=$2; WL=$3; ST=${4:-3}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$rep -- python3 $R/bench.py --workload $WL --steps $ST --warmup 1 --no-cpu-baseline > $O/p$rep.log 2>&1
  export SMX_ALLOW_LIB_OVERRIDE=1 SMX_LIB_PATH=$R/stereo_matching_cuda_amd/_build_exp/$V/libsmx_hip.so
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/v$rep -- python3 $R/bench.py --workload $WL --steps $ST --warmup 1 --no-cpu-baseline > $O/v$rep.log 2>&1
  unset SMX_ALLOW_LIB_OVERRIDE SMX_LIB_PATH
done
for d in p1 v1 p2 v2; do echo "== $d"; sed -n 2,3p $O/$d/*/*kernel_stats.csv | cut -d, -f1,2,4 | cut -c1-110; done
