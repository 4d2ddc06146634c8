#!/bin/bash
# rocprofv3 kernel trace + HBM counters of bench.py for one workload (run on the GPU box).
# usage: tools/gpu_profile_workload.sh <tag> <workload> [steps]
TAG=$1; WL=$2; STEPS=${3:-3}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 python3 $R/bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu-baseline > $O/kt.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $C --output-format csv -d $O/pmc_$C -- python3 $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_$C.log 2>&1
done
head -5 $O/kt/*/*kernel_stats.csv | cut -c1-150
python3 $R/tools/traffic.py $O $O/traffic.json 2>&1 | tail -6
cut -c1-400 $O/bench.json
