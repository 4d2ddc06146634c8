#!/bin/bash
# One GPU-box call: the -m gpu suite, then the bench under rocprofv3 --kernel-trace (no counters).  usage: tools/gpu_quick.sh <tag>
TAG=${1:-q}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 python3 $R/bench.py --steps 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline --steps 10 > $O/kt.log 2>&1
cat $O/kt/*/*kernel_stats.csv | cut -c1-130
cut -c1-400 $O/bench.json
