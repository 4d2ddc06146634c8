#!/bin/bash
# One GPU-box call: the -m gpu test suite, then bench.py under rocprofv3 (kernel trace + HBM counters).
# usage (from the repo root on the GPU box): tools/gpu_suite.sh <tag> [pytest -k expression]
TAG=${1:-run}; KEXPR=${2:-}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
if [ -n "$KEXPR" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
else
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
fi
tail -5 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 python3 $R/bench.py --steps 20 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline --steps 10 > $O/kt.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $O/pmc_$C -- python3 $R/bench.py --no-cpu-baseline --steps 5 > $O/pmc_$C.log 2>&1
done
cat $O/kt/*/*kernel_stats.csv | cut -c1-150
python3 $R/tools/traffic.py $O $O/traffic.json 2>&1 | tail -12
cut -c1-900 $O/bench.json
