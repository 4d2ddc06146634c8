#!/bin/bash
# Instruction-cache counters of the walker (one --pmc pass of bench.py).  usage: tools/pmc_icache.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/p -- python3 $R/bench.py --no-cpu-baseline --steps 5 > $O/p.log 2>&1; echo rc=$?
tail -2 $O/p.log | cut -c1-300
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$O/p/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-30:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in agg:
    if "walk" in k or "wta" in k:
        print(k, {c: round(agg[k][c] / cnt[k][c]) for c in agg[k]})
PY
