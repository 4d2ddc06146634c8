#!/bin/bash
# Copy one profile set from gpurun_out/ into profiles/r05_* (run here, in the repo root, with the build the runs used still in
# place: traffic.py / valu.py tag their files with smx_version()).
# usage: tools/collect_profiles.sh <suite tag> <motorcycle tag> <4k tag> <pmc tag>
set -e
G=gpurun_out; S=$1; M=$2; K=$3; P=$4
cp $G/$S/bench.json profiles/r05_bench.json
cp $G/$S/kt/*/*kernel_stats.csv profiles/r05_kernel_stats.csv
python3 tools/traffic.py $G/$S profiles/r05_traffic.json | tail -1
for wl in motorcycle:$M 4k:$K; do
  n=${wl%%:*}; t=${wl##*:}
  cp $G/$t/bench.json profiles/r05_${n}_bench.json
  cp $G/$t/kt/*/*kernel_stats.csv profiles/r05_${n}_kernel_stats.csv
  python3 tools/traffic.py $G/$t profiles/r05_${n}_traffic.json | tail -1
done
cp $G/$P/summary.txt profiles/r05_pmc_summary.txt
python3 tools/isa_budget.py profiles/r05_isa_budget.txt > /dev/null 2>&1
python3 tools/valu.py profiles/r05_pmc_summary.txt profiles/r05_isa_budget.txt profiles/r05_valu.json > /dev/null
python3 - <<'PY'
import json
for f in ["r05_bench", "r05_motorcycle_bench", "r05_4k_bench"]:
    r = json.loads([l for l in open(f"profiles/{f}.json") if l.startswith("{")][0]); ro = r["roofline"]
    print(f, "MPix/s", round(r["value"], 1), "ms", round(r["ms_per_step"], 4), "walker ms", round(ro["avg_launch_ms"], 4), "frac", round(ro["frac"], 4),
          "op frac", round(ro["operator"]["frac"], 4), "stages", {k: round(v, 4) for k, v in ro["stage_ms_per_call"].items()})
PY
for f in profiles/r05_kernel_stats.csv profiles/r05_motorcycle_kernel_stats.csv profiles/r05_4k_kernel_stats.csv; do echo $f; sed -n 2,7p $f | cut -d, -f1,2,4 | cut -c1-140; done
