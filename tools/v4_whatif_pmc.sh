#!/bin/bash
# instruction counts of k_v4_walk for what-if variants (tools/exp_build.sh wi<bits> -DSMX_V4_WHATIF=<bits>): the
# difference to the full kernel is the number of instructions the left-out part executes.
# usage (GPU box): tools/v4_whatif_pmc.sh <variant dir names...>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wipmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export SMX_ALLOW_LIB_OVERRIDE=1
for v in "$@"; do
  export SMX_LIB_PATH=$R/stereo_matching_cuda_amd/_build_exp/$v/libsmx_hip.so
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/$v -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/$v.log 2>&1 || echo "$v failed"
  python3 - <<PY
import csv, glob, collections
a = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("$O/$v/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_v4_walk" in r["Kernel_Name"]:
            a[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print("$v".ljust(8), " ".join(f"{k[3:]}={a[k]/max(1,n[k])/1e6:8.1f}M" for k in sorted(a)))
PY
done
