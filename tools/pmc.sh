#!/bin/bash
# Collect rocprofv3 PMC counters for bench.py in separate passes (run on the GPU box via gpurun).
# usage: tools/pmc.sh <outdir-under-gpurun_out> [bench args...]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT SQ_INSTS_GDS SQ_INSTS_BRANCH" \
         "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
with open("$OUT/summary.txt", "w") as o:
    for k in sorted(agg):
        if "walk" not in k and "wta" not in k: continue
        o.write(k + "\n")
        for c in sorted(agg[k]):
            o.write(f"   {c:28s} per-dispatch {agg[k][c]/cnt[k][c]:16.1f}  (n={cnt[k][c]})\n")
print(open("$OUT/summary.txt").read())
PY
