#!/bin/bash
# dev tool (GPU box): SQ counters of k_v5_walk with the role priorities on / off (SMX_V5_PRIO=1/0) on one workload,
# separate --pmc passes.  usage: tools/prio_pmc.sh <outdir-under-gpurun_out> <workload>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; WL=$2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
for P in 1 0; do
  i=0
  for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    SMX_V5_PRIO=$P timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/prio${P}_p$i -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --no-cpu-baseline --steps 2 --warmup 1 --preheat-s 0 > $OUT/prio${P}_p$i.log 2>&1 || echo "prio $P pass $i failed"
  done
done
python3 - <<PY
import csv, glob, collections
res = {}
for P in (1, 0):
    agg = collections.defaultdict(float); cnt = collections.defaultdict(int)
    for f in glob.glob("$OUT/prio%d_p*/*/*counter_collection.csv" % P):
        for r in csv.DictReader(open(f)):
            if "k_v5_walk" not in r["Kernel_Name"]: continue
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    res[P] = {c: agg[c] / cnt[c] for c in agg}
with open("$OUT/summary.txt", "w") as o:
    o.write("k_v5_walk on $WL, per dispatch: role priorities on (SMX_V5_PRIO=1) / off (0) / ratio\n")
    for c in sorted(set(res[1]) | set(res[0])):
        a, b = res[1].get(c, float("nan")), res[0].get(c, float("nan"))
        o.write(f"   {c:28s} {a:18.1f} {b:18.1f}   {a / b if b else float('nan'):7.3f}\n")
print(open("$OUT/summary.txt").read())
PY
