#!/bin/bash
# dev tool (GPU box): a few SQ counters of k_v5_walk for the product build and one variant (_build_exp/<name>), KITTI bench
# usage: tools/pmc_ab.sh <outdir-under-gpurun_out> <variant>
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; V=$2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for B in product $V; do
  i=0
  for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
    i=$((i+1))
    if [ $B = product ]; then
      timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/${B}_p$i -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --preheat-s 0 > $OUT/${B}_p$i.log 2>&1
    else
      SMX_ALLOW_LIB_OVERRIDE=1 SMX_LIB_PATH=$GRAFT_REPO_ROOT/stereo_matching_cuda_amd/_build_exp/$V/libsmx_hip.so timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/${B}_p$i -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --preheat-s 0 > $OUT/${B}_p$i.log 2>&1
    fi
  done
done
python3 - <<PY
import csv, glob, collections
res = {}
for B in ("product", "$V"):
    agg = collections.defaultdict(float); cnt = collections.defaultdict(int)
    for f in glob.glob("$OUT/%s_p*/*/*counter_collection.csv" % B):
        for r in csv.DictReader(open(f)):
            if "k_v5_walk" not in r["Kernel_Name"]: continue
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    res[B] = {c: agg[c] / cnt[c] for c in agg}
for c in sorted(set(res["product"]) | set(res["$V"])):
    a, b = res["product"].get(c, float("nan")), res["$V"].get(c, float("nan"))
    print(f"   {c:24s} product {a:16.0f}   $V {b:16.0f}   ratio {a / b if b else float('nan'):6.3f}")
PY
