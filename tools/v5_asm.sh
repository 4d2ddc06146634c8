#!/bin/bash
# dev tool (CPU): the gfx950 assembly of smx_agg_v5.hip with the product flags + -DSMX_V5_MARK + extra flags -> $1
# usage: tools/v5_asm.sh out.s [-Dflags...]
set -e
ROOT=$(cd $(dirname $0)/.. && pwd); OUT=$(realpath $1); shift
T=$(mktemp -d); cd $T
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
  -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None -fvisibility=hidden -DSMX_V5_MARK "$@" -I$ROOT/include -I$ROOT/stereo_matching_cuda_amd/csrc \
  --save-temps -c $ROOT/stereo_matching_cuda_amd/csrc/smx_agg_v5.hip -o x.o >/dev/null 2>&1
cp smx_agg_v5-hip-amdgcn-amd-amdhsa-gfx950.s $OUT; cd /; rm -rf $T
