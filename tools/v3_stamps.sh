#!/bin/bash
# Diagnostic build of libsmx_hip.so with in-kernel stamps for work item $1 (default 3000) -> _build_stamps/
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
ITEM=${1:-3000}
OUT=$ROOT/stereo_matching_cuda_amd/_build_stamps
mkdir -p $OUT
cd $ROOT/stereo_matching_cuda_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None -fvisibility=hidden -I$ROOT/include -DSMX_V3_STAMPS=$ITEM"
for f in smx_kernels smx_agg_v3 smx_capi; do /opt/rocm/bin/hipcc $FL -c $f.hip -o $OUT/$f.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libsmx_hip.so $OUT/*.o
