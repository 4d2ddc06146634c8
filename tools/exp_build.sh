#!/bin/bash
# dev tool: build a variant of libsmx_hip.so into _build_exp/<name>/ with extra flags
set -e
ROOT=$(cd $(dirname $0)/.. && pwd); NAME=$1; shift
OUT=$ROOT/stereo_matching_cuda_amd/_build_exp/$NAME; mkdir -p $OUT
cd $ROOT/stereo_matching_cuda_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -mllvm -amdgpu-atomic-optimizer-strategy=None -fvisibility=hidden -I$ROOT/include $*"
for f in smx_kernels smx_agg_v3 smx_capi; do /opt/rocm/bin/hipcc $FL -c $f.hip -o $OUT/$f.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libsmx_hip.so $OUT/*.o
