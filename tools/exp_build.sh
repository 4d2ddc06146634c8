#!/bin/bash
# Build timing-experiment variants of libsmx_hip.so (results are WRONG by design; only for
# attributing time inside the walker kernels).  usage: tools/exp_build.sh 1 2 3
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
cd $ROOT/stereo_matching_cuda_amd/csrc
for n in "$@"; do
  out=$ROOT/stereo_matching_cuda_amd/_build/exp$n
  mkdir -p $out
  for f in smx_kernels smx_agg_v2 smx_capi; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off \
      -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fvisibility=hidden -I$ROOT/include \
      -DSMX_EXP=$n -c $f.hip -o $out/$f.o
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libsmx_hip.so $out/*.o
done
