#!/bin/bash
# dev tool: build a variant of libsmx_hip.so into _build_exp/<name>/ with extra flags ON TOP of the product
# flags of csrc/Makefile (so a variant differs from the product build by exactly those flags)
set -e
ROOT=$(cd $(dirname $0)/.. && pwd); NAME=$1; shift
OUT=$ROOT/stereo_matching_cuda_amd/_build_exp/$NAME; mkdir -p $OUT
make -s -C $ROOT/stereo_matching_cuda_amd/csrc OUT=$OUT EXTRA="$*" $OUT/libsmx_hip.so
