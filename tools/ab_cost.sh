cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
echo "product: $(SMX_SRC=cost timeout -k 10 120 python tools/pair_time.py 0 2 kitti 2>&1 | grep path)"
echo "x5prev: $(SMX_SRC=cost SMX_ALLOW_LIB_OVERRIDE=1 SMX_LIB_PATH=$PWD/stereo_matching_cuda_amd/_build_exp/x5prev/libsmx_hip.so timeout -k 10 120 python tools/pair_time.py 0 2 kitti 2>&1 | grep path)"
done
