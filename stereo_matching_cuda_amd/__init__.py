"""MI355X-native stereo-pair -> disparity-map path (drop-in for hamza1030/stereo_matching_cuda).

Layout
  csrc/     hand-written HIP kernels (gfx950) + the C-ABI of include/smx.h -> _build/libsmx_hip.so
  host/     C++ mirror of the reference's entry point and per-stage headers (main.cpp,
            costVolume.cuh, guidedFilter.cuh, filter.cuh, integral.cuh, winner_take_all.cuh,
            occlusion.cuh, ...) calling the C-ABI
  stages    numpy front end with the reference's stage names (host pointers in/out)
  device    device-resident pipeline on torch-allocated HBM buffers and HIP streams
  sharded   disparity-slice sharding over ranks + packed-key min all-reduce (RCCL / gloo)
  synth     seeded synthetic stereo pairs

Importing the package does not load the HIP library; the first compute call does, and fails
loudly if it has not been built.
"""
from ._lib import Params, SmxError, build, check, default_params, lib  # noqa: F401
from .stages import (compute_cost, compute_guided_filter, detect_occlusion,  # noqa: F401
                     fill_occlusion, filter, init_wta, integral, rgb_to_grayscale, stereo_pair, write_mat)

__all__ = ["Params", "SmxError", "build", "default_params", "lib", "rgb_to_grayscale",
           "compute_cost", "compute_guided_filter", "integral", "detect_occlusion", "fill_occlusion",
           "init_wta", "stereo_pair", "write_mat", "filter", "check"]
