// guidedFilter.cuh -- reference stereo_matching_cuda/guidedFilter.cuh:7
#pragma once
#include "SystemIncludes.h"
#include "helpers.cuh"
#include "integral.cuh"

// filter_cost / disp_map are in/out (dispSelectOnGPU, guidedFilter.cu:403-411).
void compute_guided_filter(unsigned char* i, float* cost, float* filter_cost, float* disp_map,
                           unsigned char* mean, const int w, const int h, const int size_d, int dmin,
                           bool host_gpu_compare);
