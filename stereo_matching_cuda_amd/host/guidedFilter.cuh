// guidedFilter.cuh -- reference stereo_matching_cuda/guidedFilter.cuh:7-39
#pragma once
#include "SystemIncludes.h"
#include "helpers.cuh"
#include "integral.cuh"

// filter_cost / disp_map are in/out (dispSelectOnGPU, guidedFilter.cu:403-411).
// host_gpu_compare: also run guided_filter_onCpu on copies of the in/out arrays and check_errors() the
// device results against it (the reference runs its twin from main.cu:136-139 without comparing).
void compute_guided_filter(unsigned char* i, float* cost, float* filter_cost, float* disp_map,
                           unsigned char* mean, const int w, const int h, const int size_d, int dmin,
                           bool host_gpu_compare);

// CPU twins (cpu_twins.cpp), reference declarations guidedFilter.cuh:9-20,37-39
void dispSelectOnCPU(float* q, float* filter_cost, float* dmap, const int n, int label);
void computeBoxFilterOnCPU(float* image, float* integral, float* mean, const int w, const int h);
float computeMeanOnCPU(float* I, float* S, int idx, int idy, const int w, const int h);
void chToFlOnCPU(unsigned char* image, float* result, int len);
void flToChOnCPU(float* image, unsigned char* result, int len);
void pixelMultOnCPU(float* image1, float* image2, float* result, int len);
void pixelSousOnCPU(float* image1, float* image2, float* result, int len);
void pixelAddOnCPU(float* image1, float* image2, float* result, int len);
void pixelDivOnCPU(float* image1, float* image2, float* result, int len);
void guided_filter_onCpu(unsigned char* im1, float* cost, float* filtered_cost, float* dmap,
                         unsigned char* mean, const int w, const int h, const int size_d, int dmin);
