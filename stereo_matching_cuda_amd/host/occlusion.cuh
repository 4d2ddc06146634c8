// occlusion.cuh -- reference stereo_matching_cuda/occlusion.cuh:8-19
#pragma once
#include "SystemIncludes.h"
#include "helpers.cuh"

// dmapl / dmapr are dead in the reference (occlusion.cu:51-52) and untouched here.
void detect_occlusion(float* disparityLeft, float* disparityRight, const int dOcclusion,
                      unsigned char* dmapl, unsigned char* dmapr, const int w, const int h);
void fill_occlusion(float* disparity, const int w, const int h, const float vMin);
// CPU twins (cpu_twins.cpp)
void detect_occlusionOnCPU(float* disparityLeft, float* disparityRight, const int dOcclusion, const int w,
                           const int h);
void fill_occlusionOnCPU(float* disparity, const int w, const int h, const float vMin);
