// integral.cuh -- reference stereo_matching_cuda/integral.cuh:3,7
#pragma once
#include "SystemIncludes.h"

void integral(float* image, float* integral, int width, int height);
void integralOnCPU(float* in, float* out, const int w, const int h);   // CPU twin (cpu_twins.cpp)
