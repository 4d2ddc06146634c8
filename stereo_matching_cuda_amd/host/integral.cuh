// integral.cuh -- reference stereo_matching_cuda/integral.cuh:3
#pragma once
#include "SystemIncludes.h"

void integral(float* image, float* integral, int width, int height);
