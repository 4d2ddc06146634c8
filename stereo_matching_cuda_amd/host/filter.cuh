// filter.cuh -- reference stereo_matching_cuda/filter.cuh:12.  filter() is dead code in the
// reference (never called from main.cu); here it is a standalone op on the device:
// mean = truncated zero-padded (2R+1)^2 box mean (u8), var = truncated box mean of I*I minus mean*mean
// (filter.cu:39-115, 143-181).  `cuda` is accepted for signature compatibility.
#pragma once
#include "SystemIncludes.h"
#include "helpers.cuh"

void filter(unsigned char* image, int width, int height, unsigned char* mean, float* var, bool cuda);
