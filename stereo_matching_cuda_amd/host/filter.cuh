// filter.cuh -- reference stereo_matching_cuda/filter.cuh:12.  filter() is dead code in the
// reference (never called from main.cu) and is outside the hot path; the declaration is kept so that
// code including this header still compiles.  Calling it reports "not on the stereo path" and exits.
#pragma once
#include "SystemIncludes.h"
#include "helpers.cuh"

void filter(unsigned char* image, int width, int height, unsigned char* mean, float* var, bool cuda);
