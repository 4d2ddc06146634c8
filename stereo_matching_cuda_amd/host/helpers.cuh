// helpers.cuh -- reference stereo_matching_cuda/helpers.cuh:5-6 (exact-equality compare + print)
#pragma once
#include "SystemIncludes.h"

bool check_errors(float* resCPU, float* resGPU, int len);
bool check_errors(unsigned char* resCPU, unsigned char* resGPU, int len);
