// costVolume.cuh -- reference stereo_matching_cuda/costVolume.cuh:7
#pragma once
#include "SystemIncludes.h"
#include "helpers.cuh"

// cost: size_d*w1*h1 floats, [z][y][x], size_d = d_max - d_min + 1 taken from the configuration exactly
// as the reference takes it from its macros (costVolume.cu:5); slice z has label dmin + z.
void compute_cost(unsigned char* i1, unsigned char* i2, float* cost, int w1, int w2, int h1, int h2,
                  int dmin, bool host_gpu_compare);
