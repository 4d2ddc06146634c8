// costVolume.cuh -- reference stereo_matching_cuda/costVolume.cuh:7-14
#pragma once
#include "SystemIncludes.h"
#include "helpers.cuh"

// cost: size_d*w1*h1 floats, [z][y][x], size_d = d_max - d_min + 1 taken from the configuration exactly
// as the reference takes it from its macros (costVolume.cu:5); slice z has label dmin + z.
// host_gpu_compare: also run the CPU twin and check_errors() the device result against it
// (costVolume.cu:56-74).
void compute_cost(unsigned char* i1, unsigned char* i2, float* cost, int w1, int w2, int h1, int h2,
                  int dmin, bool host_gpu_compare);

// CPU twins (cpu_twins.cpp), reference declarations costVolume.cuh:8-14
void costVolumeOnCPU(unsigned char* i1, unsigned char* i2, float* cost, int w1, int w2, int h1, int h2,
                     int size_d, int dmin);
float x_derivativeCPU(unsigned char* im, int col_index, int index, int width);
int iDivUp(int a, int b);
void compute_costVolumeOnCpu(unsigned char* i1, unsigned char* i2, float* cost, float* derivative1,
                             float* derivative2, int w1, int w2, int h1, int h2, int size_d, int dmin);
void x_derivativeOnCpu(unsigned char* in, float* out, int w, int h);
