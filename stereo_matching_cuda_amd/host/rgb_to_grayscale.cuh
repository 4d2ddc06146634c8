// rgb_to_grayscale.cuh -- reference stereo_matching_cuda/rgb_to_grayscale.cuh:5-8
#pragma once
#include "SystemIncludes.h"

// Returns a malloc()ed n-byte gray image the caller frees (reference rgb_to_grayscale.cu:31,72).
// host_gpu_compare: also run sumArraysOnHost and compare (rgb_to_grayscale.cu:60-65).
unsigned char* rgb_to_grayscale(unsigned char* h_rgb, const int n, int channels, bool host_gpu_compare);
void sumArraysOnHost(unsigned char* image, unsigned char* gray, const int N, int channels);
bool check_errors_grayscale(unsigned char* host, unsigned char* gpu, int len);
