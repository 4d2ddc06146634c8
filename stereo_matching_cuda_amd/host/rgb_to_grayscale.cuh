// rgb_to_grayscale.cuh -- reference stereo_matching_cuda/rgb_to_grayscale.cuh:7
#pragma once
#include "SystemIncludes.h"

// Returns a malloc()ed n-byte gray image the caller frees (reference rgb_to_grayscale.cu:31,72).
// host_gpu_compare is accepted for signature compatibility; the reference's CPU self-check is not
// part of the product path (tests/ compare against the oracle instead).
unsigned char* rgb_to_grayscale(unsigned char* h_rgb, const int n, int channels, bool host_gpu_compare);
