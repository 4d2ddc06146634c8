// filter.cuh -- reference stereo_matching_cuda/filter.cuh:12.  filter() is dead code in the
// reference (never called from main.cu); here it is a standalone op on the device:
// mean = truncated zero-padded (2R+1)^2 box mean (u8), var = truncated box mean of I*I minus mean*mean
// (filter.cu:39-115, 143-181).  `cuda` is accepted for signature compatibility.
#pragma once
#include "SystemIncludes.h"
#include "helpers.cuh"

// filter.cuh:6.  The reference's body cannot work (filter.cu:3-27: `size_t` loop variables start at -RADIUS, so the window
// loops never run; the `continue` skips every pixel but a corner block, which gets 0; the division is by (2R+1) and then
// multiplied by (2R+1)).  This is the CPU twin of what filter()'s device path computes for `mean` -- the truncated,
// zero-padded (2R+1)^2 box mean, x offset outer / y offset inner in f32 (filter.cu:57-64) -- so that check_errors() can
// hold it against the GPU result (tests/host_wrappers_check.cpp).
void boxFilterOnCPU(unsigned char* image, unsigned char* mean, int width, int height);

void filter(unsigned char* image, int width, int height, unsigned char* mean, float* var, bool cuda);
