// Test shim (built by tests/test_host_mirror.py): exposes the CPU twins of
// stereo_matching_cuda_amd/host/cpu_twins.cpp with C linkage so that a CPU-only test can hold them
// against the oracle through ctypes.
#include "costVolume.cuh"
#include "guidedFilter.cuh"
#include "occlusion.cuh"
#include "rgb_to_grayscale.cuh"

extern "C" {
void twin_set_range(int dmin, int dmax) { smx_config().d_min = dmin; smx_config().d_max = dmax; }
void twin_gray(unsigned char* rgb, unsigned char* gray, int n, int ch) { sumArraysOnHost(rgb, gray, n, ch); }
void twin_cost(unsigned char* i1, unsigned char* i2, float* cost, int w, int h, int size_d, int dmin) {
    costVolumeOnCPU(i1, i2, cost, w, w, h, h, size_d, dmin);
}
void twin_integral(float* in, float* out, int w, int h) { integralOnCPU(in, out, w, h); }
void twin_guided(unsigned char* I, float* cost, float* best, float* dmap, unsigned char* mean, int w, int h,
                 int size_d, int dmin) {
    guided_filter_onCpu(I, cost, best, dmap, mean, w, h, size_d, dmin);
}
void twin_detect(float* dl, float* dr, int docc, int w, int h) { detect_occlusionOnCPU(dl, dr, docc, w, h); }
void twin_fill(float* d, int w, int h, float vmin) { fill_occlusionOnCPU(d, w, h, vmin); }
}
