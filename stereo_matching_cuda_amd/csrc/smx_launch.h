// Launchers implemented in smx_kernels.hip (all asynchronous on `st`).
#pragma once
#include "smx_common.h"

namespace smx {
int launch_gray(const smx_params* p, const uint8_t* rgb, int64_t n, int ch, uint8_t* gray, hipStream_t st);
int launch_cost(const smx_params* p, const uint8_t* i1, const uint8_t* i2, float* cost, int w, int h,
                int d0, int count, hipStream_t st);
int launch_integral(int mode, const float* in0, const float* in1, float* out0, float* out1, int w,
                    int h, int nplanes, hipStream_t st);
int launch_guid_prep(const uint8_t* I, float* im, float* sq, int64_t n, hipStream_t st);
int launch_guid_finish(const smx_params* p, const float* S_im, const float* S_sq, float* mean_im,
                       float* cinv, uint8_t* mean_u8, int w, int h, hipStream_t st);
int launch_ab(const smx_params* p, const float* Sp, const float* SIp, const float* mean_im,
              const float* cinv, float* A, float* B, int w, int h, int nplanes, hipStream_t st);
int launch_q_wta(const smx_params* p, const float* Sa, const float* Sb, const float* im,
                 int64_t* keys, float* agg, int w, int h, int count, int slice0, hipStream_t st);
int launch_init_keys(int64_t* keys, int64_t n, hipStream_t st);
int launch_init_wta(float* best, float* dmap, int64_t n, hipStream_t st);
int launch_apply_keys(const int64_t* keys, int64_t n, int dmin, float* best, float* dmap, hipStream_t st);
int launch_detect_occlusion(const smx_params* p, float* dL, const float* dR, int dOcc, int w, int h,
                            hipStream_t st);
int launch_fill_occlusion(const float* src, float* disp, int w, int h, float vMin, hipStream_t st);
int launch_finish_keys(const int64_t* keys, int64_t n, int dminl, int dminr, float* best, float* dmap,
                       float* occlusion, hipStream_t st);
int launch_filter(const smx_params* p, const uint8_t* I, uint8_t* mean, float* var, int w, int h,
                  hipStream_t st);
bool finish_pair_row_supported(int w);
int launch_finish_pair_row(const smx_params* p, const int64_t* keys, int w, int h, int dminl, int dminr, int dOcc,
                           float vMin, float* best, float* dmap, float* occlusion, float* filled, hipStream_t st);
}  // namespace smx
