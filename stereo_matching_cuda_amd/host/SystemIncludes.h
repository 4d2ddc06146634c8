// SystemIncludes.h -- configuration of the drop-in host layer.
// Mirrors the reference's stereo_matching_cuda/SystemIncludes.h:6-52: the same macro names and
// defaults, but every tunable is overridable at build time (-DD_MIN=-191) and the HIP runtime is
// reached only through the C-ABI of include/smx.h (no device code in this layer).
#pragma once

#ifndef R_W
#define R_W 0.299
#endif
#ifndef G_W
#define G_W 0.587
#endif
#ifndef B_W
#define B_W 0.0721   /* sic, reference SystemIncludes.h:9 */
#endif
#ifndef ALPHA
#define ALPHA 0.9
#endif
#ifndef D_MAX
#define D_MAX 0
#endif
#ifndef D_MIN
#define D_MIN -15
#endif
#ifndef TH_grad
#define TH_grad 2
#endif
#ifndef TH_color
#define TH_color 7
#endif
#ifndef RADIUS
#define RADIUS 9
#endif
#ifndef EPS
#define EPS 6.5025
#endif
#ifndef D_LR
#define D_LR 0
#endif

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <string>

#include "smx.h"

// Reference CHECK (SystemIncludes.h:46-52) prints the CUDA error and exits; here the status comes
// from the C-ABI.  The reference exits with code 0 (sic); a failure exits with 1 here.
#define CHECK(call)                                                                          \
    do {                                                                                     \
        int smx_rc__ = (call);                                                               \
        if (smx_rc__ != SMX_OK) {                                                            \
            fprintf(stderr, "SMX ERROR! file: %s[%i] -> %s\n", __FILE__, __LINE__,           \
                    smx_last_error());                                                       \
            exit(1);                                                                         \
        }                                                                                    \
    } while (0)

// Runtime view of the macros above (main() may override the disparity range from argv).
struct smx_host_config {
    smx_params params;
    int d_min, d_max;
};
smx_host_config& smx_config();
